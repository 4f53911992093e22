// capi.hip -- library lifecycle + pattern-table accessors of the C-ABI (include/gomoku_hip.h).
#include <algorithm>
#include <mutex>
#include <cstring>
#include <string>
#include <vector>

#include "capi_common.h"

namespace gmk {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}

DeviceState& device_state() {
    static DeviceState s;
    return s;
}

}  // namespace gmk

using gmk::device_state;
using gmk::production_automaton;

extern "C" const char* gmk_last_error(void) { return gmk::g_error; }

namespace gmk {
namespace {
struct PoolBlock { void* p; size_t bytes; bool used; unsigned long long freed_at; };      // freed_at: a tick of the pool's clock when the block went idle
unsigned long long g_pool_clock = 0;
std::mutex g_pool_mutex;
std::vector<PoolBlock> g_pool;
constexpr size_t kPoolMinBytes = size_t(16) << 20, kPoolCapBytes = size_t(224) << 30;
bool g_pool_poison = false;                                        // gmk_pool_poison: a block that is handed out again is filled with 0xA5 first
// What may stay idle: at most kPoolCapBytes, and at most three quarters of what the device could hand out if the pool gave everything back
// (free + idle) -- the idle blocks are invisible to every other allocator of the process (torch's caching allocator only sees "out of memory"),
// so a quarter of the reclaimable memory (tens of GB) always stays with the driver.  (Not less: a block that goes back to the driver is cleared
// by it when it is handed out again, 1.5 s per 24 GB, to whoever allocates next -- measured: 3.9 s on the training tuples' 2.7 GB right after
// the pool had dropped a 70 GB arena.)
size_t pool_idle_cap(size_t idle) {
    size_t free_bytes = 0, total = 0;
    if (hipMemGetInfo(&free_bytes, &total) != hipSuccess) { (void)hipGetLastError(); return kPoolCapBytes; }
    return std::min(kPoolCapBytes, (free_bytes + idle) / 4 * 3);
}
size_t pool_idle_bytes() { size_t t = 0; for (const PoolBlock& b : g_pool) if (!b.used) t += b.bytes; return t; }
void pool_drop_idle(size_t keep) {                                 // gives idle blocks back to the driver, the longest-idle first, until at most `keep` bytes idle
    while (pool_idle_bytes() > keep) {                             // (what was freed last is what the next handle is most likely to ask for again)
        size_t at = g_pool.size();
        for (size_t i = 0; i < g_pool.size(); ++i) if (!g_pool[i].used && (at == g_pool.size() || g_pool[i].freed_at < g_pool[at].freed_at)) at = i;
        if (at == g_pool.size()) return;
        (void)hipFree(g_pool[at].p);
        g_pool.erase(g_pool.begin() + static_cast<long>(at));
    }
}
}  // namespace

hipError_t device_malloc_bytes(void** p, size_t bytes) {
    if (bytes < kPoolMinBytes) return hipMalloc(p, bytes);
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    size_t best = g_pool.size();
    for (size_t i = 0; i < g_pool.size(); ++i)
        if (!g_pool[i].used && g_pool[i].bytes >= bytes && g_pool[i].bytes - bytes <= bytes / 4 && (best == g_pool.size() || g_pool[i].bytes < g_pool[best].bytes)) best = i;
    if (best != g_pool.size()) {
        g_pool[best].used = true;
        *p = g_pool[best].p;
        if (g_pool_poison || (kProfileBuild && profile_env("GMK_POOL_POISON"))) return hipMemset(*p, 0xA5, g_pool[best].bytes);      // diagnostic: a reused block is NOT zero
        return hipSuccess;
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {                                          // make room: everything idle goes back to the driver, then once more
        (void)hipGetLastError();
        pool_drop_idle(0);
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) g_pool.push_back(PoolBlock{*p, bytes, true, 0ull});
    return e;
}

hipError_t device_free(void* p) {
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (PoolBlock& b : g_pool)
            if (b.p == p) {
                (void)hipDeviceSynchronize();                      // as hipFree would: nothing may still be running on the block when the next handle gets it
                b.used = false;
                b.freed_at = ++g_pool_clock;
                pool_drop_idle(pool_idle_cap(pool_idle_bytes()));
                return hipSuccess;
            }
    }
    return hipFree(p);
}

void device_pool_release() {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    pool_drop_idle(0);
}
void device_pool_poison(bool on) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pool_poison = on;
}
}  // namespace gmk

extern "C" int gmk_init(int device) {
    gmk::DeviceState& st = device_state();
    if (st.ready) return GMK_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        gmk::set_error("no HIP device is visible: libgomoku_hip has no CPU fallback");
        return GMK_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) { gmk::set_error("device %d out of range (%d devices)", device, count); return GMK_ERR_ARG; }
    GMK_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    GMK_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    st.device = device;
    st.cu_count = prop.multiProcessorCount;
    const gmk::DeviceTables& t = production_automaton().device();
    st.n_states = t.n_states;
    st.n_patterns = t.n_patterns;
    st.emit_words = static_cast<int>(t.emit_lists.size());
    st.n_records = t.n_records;
    GMK_HIP_CHECK(hipMalloc(&st.d_trans, t.dev_trans.size() * sizeof(uint32_t)));
    GMK_HIP_CHECK(hipMalloc(&st.d_records, (t.dev_records.size() + gmk::kPrefixWords) * sizeof(uint32_t)));      // the records, then dev_prefix4
    GMK_HIP_CHECK(hipMemcpy(st.d_trans, t.dev_trans.data(), t.dev_trans.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    GMK_HIP_CHECK(hipMemcpy(st.d_records, t.dev_records.data(), t.dev_records.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    GMK_HIP_CHECK(hipMemcpy(st.d_records + t.dev_records.size(), t.dev_prefix4.data(), gmk::kPrefixWords * sizeof(uint32_t), hipMemcpyHostToDevice));
    st.ready = true;
    return GMK_OK;
}

extern "C" int gmk_pool_release(void) {
    gmk::device_pool_release();
    return GMK_OK;
}

extern "C" int gmk_pool_poison(int on) {
    gmk::device_pool_poison(on != 0);
    return GMK_OK;
}

extern "C" int gmk_shutdown(void) {
    gmk::DeviceState& st = device_state();
    if (!st.ready) return GMK_OK;
    gmk::device_pool_release();
    (void)hipFree(st.d_trans);
    (void)hipFree(st.d_records);
    st = gmk::DeviceState{};
    return GMK_OK;
}

extern "C" int gmk_device_info(int* cu_count, size_t* hbm_bytes, char* name, int name_cap) {
    gmk::DeviceState& st = device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded"); return GMK_ERR_STATE; }
    hipDeviceProp_t prop;
    GMK_HIP_CHECK(hipGetDeviceProperties(&prop, st.device));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    if (name && name_cap > 0) {                                  // the marketing name can be empty on this image (no amdgpu.ids): fall back to the ISA name
        std::strncpy(name, prop.name[0] ? prop.name : prop.gcnArchName, static_cast<size_t>(name_cap) - 1);
        name[name_cap - 1] = 0;
    }
    return GMK_OK;
}

extern "C" int gmk_tables_info(gmk_table_info* info) {
    if (!info) return GMK_ERR_ARG;
    const gmk::PatternAutomaton& a = production_automaton();
    const gmk::DeviceTables& t = a.device();
    info->n_patterns = t.n_patterns;
    info->n_states = t.n_states;
    info->dat_size = static_cast<int32_t>(a.base().size());
    info->max_emissions = t.max_emissions;
    info->trans_words = static_cast<int32_t>(t.trans.size());
    info->emit_words = static_cast<int32_t>(t.emit_lists.size());
    for (int i = 0; i < 5; ++i) info->invariants[i] = a.invariants()[i];
    return GMK_OK;
}

extern "C" int gmk_tables_pattern(int i, char str[8], int* favour, int* type, int* score) {
    const auto& pats = production_automaton().patterns();
    if (i < 0 || i >= static_cast<int>(pats.size())) return GMK_ERR_ARG;
    if (str) { std::memset(str, 0, 8); std::memcpy(str, pats[i].rich.data(), pats[i].rich.size()); }
    if (favour) *favour = pats[i].favour;
    if (type) *type = pats[i].type;
    if (score) *score = pats[i].score;
    return GMK_OK;
}

extern "C" int gmk_tables_copy(uint32_t* trans, uint16_t* emit_lists, uint32_t* pattern_info) {
    const gmk::DeviceTables& t = production_automaton().device();
    if (trans) std::memcpy(trans, t.trans.data(), t.trans.size() * sizeof(uint32_t));
    if (emit_lists) std::memcpy(emit_lists, t.emit_lists.data(), t.emit_lists.size() * sizeof(uint16_t));
    if (pattern_info) std::memcpy(pattern_info, t.pattern_info.data(), t.pattern_info.size() * sizeof(uint32_t));
    return GMK_OK;
}

extern "C" int gmk_tables_copy_dat(int32_t* base, int32_t* check, int32_t* fail) {
    const gmk::PatternAutomaton& a = production_automaton();
    const size_t bytes = a.base().size() * sizeof(int32_t);
    if (base) std::memcpy(base, a.base().data(), bytes);
    if (check) std::memcpy(check, a.check().data(), bytes);
    if (fail) std::memcpy(fail, a.fail().data(), bytes);
    return GMK_OK;
}

extern "C" int gmk_tables_scan(const uint8_t* codes, int n, int32_t* patterns, int32_t* offsets, int cap) {
    if (!codes || n < 0) return GMK_ERR_ARG;
    for (int i = 0; i < n; ++i) if (codes[i] < 1 || codes[i] > 4) return GMK_ERR_ARG;
    const auto stream = production_automaton().scan(codes, n);
    int m = 0;
    for (const auto& pr : stream) {
        if (m < cap) { if (patterns) patterns[m] = pr.first; if (offsets) offsets[m] = pr.second; }
        ++m;
    }
    return m;
}
