// capi_common.h -- shared state of the C-ABI translation units (error text, device tables).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/gomoku_hip.h"
#include "pattern_tables.h"

namespace gmk {

void set_error(const char* fmt, ...);

struct DeviceState {
    bool ready = false;
    int device = -1;
    int cu_count = 0;
    uint32_t* d_trans = nullptr;        // n_states*4 words, device form (DeviceTables::dev_trans)
    uint32_t* d_records = nullptr;      // 4 words per emission record (DeviceTables::dev_records), then kPrefixWords words of DeviceTables::dev_prefix4
    int n_states = 0, n_patterns = 0, emit_words = 0, n_records = 0;
};
DeviceState& device_state();

// Profiling switches (phase masks, in-kernel timers, launch-shape overrides) exist in the -DGMK_PROFILE build only:
// `python -m gomokuai_amd.build --profile` makes libgomoku_hip_prof.so, which tools/*.sh load with GMK_HIP_LIB=prof.
// The production library never reads the environment, so a stray variable cannot change a result.
#ifdef GMK_PROFILE
inline const char* profile_env(const char* name) { return std::getenv(name); }
constexpr bool kProfileBuild = true;
#else
inline const char* profile_env(const char*) { return nullptr; }
constexpr bool kProfileBuild = false;
#endif

// Device memory for the handles.  The tree arenas are tens of GB per handle, and the driver clears what it takes back before it hands it
// out again: a 24 GB hipMalloc that follows a hipFree of that size takes ~1.5 s instead of 1 ms (tools/alloc_probe.py), which is a third
// of a self-play batch.  So a large block (>= 16 MB) that a handle gives up is kept in a process-wide pool and the next handle that asks for
// about as much (the block is at most a quarter larger) gets it back, uncleared -- no kernel reads a node it has not written -- and small
// requests go to the driver as before.  The pool holds at most kPoolCapBytes; gmk_shutdown returns everything.
hipError_t device_malloc_bytes(void** p, size_t bytes);
hipError_t device_free(void* p);
void device_pool_release();
void device_pool_poison(bool on);
template <class T>
inline hipError_t device_malloc(T** p, size_t bytes) { return device_malloc_bytes(reinterpret_cast<void**>(p), bytes); }

#define GMK_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            gmk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return GMK_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

}  // namespace gmk
