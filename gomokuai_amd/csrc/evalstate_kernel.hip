// evalstate_kernel.hip -- K2: the incrementally maintained Evaluator on the device, one state per game.
//
// Evaluator::applyMove / revertMove scripts (core/lib/src/Pattern.cpp:274-342) over device-resident states; the update
// itself lives in evalstate_device.h.  Mapping: one wavefront per game, six games per workgroup.  The 17.8 KB state
// is copied into LDS, all moves of the launch are applied there, and it is written back once.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "evalstate_device.h"

namespace {

using namespace gmk::evs;
constexpr int kGamesPerBlock = 9;                // 9 x 16.3 KB of state and scratch + 14.4 KB of automaton tables fit one CU's 160 KB of LDS (7 before the density words were packed and the matches left their LDS queue)
constexpr int kThreads = 64 * kGamesPerBlock;

__global__ __launch_bounds__(kThreads)
void evalstate_update_kernel(uint32_t* __restrict__ states, const int16_t* __restrict__ moves, int moves_per_game, int n_games,
                             const uint32_t* __restrict__ g_trans, const uint32_t* __restrict__ g_records, int trans_words, int record_words,
                             unsigned long long* prof /* profiling aid, normally null: 8 counters per game */, int phases /* profiling aid: 0x3F */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // layout: [games: kGamesPerBlock * (kStateWords + kScratchWords rounded)] [trans] [records, then the four-symbol prefix table: record_words counts both]
    constexpr int kPerGame = (kStateWords + kScratchWords + 16 + 3) & ~3;     // + 8 profiling counters
    uint32_t* s_trans = lds + kGamesPerBlock * kPerGame;
    for (int i = threadIdx.x; i < trans_words + record_words; i += kThreads) s_trans[i] = i < trans_words ? g_trans[i] : g_records[i - trans_words];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int game = blockIdx.x * kGamesPerBlock + wave;
    uint32_t* st = lds + wave * kPerGame;
    if (game < n_games) {
        const uint4* src = reinterpret_cast<const uint4*>(states + static_cast<size_t>(game) * kStateWords);
        uint4* dst = reinterpret_cast<uint4*>(st);
        // six loads in flight per lane (one by one the copy is eighteen trips to memory in a row)
        constexpr int kQuads = kStateWords / 4, kRounds = kQuads / 64;              // whole rounds of 64 lanes, then the rest
#pragma unroll
        for (int r0 = 0; r0 < kRounds + 1; r0 += 6) {
            uint4 t[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) t[u] = src[min(64 * (r0 + u) + lane, kQuads - 1)];          // (the last round is partly there: clamped, stored below only where it is)
#pragma unroll
            for (int u = 0; u < 6; ++u) if (64 * (r0 + u) + lane < kQuads) dst[64 * (r0 + u) + lane] = t[u];
        }
    }
    __syncthreads();
    if (game >= n_games) return;
    Ctx c{st, st + kStateWords, reinterpret_cast<const char*>(s_trans), reinterpret_cast<const uint4*>(s_trans + trans_words),
          reinterpret_cast<const char*>(s_trans + trans_words + record_words - gmk::kPrefixWords), lane,
          gmk::kProfileBuild && prof ? reinterpret_cast<unsigned long long*>(st + kStateWords + kScratchWords) : nullptr,
          gmk::kProfileBuild ? phases : 0x3F};                    // (the production build has neither: the update's seven timer tests and six phase tests fold away)
    if (prof && lane < 16) st[kStateWords + kScratchWords + lane] = 0u;
    // the script 64 steps at a time, one step per lane, handed out with v_readlane: a load per step would put a trip to memory in
    // front of every update (the wavefront has nothing else to do meanwhile)
    for (int m0 = 0; m0 < moves_per_game; m0 += 64) {
        const int mine = m0 + lane < moves_per_game ? moves[static_cast<size_t>(game) * moves_per_game + m0 + lane] : -1;
        const int steps = min(64, moves_per_game - m0);
        for (int j = 0; j < steps; ++j) {
            const int mv = __builtin_amdgcn_readlane(mine, j);
            evaluator_step(c, mv);                              // >= 0 apply, -2 revert, -1 nothing
            wave_phase_fence();
        }
    }
    uint4* dst = reinterpret_cast<uint4*>(states + static_cast<size_t>(game) * kStateWords);
    const uint4* src = reinterpret_cast<const uint4*>(st);
    for (int i = lane; i < kStateWords / 4; i += 64) dst[i] = src[i];
    if (prof && lane < 8) prof[static_cast<size_t>(game) * 8 + lane] = c.prof[lane];
}

}  // namespace

struct gmk_evalstate {
    int n_games = 0;
    uint32_t* d_states = nullptr;
    bool attr_set = false;
};

extern "C" int gmk_evalstate_reset(gmk_evalstate* e) {
    if (!e) return GMK_ERR_ARG;
    std::vector<uint32_t> one(kStateWords, 0u);
    fill_initial_state(one.data());
    std::vector<uint32_t> all(static_cast<size_t>(e->n_games) * kStateWords);
    for (int g = 0; g < e->n_games; ++g) std::memcpy(all.data() + static_cast<size_t>(g) * kStateWords, one.data(), kStateWords * 4);
    GMK_HIP_CHECK(hipMemcpy(e->d_states, all.data(), all.size() * 4, hipMemcpyHostToDevice));
    return GMK_OK;
}

extern "C" int gmk_evalstate_create(int n_games, gmk_evalstate** out) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (!out || n_games <= 0) { gmk::set_error("gmk_evalstate_create: bad arguments"); return GMK_ERR_ARG; }
    gmk_evalstate* e = new gmk_evalstate;
    e->n_games = n_games;
    if (hipMalloc(&e->d_states, static_cast<size_t>(n_games) * kStateWords * 4) != hipSuccess) { delete e; gmk::set_error("gmk_evalstate_create: hipMalloc failed"); return GMK_ERR_HIP; }
    const int rc = gmk_evalstate_reset(e);
    if (rc != GMK_OK) { (void)hipFree(e->d_states); delete e; return rc; }
    *out = e;
    return GMK_OK;
}

extern "C" int gmk_evalstate_destroy(gmk_evalstate* e) {
    if (!e) return GMK_OK;
    (void)hipFree(e->d_states);
    delete e;
    return GMK_OK;
}

extern "C" int gmk_evalstate_update(gmk_evalstate* e, const int16_t* d_moves, int moves_per_game, void* stream) {
    gmk::DeviceState& st = gmk::device_state();
    if (!e || !d_moves || moves_per_game < 0) { gmk::set_error("gmk_evalstate_update: bad arguments"); return GMK_ERR_ARG; }
    if (moves_per_game == 0) return GMK_OK;
    constexpr int kPerGame = (kStateWords + kScratchWords + 16 + 3) & ~3;
    const size_t lds = static_cast<size_t>(kGamesPerBlock * kPerGame + st.n_states * 4 + st.n_records * 4 + gmk::kPrefixWords) * 4;
    if (!e->attr_set) {
        GMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(evalstate_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        e->attr_set = true;
    }
    const int grid = (e->n_games + kGamesPerBlock - 1) / kGamesPerBlock;
    unsigned long long* d_prof = nullptr;
    static const bool profile = gmk::profile_env("GMK_EVS_PROFILE") != nullptr;
    static const int phases = gmk::profile_env("GMK_EVS_PHASE_MASK") ? std::atoi(gmk::profile_env("GMK_EVS_PHASE_MASK")) : 0x3F;
    if (profile) {
        GMK_HIP_CHECK(hipMalloc(&d_prof, static_cast<size_t>(e->n_games) * 8 * sizeof(unsigned long long)));
        GMK_HIP_CHECK(hipMemset(d_prof, 0, static_cast<size_t>(e->n_games) * 8 * sizeof(unsigned long long)));
    }
    hipLaunchKernelGGL(evalstate_update_kernel, dim3(grid), dim3(kThreads), lds, static_cast<hipStream_t>(stream), e->d_states, d_moves,
                       moves_per_game, e->n_games, st.d_trans, st.d_records, st.n_states * 4, st.n_records * 4 + gmk::kPrefixWords, d_prof, phases);
    GMK_HIP_CHECK(hipGetLastError());
    if (profile) {                                              // cycles per phase of Updater::updateMove, mean over games
        std::vector<unsigned long long> h(static_cast<size_t>(e->n_games) * 8);
        GMK_HIP_CHECK(hipDeviceSynchronize());
        GMK_HIP_CHECK(hipMemcpy(h.data(), d_prof, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(d_prof);
        double sum[8] = {};
        for (size_t i = 0; i < h.size(); ++i) sum[i % 8] += static_cast<double>(h[i]);
        const char* names[7] = {"match-", "compounds-", "patterns-", "board+block", "match+", "patterns+", "compounds+"};
        std::fprintf(stderr, "[GMK_EVS_PROFILE] s_memtime ticks (100 MHz) per update:");
        for (int k = 0; k < 7; ++k) std::fprintf(stderr, " %s %.1f", names[k], sum[k] / (sum[7] > 0 ? sum[7] : 1));
        std::fprintf(stderr, "\n");
    }
    return GMK_OK;
}

extern "C" int gmk_evalstate_update_host(gmk_evalstate* e, const int16_t* h_moves, int moves_per_game) {
    if (!e || !h_moves || moves_per_game < 0) { gmk::set_error("gmk_evalstate_update_host: bad arguments"); return GMK_ERR_ARG; }
    int16_t* d = nullptr;
    const size_t bytes = static_cast<size_t>(e->n_games) * moves_per_game * sizeof(int16_t);
    if (bytes == 0) return GMK_OK;
    GMK_HIP_CHECK(hipMalloc(&d, bytes));
    int rc = GMK_OK;
    if (hipMemcpy(d, h_moves, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = GMK_ERR_HIP;
    if (rc == GMK_OK) rc = gmk_evalstate_update(e, d, moves_per_game, nullptr);
    if (rc == GMK_OK && hipDeviceSynchronize() != hipSuccess) rc = GMK_ERR_HIP;
    (void)hipFree(d);
    return rc;
}

extern "C" int gmk_evalstate_read(gmk_evalstate* e, int32_t* h_scores, int32_t* h_density, uint32_t* h_pattern_dist, uint32_t* h_compound_dist,
                                  int32_t* h_meta, uint8_t* h_record) {
    if (!e) return GMK_ERR_ARG;
    std::vector<uint32_t> all(static_cast<size_t>(e->n_games) * kStateWords);
    GMK_HIP_CHECK(hipDeviceSynchronize());
    GMK_HIP_CHECK(hipMemcpy(all.data(), e->d_states, all.size() * 4, hipMemcpyDeviceToHost));
    for (int g = 0; g < e->n_games; ++g) {
        const uint32_t* s = all.data() + static_cast<size_t>(g) * kStateWords;
        if (h_scores) std::memcpy(h_scores + static_cast<size_t>(g) * 900, s + oScores, 3600);
        if (h_density) unpack_density(s, h_density + static_cast<size_t>(g) * 900);
        if (h_pattern_dist)
            for (int cell = 0; cell < 226; ++cell) std::memcpy(h_pattern_dist + (static_cast<size_t>(g) * 226 + cell) * 8, s + oPdist + pdist_index(cell, 0), 32);
        if (h_compound_dist) std::memcpy(h_compound_dist + static_cast<size_t>(g) * 226 * 3, s + oCdist, 226 * 3 * 4);
        if (h_meta) std::memcpy(h_meta + static_cast<size_t>(g) * 4, s + oMeta, 16);
        if (h_record) std::memcpy(h_record + static_cast<size_t>(g) * 228, s + oRecord, 228);
    }
    return GMK_OK;
}
