// noise_device.h -- Default::AddNoise (core/lib/include/algorithms/MonteCarlo.hpp:97-108, Statistical.hpp:29-34) by ONE WAVEFRONT, with the
// counter-based sampler of include/gomoku_noise.h: lane l holds the root's priors of the cells l, l + 64, l + 128, l + 192, draws their gamma
// variates side by side, and the squares are added in the order gmk_noise_sum225 states (no exchange through memory: DPP row adds, then the four
// row leaders).  The serial statement of the same computation is gmk_noise_mix225; the kernels that search (K3 mcts_kernel.hip, K6
// trad_kernel.hip) call this between two searches of their persistent self-play loops, so no root ever waits for the host to draw its noise.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/gomoku_noise.h"

namespace gmk {
namespace noise {

// lane i of a 16-lane row reads lane i + N of the same row (DPP row_shl; 0 beyond the row)
template <int N>
__device__ __forceinline__ float row_down(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xF, 0xF, true));
}

// the one summation order of the float reductions (oracle/go_trad.c: sum225, include/gomoku_noise.h: gmk_noise_sum225): the caller has added its
// cells l, l + 64, l + 128, l + 192 in that order into `p`; then a binary tree inside every row of 16 lanes (offsets 8, 4, 2, 1: four DPP adds),
// then (row 0 + row 1) + (row 2 + row 3); every lane gets the result
__device__ __forceinline__ float tree_sum(float p) {
    p += row_down<8>(p);
    p += row_down<4>(p);
    p += row_down<2>(p);
    p += row_down<1>(p);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 48));
    return (r0 + r1) + (r2 + r3);
}

// one draw, kept out of line: the sampler is a loop around a Philox block and a dozen binary64 divisions, the callers are kernels that have no
// registers to spare, and the call sits outside their playout loops
static __device__ __noinline__ float gamma_draw(float alpha, uint32_t game_id, uint32_t stones, uint32_t cell, uint32_t seed_lo, uint32_t seed_hi) {
    return gmk_noise_gamma(alpha, game_id, stones, cell, seed_lo, seed_hi);
}

// p[j] = the prior of cell lane + 64 j (0: no child; entries beyond cell 224 must be 0), all 64 lanes of the wavefront call this together.
// On return p[j] is the mixed prior.
__device__ __forceinline__ void mix_root_priors(float (&p)[4], int lane, float alpha, float epsilon, uint32_t game_id, uint32_t stones, uint32_t seed_lo, uint32_t seed_hi) {
    float nz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        p[j] *= 1 - epsilon;                                      // prior_probs *= 1 - epsilon
        nz[j] = p[j] != 0.0f ? gamma_draw(alpha, game_id, stones, static_cast<uint32_t>(lane + 64 * j), seed_lo, seed_hi) : 0.0f;
    }
    float sq = nz[0] * nz[0];
#pragma unroll
    for (int j = 1; j < 4; ++j) if (lane + 64 * j < 225) sq += nz[j] * nz[j];
    const float z = tree_sum(sq);
    if (z > 0.0f) {                                               // normalized(): a zero vector stays zero
        const float nrm = sqrtf(z);
#pragma unroll
        for (int j = 0; j < 4; ++j) nz[j] = nz[j] / nrm;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] += epsilon * nz[j];
}

}  // namespace noise
}  // namespace gmk
