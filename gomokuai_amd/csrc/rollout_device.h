// rollout_device.h -- Default::RandomRollout (core/lib/include/algorithms/MonteCarlo.hpp:37-47) on bit-board line words,
// shared by the search kernels that play random games to the end (mcts_kernel.hip K3, rave_kernel.hip K8).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "board_device.h"
#include "philox.h"

namespace gmk {
namespace rollout {

// Line words of a position (the layout of K1): word = black | white << 16, bit = position along the line.
// rows [0,15), columns [16,31), diagonals x-y+14 at [32,61), anti-diagonals x+y at [61,90).
constexpr int kLineWords = 92;
constexpr int kColBase = 16, kDiagBase = 32, kAntiBase = 61;

// One Philox block = eight 16-bit draws = the uniform cell draws of eight plies (Board::getRandomMove, Game.cpp:64-73:
// cell = (draw * 225) >> 16), packed as bytes y | x << 4: plies 0-3 in .x, plies 4-7 in .y.
__device__ __forceinline__ uint2 rollout_cells(uint32_t game_id, uint32_t playout, uint32_t c2, uint32_t block, uint32_t k0, uint32_t k1) {
    const gmk::Philox4 p = gmk::philox4x32_10(game_id, playout, c2, block, k0, k1);
    uint32_t cells_lo = 0, cells_hi = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {                                    // eight independent chains
        const uint32_t word = p.v[j >> 1], half = (j & 1) ? (word >> 16) : (word & 0xFFFFu);
        const uint32_t r = (half * 225u) >> 16, yy = r / 15u, byte = yy | ((r - 15u * yy) << 4);
        if (j < 4) cells_lo |= byte << (8 * j); else cells_hi |= byte << (8 * (j - 4));
    }
    return make_uint2(cells_lo, cells_hi);
}

// Default::RandomRollout (MonteCarlo.hpp:37-47) on a lane-private set of line words; returns the winner (+1 / -1 / 0).
// A move sets one bit in the four lines through its cell; five-in-a-row through the new stone
// (Board::checkGameEnd, Game.cpp:88-136) is a run of five in the mover's half of one of those four words.
// Draws: block k >> 3 of the Philox stream (game_id, playout, c2), or, with Precomputed, cells[k >> 3] as rollout_cells
// made them (a caller with idle lanes takes the generator off the serial chain that way).
template <bool Precomputed>
__device__ inline int random_rollout_impl(uint32_t* lines /* [word << stride_log2] */, int stride_log2, int to_move, int stones,
                                          uint32_t game_id, uint32_t playout, uint32_t c2, uint32_t k0, uint32_t k1, const uint2* cells) {
    uint32_t cells_lo = 0, cells_hi = 0;                              // the next eight draws as bytes y | x << 4
    uint2 ahead = Precomputed ? cells[0] : make_uint2(0u, 0u);
    uint32_t stone = to_move > 0 ? 1u : 0x10000u, halves = to_move > 0 ? 0x05040100u : 0x07060302u;
    // All rollouts of the wavefront step together (k is the same for all of them); a finished one is switched off by `live`
    // and the loop ends on a wave-uniform test, so the back edge is a scalar branch instead of per-lane exec bookkeeping.
    bool live = true;
    int result = 0;
    for (uint32_t k = 0;; ++k) {
        if ((k & 7u) == 0u) {
            if (Precomputed) {
                cells_lo = ahead.x; cells_hi = ahead.y;
                ahead = cells[min((k >> 3) + 1u, 28u)];               // in flight during the next eight plies
            } else {
                const uint2 c = rollout_cells(game_id, playout, c2, k >> 3, k0, k1);
                cells_lo = c.x; cells_hi = c.y;
            }
        }
        if (live) {
        const uint32_t cell_byte = (((k & 4u) ? cells_hi : cells_lo) >> (8u * (k & 3u))) & 0xFFu;
        int y = static_cast<int>(cell_byte & 15u);
        int x = static_cast<int>(cell_byte >> 4);
        uint32_t rw = lines[y << stride_log2];
        uint32_t open = ~(rw | (rw >> 16)) & 0x7FFFu & (0x7FFFu << x);
        while (!open) {                                               // linear probe with wrap
            y = (y == 14) ? 0 : y + 1;
            rw = lines[y << stride_log2];
            open = ~(rw | (rw >> 16)) & 0x7FFFu;
        }
        x = __ffs(open) - 1;
        uint32_t* col = lines + ((kColBase + x) << stride_log2);
        uint32_t* dia = lines + ((kDiagBase + x - y + 14) << stride_log2);
        uint32_t* ant = lines + ((kAntiBase + x + y) << stride_log2);
        const uint32_t r_new = rw | (stone << x);
        const uint32_t c_new = *col | (stone << y);
        const uint32_t d_new = *dia | (stone << min(x, y));
        const uint32_t a_new = *ant | (stone << min(14 - x, y));
        lines[y << stride_log2] = r_new; *col = c_new; *dia = d_new; *ant = a_new;
        ++stones;
        // the mover's halves of two line words side by side (bits 15 and 31 are gaps), one run test each
        const uint32_t rc = __builtin_amdgcn_perm(c_new, r_new, halves), da = __builtin_amdgcn_perm(a_new, d_new, halves);
        const uint32_t fives = (rc & (rc >> 1) & (rc >> 2) & (rc >> 3) & (rc >> 4)) | (da & (da >> 1) & (da >> 2) & (da >> 3) & (da >> 4));
        if (fives != 0u || stones == 225) { result = fives ? to_move : 0; live = false; }     // one exit test per move
        to_move = -to_move;
        stone ^= 0x10001u;                                            // bit 0 for black, bit 16 for white
        halves ^= 0x02020202u;                                        // byte selector: the low halves for black, the high halves for white
        }
        if (__ballot(live) == 0ull) return result;
    }
}

__device__ inline int random_rollout(uint32_t* lines, int stride_log2, int to_move, int stones,
                                     uint32_t game_id, uint32_t playout, uint32_t c2, uint32_t k0, uint32_t k1) {
    return random_rollout_impl<false>(lines, stride_log2, to_move, stones, game_id, playout, c2, k0, k1, nullptr);
}
__device__ inline int random_rollout_cells(uint32_t* lines, int stride_log2, int to_move, int stones, const uint2* cells /* [29] */) {
    return random_rollout_impl<true>(lines, stride_log2, to_move, stones, 0u, 0u, 0u, 0u, 0u, cells);
}

// five or more through cell (x, y) for the colour in bits [shift, shift+15), from line words at lines[word * Stride]
template <int Stride>
__device__ __forceinline__ bool five_on_lines(const uint32_t* lines, int x, int y, int shift) {
    const uint32_t rc = ((lines[y * Stride] >> shift) & 0x7FFFu) | (((lines[(kColBase + x) * Stride] >> shift) & 0x7FFFu) << 16);
    const uint32_t da = ((lines[(kDiagBase + x - y + 14) * Stride] >> shift) & 0x7FFFu) | (((lines[(kAntiBase + x + y) * Stride] >> shift) & 0x7FFFu) << 16);
    return run_of_five(rc) || run_of_five(da);
}

}  // namespace rollout
}  // namespace gmk
