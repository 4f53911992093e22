// rollout_device.h -- Default::RandomRollout (core/lib/include/algorithms/MonteCarlo.hpp:37-47) on bit-board line words,
// shared by the search kernels that play random games to the end (mcts_kernel.hip K3, rave_kernel.hip K8).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "board_device.h"
#include "philox.h"

namespace gmk {
namespace rollout {

// Line words of a position: word = black | white << 16; rows [0,15) bit x, columns [16,31) bit y, diagonals x-y+14 at [32,61) and
// anti-diagonals x+y at [61,90) bit x (neighbours along any line sit in neighbouring bits, which is all a run test needs; x for both
// diagonal kinds makes a move's stone bit the same for three of its four words).
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

// Default::RandomRollout (MonteCarlo.hpp:37-47) on a lane-private set of line words in LDS; returns the winner (+1 / -1 / 0).
// A move sets one bit in the four lines through its cell; five-in-a-row through the new stone
// (Board::checkGameEnd, Game.cpp:88-136) is a run of five in the mover's half of one of those four words.
// Draws: fetch(b) = block b of the lane's Philox stream as rollout_cells made it (the callers compute the blocks with all their lanes
// before the rollout starts: the generator is ~150 instructions per eight plies, and a wavefront that is alone on its SIMD pays for
// every instruction of the chain).  The loop body is one block = eight plies, unrolled: the cell's byte position, the mover's stone
// bit and byte selector are then constants of the ply's position in the block, and line addresses are one multiply-add each.
using lds_u32 = __attribute__((address_space(3))) uint32_t;
__device__ __forceinline__ lds_u32* lds_at(uint32_t byte_address) { return reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(byte_address)); }

template <class Fetch>
__device__ inline int random_rollout_blocks(uint32_t* lines /* [word * stride], in LDS */, uint32_t stride /* words between a lane's line words */,
                                            int to_move, int stones, int no_tie_before /* wave-uniform: no rollout of the wavefront fills its board before this ply
                                            (the smallest 224 - stones among them; 0 is always right) */, Fetch fetch) {
    const uint32_t sb = 4u * stride;                               // bytes (< 2^24: the multiply-adds below are 24-bit)
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u32*)lines));
    uint32_t col0 = base + kColBase * sb, dia0 = base + (kDiagBase + 14) * sb, ant0 = base + kAntiBase * sb;
    asm volatile("" : "+v"(col0), "+v"(dia0), "+v"(ant0));         // kept in registers: a line's address is one multiply-add
    // the movers of the even and of the odd plies: stone bit (bit 0 black, bit 16 white) and v_perm selector of the mover's halves
    const uint32_t stone_even = to_move > 0 ? 1u : 0x10000u, stone_odd = stone_even ^ 0x10001u;
    const uint32_t halves_even = to_move > 0 ? 0x05040100u : 0x07060302u, halves_odd = halves_even ^ 0x02020202u;
    const int last_ply = 224 - stones;                             // the ply that fills the board
    uint32_t won = 0;                                              // 2 | parity of the ply that made five, 0: none (yet)
    // All rollouts of the wavefront step together; a finished one is switched off by `live` and the loop ends on a wave-uniform test
    // (twice per block: the plies in between find no live lane and are skipped).
    // The row word of a ply's cell is read one ply ahead, together with the three other lines of the ply before (one LDS round trip
    // per ply instead of two: the wavefront is alone on its SIMD, nothing else covers the wait); that read is issued before the
    // ply's writes, so the row the ply itself changes is taken from its registers instead.
    bool live = true;
    uint2 cur = fetch(0u);
    uint32_t rw_ahead = *lds_at(base + __umul24(cur.x & 15u, sb));
    // one block of eight plies; Tie: a board may fill up during it (the test costs three instructions a ply, and full boards are rare:
    // the blocks before `no_tie_before` run without it)
    auto play_block = [&](auto tie_tag, uint32_t b, const uint2 ahead) -> bool {
        constexpr bool kTie = decltype(tie_tag)::value;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (live) {
                const uint32_t cw = j < 4 ? cur.x : cur.y, cw_next = j + 1 < 4 ? cur.x : j + 1 < 8 ? cur.y : ahead.x;
                uint32_t y = (cw >> (8 * (j & 3))) & 15u, x = (cw >> (8 * (j & 3) + 4)) & 15u;
                const uint32_t y_next = (cw_next >> (8 * ((j + 1) & 3))) & 15u;
                uint32_t row_at = base + __umul24(y, sb);
                uint32_t rw = rw_ahead;
                uint32_t open = ~(rw | (rw >> 16)) & 0x7FFFu & (0x7FFFu << x);
                while (!open) {                                     // linear probe with wrap (Board::getRandomMove, Game.cpp:64-73)
                    y = (y == 14u) ? 0u : y + 1u;
                    row_at = base + __umul24(y, sb);
                    rw = *lds_at(row_at);
                    open = ~(rw | (rw >> 16)) & 0x7FFFu;
                }
                x = static_cast<uint32_t>(__ffs(open)) - 1u;
                const uint32_t col_at = col0 + __umul24(x, sb), dia_at = dia0 + static_cast<uint32_t>(__mul24(static_cast<int>(x) - static_cast<int>(y), static_cast<int>(sb))),
                               ant_at = ant0 + __umul24(x + y, sb);
                const uint32_t stone = (j & 1) ? stone_odd : stone_even, halves = (j & 1) ? halves_odd : halves_even;
                const uint32_t c_old = *lds_at(col_at), d_old = *lds_at(dia_at), a_old = *lds_at(ant_at);
                const uint32_t next_row = *lds_at(base + __umul24(y_next, sb));
                const uint32_t r_new = rw | (stone << x);
                const uint32_t c_new = c_old | (stone << y);
                const uint32_t d_new = d_old | (stone << x);
                const uint32_t a_new = a_old | (stone << x);
                *lds_at(row_at) = r_new; *lds_at(col_at) = c_new; *lds_at(dia_at) = d_new; *lds_at(ant_at) = a_new;
                rw_ahead = y_next == y ? r_new : next_row;
                // the mover's halves of two line words side by side (bits 15 and 31 are gaps), one run test each
                const uint32_t rc = __builtin_amdgcn_perm(c_new, r_new, halves), da = __builtin_amdgcn_perm(a_new, d_new, halves);
                const uint32_t rc2 = rc & (rc >> 1), da2 = da & (da >> 1);                       // runs of two, of four, of five
                const uint32_t rc4 = rc2 & (rc2 >> 2), da4 = da2 & (da2 >> 2);
                const uint32_t fives = (rc4 & (rc >> 4)) | (da4 & (da >> 4));
                if (fives != 0u) won = 2u | static_cast<uint32_t>(j & 1);
                live = fives == 0u && (!kTie || static_cast<int>(8u * b) + j != last_ply);
            }
            if ((j & 3) == 3 && __ballot(live) == 0ull) return true;
        }
        return false;
    };
    for (uint32_t b = 0;; ++b) {
        const uint2 ahead = fetch(b + 1u);                          // in flight during these eight plies
        const bool over = static_cast<int>(8u * b) + 7 < no_tie_before ? play_block(std::false_type{}, b, ahead) : play_block(std::true_type{}, b, ahead);
        if (over) return won ? ((won & 1u) ? -to_move : to_move) : 0;
        cur = ahead;
    }
}

// What a rollout carries from one span of blocks to the next (random_rollout_quads_span / _pairs_span): the callers generate a rollout's draws in
// STAGES -- the first blocks for every rollout, the rest only for the rollouts that are still running after them (a rollout ends after ~100 of up to
// 220 plies, and a Philox block is ~150 instructions) -- and the rollout stops at the end of a stage and goes on after the next one.
struct RolloutState {
    bool live = true;
    uint32_t won = 0;                              // 2 | parity of the ply that made five, 0: none (yet)
    __device__ __forceinline__ int winner(int to_move) const { return won ? ((won & 1u) ? -to_move : to_move) : 0; }
};

// The same rollout on FOUR lanes (an aligned quad of the wavefront, all four called with the same arguments): lane d of the quad keeps
// the line of direction d through the move's cell (0 row, 1 column, 2 diagonal, 3 anti-diagonal) -- one address, one read, one write and
// one run test per lane and ply where the single-lane form does four of each, and the quad ORs its four verdicts with two DPP
// instructions.  What a ply costs a wavefront is its instruction count (a wave64 instruction occupies the SIMD four cycles whatever
// the number of live lanes): ~30 instead of ~42, for lanes that would idle otherwise.  Every lane of the quad follows the row word
// (the probe rule needs it), so the move's cell is known to all four without an exchange.
template <class Fetch>
__device__ inline int random_rollout_quads(uint32_t* lines /* [word * stride], in LDS */, uint32_t stride, int to_move, int stones, int no_tie_before, Fetch fetch) {
    uint32_t sb = 4u * stride;
    uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u32*)lines));
    asm volatile("" : "+v"(sb), "+v"(base));                       // opaque: what derives from them is computed here, per call (hoisted out of the caller's
                                                                   // loop it would live in registers the caller does not have, i.e. in scratch)
    const uint32_t d = threadIdx.x & 3u;
    // this lane's line of a move at (x, y): word c0 + ax * x + ay * y, stone bit = y for the column, x otherwise
    int line0 = static_cast<int>(base + (d == 0u ? 0u : d == 1u ? kColBase : d == 2u ? kDiagBase + 14u : kAntiBase) * sb);
    int per_x = d == 0u ? 0 : static_cast<int>(sb), per_y = d == 0u ? static_cast<int>(sb) : d == 1u ? 0 : d == 2u ? -static_cast<int>(sb) : static_cast<int>(sb);
    asm volatile("" : "+v"(line0), "+v"(per_x), "+v"(per_y));
    uint32_t column_mask = d == 1u ? 0xFFFFFFFFu : 0u;            // the stone's bit position: y on the column's lane, x elsewhere -- one v_bfi with a mask in a register
    asm volatile("" : "+v"(column_mask));                        // (as a lane predicate it sat in a spilled scalar pair: two v_readlane a ply)
    const uint32_t stone_even = to_move > 0 ? 1u : 0x10000u, stone_odd = stone_even ^ 0x10001u;
    const uint32_t half_even = to_move > 0 ? 0u : 16u, half_odd = half_even ^ 16u;          // where the mover's fifteen bits start
    const int last_ply = 224 - stones;
    uint32_t won = 0;
    bool live = true;
    uint2 cur = fetch(0u);
    uint32_t rw_ahead = *lds_at(base + __umul24(cur.x & 15u, sb));
    auto play_block = [&](auto tie_tag, uint32_t b, const uint2 ahead) -> bool {
        constexpr bool kTie = decltype(tie_tag)::value;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (live) {                                             // quad-uniform: the four lanes share the verdicts
                const uint32_t cw = j < 4 ? cur.x : cur.y, cw_next = j + 1 < 4 ? cur.x : j + 1 < 8 ? cur.y : ahead.x;
                uint32_t y = (cw >> (8 * (j & 3))) & 15u, x = (cw >> (8 * (j & 3) + 4)) & 15u;
                const uint32_t y_next = (cw_next >> (8 * ((j + 1) & 3))) & 15u;
                uint32_t rw = rw_ahead;
                uint32_t open = ~(rw | (rw >> 16)) & 0x7FFFu & (0x7FFFu << x);
                while (!open) {                                     // linear probe with wrap (Board::getRandomMove, Game.cpp:64-73)
                    y = (y == 14u) ? 0u : y + 1u;
                    rw = *lds_at(base + __umul24(y, sb));
                    open = ~(rw | (rw >> 16)) & 0x7FFFu;
                }
                x = static_cast<uint32_t>(__ffs(open)) - 1u;
                int at_row = __mul24(static_cast<int>(y), per_y) + line0;              // two multiply-adds (summed in one expression they become two multiplies and an add3)
                asm volatile("" : "+v"(at_row));
                const uint32_t at = static_cast<uint32_t>(__mul24(static_cast<int>(x), per_x) + at_row);
                const uint32_t stone = (j & 1) ? stone_odd : stone_even, half = (j & 1) ? half_odd : half_even;
                const uint32_t old = *lds_at(at);
                const uint32_t next_row = *lds_at(base + __umul24(y_next, sb));
                const uint32_t mine = old | (stone << ((y & column_mask) | (x & ~column_mask)));
                *lds_at(at) = mine;
                rw_ahead = y_next == y ? (rw | (stone << x)) : next_row;
                const uint32_t h = (mine >> half) & 0x7FFFu;
                const uint32_t h2 = h & (h >> 1), h4 = h2 & (h2 >> 2);
                uint32_t fives = h4 & (h >> 4);
                fives |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(fives), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
                fives |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(fives), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
                if (fives != 0u) won = 2u | static_cast<uint32_t>(j & 1);
                live = fives == 0u && (!kTie || static_cast<int>(8u * b) + j != last_ply);
            }
            if ((j & 3) == 3 && __ballot(live) == 0ull) return true;
        }
        return false;
    };
    for (uint32_t b = 0;; ++b) {
        const uint2 ahead = fetch(b + 1u);
        const bool over = static_cast<int>(8u * b) + 7 < no_tie_before ? play_block(std::false_type{}, b, ahead) : play_block(std::true_type{}, b, ahead);
        if (over) return won ? ((won & 1u) ? -to_move : to_move) : 0;
        cur = ahead;
    }
}

// ... and on TWO lanes (an aligned pair, both called with the same arguments), for wavefronts whose rollouts leave no room for quads: the even
// lane keeps the row and the column through the move's cell, the odd lane the diagonal and the anti-diagonal; two addresses, two reads, two
// writes and one run test (both words' mover halves side by side) per lane and ply, one DPP instruction for the pair's verdict: ~31
// instructions a ply.
template <class Fetch>
__device__ inline bool random_rollout_pairs_span(uint32_t* lines /* [word * stride], in LDS */, uint32_t stride, int to_move, int stones, int no_tie_before, Fetch fetch,
                                                 RolloutState& state, uint32_t b_from, uint32_t b_to) {
    uint32_t sb = 4u * stride;
    uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u32*)lines));
    asm volatile("" : "+v"(sb), "+v"(base));                       // opaque, as in random_rollout_quads
    const bool odd = (threadIdx.x & 1u) != 0u;
    // this lane's two lines of a move at (x, y): first = row | diagonal (stone bit x), second = column (bit y) | anti-diagonal (bit x)
    int first0 = static_cast<int>(base + (odd ? (kDiagBase + 14u) * sb : 0u)), second0 = static_cast<int>(base + (odd ? kAntiBase : kColBase) * sb);
    int first_x = odd ? static_cast<int>(sb) : 0, first_y = odd ? -static_cast<int>(sb) : static_cast<int>(sb), second_y = odd ? static_cast<int>(sb) : 0;
    asm volatile("" : "+v"(first0), "+v"(second0), "+v"(first_x), "+v"(first_y), "+v"(second_y));
    const uint32_t stone_even = to_move > 0 ? 1u : 0x10000u, stone_odd = stone_even ^ 0x10001u;
    const uint32_t halves_even = to_move > 0 ? 0x05040100u : 0x07060302u, halves_odd = halves_even ^ 0x02020202u;
    uint32_t odd_mask = odd ? 0xFFFFFFFFu : 0u;                   // second word's bit position: x on the odd lane (anti-diagonal), y on the even one (column)
    asm volatile("" : "+v"(odd_mask));
    const int last_ply = 224 - stones;
    uint32_t won = state.won;
    bool live = state.live;
    uint2 cur = fetch(b_from);
    uint32_t rw_ahead = *lds_at(base + __umul24(cur.x & 15u, sb));
    auto play_block = [&](auto tie_tag, uint32_t b, const uint2 ahead) -> bool {
        constexpr bool kTie = decltype(tie_tag)::value;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (live) {                                             // pair-uniform
                const uint32_t cw = j < 4 ? cur.x : cur.y, cw_next = j + 1 < 4 ? cur.x : j + 1 < 8 ? cur.y : ahead.x;
                uint32_t y = (cw >> (8 * (j & 3))) & 15u, x = (cw >> (8 * (j & 3) + 4)) & 15u;
                const uint32_t y_next = (cw_next >> (8 * ((j + 1) & 3))) & 15u;
                uint32_t rw = rw_ahead;
                uint32_t open = ~(rw | (rw >> 16)) & 0x7FFFu & (0x7FFFu << x);
                while (!open) {                                     // linear probe with wrap (Board::getRandomMove, Game.cpp:64-73)
                    y = (y == 14u) ? 0u : y + 1u;
                    rw = *lds_at(base + __umul24(y, sb));
                    open = ~(rw | (rw >> 16)) & 0x7FFFu;
                }
                x = static_cast<uint32_t>(__ffs(open)) - 1u;
                int first_row = __mul24(static_cast<int>(y), first_y) + first0, second_row = __mul24(static_cast<int>(y), second_y) + second0;
                asm volatile("" : "+v"(first_row), "+v"(second_row));             // (multiply-adds: see random_rollout_quads)
                const uint32_t first_at = static_cast<uint32_t>(__mul24(static_cast<int>(x), first_x) + first_row);
                const uint32_t second_at = static_cast<uint32_t>(static_cast<int>(__umul24(x, sb)) + second_row);
                const uint32_t stone = (j & 1) ? stone_odd : stone_even, halves = (j & 1) ? halves_odd : halves_even;
                const uint32_t first_old = *lds_at(first_at), second_old = *lds_at(second_at);
                const uint32_t next_row = *lds_at(base + __umul24(y_next, sb));
                const uint32_t put = stone << x;
                const uint32_t first_new = first_old | put, second_new = second_old | (stone << ((x & odd_mask) | (y & ~odd_mask)));
                *lds_at(first_at) = first_new; *lds_at(second_at) = second_new;
                rw_ahead = y_next == y ? (rw | put) : next_row;
                const uint32_t both = __builtin_amdgcn_perm(second_new, first_new, halves);          // the mover's halves; bits 15 and 31 are gaps
                const uint32_t b2 = both & (both >> 1), b4 = b2 & (b2 >> 2);
                uint32_t fives = b4 & (both >> 4);
                fives |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(fives), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]: the pair's other lane
                if (fives != 0u) won = 2u | static_cast<uint32_t>(j & 1);
                live = fives == 0u && (!kTie || static_cast<int>(8u * b) + j != last_ply);
            }
            if ((j & 3) == 3 && __ballot(live) == 0ull) return true;
        }
        return false;
    };
    for (uint32_t b = b_from; b < b_to; ++b) {
        const uint2 ahead = b + 1u < b_to ? fetch(b + 1u) : make_uint2(0u, 0u);      // (see random_rollout_quads_span)
        const bool over = static_cast<int>(8u * b) + 7 < no_tie_before ? play_block(std::false_type{}, b, ahead) : play_block(std::true_type{}, b, ahead);
        if (over) { state.live = false; state.won = won; return true; }
        cur = ahead;
    }
    state.live = live; state.won = won;
    return false;
}

template <class Fetch>
__device__ inline int random_rollout_pairs(uint32_t* lines, uint32_t stride, int to_move, int stones, int no_tie_before, Fetch fetch) {
    RolloutState state;
    (void)random_rollout_pairs_span(lines, stride, to_move, stones, no_tie_before, fetch, state, 0u, 30u);
    return state.winner(to_move);
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
