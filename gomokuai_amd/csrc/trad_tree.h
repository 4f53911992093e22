// trad_tree.h -- the "first child" search tree shared by the kernels whose select stage is RAVE::Select
// (core/lib/include/algorithms/MonteCarlo.hpp:149-152): K6 (trad_kernel.hip, TraditionalPolicy) and K8 (rave_kernel.hip,
// PoolRAVEPolicy).  One arena of `cap` nodes per game in HBM, the root at index 0, children of a node consecutive;
// what RAVE::BackPropogate's swaps change is kept as the node's position in its parent's order (`ord`) and the parent's
// record of its current first child (`front`).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace gmk {
namespace tree {

constexpr uint32_t kNoParent = 0xFFFFFFu;

constexpr uint32_t kStatusIdleSlot = 16u;        // continuous batching (gmk_trad_selfplay_run): the slot's games have run out, the searches skip it

struct TradHeader {                              // 64 B per game in HBM
    uint32_t n_nodes, init_acts, status, fresh;  // status: bit 0 node capacity reached, bit 1 evaluator error, bit 2 board-only revert met, bit 3 illegal step, bit 4 idle slot
                                                 // fresh: 1 = new root + evaluator sync, 2 = the tree was re-rooted (kept): evaluator sync only
    uint32_t playouts_done, root_black, pad0, pad1;
    unsigned long long evaluator_updates, pad2;
    uint32_t prof[4];                            // GMK_TRAD_PROFILE: shader clocks (>> 10) in select + evaluator moves, simulate + expand, backup, all
};
static_assert(sizeof(TradHeader) == 64, "TradHeader layout");

struct TradArena {
    uint2* stat;                                 // [n_games][cap] {visits, value bits}
    uint2* info;                                 // [n_games][cap] {parent | cell << 24, prior bits}
    uint32_t* link;                              // [n_games][cap] first child | children << 24
    uint2* front;                                // [n_games][cap] the child that is first in the CURRENT order: {id | cell << 24, its link word}
    uint8_t* ord;                                // [n_games][cap] the node's position in its parent's current child order
    uint2* amaf;                                 // [n_games][cap] {amaf_visits, amaf_value bits} (AMAFNode, MonteCarlo.hpp:113-122); null for K6
};

// What the device-resident self-play loop (gmk_trad_selfplay_run) hands its step kernel: the game a slot plays and the hand-over of a
// finished game's slot to the next unstarted game, the openings, and the records by GAME (not by slot).
struct TradSelfPlay {
    int32_t* slot_game;                          // [n_slots] the game (0 .. n_total-1) a slot plays, -1: none
    int32_t* next_game;                          // the next unstarted game
    int n_total;
    const uint8_t* open_moves;                   // [n_total][open_stride], may be null
    const int32_t* open_lens;                    // [n_total]
    int open_stride;
    uint32_t* game_ids;                          // [n_slots] = slot_game as the searches and the root noise key their streams
    uint8_t* rec_moves;                          // [n_total][225]
    int32_t* rec_lens;                           // [n_total]
    uint16_t* rec_visits;                        // [n_total][225][225] or null: root visit counts of every searched ply
    int8_t* rec_winner;                          // [n_total]
    int32_t* unfinished;                         // slots that still have a game after this step
    int32_t* overflow;                           // set when a search stopped at its node capacity
    // the persistent loop with the reference agent's per-move semantics (MCTS.cpp:129-147, 179-183): the chosen child's subtree is kept -- compacted
    // into the slot's OTHER arena, arena_stride nodes further on, inside the launch -- and Default::AddNoise runs before every search, drawn by the
    // wavefront itself from the counter-based sampler (include/gomoku_noise.h), keyed by (seed; first_game_id + game, stones, cell)
    int reuse;
    float noise_alpha, noise_epsilon;            // alpha == 0: no noise
    size_t arena_stride;                         // nodes between a slot's two arenas (0: one arena)
    uint32_t seed_lo, seed_hi, first_game_id;
};

// ---- wave-wide reductions without LDS round trips ----
// DPP row_shl by N lanes for 64-bit and unsigned values; a lane without a source inside its row of 16 keeps its own value
template <int N>
__device__ __forceinline__ double row_down_keep(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x100 + N, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(lo, lo, 0x100 + N, 0xF, 0xF, false));
}
template <int N>
__device__ __forceinline__ uint32_t row_down_keep(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x100 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ double lane_value(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// wave-wide maximum of doubles that are never NaN / minimum of unsigned values: DPP inside the rows, the four row leaders by readlane
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, row_down_keep<8>(v)); v = fmax(v, row_down_keep<4>(v)); v = fmax(v, row_down_keep<2>(v)); v = fmax(v, row_down_keep<1>(v));
    return fmax(fmax(lane_value(v, 0), lane_value(v, 16)), fmax(lane_value(v, 32), lane_value(v, 48)));
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
    v = min(v, row_down_keep<8>(v)); v = min(v, row_down_keep<4>(v)); v = min(v, row_down_keep<2>(v)); v = min(v, row_down_keep<1>(v));
    const uint32_t a = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 0)), b = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 16));
    const uint32_t c = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 32)), d = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 48));
    return min(min(a, b), min(c, d));
}

}  // namespace tree
}  // namespace gmk

struct gmk_trad {
    int n_games = 0, cap = 0;
    uint32_t* d_states = nullptr;
    uint2 *d_stat = nullptr, *d_info = nullptr;
    uint32_t* d_link = nullptr;
    uint2* d_front = nullptr;
    uint8_t* d_ord = nullptr;
    uint2* d_amaf = nullptr;                                                // allocated by the first gmk_trad_run_poolrave
    uint2 *d_stat2 = nullptr, *d_info2 = nullptr, *d_front2 = nullptr;      // second arena, allocated by the first gmk_trad_step
    uint32_t* d_link2 = nullptr;
    uint8_t* d_ord2 = nullptr;
    uint2* d_amaf2 = nullptr;
    int16_t* d_forced = nullptr;
    float* d_priors = nullptr;
    gmk::tree::TradHeader* d_hdr = nullptr;
    uint8_t* d_moves = nullptr;
    int32_t* d_lens = nullptr;
    std::vector<uint32_t> game_ids;                                         // the game a slot is playing, relative to the callers' first_game_id (default: the slot number)
    uint32_t* d_game_ids = nullptr;
    uint32_t* d_path_spill = nullptr;            // [n_games][kPathSpill] K6: the child ranges of path levels the LDS copy has no room for
    bool attr_set = false, attr_set_selfplay = false, positioned = false, second_arena = false;
    // gmk_trad_set_option
    int noise_sampler = 0;                       // GMK_NOISE_SAMPLER_STD: where Default::AddNoise draws from
    int lockstep = 0;
    // the persistent loop with kept subtrees wants a slot's two arenas a fixed distance apart: both halves of ONE allocation per array
    bool paired = false;
    uint2 *block_stat = nullptr, *block_info = nullptr, *block_front = nullptr;
    uint32_t* block_link = nullptr;
    uint8_t* block_ord = nullptr;
    size_t arena_stride() const { return (paired && d_stat2 > d_stat) ? static_cast<size_t>(d_stat2 - d_stat) : 0; }
    int policy = 0;                                                          // 0 not searched yet, 1 TraditionalPolicy (gmk_trad_run), 2 PoolRAVEPolicy (gmk_trad_run_poolrave): one per handle

    gmk::tree::TradArena arena() const { return {d_stat, d_info, d_link, d_front, d_ord, d_amaf}; }
    gmk::tree::TradArena arena2() const { return {d_stat2, d_info2, d_link2, d_front2, d_ord2, d_amaf2}; }
};
