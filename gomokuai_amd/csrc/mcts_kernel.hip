// mcts_kernel.hip -- K3: batched MCTS playouts (RandomPolicy) as one persistent launch on gfx950.
//
// Per game the loop is the reference's MCTS::playout (core/lib/src/MCTS.cpp:158-177) with
// Default::Select / Expand / BackPropogate (core/lib/include/algorithms/MonteCarlo.hpp:57-95) and
// RandomPolicy::averagedSimulate (core/lib/include/policies/Random.h:22-35); the numerics (f64 PUCB from
// f32 operands, f32 running mean, first-maximum tie break, children in ascending cell order) are kept so
// that visit counts and Q are bit-identical to the CPU restatement under the shared Philox stream.
//
// Mapping onto CDNA4
//   * one 64-lane wavefront (= one workgroup) owns G <= 4 games (gmk_mcts_create: as many as keep two wavefronts per SIMD busy; two for
//     BASELINE configs[2]'s 4 096 games) for the whole search: all `playouts` iterations run inside ONE launch, no host round trips; in the
//     persistent self-play form (SearchParams::persistent) the launch also steps the games and plays them to their end, turn by turn;
//   * tree phases (select / expand / backup) give each game a quarter-wave (16 lanes = one DPP row): the
//     <= 225 children of a node are scored 16 at a time with f64 PUCB (the bonus explore / (n + 1) from a per-level table for n < 16) and
//     reduced with row shuffles, four games proceed concurrently per wave so that their dependent HBM loads overlap;
//   * the simulate phase gives every ROLLOUT a lane: lane = (game, rollout); each lane plays its random
//     game on a private bit-board kept in LDS (column layout [row][lane]: conflict-free), draws come from
//     Philox4x32-10 keyed by (seed; game, playout, root stones << 8 | rollout, ply >> 3), eight 16-bit draws per block;
//   * the tree lives in HBM, one arena per game, structure-of-arrays: stats {visits u32, value f32} (8 B,
//     what select reads per child), link {first child << 8 | cell} and parent (4 B each).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "capi_common.h"
#include "board_device.h"
#include "philox.h"
#include "rollout_device.h"
#include "root_noise.h"
#include "noise_device.h"

namespace {

using gmk::five_through;
using gmk::run_of_five;
using namespace gmk::rollout;

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kMaxGamesPerBlock = 16;
constexpr int kPathCap = 32;           // deeper descents fall back to walking parent[] with loads
constexpr int kCellBlocks = 30;        // Philox blocks of a rollout: 29 cover 225 plies, one more is read ahead
#ifndef GMK_STAGE_BLOCKS
#define GMK_STAGE_BLOCKS 16
#endif
constexpr int kStageBlocks = GMK_STAGE_BLOCKS;       // ... of which the first sixteen (128 plies; 12 .. 18 measure within 1 %, 16 best: profiles/r04_k3_stage_blocks.txt) are generated for every rollout, the others for those still running then

struct GameHeader {                 // 128 B per game, in HBM
    uint32_t rows[16];              // root position: black | white << 16 per row
    uint32_t root;                  // node index of the root inside the arena
    uint32_t n_nodes;               // nodes in use (MCTS::m_size)
    uint32_t stones;                // stones on the root board (= Policy::m_initActs)
    uint32_t last_move;             // cell of the last move, 255 if none
    uint32_t game_id;               // global game id (RNG counter word 0)
    uint32_t status;                // bit 0: the game is over, bit 1: an arena of this slot filled up (sticky over a slot's games), bit 2: illegal move requested
    uint64_t alg_bytes;             // algorithmic tree bytes of the last run
    uint32_t playouts_done;         // playouts already run from this root (RNG counter word 1 continues across launches)
    uint32_t noise;                 // 1: the root's children take their priors from root_prior[] (Default::AddNoise ran)
    uint32_t root_expanded;         // scratch for gmk_mcts_add_root_noise
    uint32_t pad[4];                // (diagnostic build: clock sums of the four phases)
    uint32_t arena;                 // the persistent self-play loop with kept subtrees: which of the game's two arenas holds its tree (0 / 1); 0 everywhere else
};
static_assert(sizeof(GameHeader) == 128, "GameHeader layout");

// Continuous batching for whole-game self-play (gmk_selfplay_run): the handle's games are SLOTS; slot g plays game slot_game[g] of
// n_total, its records go to that game's rows, and when the game ends the slot takes the next game nobody has started (a counter in
// device memory): its opening position becomes the slot's root, its global id the slot's random-number key.  So the search
// launches stay full until fewer games than slots remain, instead of waiting for the longest game of a fixed batch.
struct SlotRefill {
    int32_t* slot_game;             // [n_slots] game played by each slot, -1 = none (null: slot g plays game g and is not refilled)
    int32_t* next_game;             // [1] first game not started yet
    int n_total;
    const uint8_t* open_moves;      // [n_total][open_stride] opening moves (black first), may be null
    const int32_t* open_lens;       // [n_total] (<= 8: an opening cannot be a finished game)
    int open_stride;
    uint32_t first_game_id;
};

struct SearchParams {
    double c_puct;
    uint32_t seed_lo, seed_hi;
    int c_rollouts;
    int games_per_block;
    int node_capacity;
    int n_games;
    int playouts;
    int profile;                    // GMK_MCTS_PROFILE=1: per-phase shader-clock sums into GameHeader::pad (diagnostic runs only)
    const float* value_table;       // float(double(sum) / double(c_rollouts)) at index sum + c_rollouts (Random.h:30-33): owned by the HANDLE, since
                                    // handles with different c_rollouts run side by side (supervisor against candidate)
    // The persistent self-play loop (gmk_selfplay_run without kept subtrees and root noise): the kernel is a loop of TURNS -- the search of its
    // games' current roots, then for each of them the step mcts_advance_kernel does (the move, the record, the end-of-game check, a new root, a
    // finished game's slot taking the next unstarted game) -- until none of the wavefront's slots has a game left: ONE launch, no wavefront
    // waits for another's search.  The records below are the step's outputs; a game's record does not depend on where or when it was played
    // (its random streams are keyed by its global id), so this loop and the lock-step one write the same bytes.
    int persistent;
    uint8_t* rec_moves;
    uint16_t* rec_visits;
    int32_t* rec_lens;
    int8_t* rec_winner;
    int32_t* unfinished;            // (written, never read, in this mode)
    SlotRefill slots;
    // ... with the reference agent's per-move semantics (agents/mcts.py:17-21, MCTS.cpp:129-147, 179-183): the chosen child's subtree is kept --
    // compacted into the game's OTHER arena (arena_stride nodes further on; GameHeader::arena says which one is live), per game, inside the launch --
    // and Default::AddNoise runs before every search, drawn by the wavefront itself from the counter-based sampler of include/gomoku_noise.h
    int reuse;
    float noise_alpha, noise_epsilon;   // alpha == 0: no noise
    size_t arena_stride;                // nodes between a game's two arenas (0: there is one arena)
};

__device__ bool advance_one(int g, int lane, GameHeader* headers, uint2* stats, uint32_t* link, uint32_t* parent, uint2* stats2, uint32_t* link2,
                            uint32_t* parent2, size_t cap, uint8_t* rec_moves, uint16_t* rec_visits, int32_t* rec_lens, int8_t* rec_winner,
                            int32_t* unfinished, int reuse, int flip_all, const int16_t* forced, const SlotRefill& slots);
__device__ void root_noise_one(int g, int lane, GameHeader* headers, const uint32_t* link, float* root_prior, size_t cap, size_t arena_stride,
                               float alpha, float epsilon, uint32_t seed_lo, uint32_t seed_hi);

__constant__ float c_prior[226];          // 1.0f / float(n) evaluated on the host (MonteCarlo.hpp:50-55)

// Quarter-wave (one DPP row of 16 lanes) primitives for the select / expand phases: DPP moves cost an ALU cycle each where
// a ds_bpermute shuffle is a ~100-cycle LDS round trip, and this kernel runs one wave per SIMD with nothing to hide it.
template <int N>
__device__ __forceinline__ void row_best(double& best, int& best_i) {   // rotate the row right by N lanes, keep the better
    const int lo = __double2loint(best), hi = __double2hiint(best);
    const double o = __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x120 + N, 0xF, 0xF, false),
                                      __builtin_amdgcn_update_dpp(lo, lo, 0x120 + N, 0xF, 0xF, false));
    const int oi = __builtin_amdgcn_update_dpp(best_i, best_i, 0x120 + N, 0xF, 0xF, false);
    const bool take = o > best || (o == best && oi < best_i);
    best = take ? o : best;
    best_i = take ? oi : best_i;
}

// the better of the two DPP rows of a half-wavefront, in all of its lanes: v_permlane16_swap hands every lane the even row's and the odd
// row's value of its pair of rows (gfx950)
__device__ __forceinline__ void rows_best(double& best, int& best_i) {
    auto swap = [](unsigned v) { return __builtin_amdgcn_permlane16_swap(v, v, false, false); };
    const auto lo = swap(static_cast<unsigned>(__double2loint(best))), hi = swap(static_cast<unsigned>(__double2hiint(best))), ix = swap(static_cast<unsigned>(best_i));
    const double even = __hiloint2double(static_cast<int>(hi[0]), static_cast<int>(lo[0])), odd = __hiloint2double(static_cast<int>(hi[1]), static_cast<int>(lo[1]));
    const int even_i = static_cast<int>(ix[0]), odd_i = static_cast<int>(ix[1]);
    const bool take_odd = odd > even || (odd == even && odd_i < even_i);
    best = take_odd ? odd : even;
    best_i = take_odd ? odd_i : even_i;
}

__device__ __forceinline__ int row_scan(int v) {                        // inclusive prefix sum over the row (row_shr, zero fill)
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    return v;
}

// kTurns = false: one search (gmk_mcts_run); true: the persistent self-play loop, search after search with the step (and the root noise) in between.
// Two instantiations, so that the bare search does not carry the registers of the turn loop through its playouts.
// kStaged: the rollouts' draws are generated in two stages (the pairs form: wavefronts of three or four games); a separate instantiation, so that
// the other forms -- BASELINE configs[2] runs on quads -- are the code they were, to the register.
template <bool kTurns, bool kStaged>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))      // two wavefronts per SIMD: at most 256 registers
void mcts_playouts_kernel(GameHeader* __restrict__ headers, uint2* __restrict__ stats, uint32_t* __restrict__ link,
                          uint32_t* __restrict__ parent, float* root_prior /* written by this kernel's own root_noise_one between two turns: neither const nor restrict */,
                          SearchParams prm) {
    // dynamic LDS, sized by the games G and rollouts R of the block (lds_words() below): the rollout positions as line words
    // [word][rollout lane], the rollouts' random cells [block of eight plies][rollout lane], the games' leaf positions, their paths
    extern __shared__ uint32_t s_lane_lines[];
    __shared__ uint32_t s_cur[kMaxGamesPerBlock], s_ply[kMaxGamesPerBlock], s_last[kMaxGamesPerBlock];
    __shared__ uint32_t s_need[kMaxGamesPerBlock];               // 1: leaf needs rollouts, 0: terminal
    __shared__ float s_term_value[kMaxGamesPerBlock];
    __shared__ int s_sum[kMaxGamesPerBlock];
    __shared__ uint32_t s_nodes[kMaxGamesPerBlock], s_status[kMaxGamesPerBlock];
    __shared__ unsigned long long s_bytes[kMaxGamesPerBlock];
    __shared__ uint32_t s_active[kMaxGamesPerBlock];
    // the descent path with the statistics select already loaded: backup needs no loads (MonteCarlo.hpp:90-95)
    __shared__ uint32_t s_depth[kMaxGamesPerBlock];             // 0: the game is over (status bit 0), nothing to search
    // the root as the launch found it (header fields) and as the playouts leave it (statistics, link word), and the leaf's link word:
    // a playout then starts and expands without waiting for a load of its own
    __shared__ uint32_t s_root[kMaxGamesPerBlock], s_root_stones[kMaxGamesPerBlock], s_root_last[kMaxGamesPerBlock], s_root_noise[kMaxGamesPerBlock];
    __shared__ uint32_t s_root_rows[kMaxGamesPerBlock][16];
    __shared__ uint2 s_root_stats[kMaxGamesPerBlock];
    __shared__ uint32_t s_root_link[kMaxGamesPerBlock], s_leaf_link[kMaxGamesPerBlock];
    __shared__ uint32_t s_rng_game[kMaxGamesPerBlock], s_rng_playout[kMaxGamesPerBlock], s_rng_stones[kMaxGamesPerBlock];   // Philox counter words of a game's rollouts
    __shared__ uint32_t s_running[64], s_inverse[65];             // the rollouts that go into the second stage; 2^32 / n rounded up, n = 1 .. 64
    __shared__ size_t s_arena[kMaxGamesPerBlock];                // first node of the game's LIVE arena (the persistent loop with kept subtrees flips it per game)

    const int lane = threadIdx.x, quarter = lane >> 4, l16 = lane & 15;
    if constexpr (kStaged) s_inverse[lane + 1] = lane == 0 ? 0u : 0xFFFFFFFFu / static_cast<uint32_t>(lane + 1) + 1u;      // (n = 1 takes the dividend as it is; read after the turn's first barrier)
    const int G = prm.games_per_block, R = prm.c_rollouts;
    const int game0 = blockIdx.x * G;
    const int games_here = min(G, prm.n_games - game0);
    const int rounds = (G + 3) >> 2;
    const size_t cap = static_cast<size_t>(prm.node_capacity);

    for (;;) {                                                  // one turn = one search of the wavefront's games (the only turn unless prm.persistent)
    if (lane < kMaxGamesPerBlock) {
        const bool ok = lane < games_here;
        s_nodes[lane] = ok ? headers[game0 + lane].n_nodes : 0;
        s_status[lane] = ok ? headers[game0 + lane].status : 0;
        s_bytes[lane] = 0;
        s_active[lane] = (ok && !(headers[game0 + lane].status & 1u)) ? 1u : 0u;
        s_rng_game[lane] = ok ? headers[game0 + lane].game_id : 0u;
        s_rng_playout[lane] = ok ? headers[game0 + lane].playouts_done : 0u;
        s_rng_stones[lane] = ok ? headers[game0 + lane].stones << 8 : 0u;
        const uint32_t root = ok ? headers[game0 + lane].root : 0u;
        const size_t arena = static_cast<size_t>(game0 + lane) * cap + ((kTurns && ok && headers[game0 + lane].arena) ? prm.arena_stride : 0);
        s_arena[lane] = arena;
        s_root[lane] = root;
        s_root_stones[lane] = ok ? headers[game0 + lane].stones : 0u;
        s_root_last[lane] = ok ? headers[game0 + lane].last_move : 255u;
        s_root_noise[lane] = ok ? headers[game0 + lane].noise : 0u;
        s_root_stats[lane] = ok ? stats[arena + root] : make_uint2(0u, 0u);
        s_root_link[lane] = ok ? link[arena + root] : 0u;
    }
    for (int i = lane; i < kMaxGamesPerBlock * 16; i += 64) s_root_rows[i >> 4][i & 15] = (i >> 4) < games_here ? headers[game0 + (i >> 4)].rows[i & 15] : 0u;
    __syncthreads();
    const int n_rollout_lanes = G * R;
    uint2* const s_cells = reinterpret_cast<uint2*>(s_lane_lines + kLineWords * n_rollout_lanes);
    uint32_t (*const s_leaf)[kLineWords] = reinterpret_cast<uint32_t (*)[kLineWords]>(s_lane_lines + (kLineWords + 2 * kCellBlocks) * n_rollout_lanes);   // leaf position of each game as line words (rows first)
    uint32_t (*const s_path_node)[kPathCap] = reinterpret_cast<uint32_t (*)[kPathCap]>(&s_leaf[G][0]);
    uint32_t (*const s_path_visits)[kPathCap] = s_path_node + G;
    float (*const s_path_value)[kPathCap] = reinterpret_cast<float (*)[kPathCap]>(s_path_visits + G);
    // t / d = umulhi(t, 2^32 / d rounded up) for the small t here; the constant does not exist for d == 1 (it would be 2^32: it wraps to 0 and
    // every quotient with it), so a divisor of one -- RandomPolicy(c, 1), or one game with one rollout -- takes the dividend as it is
    const uint32_t inv_lanes = 0xFFFFFFFFu / static_cast<uint32_t>(n_rollout_lanes) + 1u, inv_r = 0xFFFFFFFFu / static_cast<uint32_t>(R) + 1u;
    auto div_lanes = [&](uint32_t t) -> uint32_t { return n_rollout_lanes == 1 ? t : __umulhi(t, inv_lanes); };
    auto div_r = [&](uint32_t t) -> uint32_t { return R == 1 ? t : __umulhi(t, inv_r); };

    unsigned long long prof[4] = {0, 0, 0, 0}, t_mark = (gmk::kProfileBuild && prm.profile) ? __builtin_amdgcn_s_memtime() : 0ull;   // diagnostic build only
    for (int playout = 0; playout < prm.playouts; ++playout) {
        if (lane < kMaxGamesPerBlock) s_sum[lane] = 0;

        // ---- select: descend to a leaf (MCTS.cpp:160-163), a quarter-wave per game -- or half a wavefront per game when the wavefront has two
        //      games at most (W = 32: eight children a lane instead of fifteen; the board rows and the bonus table are kept by both of the
        //      half's DPP rows, the same values twice.  All 64 lanes for a wavefront's only game: 3 % on a one-game search, not kept) ----
        auto descend = [&](auto width_tag, const int gs) {
            constexpr int W = decltype(width_tag)::value, kPerLane = (225 + W - 1) / W;
            const int lw = lane & (W - 1);
            {
                const size_t base = s_arena[gs];
                uint32_t row = s_root_rows[gs][l16];                 // lane y holds row y of the board
                const uint32_t root = s_root[gs];
                uint32_t cur = root, ply = s_root_stones[gs], last = s_root_last[gs];
                unsigned long long bytes = 0;
                uint2 cur_stats = s_root_stats[gs];
                uint32_t cur_link = s_root_link[gs];
                uint32_t depth = 0;
                for (;;) {
                    if (l16 == 0 && depth < kPathCap) { s_path_node[gs][depth] = cur; s_path_visits[gs][depth] = cur_stats.x; s_path_value[gs][depth] = __uint_as_float(cur_stats.y); }
                    const uint32_t first = cur_link >> 8;
                    if (!first) break;                               // Node::isLeaf
                    const int n_child = 225 - static_cast<int>(ply);
                    const double n_parent = static_cast<double>(cur_stats.x);
                    const double root_n = sqrt(n_parent);
                    const double explore = prm.c_puct * static_cast<double>(c_prior[n_child]) * root_n;   // MonteCarlo.hpp:23-28
                    const bool noisy = s_root_noise[gs] && cur == root;  // only the root's children ever carry non-uniform priors
                    // Default::Select starts from max_score = -1.0 with index 0 and takes strictly greater scores; every
                    // score is >= -1 (Q in [-1,1], bonus >= 0), so that equals "first maximum", which -2.0 yields lane-locally
                    double best = -2.0;
                    int best_i = 0;
                    uint2 best_st = make_uint2(0u, 0u);
                    uint32_t best_link = 0u;
                    // all of this lane's children (i = l16, l16 + 16, ...: at most 15) are fetched before any is scored, so
                    // the loads overlap instead of queueing behind each other's f64 divide; their link words come with them:
                    // the chosen child's is then at hand and the next level starts without a dependent load of its own
                    uint2 st[kPerLane];
                    uint32_t lk[kPerLane];
#pragma unroll
                    for (int j = 0; j < kPerLane; ++j) {
                        const int i = lw + W * j;
                        st[j] = (i < n_child) ? stats[base + first + i] : make_uint2(0u, 0u);
                        lk[j] = (i < n_child) ? link[base + first + i] : 0u;
                    }
                    // the scores of all fifteen first, without a branch: fifteen independent f64 divide chains that the scheduler
                    // interleaves (inside `if (i < n_child)` they ran one behind the other); a slot without a child scores explore / 1
                    double score[kPerLane];
                    if (__ballot(noisy) == 0ull) {
                        // explore / (n + 1) depends on the child's visit count only, and nearly all children of a node have few visits: lane
                        // l16 of the game's quarter computes the quotient for n = l16 ONCE, a child with n < 16 fetches it from that lane (two
                        // ds_bpermute), the few others divide as before -- the same division on the same operands either way, bit for bit
                        const double by_visits = explore / static_cast<double>(l16 + 1);
                        const int quarter_base = (lane & 48) << 2;          // byte address of the quarter's lane 0 for ds_bpermute
                        const int lo = __double2loint(by_visits), hi = __double2hiint(by_visits);
#pragma unroll
                        for (int j = 0; j < kPerLane; ++j) {
                            const uint32_t n_j = st[j].x;
                            const int from = quarter_base + 4 * static_cast<int>(min(n_j, 15u));
                            double bonus = __hiloint2double(__builtin_amdgcn_ds_bpermute(from, hi), __builtin_amdgcn_ds_bpermute(from, lo));
                            if (n_j >= 16u) bonus = explore / static_cast<double>(n_j + 1u);
                            score[j] = static_cast<double>(__uint_as_float(st[j].y)) + bonus;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < kPerLane; ++j) {
                            const int i = lw + W * j;
                            double bonus = explore;
                            if (noisy && i < n_child) bonus = prm.c_puct * static_cast<double>(root_prior[static_cast<size_t>(game0 + gs) * 225 + i]) * root_n;
                            score[j] = static_cast<double>(__uint_as_float(st[j].y)) + bonus / static_cast<double>(st[j].x + 1u);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < kPerLane; ++j) {
                        const int i = lw + W * j;
                        const bool take = i < n_child && score[j] > best;
                        best = take ? score[j] : best;
                        best_i = take ? i : best_i;
                        best_st.x = take ? st[j].x : best_st.x;
                        best_st.y = take ? st[j].y : best_st.y;
                        best_link = take ? lk[j] : best_link;
                    }
                    // first maximum wins (strict > in ascending order): (score, -index) is a total order, so rotating the
                    // row of 16 lanes by 8, 4, 2, 1 leaves the same winner in every lane - DPP moves, no LDS permute
                    row_best<8>(best, best_i); row_best<4>(best, best_i); row_best<2>(best, best_i); row_best<1>(best, best_i);
                    if constexpr (W == 32) rows_best(best, best_i);           // ... and the better of the half's two rows
                    bytes += static_cast<unsigned long long>(n_child) * 8ull;
                    cur = first + static_cast<uint32_t>(best_i);
                    // the statistics of the chosen child sit in the lane that scored it (children i = lane mod W)
                    cur_stats.x = __shfl(best_st.x, best_i & (W - 1), W);
                    cur_stats.y = __shfl(best_st.y, best_i & (W - 1), W);
                    cur_link = __shfl(best_link, best_i & (W - 1), W);
                    // its cell = the best_i-th empty cell of the current position in ascending order (children are created
                    // that way, MonteCarlo.hpp:71-80): no dependent load of link[]
                    {
                        const uint32_t open = (l16 < 15) ? (~(row | (row >> 16)) & 0x7FFFu) : 0u;
                        const int mine = __popc(open);
                        const int incl = row_scan(mine);
                        const int skip = best_i - (incl - mine);
                        const bool owner = skip >= 0 && skip < mine;
                        uint32_t m = open;
                        for (int t = 0; t < skip && owner; ++t) m &= m - 1u;
                        const uint32_t cell = static_cast<uint32_t>(l16) * 15u + static_cast<uint32_t>(__ffs(m) - 1);
                        const unsigned long long who = __ballot(owner) >> (quarter * 16);
                        last = __shfl(cell, __ffsll(static_cast<long long>(who & 0xFFFFull)) - 1, 16);
                        if (owner) row |= 1u << ((cell % 15u) + ((ply & 1u) ? 16u : 0u));       // Policy::applyMove, no victory check
                    }
                    ++ply;
                    ++depth;
                }
                // leaf position -> line words: rows as they are, one OR per stone into column / diagonal / anti-diagonal
                for (int w = l16; w < kLineWords; w += 16) s_leaf[gs][w] = 0u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (l16 < 15) {
                    s_leaf[gs][l16] = row;
                    for (uint32_t m = (row | (row >> 16)) & 0x7FFFu; m; m &= m - 1u) {
                        const int x = __ffs(m) - 1, y = l16;
                        const uint32_t cb = ((row >> x) & 1u) ? 0u : 16u;
                        atomicOr(&s_leaf[gs][kColBase + x], 1u << (y + cb));
                        atomicOr(&s_leaf[gs][kDiagBase + x - y + 14], 1u << (x + cb));
                        atomicOr(&s_leaf[gs][kAntiBase + x + y], 1u << (x + cb));
                    }
                }
                if (lw == 0) { s_cur[gs] = cur; s_ply[gs] = ply; s_last[gs] = last; s_bytes[gs] += bytes; s_depth[gs] = depth; s_leaf_link[gs] = cur_link; }
            }
        };
        if (G <= 2) {
            if ((lane >> 5) < games_here && s_active[lane >> 5]) descend(std::integral_constant<int, 32>{}, lane >> 5);
        } else {
            for (int round = 0; round < rounds; ++round) {
                const int gs = round * 4 + quarter;
                if (gs < games_here && s_active[gs]) descend(std::integral_constant<int, 16>{}, gs);
            }
        }
        __syncthreads();
        if (gmk::kProfileBuild && prm.profile) { const unsigned long long t = __builtin_amdgcn_s_memtime(); prof[0] += t - t_mark; t_mark = t; }

        // ---- terminal test at the leaf (Policy::checkGameEnd, MCTS.cpp:166) ----
        if (lane < games_here && s_active[lane]) {
            const uint32_t ply = s_ply[lane], last = s_last[lane];
            bool five = false;
            if (ply > 0 && last < 225u) five = five_on_lines<1>(s_leaf[lane], static_cast<int>(last % 15u), static_cast<int>(last / 15u), (ply & 1u) ? 0 : 16);
            const bool over = five || ply == 225u;
            s_need[lane] = over ? 0u : 1u;
            s_term_value[lane] = five ? 1.0f : 0.0f;                 // CalcScore(node->player, winner): the mover won, or a tie
        }
        __syncthreads();
        if (gmk::kProfileBuild && prm.profile) { const unsigned long long t = __builtin_amdgcn_s_memtime(); prof[1] += t - t_mark; t_mark = t; }

        // ---- simulate (Random.h:22-35).  First the random cells of the rollouts, with all 64 lanes: one Philox block = eight plies per task,
        //      block-major, so that the rollout loop below only plays (the generator is ~150 instructions a block, and a third of the lanes
        //      roll out) -- in TWO STAGES where the rollouts can stop and go on (quads, pairs): blocks 0 .. kStageBlocks - 1 for every rollout,
        //      the others only for the rollouts that are still running after those 96 plies (a rollout ends after ~100 of up to 220 plies:
        //      round 3 generated 27 blocks per rollout and played 13 of them on average) ----
        // Measured (tools/mcts_time.py): with 20 rollouts per wavefront (four games, pairs) 9 generator passes become 7: 75.2 -> 73.7 ms at 16 384 games;
        // with 10 (two games, quads: BASELINE configs[2]) 5 become 4 and the second stage's bookkeeping costs more than that pass: 28.0 -> 28.5 ms.
        // So: staged for the pairs form only; quads and the one-lane form generate everything and play straight through.
        constexpr bool staged = kStaged;                            // (the host instantiates it for 2 * rollouts <= 64 < 4 * rollouts)
        const int first_blocks = staged ? kStageBlocks : 29;
        for (int t = lane; t < first_blocks * n_rollout_lanes; t += 64) {
            const uint32_t b = div_lanes(static_cast<uint32_t>(t)), rl = static_cast<uint32_t>(t) - b * static_cast<uint32_t>(n_rollout_lanes);
            const uint32_t gs = div_r(rl), r = rl - gs * static_cast<uint32_t>(R);
            if (static_cast<int>(gs) < games_here && s_active[gs] && s_need[gs] && 8u * b < 225u - s_ply[gs])
                s_cells[b * static_cast<uint32_t>(n_rollout_lanes) + rl] = rollout_cells(s_rng_game[gs], s_rng_playout[gs] + static_cast<uint32_t>(playout), s_rng_stones[gs] | r,
                                                                                       b, prm.seed_lo, prm.seed_hi);
        }
        __syncthreads();
        // the rollouts: four lanes each (random_rollout_quads) while the wavefront has the lanes for it, else two (random_rollout_pairs), else one
        {
            int no_tie_before = 224;                                 // the first ply at which a board of this wavefront can fill up
            for (int g = 0; g < games_here; ++g)
                if (s_active[g] && s_need[g]) no_tie_before = min(no_tie_before, 224 - static_cast<int>(s_ply[g]));
            no_tie_before = __builtin_amdgcn_readfirstlane(no_tie_before);
            const int share = 4 * n_rollout_lanes <= 64 ? 2 : 2 * n_rollout_lanes <= 64 ? 1 : 0;       // log2 of the lanes per rollout
            const int rl = lane >> share, part = lane & ((1 << share) - 1), parts = 1 << share;      // the rollout this lane works on, and its share of it
            const int gs = static_cast<int>(div_r(static_cast<uint32_t>(rl)));
            const bool rolls = rl < n_rollout_lanes && gs < games_here && s_active[gs] && s_need[gs];
            RolloutState state;
            state.live = rolls;
            if (rolls) {
                // the leaf's line words into the rollout's position, 23 reads in flight at a time (one by one each copy is a
                // round trip: the compiler cannot tell that the two regions are apart); the lanes of a rollout copy a share each
                static_assert(kLineWords == 4 * 23, "copy batches");
                for (int w0 = 23 * part; w0 < kLineWords; w0 += 23 * parts) {
                    uint32_t t[23];
    #pragma unroll
                    for (int i = 0; i < 23; ++i) t[i] = s_leaf[gs][w0 + i];
    #pragma unroll
                    for (int i = 0; i < 23; ++i) s_lane_lines[(w0 + i) * n_rollout_lanes + rl] = t[i];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t ply = s_ply[gs];
                const int init_player = (ply & 1u) ? -1 : 1;         // black moves on even stone counts
                const uint2* my_cells = s_cells + rl;
                auto fetch = [&](uint32_t b) { return my_cells[b * static_cast<uint32_t>(n_rollout_lanes)]; };
                uint32_t* const position = &s_lane_lines[rl];
                const uint32_t stride = static_cast<uint32_t>(n_rollout_lanes);
                if constexpr (kStaged) {
                    (void)random_rollout_pairs_span(position, stride, init_player, static_cast<int>(ply), no_tie_before, fetch, state, 0u, static_cast<uint32_t>(kStageBlocks));
                } else {
                    // (quads and the one-lane form: every block is there, the rollout plays straight through -- the code of round 3, untouched:
                    // BASELINE configs[2] runs here, and the resumable form costs it 3 %)
                    const int winner = share == 2 ? random_rollout_quads(position, stride, init_player, static_cast<int>(ply), no_tie_before, fetch)
                                     : share == 1 ? random_rollout_pairs(position, stride, init_player, static_cast<int>(ply), no_tie_before, fetch)
                                                  : random_rollout_blocks(position, stride, init_player, static_cast<int>(ply), no_tie_before, fetch);
                    if (part == 0) atomicAdd(&s_sum[gs], init_player * winner);          // CalcScore(init_player, winner)
                    state.live = false;
                }
            }
            // second stage (pairs): the blocks from kStageBlocks on, for the rollouts that are still running (wave-uniform: most playouts have some)
            const unsigned long long running = kStaged ? __ballot(state.live && part == 0) : 0ull;
            if (kStaged && running) {
                const uint32_t n_running = static_cast<uint32_t>(__popcll(running));
                if (state.live && part == 0)
                    s_running[__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(running >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(running), 0u))] = static_cast<uint32_t>(rl);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t inv_running = s_inverse[n_running];      // t / n_running = umulhi(t, 2^32 / n_running rounded up) for the small t here
                for (uint32_t t = static_cast<uint32_t>(lane); t < static_cast<uint32_t>(29 - kStageBlocks) * n_running; t += 64u) {
                    const uint32_t k = n_running == 1u ? t : __umulhi(t, inv_running), trl = s_running[t - k * n_running], b = static_cast<uint32_t>(kStageBlocks) + k;
                    const uint32_t tg = div_r(trl), r = trl - tg * static_cast<uint32_t>(R);
                    if (8u * b < 225u - s_ply[tg])
                        s_cells[b * static_cast<uint32_t>(n_rollout_lanes) + trl] = rollout_cells(s_rng_game[tg], s_rng_playout[tg] + static_cast<uint32_t>(playout), s_rng_stones[tg] | r,
                                                                                                b, prm.seed_lo, prm.seed_hi);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (state.live) {
                    const uint32_t ply = s_ply[gs];
                    const int init_player = (ply & 1u) ? -1 : 1;
                    const uint2* my_cells = s_cells + rl;
                    auto fetch = [&](uint32_t b) { return my_cells[b * static_cast<uint32_t>(n_rollout_lanes)]; };
                    uint32_t* const position = &s_lane_lines[rl];
                    const uint32_t stride = static_cast<uint32_t>(n_rollout_lanes);
                    (void)random_rollout_pairs_span(position, stride, init_player, static_cast<int>(ply), no_tie_before, fetch, state, static_cast<uint32_t>(kStageBlocks), 30u);
                }
            }
            if (kStaged && rolls && part == 0) {
                const int init_player = (s_ply[gs] & 1u) ? -1 : 1;
                atomicAdd(&s_sum[gs], init_player * state.winner(init_player));          // CalcScore(init_player, winner)
            }
        }
        __syncthreads();
        if (gmk::kProfileBuild && prm.profile) { const unsigned long long t = __builtin_amdgcn_s_memtime(); prof[2] += t - t_mark; t_mark = t; }

        // ---- expand + backup: quarter-wave per game (MonteCarlo.hpp:71-95) ----
        for (int round = 0; round < rounds; ++round) {
            const int gs = round * 4 + quarter;
            if (gs < games_here && s_active[gs]) {
                const size_t base = s_arena[gs];
                const uint32_t cur = s_cur[gs], ply = s_ply[gs];
                float value;
                unsigned long long bytes = 0;
                if (s_need[gs]) {
                    const float state_value = prm.value_table[s_sum[gs] + R];
                    value = -state_value;                            // MCTS.cpp:169
                    const uint32_t rw = s_leaf[gs][l16];
                    const uint32_t open = (l16 < 15) ? (~(rw | (rw >> 16)) & 0x7FFFu) : 0u;
                    const int mine = __popc(open);
                    const int before = row_scan(mine) - mine;         // exclusive scan over the 16 lanes
                    const uint32_t n_child = 225u - ply;
                    const uint32_t first = s_nodes[gs];
                    if (static_cast<size_t>(first) + n_child <= cap) {
                        uint32_t idx = first + static_cast<uint32_t>(before);
                        for (uint32_t m = open; m; m &= m - 1u, ++idx) {
                            const uint32_t cell = static_cast<uint32_t>(l16) * 15u + static_cast<uint32_t>(__ffs(m) - 1);
                            stats[base + idx] = make_uint2(0u, 0u);
                            link[base + idx] = cell;
                            parent[base + idx] = cur;
                        }
                        if (l16 == 0) {
                            const uint32_t expanded = (first << 8) | (s_leaf_link[gs] & 0xFFu);
                            link[base + cur] = expanded;
                            if (cur == s_root[gs]) s_root_link[gs] = expanded;
                            s_nodes[gs] = first + n_child;
                        }
                        bytes += static_cast<unsigned long long>(n_child) * 16ull;
                    } else if (l16 == 0) {
                        s_status[gs] |= 2u;                          // arena full: the leaf stays a leaf
                    }
                } else {
                    value = s_term_value[gs];                        // MCTS.cpp:172
                }
                // Default::BackPropogate: level d of the path gets value * (-1)^(depth - d); the old statistics came with
                // select, so every level is one independent store (one lane per level)
                const uint32_t depth = s_depth[gs];
                if (depth < kPathCap) {
                    for (uint32_t d = l16; d <= depth; d += 16) {
                        const uint32_t visits = s_path_visits[gs][d] + 1u;
                        float q = s_path_value[gs][d];
                        const float v = ((depth - d) & 1u) ? -value : value;
                        q += (v - q) / static_cast<float>(visits);
                        stats[base + s_path_node[gs][d]] = make_uint2(visits, __float_as_uint(q));
                        if (d == 0u) s_root_stats[gs] = make_uint2(visits, __float_as_uint(q));          // level 0 of the path is the root
                    }
                    bytes += 16ull * (depth + 1u);
                } else if (l16 == 0) {                               // very deep path: walk parent[] (loads)
                    for (uint32_t node = cur; node != kNone; node = parent[base + node], value = -value) {
                        uint2 st = stats[base + node];
                        st.x += 1u;
                        float q = __uint_as_float(st.y);
                        q += (value - q) / static_cast<float>(st.x);
                        st.y = __float_as_uint(q);
                        stats[base + node] = st;
                        if (node == s_root[gs]) s_root_stats[gs] = st;
                        bytes += 16ull;
                    }
                }
                if (l16 == 0) {
                    s_bytes[gs] += bytes;
                }
            }
        }
        __syncthreads();
        if (gmk::kProfileBuild && prm.profile) { const unsigned long long t = __builtin_amdgcn_s_memtime(); prof[3] += t - t_mark; t_mark = t; }
    }

    if (lane < games_here) {
        headers[game0 + lane].n_nodes = s_nodes[lane];
        headers[game0 + lane].status = s_status[lane];
        headers[game0 + lane].alg_bytes = s_bytes[lane];
        if (s_active[lane]) headers[game0 + lane].playouts_done += static_cast<uint32_t>(prm.playouts);
        if (gmk::kProfileBuild && prm.profile) for (int k = 0; k < 4; ++k) headers[game0 + lane].pad[k] = static_cast<uint32_t>(prof[k] >> 10);
    }
    if (!kTurns) break;
    // the step of every game of this wavefront (headers and trees were written by other lanes than the ones that read them now, and the
    // next turn reads what the step writes: an agent-scope release / acquire pair either side)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    {
        // (opaque copies: what the step derives from the kernel's arguments is computed here, not hoisted out of the turn loop and carried
        // through the playouts in registers -- the search runs two wavefronts per SIMD on 256 registers and has none to spare)
        GameHeader* h_ = headers; uint2* st_ = stats; uint32_t* lk_ = link; uint32_t* pa_ = parent;
        int first_ = game0, lane_ = lane;
        asm volatile("" : "+s"(h_), "+s"(st_), "+s"(lk_), "+s"(pa_), "+s"(first_), "+v"(lane_));
#pragma unroll 1
        for (int g = 0; g < games_here; ++g) {
            // the game's live arena and its other one (one and the same without kept subtrees: arena_stride = 0)
            const size_t live = h_[first_ + g].arena ? prm.arena_stride : 0, other = prm.arena_stride - live;
            const bool flipped = advance_one(first_ + g, lane_, h_, st_ + live, lk_ + live, pa_ + live, st_ + other, lk_ + other, pa_ + other, cap, prm.rec_moves, prm.rec_visits,
                                             prm.rec_lens, prm.rec_winner, prm.unfinished, prm.reuse, 0, nullptr, prm.slots);
            if (flipped && lane_ == 0) h_[first_ + g].arena ^= 1u;     // the kept subtree was compacted into the other arena: that one is the tree now
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (prm.noise_alpha > 0.0f) {                                   // Default::AddNoise before the next search (MCTS.cpp:182), on the roots that have children
        GameHeader* h_ = headers; uint32_t* lk_ = link; float* rp_ = root_prior;
        int first_ = game0, lane_ = lane;
        asm volatile("" : "+s"(h_), "+s"(lk_), "+s"(rp_), "+s"(first_), "+v"(lane_));
#pragma unroll 1
        for (int g = 0; g < games_here; ++g)
            root_noise_one(first_ + g, lane_, h_, lk_, rp_, cap, prm.arena_stride, prm.noise_alpha, prm.noise_epsilon, prm.seed_lo, prm.seed_hi);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    bool any = false;
    for (int g = 0; g < games_here; ++g) any |= !(headers[game0 + g].status & 1u);
    if (!any) break;
    }
}

// fresh single-node trees (MCTS::reset, MCTS.cpp:149-156)
__global__ void mcts_init_roots_kernel(const GameHeader* __restrict__ headers, uint2* __restrict__ stats, uint32_t* __restrict__ link,
                                       uint32_t* __restrict__ parent, size_t cap, int n_games) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_games) return;
    const size_t base = static_cast<size_t>(g) * cap;
    stats[base] = make_uint2(0u, 0u);
    link[base] = headers[g].last_move;
    parent[base] = kNone;
}

// One self-play move per unfinished game: MCTS::stepForward() (MCTS.cpp:129-134) + Board::applyMove with the
// victory check (Game.cpp:37-47, 88-136) on the root position, the (move, visit counts) record of
// agents/utils.py:29-41, and the new root.  One wavefront per game.
// reuse: the chosen child's subtree is kept (copied into arena 2); flip_all: the caller swaps the arenas of ALL games afterwards (the lock-step
// loop: whatever a game's tree is after this step has to be in arena 2), otherwise only a game that kept a subtree moves, and the return value
// says so (the persistent loop: GameHeader::arena).  Returns (wave-uniform) whether the game's tree is in arena 2 now.
__device__ bool advance_one(int g, int lane, GameHeader* headers, uint2* stats, uint32_t* link, uint32_t* parent, uint2* stats2, uint32_t* link2,
                            uint32_t* parent2, size_t cap, uint8_t* rec_moves, uint16_t* rec_visits, int32_t* rec_lens, int8_t* rec_winner,
                            int32_t* unfinished, int reuse, int flip_all,
                            const int16_t* forced /* null, or per game: the move to step to (MCTS::stepForward(move)), -1 = the most visited child */,
                            const SlotRefill& slots) {
    GameHeader& hdr = headers[g];
    const int rec = slots.slot_game ? slots.slot_game[g] : g;      // the records' row of this slot's game
    if (rec < 0) return false;                                      // a slot that never got a game
    const size_t base = static_cast<size_t>(g) * cap;
    const uint32_t root = hdr.root, stones = hdr.stones;
    const uint32_t first = link[base + root] >> 8;
    const int want = forced ? forced[g] : -1;
    const uint32_t row_want = (want >= 0 && want < 225) ? hdr.rows[want / 15] >> (want % 15) : 0x10001u;
    const bool can_force = want >= 0 && want < 225 && !(row_want & 0x10001u);
    if (want >= 0 && !can_force && lane == 0) hdr.status |= 4u;     // the move is not legal on the root position: nothing is played
    if ((hdr.status & 1u) || (!first && !can_force) || (want >= 0 && !can_force)) {     // already over, never searched, or an illegal request: nothing to play
        if (reuse && flip_all && lane == 0) {                       // the live arena flips for every game: carry the root over
            stats2[base] = stats[base + root];
            link2[base] = link[base + root] & 0xFFu;
            parent2[base] = kNone;
            hdr.root = 0;
            hdr.n_nodes = 1;
        }
        // The persistent loop (flip_all = 0, no forced moves) ends when every slot's game has: a game whose root has no child after its search -- an
        // arena too small for one expansion -- would never end.  Its slot stops here, reported as an arena that filled up.
        if (!flip_all && !forced && !(hdr.status & 1u) && lane == 0) hdr.status |= 1u | 2u;
        return reuse && flip_all;
    }
    const int n_child = 225 - static_cast<int>(stones);
    long long best = -1;
    int best_i = 0;
    if (first) {
        for (int i = lane; i < n_child; i += 64) {
            // stepForward(): std::max_element, the first maximum; stepForward(move): the child of that move
            const long long v = can_force ? ((link[base + first + i] & 0xFFu) == static_cast<uint32_t>(want) ? 1 : 0) : static_cast<long long>(stats[base + first + i].x);
            if (v > best) { best = v; best_i = i; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const long long o = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(best_i, m, 64);
            if (o > best || (o == best && oi < best_i)) { best = o; best_i = oi; }
        }
    }
    const uint32_t child = first + static_cast<uint32_t>(best_i);
    const uint32_t cell = first ? (link[base + child] & 0xFFu) : static_cast<uint32_t>(want);
    const bool keep = reuse && first;                               // an unexpanded root has no subtree to keep (MCTS.cpp:140-145 creates a node)
    const int len = rec_lens[rec];
    if (rec_visits) {
        uint16_t* rv = rec_visits + (static_cast<size_t>(rec) * 225 + static_cast<size_t>(len)) * 225;
        for (int i = lane; i < 225; i += 64) rv[i] = 0;
        __syncthreads();
        if (first)
            for (int i = lane; i < n_child; i += 64)
                rv[link[base + first + i] & 0xFFu] = static_cast<uint16_t>(min(stats[base + first + i].x, 65535u));
    }
    __syncthreads();
    bool refilled = false;                                          // (lane 0) the slot starts a new game: fresh root whatever `keep` says
    if (lane == 0) {
        rec_moves[static_cast<size_t>(rec) * 225 + len] = static_cast<uint8_t>(cell);
        rec_lens[rec] = len + 1;
        const int x = static_cast<int>(cell % 15u), y = static_cast<int>(cell / 15u);
        const int shift = (stones & 1u) ? 16 : 0;                   // black moves on even stone counts
        hdr.rows[y] |= 1u << (x + shift);
        hdr.stones = stones + 1;
        hdr.last_move = cell;
        const bool five = five_through<1>(hdr.rows, x, y, shift);
        uint32_t root_cell = cell;
        if (five || stones + 1 == 225u) {
            hdr.status |= 1u;
            rec_winner[rec] = five ? static_cast<int8_t>(shift ? -1 : 1) : static_cast<int8_t>(0);
            const int next = slots.slot_game ? atomicAdd(slots.next_game, 1) : slots.n_total;
            if (slots.slot_game && next < slots.n_total) {          // the slot takes the next unstarted game: its opening is the new root
                refilled = true;
                slots.slot_game[g] = next;
                const int olen = slots.open_lens ? slots.open_lens[next] : 0;
                for (int y = 0; y < 16; ++y) hdr.rows[y] = 0u;
                for (int i = 0; i < olen; ++i) {
                    const uint32_t c = slots.open_moves[static_cast<size_t>(next) * slots.open_stride + i];
                    hdr.rows[c / 15u] |= 1u << (c % 15u + ((i & 1) ? 16u : 0u));
                    rec_moves[static_cast<size_t>(next) * 225 + i] = static_cast<uint8_t>(c);
                    root_cell = c;
                }
                if (olen == 0) root_cell = 255u;
                rec_lens[next] = olen;
                hdr.stones = static_cast<uint32_t>(olen);
                hdr.last_move = root_cell;
                hdr.game_id = slots.first_game_id + static_cast<uint32_t>(next);
                hdr.status &= 2u;                                   // a new game: not over; "an arena of this slot filled up" stays set for the whole
                                                                    // run (the caller reads it once, afterwards: dropped playouts must not go unreported)
                atomicAdd(unfinished, 1);
            }
        } else {
            atomicAdd(unfinished, 1);
        }
        hdr.playouts_done = 0;
        hdr.noise = 0;
        if (!keep || refilled) {                                    // MCTS::reset + syncWithBoard: a fresh one-node tree
            hdr.root = 0;
            hdr.n_nodes = 1;
            ((reuse && flip_all) ? stats2 : stats)[base] = make_uint2(0u, 0u);    // in the lock-step loop with reuse the live arena flips for every game
            ((reuse && flip_all) ? link2 : link)[base] = root_cell;
            ((reuse && flip_all) ? parent2 : parent)[base] = kNone;
        }
    }
    refilled = __shfl(static_cast<int>(refilled), 0, 64) != 0;
    if (!keep || refilled) return reuse && flip_all;

    // The subtree of the move becomes the tree (updateRoot, MCTS.cpp:63-67); the reference frees the siblings, here
    // the kept subtree is copied level by level into the other arena so that node indices stay dense.  In the new
    // arena a node's `link` carries its OLD first-child index until the scan reaches it and copies its children.
    if (lane == 0) {
        stats2[base] = stats[base + child];
        link2[base] = link[base + child];
        parent2[base] = kNone;
    }
    __syncthreads();
    uint32_t next = 1, level_end = 1;
    int depth = 0;                                                  // of the nodes being scanned; root = 0
    for (uint32_t i0 = 0; i0 < next;) {
        if (i0 == level_end) { ++depth; level_end = next; }
        const uint32_t chunk = min(64u, level_end - i0);
        const uint32_t old_first = (static_cast<uint32_t>(lane) < chunk) ? (link2[base + i0 + lane] >> 8) : 0u;
        unsigned long long todo = __ballot(old_first != 0u);
        const uint32_t n = 225u - (stones + 1u + static_cast<uint32_t>(depth));     // children of a node at this depth
        while (todo) {
            const int j = __ffsll(static_cast<long long>(todo)) - 1;
            todo &= todo - 1ull;
            const uint32_t of = __shfl(old_first, j, 64), node = i0 + static_cast<uint32_t>(j);
            for (uint32_t k = lane; k < n; k += 64) {
                stats2[base + next + k] = stats[base + of + k];
                link2[base + next + k] = link[base + of + k];
                parent2[base + next + k] = node;
            }
            if (lane == 0) link2[base + node] = (next << 8) | (link2[base + node] & 0xFFu);
            next += n;
        }
        __syncthreads();                                            // children written above are scanned below
        i0 += chunk;
    }
    if (lane == 0) { hdr.root = 0; hdr.n_nodes = next; }
    return true;
}

// Default::AddNoise (MonteCarlo.hpp:97-108) on the root of game g, by one wavefront, with the counter-based sampler (noise_device.h): the root's
// children are its empty cells in ascending order, every one with Default::UniformProbs' prior 1 / float(children); the mixed priors go to
// root_prior[g][child] and GameHeader::noise tells select to read them.  A finished game or a root without children takes none (the loop
// over node->children is empty there).  Draws are keyed by (seed; global game id, stones on the root board, cell).
__device__ void root_noise_one(int g, int lane, GameHeader* headers, const uint32_t* link, float* root_prior, size_t cap, size_t arena_stride,
                               float alpha, float epsilon, uint32_t seed_lo, uint32_t seed_hi) {
    GameHeader& hdr = headers[g];
    const size_t base = static_cast<size_t>(g) * cap + (hdr.arena ? arena_stride : 0);
    const bool apply = !(hdr.status & 1u) && (link[base + hdr.root] >> 8) != 0u;
    if (apply) {
        const uint32_t stones = hdr.stones;
        const float uniform = c_prior[225u - stones];
        float p[4];
        int rank[4];
        bool child[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cell = lane + 64 * j, y = min(cell, 224) / 15, x = min(cell, 224) % 15;
            int before = 0;                                         // empty cells in front of this one = the child's index
            for (int yy = 0; yy < y; ++yy) { const uint32_t r = hdr.rows[yy]; before += __popc(~(r | (r >> 16)) & 0x7FFFu); }
            const uint32_t r = hdr.rows[y], open = ~(r | (r >> 16)) & 0x7FFFu;
            rank[j] = before + __popc(open & ((1u << x) - 1u));
            child[j] = cell < 225 && ((open >> x) & 1u);
            p[j] = child[j] ? uniform : 0.0f;
        }
        gmk::noise::mix_root_priors(p, lane, alpha, epsilon, hdr.game_id, stones, seed_lo, seed_hi);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (child[j]) root_prior[static_cast<size_t>(g) * 225 + rank[j]] = p[j];
    }
    if (lane == 0) hdr.noise = apply ? 1u : 0u;
}

__global__ __launch_bounds__(64)
void mcts_root_noise_kernel(GameHeader* __restrict__ headers, const uint32_t* __restrict__ link, float* __restrict__ root_prior, size_t cap, int n_games,
                            float alpha, float epsilon, uint32_t seed_lo, uint32_t seed_hi) {
    if (static_cast<int>(blockIdx.x) < n_games) root_noise_one(blockIdx.x, threadIdx.x, headers, link, root_prior, cap, 0, alpha, epsilon, seed_lo, seed_hi);
}

__global__ __launch_bounds__(64)
void mcts_advance_kernel(GameHeader* __restrict__ headers, uint2* __restrict__ stats, uint32_t* __restrict__ link,
                         uint32_t* __restrict__ parent, uint2* __restrict__ stats2, uint32_t* __restrict__ link2,
                         uint32_t* __restrict__ parent2, size_t cap, int n_games,
                         uint8_t* __restrict__ rec_moves, uint16_t* __restrict__ rec_visits, int32_t* __restrict__ rec_lens,
                         int8_t* __restrict__ rec_winner, int32_t* __restrict__ unfinished, int reuse,
                         const int16_t* __restrict__ forced, SlotRefill slots, int32_t* __restrict__ counters) {
    if (static_cast<int>(blockIdx.x) >= n_games) return;
    // counters[0] counts the games that go on, counters[1] the workgroups that are through: the last one hands the count to the caller and
    // leaves both at zero for the next launch -- no memset in front of every step (a four-byte fill is a launch of its own, and beside a
    // second handle's search it waited milliseconds for a wavefront slot)
    advance_one(blockIdx.x, threadIdx.x, headers, stats, link, parent, stats2, link2, parent2, cap, rec_moves, rec_visits, rec_lens, rec_winner, &counters[0], reuse, 1, forced, slots);
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&counters[1], 1) == n_games - 1) {
            *unfinished = atomicExch(&counters[0], 0);
            atomicExch(&counters[1], 0);
        }
    }
}

__global__ void mcts_root_flags_kernel(GameHeader* __restrict__ headers, const uint32_t* __restrict__ link, size_t cap, int n_games) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_games) return;
    headers[g].root_expanded = (link[static_cast<size_t>(g) * cap + headers[g].root] >> 8) != 0u;
}

// children of the root -> visit counts by cell (MCTS::evalState, MCTS.cpp:104-110)
__global__ void mcts_root_stats_kernel(const GameHeader* __restrict__ headers, const uint2* __restrict__ stats,
                                       const uint32_t* __restrict__ link, size_t cap, size_t arena_stride, int n_games,
                                       uint32_t* __restrict__ visits, float* __restrict__ root_value,
                                       uint32_t* __restrict__ root_visits, uint32_t* __restrict__ nodes, int32_t* __restrict__ status) {
    const int g = blockIdx.x;
    if (g >= n_games) return;
    const GameHeader& hdr = headers[g];
    const size_t base = static_cast<size_t>(g) * cap + (hdr.arena ? arena_stride : 0);
    for (int i = threadIdx.x; i < 225; i += blockDim.x) visits[static_cast<size_t>(g) * 225 + i] = 0;
    __syncthreads();
    const uint32_t first = link[base + hdr.root] >> 8;
    if (first) {
        const int n_child = 225 - static_cast<int>(hdr.stones);
        for (int i = threadIdx.x; i < n_child; i += blockDim.x)
            visits[static_cast<size_t>(g) * 225 + (link[base + first + i] & 0xFFu)] = stats[base + first + i].x;
    }
    if (threadIdx.x == 0) {
        const uint2 st = stats[base + hdr.root];
        root_visits[g] = st.x;
        root_value[g] = __uint_as_float(st.y);
        nodes[g] = hdr.n_nodes;
        status[g] = static_cast<int32_t>(hdr.status);
    }
}

}  // namespace

struct gmk_mcts {
    int n_games = 0, node_capacity = 0, c_rollouts = 5, games_per_block = 12;
    double c_puct = 5.0;
    uint64_t seed = 0;
    GameHeader* d_headers = nullptr;
    uint2* d_stats = nullptr;          // live arena
    uint32_t* d_link = nullptr;
    uint32_t* d_parent = nullptr;
    uint2* d_stats2 = nullptr;         // second arena, allocated by the first advance() that keeps subtrees
    uint32_t* d_link2 = nullptr;
    uint32_t* d_parent2 = nullptr;
    float* d_root_prior = nullptr;     // [n_games][225] by child index, used while GameHeader::noise is set
    float* d_value = nullptr;          // [2 * c_rollouts + 1] rollout sum -> state value
    SlotRefill slots{};                // continuous batching (gmk_selfplay_run); all null otherwise
    struct { uint8_t* moves; uint16_t* visits; int32_t* lens; int8_t* winner; int32_t* unfinished; int reuse; float noise_alpha, noise_epsilon; } persistent_rec{};     // set while gmk_selfplay_run's ONE launch is issued
    int32_t* d_slot_state = nullptr;   // [n_games + 1] slot_game, next_game
    int32_t* d_step_counters = nullptr; // [2] mcts_advance_kernel's own (zero between launches)
    uint8_t* d_open_moves = nullptr;
    int32_t* d_open_lens = nullptr;
    void* d_step_scratch = nullptr;    // record outputs of gmk_mcts_step_host
    hipStream_t last_stream = nullptr;
    bool rooted = false;               // gmk_mcts_set_roots has run: headers and arenas hold trees
    // gmk_mcts_set_option
    int noise_sampler = GMK_NOISE_SAMPLER_STD;   // where Default::AddNoise draws from
    int lockstep = 0;                  // 1: gmk_selfplay_run alternates search and step launches even where ONE persistent launch could play the games
    // the persistent loop with kept subtrees wants a game's two arenas a fixed distance apart: both halves of ONE allocation per array
    bool paired = false;
    uint2* block_stats = nullptr;
    uint32_t *block_link = nullptr, *block_parent = nullptr;
    size_t arena_stride() const { return (paired && d_stats2 > d_stats) ? static_cast<size_t>(d_stats2 - d_stats) : 0; }
};

extern "C" int gmk_mcts_create(int n_games, int node_capacity, double c_puct, int c_rollouts, uint64_t seed, gmk_mcts** out) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (!out || n_games <= 0 || node_capacity < 2 || node_capacity >= (1 << 24) || c_rollouts < 1 || c_rollouts > 64) {
        gmk::set_error("gmk_mcts_create: bad arguments (2 <= node_capacity < 2^24: a node's first-child index shares a word with its cell)");
        return GMK_ERR_ARG;
    }
    gmk_mcts* m = new gmk_mcts;
    m->n_games = n_games; m->node_capacity = node_capacity; m->c_puct = c_puct; m->c_rollouts = c_rollouts; m->seed = seed;
    // Games per wavefront.  A game is a sequential chain (select -> rollouts -> backup), so wall time is set by the chain's latency,
    // not by lane utilisation: fewer games per wave = fewer sequential tree rounds and a shorter longest-rollout tail, as long as
    // the chip has wave slots to spare (1024 SIMDs on MI355X, two of these wavefronts each): aim at two waves per SIMD, and never
    // more than four games per wavefront (one quarter-wave round of the tree phases) -- a batch beyond 8 192 games then runs in
    // several rounds of workgroups, which is faster than longer wavefronts (M playouts/s at 800 playouts, games per wavefront 4 / 6 /
    // 8 / 12: 16 384 games 156 / 123 / 119 / 95, 32 768 games 159 / 132 / 117 / 99; 6 144 games: 3 -> 126, 4 -> 117).
    const int max_gpb = std::min(kMaxGamesPerBlock, 64 / c_rollouts);
    const int simds = std::max(1, st.cu_count * 4);
    int gpb = std::min(4, (n_games + 2 * simds - 1) / (2 * simds));
    if (const char* env = gmk::profile_env("GMK_MCTS_GAMES_PER_BLOCK")) gpb = std::atoi(env);
    m->games_per_block = std::max(1, std::min(max_gpb, gpb));
    // (the tree arenas -- tens of GB -- are allocated by the first gmk_mcts_set_roots, or by gmk_selfplay_run as the two halves of one block per array
    // when it keeps subtrees inside its persistent launch: allocating them here only to replace them there costs seconds of driver time)
    if (gmk::device_malloc(&m->d_headers, sizeof(GameHeader) * n_games) != hipSuccess) {
        gmk::set_error("gmk_mcts_create: hipMalloc of the game headers failed");
        gmk_mcts_destroy(m);
        return GMK_ERR_HIP;
    }
    static bool prior_uploaded = false;            // the same for every handle: written once, never while a search may be reading it
    if (!prior_uploaded) {
        float prior[226];
        prior[0] = 0.0f;
        for (int i = 1; i <= 225; ++i) prior[i] = 1.0f / static_cast<float>(i);
        GMK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_prior), prior, sizeof prior));
        prior_uploaded = true;
    }
    float value[129] = {};
    for (int s = -c_rollouts; s <= c_rollouts; ++s) value[s + c_rollouts] = static_cast<float>(static_cast<double>(s) / static_cast<double>(c_rollouts));
    const int32_t zeros[2] = {0, 0};                   // (a synchronous copy: the first step may be launched on any stream)
    if (gmk::device_malloc(&m->d_step_counters, 8) != hipSuccess || hipMemcpy(m->d_step_counters, zeros, 8, hipMemcpyHostToDevice) != hipSuccess ||
        gmk::device_malloc(&m->d_value, sizeof value) != hipSuccess || hipMemcpy(m->d_value, value, sizeof value, hipMemcpyHostToDevice) != hipSuccess) {
        gmk::set_error("gmk_mcts_create: value table upload failed");
        gmk_mcts_destroy(m);
        return GMK_ERR_HIP;
    }
    *out = m;
    return GMK_OK;
}

extern "C" int gmk_mcts_destroy(gmk_mcts* m) {
    if (!m) return GMK_OK;
    if (m->paired) {                                                // both arenas are halves of the three blocks
        (void)gmk::device_free(m->block_stats); (void)gmk::device_free(m->block_link); (void)gmk::device_free(m->block_parent);
        m->d_stats = m->d_stats2 = nullptr; m->d_link = m->d_link2 = nullptr; m->d_parent = m->d_parent2 = nullptr;
    }
    (void)gmk::device_free(m->d_headers); (void)gmk::device_free(m->d_stats); (void)gmk::device_free(m->d_link); (void)gmk::device_free(m->d_parent); (void)gmk::device_free(m->d_value);
    (void)gmk::device_free(m->d_slot_state); (void)gmk::device_free(m->d_open_moves); (void)gmk::device_free(m->d_open_lens); (void)gmk::device_free(m->d_step_counters);
    (void)gmk::device_free(m->d_stats2); (void)gmk::device_free(m->d_link2); (void)gmk::device_free(m->d_parent2); (void)gmk::device_free(m->d_root_prior); (void)gmk::device_free(m->d_step_scratch);
    delete m;
    return GMK_OK;
}

extern "C" int gmk_mcts_set_option(gmk_mcts* m, int option, int value) {
    if (!m) { gmk::set_error("gmk_mcts_set_option: bad arguments"); return GMK_ERR_ARG; }
    if (option == GMK_OPT_NOISE_SAMPLER && (value == GMK_NOISE_SAMPLER_STD || value == GMK_NOISE_SAMPLER_COUNTER)) { m->noise_sampler = value; return GMK_OK; }
    if (option == GMK_OPT_LOCKSTEP && (value == 0 || value == 1)) { m->lockstep = value; return GMK_OK; }
    gmk::set_error("gmk_mcts_set_option: unknown option %d or value %d", option, value);
    return GMK_ERR_ARG;
}

// Both arenas of every game as the two halves of ONE block per array (the persistent loop with kept subtrees adds a fixed stride to a game's
// node indices to reach its other arena).  The trees are lost: call before gmk_mcts_set_roots.
static int pair_arenas(gmk_mcts* m) {
    const size_t nodes = static_cast<size_t>(m->n_games) * static_cast<size_t>(m->node_capacity);
    if (m->paired) {
        m->d_stats = m->block_stats; m->d_link = m->block_link; m->d_parent = m->block_parent;       // (the lock-step loop may have left them swapped)
        m->d_stats2 = m->block_stats + nodes; m->d_link2 = m->block_link + nodes; m->d_parent2 = m->block_parent + nodes;
        return GMK_OK;
    }
    (void)gmk::device_free(m->d_stats); (void)gmk::device_free(m->d_link); (void)gmk::device_free(m->d_parent);
    (void)gmk::device_free(m->d_stats2); (void)gmk::device_free(m->d_link2); (void)gmk::device_free(m->d_parent2);
    m->d_stats = m->d_stats2 = nullptr; m->d_link = m->d_link2 = nullptr; m->d_parent = m->d_parent2 = nullptr;
    m->rooted = false;
    if (gmk::device_malloc(&m->block_stats, 2 * nodes * sizeof(uint2)) != hipSuccess || gmk::device_malloc(&m->block_link, 2 * nodes * 4) != hipSuccess ||
        gmk::device_malloc(&m->block_parent, 2 * nodes * 4) != hipSuccess) {
        (void)gmk::device_free(m->block_stats); (void)gmk::device_free(m->block_link); (void)gmk::device_free(m->block_parent);
        m->block_stats = nullptr; m->block_link = nullptr; m->block_parent = nullptr;
        (void)hipGetLastError();
        gmk::set_error("gmk_selfplay_run: hipMalloc of two arenas per game (%zu nodes, %.1f GB) failed", 2 * nodes, 2 * nodes * 16.0 / 1e9);
        return GMK_ERR_HIP;
    }
    m->paired = true;
    m->d_stats = m->block_stats; m->d_link = m->block_link; m->d_parent = m->block_parent;
    m->d_stats2 = m->block_stats + nodes; m->d_link2 = m->block_link + nodes; m->d_parent2 = m->block_parent + nodes;
    return GMK_OK;
}

static int one_arena(gmk_mcts* m) {
    if (m->d_stats) return GMK_OK;
    const size_t nodes = static_cast<size_t>(m->n_games) * static_cast<size_t>(m->node_capacity);
    if (gmk::device_malloc(&m->d_stats, nodes * sizeof(uint2)) != hipSuccess || gmk::device_malloc(&m->d_link, nodes * 4) != hipSuccess ||
        gmk::device_malloc(&m->d_parent, nodes * 4) != hipSuccess) {
        (void)gmk::device_free(m->d_stats); (void)gmk::device_free(m->d_link); (void)gmk::device_free(m->d_parent);
        m->d_stats = nullptr; m->d_link = nullptr; m->d_parent = nullptr;
        (void)hipGetLastError();
        gmk::set_error("gmk_mcts: hipMalloc of %zu nodes (%.1f GB) failed", nodes, nodes * 16.0 / 1e9);
        return GMK_ERR_HIP;
    }
    return GMK_OK;
}

// The handle's tree arenas, now: one arena per game (what gmk_mcts_set_roots would allocate at its first call) or, two_arenas != 0, the two
// arenas per game of the persistent loop with kept subtrees (what gmk_selfplay_run would).  For callers that want the allocation -- tens of GB; the
// driver clears memory it has handed out before, seconds per 24 GB -- outside a region they time, or want the blocks in the library's pool
// before a batch starts (create, reserve, destroy: the next handle of that shape finds them there).
extern "C" int gmk_mcts_reserve(gmk_mcts* m, int two_arenas) {
    if (!m) { gmk::set_error("gmk_mcts_reserve: bad arguments"); return GMK_ERR_ARG; }
    return two_arenas ? pair_arenas(m) : one_arena(m);
}

extern "C" int gmk_mcts_set_roots(gmk_mcts* m, const uint16_t* h_planes, const int16_t* h_last_move, uint32_t first_game_id) {
    if (!m || !h_planes || !h_last_move) { gmk::set_error("gmk_mcts_set_roots: bad arguments"); return GMK_ERR_ARG; }
    if (const int rc = one_arena(m); rc != GMK_OK) return rc;       // the first roots of this handle: its tree arena
    std::vector<GameHeader> hdr(static_cast<size_t>(m->n_games));
    for (int g = 0; g < m->n_games; ++g) {
        GameHeader& h = hdr[static_cast<size_t>(g)];
        std::memset(&h, 0, sizeof h);
        uint32_t stones = 0;
        for (int y = 0; y < 15; ++y) {
            const uint32_t b = h_planes[static_cast<size_t>(g) * 32 + y] & 0x7FFFu, w = h_planes[static_cast<size_t>(g) * 32 + 16 + y] & 0x7FFFu;
            h.rows[y] = b | (w << 16);
            stones += static_cast<uint32_t>(__builtin_popcount(b) + __builtin_popcount(w));
        }
        h.root = 0; h.n_nodes = 1; h.stones = stones;
        h.last_move = (h_last_move[g] >= 0 && h_last_move[g] < 225) ? static_cast<uint32_t>(h_last_move[g]) : 255u;
        h.game_id = first_game_id + static_cast<uint32_t>(g);
    }
    GMK_HIP_CHECK(hipMemcpy(m->d_headers, hdr.data(), sizeof(GameHeader) * hdr.size(), hipMemcpyHostToDevice));
    // root node of every arena: unvisited, no children, no parent
    hipLaunchKernelGGL(mcts_init_roots_kernel, dim3((m->n_games + 255) / 256), dim3(256), 0, nullptr, m->d_headers, m->d_stats, m->d_link,
                       m->d_parent, static_cast<size_t>(m->node_capacity), m->n_games);
    GMK_HIP_CHECK(hipGetLastError());
    GMK_HIP_CHECK(hipDeviceSynchronize());
    m->rooted = true;
    return GMK_OK;
}

extern "C" int gmk_mcts_set_game_ids(gmk_mcts* m, const uint32_t* h_ids) {
    if (!m || !h_ids) { gmk::set_error("gmk_mcts_set_game_ids: bad arguments"); return GMK_ERR_ARG; }
    if (!m->rooted) { gmk::set_error("gmk_mcts_set_game_ids: gmk_mcts_set_roots has not been called"); return GMK_ERR_STATE; }
    GMK_HIP_CHECK(hipDeviceSynchronize());
    std::vector<GameHeader> hdr(static_cast<size_t>(m->n_games));
    GMK_HIP_CHECK(hipMemcpy(hdr.data(), m->d_headers, hdr.size() * sizeof(GameHeader), hipMemcpyDeviceToHost));
    for (int g = 0; g < m->n_games; ++g) hdr[static_cast<size_t>(g)].game_id = h_ids[g];
    GMK_HIP_CHECK(hipMemcpy(m->d_headers, hdr.data(), hdr.size() * sizeof(GameHeader), hipMemcpyHostToDevice));
    return GMK_OK;
}

// dynamic LDS of mcts_playouts_kernel in words: line words and cell blocks per rollout lane, leaf position and path per game
static size_t lds_words(int games_per_block, int c_rollouts) {
    return static_cast<size_t>(kLineWords + 2 * kCellBlocks) * games_per_block * c_rollouts + static_cast<size_t>(kLineWords + 3 * kPathCap) * games_per_block;
}

extern "C" int gmk_mcts_run(gmk_mcts* m, int playouts, void* stream) {
    if (m && !m->rooted) { gmk::set_error("gmk_mcts_run: gmk_mcts_set_roots has not been called"); return GMK_ERR_STATE; }
    if (!m || playouts < 0) { gmk::set_error("gmk_mcts_run: bad arguments"); return GMK_ERR_ARG; }
    SearchParams prm;
    prm.c_puct = m->c_puct;
    prm.seed_lo = static_cast<uint32_t>(m->seed); prm.seed_hi = static_cast<uint32_t>(m->seed >> 32);
    prm.c_rollouts = m->c_rollouts; prm.games_per_block = m->games_per_block;
    prm.node_capacity = m->node_capacity; prm.n_games = m->n_games; prm.playouts = playouts;
    const int grid = (m->n_games + m->games_per_block - 1) / m->games_per_block;
    prm.profile = gmk::profile_env("GMK_MCTS_PROFILE") ? 1 : 0;
    prm.value_table = m->d_value;
    prm.persistent = 0; prm.rec_moves = nullptr; prm.rec_visits = nullptr; prm.rec_lens = nullptr; prm.rec_winner = nullptr; prm.unfinished = nullptr; prm.slots = SlotRefill{};
    prm.reuse = 0; prm.noise_alpha = 0.0f; prm.noise_epsilon = 0.0f; prm.arena_stride = m->arena_stride();
    if (m->persistent_rec.moves) {                              // gmk_selfplay_run's persistent form: this launch plays the games to their end
        prm.persistent = 1;
        prm.rec_moves = m->persistent_rec.moves; prm.rec_visits = m->persistent_rec.visits; prm.rec_lens = m->persistent_rec.lens; prm.rec_winner = m->persistent_rec.winner;
        prm.unfinished = m->persistent_rec.unfinished; prm.slots = m->slots;
        prm.reuse = m->persistent_rec.reuse; prm.noise_alpha = m->persistent_rec.noise_alpha; prm.noise_epsilon = m->persistent_rec.noise_epsilon;
    }
    m->last_stream = static_cast<hipStream_t>(stream);
    const int rollout_lanes = m->games_per_block * m->c_rollouts;
    const bool staged = 2 * rollout_lanes <= 64 && 4 * rollout_lanes > 64;      // the pairs form (mcts_playouts_kernel: kStaged)
    auto* kernel = prm.persistent ? (staged ? mcts_playouts_kernel<true, true> : mcts_playouts_kernel<true, false>)
                                  : (staged ? mcts_playouts_kernel<false, true> : mcts_playouts_kernel<false, false>);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), lds_words(m->games_per_block, m->c_rollouts) * 4, m->last_stream, m->d_headers, m->d_stats, m->d_link, m->d_parent,
                       m->d_root_prior, prm);
    GMK_HIP_CHECK(hipGetLastError());
    return GMK_OK;
}

extern "C" int gmk_mcts_step(gmk_mcts* m, const int16_t* d_forced_moves, uint8_t* d_moves, uint16_t* d_visits, int32_t* d_lens, int8_t* d_winner,
                             int32_t* d_unfinished, int reuse_subtree, void* stream);

extern "C" int gmk_mcts_advance(gmk_mcts* m, uint8_t* d_moves, uint16_t* d_visits, int32_t* d_lens, int8_t* d_winner,
                                int32_t* d_unfinished, int reuse_subtree, void* stream) {
    return gmk_mcts_step(m, nullptr, d_moves, d_visits, d_lens, d_winner, d_unfinished, reuse_subtree, stream);
}

extern "C" int gmk_mcts_step(gmk_mcts* m, const int16_t* d_forced_moves, uint8_t* d_moves, uint16_t* d_visits, int32_t* d_lens, int8_t* d_winner,
                             int32_t* d_unfinished, int reuse_subtree, void* stream) {
    if (m && !m->rooted) { gmk::set_error("gmk_mcts_step: gmk_mcts_set_roots has not been called"); return GMK_ERR_STATE; }
    if (!m || !d_moves || !d_lens || !d_winner || !d_unfinished) { gmk::set_error("gmk_mcts_step: bad arguments"); return GMK_ERR_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    m->last_stream = s;
    if (reuse_subtree && !m->d_stats2) {
        const size_t nodes = static_cast<size_t>(m->n_games) * static_cast<size_t>(m->node_capacity);
        if (gmk::device_malloc(&m->d_stats2, nodes * sizeof(uint2)) != hipSuccess || gmk::device_malloc(&m->d_link2, nodes * 4) != hipSuccess ||
            gmk::device_malloc(&m->d_parent2, nodes * 4) != hipSuccess) {
            (void)gmk::device_free(m->d_stats2); (void)gmk::device_free(m->d_link2); (void)gmk::device_free(m->d_parent2);      // all or nothing: the guard above tests d_stats2
            m->d_stats2 = nullptr; m->d_link2 = nullptr; m->d_parent2 = nullptr;
            (void)hipGetLastError();
            gmk::set_error("gmk_mcts_step: hipMalloc of the second arena (%zu nodes) failed", nodes);
            return GMK_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(mcts_advance_kernel, dim3(m->n_games), dim3(64), 0, s, m->d_headers, m->d_stats, m->d_link, m->d_parent,
                       m->d_stats2, m->d_link2, m->d_parent2, static_cast<size_t>(m->node_capacity), m->n_games,
                       d_moves, d_visits, d_lens, d_winner, d_unfinished, reuse_subtree, d_forced_moves, m->slots, m->d_step_counters);
    GMK_HIP_CHECK(hipGetLastError());
    if (reuse_subtree) {                                            // the copy is the live tree from here on
        std::swap(m->d_stats, m->d_stats2);
        std::swap(m->d_link, m->d_link2);
        std::swap(m->d_parent, m->d_parent2);
    }
    return GMK_OK;
}

// gmk_mcts_step for callers without device buffers of their own (the one-game searcher behind CorePyExt): the moves
// come from the host, the record outputs go to scratch buffers owned by the handle.
// Whole games with continuous batching, resident on the device: see SlotRefill.  The host's part is the launch loop (one search and one
// step per move of the slots) and a 4-byte readback per move.
extern "C" int gmk_selfplay_run(gmk_mcts* m, int n_total, uint32_t first_game_id, int playouts, int reuse_subtree, float noise_alpha, float noise_epsilon,
                                const uint8_t* h_open_moves, int open_stride, const int32_t* h_open_lens,
                                uint8_t* d_moves, uint16_t* d_visits, int32_t* d_lens, int8_t* d_winner, int32_t* h_steps, void* stream) {
    // (playouts >= 1: a root that was never searched has no child to play, and the persistent launch ends only when every game has)
    if (!m || n_total <= 0 || playouts < 1 || !d_moves || !d_lens || !d_winner || (h_open_moves && (!h_open_lens || open_stride <= 0))) {
        gmk::set_error("gmk_selfplay_run: bad arguments (playouts >= 1)");
        return GMK_ERR_ARG;
    }
    const int n_slots = m->n_games;
    const size_t nt = static_cast<size_t>(n_total);
    std::vector<int32_t> open_lens(nt, 0);
    if (h_open_moves)
        for (size_t g = 0; g < nt; ++g) {
            if (h_open_lens[g] < 0 || h_open_lens[g] > 8 || h_open_lens[g] > open_stride) { gmk::set_error("gmk_selfplay_run: opening of game %zu has %d moves (0 .. 8)", g, h_open_lens[g]); return GMK_ERR_ARG; }
            open_lens[g] = h_open_lens[g];
        }
    hipStream_t s = static_cast<hipStream_t>(stream);
    GMK_HIP_CHECK(hipStreamSynchronize(s));
    // the first min(n_slots, n_total) games start in the slots (fresh roots from their openings), the rest wait for a free slot
    std::vector<uint16_t> planes(static_cast<size_t>(n_slots) * 32, 0);
    std::vector<int16_t> last(static_cast<size_t>(n_slots), -1);
    std::vector<int32_t> slot_state(static_cast<size_t>(n_slots) + 1, -1);
    std::vector<uint8_t> first_moves(nt * 225, 0);
    for (size_t g = 0; g < nt; ++g)
        for (int i = 0; i < open_lens[g]; ++i) {
            const uint8_t c = h_open_moves[g * open_stride + i];
            if (c >= 225) { gmk::set_error("gmk_selfplay_run: opening of game %zu holds cell %d", g, c); return GMK_ERR_ARG; }
            first_moves[g * 225 + i] = c;
            if (g < static_cast<size_t>(n_slots)) { planes[g * 32 + (i & 1) * 16 + c / 15] |= static_cast<uint16_t>(1u << (c % 15)); last[g] = c; }
        }
    const int started = std::min(n_slots, n_total);
    for (int g = 0; g < started; ++g) slot_state[static_cast<size_t>(g)] = g;
    slot_state[static_cast<size_t>(n_slots)] = started;                         // next_game
    // ONE launch (SearchParams::persistent) wherever nothing has to come from the host between two searches: every wavefront plays its slots'
    // games turn by turn at its own pace; the records are the same bytes either way (a game's random streams are keyed by its global id).
    const bool noisy = noise_alpha > 0.0f && reuse_subtree;                     // (a new root has no children: AddNoise is a no-op without kept subtrees)
    const bool persistent = !m->lockstep && !(noisy && m->noise_sampler != GMK_NOISE_SAMPLER_COUNTER);
    int rc = GMK_OK;
    if (persistent && reuse_subtree) rc = pair_arenas(m);                       // a game's two arenas, a fixed stride apart
    if (rc != GMK_OK) return rc;
    rc = gmk_mcts_set_roots(m, planes.data(), last.data(), first_game_id);
    if (rc != GMK_OK) return rc;
    if (noisy && !m->d_root_prior) GMK_HIP_CHECK(gmk::device_malloc(&m->d_root_prior, static_cast<size_t>(n_slots) * 225 * sizeof(float)));
    if (!m->d_slot_state) GMK_HIP_CHECK(gmk::device_malloc(&m->d_slot_state, (static_cast<size_t>(n_slots) + 1) * 4));
    (void)gmk::device_free(m->d_open_moves); (void)gmk::device_free(m->d_open_lens);
    m->d_open_moves = nullptr; m->d_open_lens = nullptr;
    GMK_HIP_CHECK(gmk::device_malloc(&m->d_open_lens, nt * 4));
    GMK_HIP_CHECK(hipMemcpy(m->d_open_lens, open_lens.data(), nt * 4, hipMemcpyHostToDevice));
    if (h_open_moves) {
        GMK_HIP_CHECK(gmk::device_malloc(&m->d_open_moves, nt * static_cast<size_t>(open_stride)));
        GMK_HIP_CHECK(hipMemcpy(m->d_open_moves, h_open_moves, nt * static_cast<size_t>(open_stride), hipMemcpyHostToDevice));
    }
    GMK_HIP_CHECK(hipMemcpy(m->d_slot_state, slot_state.data(), slot_state.size() * 4, hipMemcpyHostToDevice));
    GMK_HIP_CHECK(hipMemcpy(d_moves, first_moves.data(), nt * 225, hipMemcpyHostToDevice));
    GMK_HIP_CHECK(hipMemcpy(d_lens, open_lens.data(), nt * 4, hipMemcpyHostToDevice));
    GMK_HIP_CHECK(hipMemset(d_winner, 0, nt));
    if (n_slots > n_total) {                                                    // slots without a game: finished from the start
        std::vector<GameHeader> hdr(static_cast<size_t>(n_slots));
        GMK_HIP_CHECK(hipMemcpy(hdr.data(), m->d_headers, hdr.size() * sizeof(GameHeader), hipMemcpyDeviceToHost));
        for (int g = n_total; g < n_slots; ++g) hdr[static_cast<size_t>(g)].status |= 1u;
        GMK_HIP_CHECK(hipMemcpy(m->d_headers, hdr.data(), hdr.size() * sizeof(GameHeader), hipMemcpyHostToDevice));
    }
    m->slots = SlotRefill{m->d_slot_state, m->d_slot_state + n_slots, n_total, m->d_open_moves, m->d_open_lens, open_stride, first_game_id};
    int32_t* d_unfinished = nullptr;
    GMK_HIP_CHECK(gmk::device_malloc(&d_unfinished, 4));
    int32_t steps = 0;
    rc = GMK_OK;
    if (persistent) {
        m->persistent_rec = {d_moves, d_visits, d_lens, d_winner, d_unfinished, reuse_subtree ? 1 : 0, noisy ? noise_alpha : 0.0f, noise_epsilon};
        rc = gmk_mcts_run(m, playouts, s);
        m->persistent_rec = {};
        if (rc == GMK_OK && hipStreamSynchronize(s) != hipSuccess) { gmk::set_error("gmk_selfplay_run: the persistent launch failed"); rc = GMK_ERR_HIP; }
        steps = 1;
    }
    for (long long step = 0; step < 226ll * (n_total / n_slots + 2) && !persistent; ++step) {
        if (noisy) rc = gmk_mcts_add_root_noise(m, noise_alpha, noise_epsilon, s);      // Default::AddNoise at the start of every search (MCTS.cpp:182)
        if (rc == GMK_OK) rc = gmk_mcts_run(m, playouts, s);
        if (rc == GMK_OK) rc = gmk_mcts_step(m, nullptr, d_moves, d_visits, d_lens, d_winner, d_unfinished, reuse_subtree, s);
        if (rc != GMK_OK) break;
        ++steps;
        int32_t unfinished = 0;
        if (hipMemcpyAsync(&unfinished, d_unfinished, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { gmk::set_error("gmk_selfplay_run: readback failed"); rc = GMK_ERR_HIP; break; }
        if (unfinished == 0) break;
    }
    (void)gmk::device_free(d_unfinished);
    m->slots = SlotRefill{};
    if (h_steps) *h_steps = steps;
    return rc;
}

extern "C" int gmk_mcts_step_host(gmk_mcts* m, const int16_t* h_moves, int reuse_subtree) {
    if (!m || !h_moves) { gmk::set_error("gmk_mcts_step_host: bad arguments"); return GMK_ERR_ARG; }
    const size_t n = static_cast<size_t>(m->n_games);
    if (!m->d_step_scratch) GMK_HIP_CHECK(gmk::device_malloc(&m->d_step_scratch, n * (225 + 4 + 2 + 1) + 64));
    uint8_t* base = static_cast<uint8_t*>(m->d_step_scratch);
    int32_t* d_lens = reinterpret_cast<int32_t*>(base);
    int32_t* d_unfinished = reinterpret_cast<int32_t*>(base + n * 4);
    int16_t* d_forced = reinterpret_cast<int16_t*>(base + n * 4 + 16);
    int8_t* d_winner = reinterpret_cast<int8_t*>(base + n * 4 + 16 + n * 2);
    uint8_t* d_moves = base + n * 4 + 16 + n * 2 + ((n + 15) & ~size_t(15));
    GMK_HIP_CHECK(hipDeviceSynchronize());
    GMK_HIP_CHECK(hipMemset(d_lens, 0, n * 4));
    GMK_HIP_CHECK(hipMemcpy(d_forced, h_moves, n * sizeof(int16_t), hipMemcpyHostToDevice));
    const int rc = gmk_mcts_step(m, d_forced, d_moves, nullptr, d_lens, d_winner, d_unfinished, reuse_subtree, nullptr);
    if (rc != GMK_OK) return rc;
    GMK_HIP_CHECK(hipDeviceSynchronize());
    return GMK_OK;
}

// Default::AddNoise (MonteCarlo.hpp:97-108) for every unfinished game whose root already has children:
//   P <- (1 - epsilon) * P + epsilon * normalized(gamma(alpha, 1) per child)        (Statistical.hpp:29-34)
// (root_noise.h).  Host side: 225 floats a game.
extern "C" int gmk_mcts_add_root_noise(gmk_mcts* m, float alpha, float epsilon, void* stream) {
    if (m && !m->rooted) { gmk::set_error("gmk_mcts_add_root_noise: gmk_mcts_set_roots has not been called"); return GMK_ERR_STATE; }
    if (!m || !(alpha > 0.0f)) { gmk::set_error("gmk_mcts_add_root_noise: bad arguments"); return GMK_ERR_ARG; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    m->last_stream = s;
    const size_t n = static_cast<size_t>(m->n_games);
    if (!m->d_root_prior) GMK_HIP_CHECK(gmk::device_malloc(&m->d_root_prior, n * 225 * sizeof(float)));
    if (m->noise_sampler == GMK_NOISE_SAMPLER_COUNTER) {            // drawn on the device, one wavefront per game (noise_device.h): nothing comes back to the host
        hipLaunchKernelGGL(mcts_root_noise_kernel, dim3(m->n_games), dim3(64), 0, s, m->d_headers, m->d_link, m->d_root_prior, static_cast<size_t>(m->node_capacity), m->n_games,
                           alpha, epsilon, static_cast<uint32_t>(m->seed), static_cast<uint32_t>(m->seed >> 32));
        GMK_HIP_CHECK(hipGetLastError());
        return GMK_OK;
    }
    hipLaunchKernelGGL(mcts_root_flags_kernel, dim3((m->n_games + 255) / 256), dim3(256), 0, s, m->d_headers, m->d_link,
                       static_cast<size_t>(m->node_capacity), m->n_games);
    GMK_HIP_CHECK(hipGetLastError());
    std::vector<GameHeader> hdr(n);
    GMK_HIP_CHECK(hipMemcpyAsync(hdr.data(), m->d_headers, sizeof(GameHeader) * n, hipMemcpyDeviceToHost, s));
    GMK_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<float> prior(n * 225, 0.0f);
    const uint64_t seed = m->seed;
    gmk::for_each_game(n, [&](size_t g) {
        GameHeader& h = hdr[g];
        h.noise = 0;
        if ((h.status & 1u) || !h.root_expanded) return;             // AddNoise is a no-op on a childless root
        const int n_child = 225 - static_cast<int>(h.stones);
        float* p = &prior[g * 225];
        const float uniform = 1.0f / static_cast<float>(n_child);     // Default::UniformProbs: the priors Expand gave the children
        for (int i = 0; i < n_child; ++i) p[i] = uniform;
        gmk::mix_root_noise(p, n_child, alpha, epsilon, gmk::root_noise_engine_seed(seed, h.game_id, h.stones));
        h.noise = 1;
    });
    GMK_HIP_CHECK(hipMemcpyAsync(m->d_root_prior, prior.data(), prior.size() * sizeof(float), hipMemcpyHostToDevice, s));
    GMK_HIP_CHECK(hipMemcpyAsync(m->d_headers, hdr.data(), sizeof(GameHeader) * n, hipMemcpyHostToDevice, s));
    GMK_HIP_CHECK(hipStreamSynchronize(s));
    return GMK_OK;
}

extern "C" int gmk_mcts_launch_info(gmk_mcts* m, int* grid, int* block, int* lds_bytes) {
    if (!m) return GMK_ERR_ARG;
    if (grid) *grid = (m->n_games + m->games_per_block - 1) / m->games_per_block;
    if (block) *block = 64;
    if (lds_bytes) *lds_bytes = static_cast<int>(lds_words(m->games_per_block, m->c_rollouts) * 4) + kMaxGamesPerBlock * 60;
    return GMK_OK;
}

extern "C" int gmk_mcts_root_stats(gmk_mcts* m, uint32_t* h_visits, float* h_root_value, uint32_t* h_root_visits,
                                   uint32_t* h_nodes, int32_t* h_status) {
    if (m && !m->rooted) { gmk::set_error("gmk_mcts_root_stats: gmk_mcts_set_roots has not been called"); return GMK_ERR_STATE; }
    if (!m) return GMK_ERR_ARG;
    const size_t n = static_cast<size_t>(m->n_games);
    uint32_t *d_visits = nullptr, *d_rv = nullptr, *d_nodes = nullptr;
    float* d_q = nullptr;
    int32_t* d_status = nullptr;
    auto cleanup = [&]() { (void)gmk::device_free(d_visits); (void)gmk::device_free(d_rv); (void)gmk::device_free(d_nodes); (void)gmk::device_free(d_q); (void)gmk::device_free(d_status); };
#define GMK_TRY(expr) do { if ((expr) != hipSuccess) { gmk::set_error("%s failed", #expr); cleanup(); return GMK_ERR_HIP; } } while (0)
    GMK_TRY(gmk::device_malloc(&d_visits, n * 225 * 4));
    GMK_TRY(gmk::device_malloc(&d_rv, n * 4));
    GMK_TRY(gmk::device_malloc(&d_nodes, n * 4));
    GMK_TRY(gmk::device_malloc(&d_q, n * 4));
    GMK_TRY(gmk::device_malloc(&d_status, n * 4));
    hipLaunchKernelGGL(mcts_root_stats_kernel, dim3(m->n_games), dim3(64), 0, m->last_stream, m->d_headers, m->d_stats, m->d_link,
                       static_cast<size_t>(m->node_capacity), m->arena_stride(), m->n_games, d_visits, d_q, d_rv, d_nodes, d_status);
    GMK_TRY(hipGetLastError());
    GMK_TRY(hipStreamSynchronize(m->last_stream));
    if (h_visits) GMK_TRY(hipMemcpy(h_visits, d_visits, n * 225 * 4, hipMemcpyDeviceToHost));
    if (h_root_value) GMK_TRY(hipMemcpy(h_root_value, d_q, n * 4, hipMemcpyDeviceToHost));
    if (h_root_visits) GMK_TRY(hipMemcpy(h_root_visits, d_rv, n * 4, hipMemcpyDeviceToHost));
    if (h_nodes) GMK_TRY(hipMemcpy(h_nodes, d_nodes, n * 4, hipMemcpyDeviceToHost));
    if (h_status) GMK_TRY(hipMemcpy(h_status, d_status, n * 4, hipMemcpyDeviceToHost));
#undef GMK_TRY
    cleanup();
    return GMK_OK;
}

extern "C" int gmk_mcts_alg_bytes(gmk_mcts* m, uint64_t* bytes) {
    if (!m || !bytes) return GMK_ERR_ARG;
    std::vector<GameHeader> hdr(static_cast<size_t>(m->n_games));
    GMK_HIP_CHECK(hipStreamSynchronize(m->last_stream));
    GMK_HIP_CHECK(hipMemcpy(hdr.data(), m->d_headers, sizeof(GameHeader) * hdr.size(), hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (const GameHeader& h : hdr) total += h.alg_bytes;
    if (gmk::profile_env("GMK_MCTS_PROFILE")) {
        double p[4] = {0, 0, 0, 0};
        for (const GameHeader& h : hdr) for (int k = 0; k < 4; ++k) p[k] += h.pad[k];
        std::fprintf(stderr, "[gmk profile] mean kilo-cycles per game-slot: select %.0f terminal %.0f rollout %.0f expand+backup %.0f\n",
                     p[0] / hdr.size(), p[1] / hdr.size(), p[2] / hdr.size(), p[3] / hdr.size());
    }
    *bytes = total;
    return GMK_OK;
}

// MCTS::evalState's pi (MCTS.cpp:112-116) with Stats::TempBasedProbs (Statistical.hpp:37-42); host side.
extern "C" int gmk_visits_to_pi(const uint32_t* visits, int stones, float* pi) {
    if (!visits || !pi) return GMK_ERR_ARG;
    const float eps = 1.1920929e-07f;                             // Eigen::NumTraits<float>::epsilon()
    float v[225], sq = 0.0f;
    for (int i = 0; i < 225; ++i) { v[i] = static_cast<float>(visits[i]); sq += v[i] * v[i]; }
    if (sq > 0.0f) { const float nrm = std::sqrt(sq); for (float& x : v) x = x / nrm; }       // VectorXf::normalized()
    for (float& x : v) x = x ? x + 1 : x;
    const float temperature = static_cast<float>(stones < 15 ? 1 : 1e-2);
    double e[225], sum = 0.0;
    for (int i = 0; i < 225; ++i) { e[i] = std::exp(static_cast<double>(std::log(v[i] + eps) / temperature)); sum += e[i]; }
    for (int i = 0; i < 225; ++i) { const float p = static_cast<float>(e[i] / sum); pi[i] = p > eps ? p : 0.0f; }
    return GMK_OK;
}
