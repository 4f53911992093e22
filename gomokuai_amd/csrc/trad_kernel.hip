// trad_kernel.hip -- K6: the pattern-guided tree search (the reference's self-play supervisor) on the device.
//
// TraditionalPolicy (core/lib/include/policies/Traditional.h:17-69) = MCTS::playout (core/lib/src/MCTS.cpp:158-177) with
//   select    RAVE::Select: the first child (MonteCarlo.hpp:149-152)
//   simulate  Heuristic::EvaluationProbs -> DecisiveFilter -> EvaluationValue on the policy's own incremental Evaluator
//             (Heuristic.hpp:16-45, 94-161), no rollout
//   expand    Default::Expand without the legality check: one child per cell with a non-zero prior (MonteCarlo.hpp:71-80)
//   backup    RAVE::BackPropogate<false>: per level, the child with the best PUCB + Q moves to the front (:160-184)
//   moves     Heuristic::CachedApplyMove / CachedRevertMove: the evaluator stays at the last leaf and is rolled back
//             (or rebuilt) only as far as the next path differs (Heuristic.hpp:165-200)
// The search has no random numbers; it is a serial chain per game, so the GPU runs MANY games: one wavefront per
// game, eight games per workgroup, the evaluator state (16.0 KB, evalstate_device.h) and the current path in LDS,
// the tree in HBM (29 B per node: statistics, parent / cell / prior, child range, the record of its FIRST child, and
// its own position in the parent's child order, which is all that BackPropogate's swaps change).  The search walks
// first children only, so the path of the previous playout stays valid down to the shallowest level whose first
// child changed: select re-reads nothing above it, and backup reads every level's children in one round of
// independent loads, the next level's issued before the current one is reduced.  Float reductions use one fixed
// order (sum225, the same as oracle/go_trad.c); PUCB and tanh are evaluated in double like the reference.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "board_device.h"
#include "evalstate_device.h"
#include "philox.h"
#include "root_noise.h"
#include "noise_device.h"
#include "trad_tree.h"

namespace {

using namespace gmk::evs;
using namespace gmk::tree;

#ifndef GMK_TRAD_GAMES
#define GMK_TRAD_GAMES 8
#endif
constexpr int kGamesPerBlock = GMK_TRAD_GAMES;
constexpr int kThreads = 64 * kGamesPerBlock;
constexpr int kPathCap = 228;                    // a path has at most 226 nodes
#ifndef GMK_TRAD_LINK_LDS
#define GMK_TRAD_LINK_LDS 48                      // (a test build with 2 sends every search of the suite through the HBM part)
#endif
constexpr int kLinkLds = GMK_TRAD_LINK_LDS;                     // levels whose child range is kept in LDS; deeper ones (no search of the suite gets there) go through HBM
constexpr int kPathSpill = kPathCap - kLinkLds;
constexpr int kRecordWords = 57;
// per-game LDS: evaluator state | evaluator scratch | path nodes (id | cell << 24) | path child ranges (the first kLinkLds levels) | record copy
// 8 x 18.2 KB + 14.4 KB of tables = 159.9 of a CU's 160 KB: what a search gains from one more game per CU it gains in full (the chains do
// not slow each other: 5 / 6 / 7 games per CU searched at 17.7 / 21.3 / 24.0 M playouts/s)
constexpr int kPerGame = (kStateWords + kScratchWords + kPathCap + kLinkLds + kRecordWords + 3) & ~3;
static_assert(kGamesPerBlock * kPerGame + 2224 + 1248 + gmk::kPrefixWords <= 160 * 256, "LDS of a workgroup (production tables: 2 224 + 1 248 words)");

struct TradParams {
    uint32_t* states;                            // [n_games][kStateWords]
    uint2* stat;                                 // [n_games][cap] {visits, value bits}
    uint2* info;                                 // [n_games][cap] {parent | cell << 24, prior bits}
    uint32_t* link;                              // [n_games][cap] first child | children << 24 (children are consecutive nodes)
    uint2* front;                                // [n_games][cap] the child that is first in the CURRENT order: {id | cell << 24, its link word}
    uint8_t* ord;                                // [n_games][cap] the node's position in its parent's current child order
    TradHeader* hdr;
    uint8_t* moves;                              // [n_games][225] position to search from (read when hdr.fresh; the persistent self-play loop appends to it)
    int32_t* lens;
    const uint32_t* g_trans;
    const uint32_t* g_records;
    int trans_words, record_words;
    uint32_t* path_spill;                        // [n_games][kPathSpill]
    int n_games, cap, playouts;
    int profile;                                 // diagnostic runs only
    double c_puct;
    int selfplay;                                // != 0: every wavefront plays whole games, search after search, until the games have run out (sp)
    TradSelfPlay sp;
};

using gmk::noise::tree_sum;                      // the one summation order of the float reductions (noise_device.h; oracle/go_trad.c: sum225)

struct Cells {                                   // a per-cell float vector: lane l holds cells l + 64 j
    float v[4];
};

__device__ __forceinline__ float sum225(const Cells& x, int lane) {
    float p = x.v[0];
#pragma unroll
    for (int j = 1; j < 4; ++j) if (lane + 64 * j < kCells) p += x.v[j];
    return tree_sum(p);
}

// MatrixBase::normalized() / normalize() (Eigen 3.3+: a zero vector stays as it is)
__device__ __forceinline__ void normalize225(Cells& x, int lane) {
    Cells sq;
#pragma unroll
    for (int j = 0; j < 4; ++j) sq.v[j] = x.v[j] * x.v[j];
    const float z = sum225(sq, lane);
    if (z > 0.0f) {
        const float n = sqrtf(z);
#pragma unroll
        for (int j = 0; j < 4; ++j) x.v[j] = x.v[j] / n;
    }
}

// Heuristic::DensityWeight (Heuristic.hpp:39-45)
__device__ __forceinline__ Cells density_weight(const uint32_t* st, int black, int lane) {
    const uint32_t* packed = st + oDensity + black * kCells;      // count | weight << 16
    Cells out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = min(lane + 64 * j, kCells - 1);
        const uint32_t w = packed[q];
        const float N = static_cast<float>(max(density_count(w), 0)), W = static_cast<float>(max(density_weight_of(w), 0));
        out.v[j] = (3.0f * W) / (1.0f + 2.0f * N);
    }
    normalize225(out, lane);
    return out;
}

// Heuristic::DecisiveFilter (Heuristic.hpp:94-161).  Candidates are (pattern, player is black): pattern < 9 a
// Pattern::Type, otherwise 9 + Compound::Type.  All lanes walk the same automaton; the mask is per cell.
__device__ __forceinline__ void decisive_filter(const uint32_t* st, int cur_black, Cells& probs, int lane) {
    enum { S4, SL3, STo44, STo43, STo33, SEnd };
    // AutomataTable[anti][state] (Heuristic.hpp:103-107) as immediates: next state in nibble anti * 6 + state, next "anti" in
    // bit anti * 6 + state (a table in memory would cost a dependent scalar load per step)
    //   anti 0: {_4,1} {To44,0} {L3,1} {To43,1} {To33,1} {End,0}     anti 1: {L3,0} {To44,1} {To43,0} {To33,0} {End,0} {End,1}
    constexpr unsigned long long kNextNibbles = 0x554321543120ull;
    constexpr uint32_t kAntiBits = 0x89Du;
    int state = S4, anti = 0;
    // the totals the automaton looks at: pattern types 4..7 and the three compound types (one round of LDS reads)
    uint32_t totals[12];
#pragma unroll
    for (int i = 0; i < 8; ++i) totals[i] = i >= 4 ? st[oPdist + pdist_index(225, i)] : 0u;
#pragma unroll
    for (int i = 0; i < 3; ++i) totals[9 + i] = st[oCdist + 225 * 3 + i];
    totals[8] = 0u;
    while (state != SEnd) {
        const uint32_t black = anti ? cur_black ^ 1 : cur_black;
        // the std::deque as 16-bit entries of one register: pattern | black << 8, entry k at bits 16 k
        uint64_t cands;
        int n, head = 0;
        if (state == S4) { cands = (7u | black << 8) | (static_cast<uint64_t>(6u | black << 8) << 16); n = 2; }           // LiveFour, DeadFour
        else if (state == SL3) { cands = 5u | black << 8; n = 1; }                                                       // LiveThree
        else { cands = static_cast<uint32_t>(9 + (STo33 - state)) | black << 8; n = 1; }
        for (; head < n; ++head) {
            const uint32_t pattern = (cands >> (16 * head)) & 0xFFu, pb = (cands >> (16 * head + 8)) & 1u;
            uint32_t total = 0;                                  // totals[pattern], without a dynamically indexed array
#pragma unroll
            for (int i = 4; i < 12; ++i) total = pattern == static_cast<uint32_t>(i) ? totals[i] : total;
            if ((total >> (16 * pb)) & 0xFFFFu) {
                if (anti && state != S4) { cands |= static_cast<uint64_t>(4u | (pb ^ 1u) << 8) << (16 * n); ++n; }       // the own DeadThree counts when answering
                break;
            }
        }
        if (head < n) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = min(lane + 64 * j, kCells - 1);
                bool keep = false;
                for (int k = head; k < n; ++k) {
                    const uint32_t pattern = (cands >> (16 * k)) & 0xFFu, pb = (cands >> (16 * k + 8)) & 1u;
                    const uint32_t field = pattern < 9 ? st[oPdist + pdist_index(q, static_cast<int>(pattern))] : st[oCdist + q * 3 + pattern - 9];
                    keep |= ((field >> (8 * group2(pb, cur_black))) & 0xFFu) != 0u;
                }
                if (!keep) probs.v[j] = 0.0f;
            }
            normalize225(probs, lane);
            state = SEnd;
        } else {
            const int at = anti * 6 + state;
            state = static_cast<int>((kNextNibbles >> (4 * at)) & 15u);
            anti = static_cast<int>((kAntiBits >> at) & 1u);
        }
    }
}

struct Game {
    Ctx c;
    uint32_t* path_node;
    uint32_t* path_link;                             // [kLinkLds] in LDS
    uint32_t* path_spill;                            // [kPathSpill] in HBM
    __device__ __forceinline__ uint32_t link_at(int d) const { return d < kLinkLds ? path_link[d] : path_spill[d - kLinkLds]; }
    __device__ __forceinline__ void set_link(int d, uint32_t v) const { if (d < kLinkLds) path_link[d] = v; else path_spill[d - kLinkLds] = v; }
    uint8_t* record_copy;
    uint2* stat;
    uint2* info;
    uint32_t* link;
    uint2* front;
    uint8_t* ord;
    int cached, init;
    unsigned long long updates;
};

// Default::AddNoise (MonteCarlo.hpp:97-108) on the root (node 0) of one game's tree by ONE wavefront, with the counter-based sampler
// (noise_device.h): the children's priors travel through `cells` (>= 225 words of LDS) into by-cell order -- lane l mixes the cells l + 64 j --
// and back.  A root without children takes none.  info / link: the game's arena.
__device__ __forceinline__ void trad_root_noise(uint2* info, const uint32_t* link, uint32_t* cells, int lane, float alpha, float epsilon,
                                                uint32_t game_id, uint32_t stones, uint32_t seed_lo, uint32_t seed_hi) {
    const uint32_t lk = link[0], first = lk & 0xFFFFFFu, n = lk >> 24;
    if (n == 0u) return;
    for (int i = lane; i < kCells; i += 64) cells[i] = 0u;
    wave_phase_fence();
    uint2 inf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = lane + 64 * k;
        inf[k] = i < n ? info[first + i] : make_uint2(0u, 0u);
        if (i < n) cells[inf[k].x >> 24] = inf[k].y;
    }
    wave_phase_fence();
    float p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = lane + 64 * j < kCells ? __uint_as_float(cells[lane + 64 * j]) : 0.0f;
    gmk::noise::mix_root_priors(p, lane, alpha, epsilon, game_id, stones, seed_lo, seed_hi);
#pragma unroll
    for (int j = 0; j < 4; ++j) if (lane + 64 * j < kCells) cells[lane + 64 * j] = __float_as_uint(p[j]);
    wave_phase_fence();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = lane + 64 * k;
        if (i < n) info[first + i] = make_uint2(inf[k].x, cells[inf[k].x >> 24]);
    }
}

template <bool kWaveOnly> __device__ uint32_t copy_subtree(const TradArena& a, const TradArena& b, size_t base, uint32_t src_root, int lane);
template <bool kWaveOnly> __device__ __forceinline__ void copy_sync();

// kSelfPlay = false: one search per game (gmk_trad_run); true: the persistent self-play loop (gmk_trad_selfplay_run, persistent = 1).  Two
// instantiations, so that the bare search does not carry the registers of the turn loop through its playouts.
template <bool kSelfPlay>
__global__ __launch_bounds__(kThreads)
void trad_playouts_kernel(TradParams prm) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // layout: [trans][records, then the four-symbol prefix table: record_words counts both][games: kGamesPerBlock * kPerGame]
    for (int i = threadIdx.x; i < prm.trans_words + prm.record_words; i += kThreads)
        lds[i] = i < prm.trans_words ? prm.g_trans[i] : prm.g_records[i - prm.trans_words];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int game = blockIdx.x * kGamesPerBlock + wave;
    uint32_t* base = lds + prm.trans_words + prm.record_words + wave * kPerGame;
    if (game < prm.n_games) {
        const uint4* src = reinterpret_cast<const uint4*>(prm.states + static_cast<size_t>(game) * kStateWords);
        uint4* dst = reinterpret_cast<uint4*>(base);
        for (int i = lane; i < kStateWords / 4; i += 64) dst[i] = src[i];
    }
    __syncthreads();                                            // tables staged; the games of a block never wait for each other again
    if (game >= prm.n_games) return;
    if (prm.hdr[game].status & kStatusIdleSlot) return;         // continuous batching: a slot whose games have run out (trad_advance_kernel)

    Game g;
    g.c = Ctx{base, base + kStateWords, reinterpret_cast<const char*>(lds), reinterpret_cast<const uint4*>(lds + prm.trans_words),
              reinterpret_cast<const char*>(lds + prm.trans_words + prm.record_words - gmk::kPrefixWords), lane};
    g.path_node = base + kStateWords + kScratchWords;
    g.path_link = g.path_node + kPathCap;
    g.path_spill = prm.path_spill + static_cast<size_t>(game) * kPathSpill;
    g.record_copy = reinterpret_cast<uint8_t*>(g.path_link + kLinkLds);
    const size_t arena = static_cast<size_t>(game) * prm.cap;
    uint32_t live = 0;                                          // which of the slot's two arenas holds its tree (the persistent loop with kept subtrees flips it)
    auto use_arena = [&](uint32_t which) {
        const size_t at = arena + ((kSelfPlay && which) ? prm.sp.arena_stride : 0);
        g.stat = prm.stat + at; g.info = prm.info + at; g.link = prm.link + at; g.front = prm.front + at; g.ord = prm.ord + at;
    };
    use_arena(0);
    g.updates = 0;
    TradHeader* hdr = prm.hdr + game;
    int32_t* meta = reinterpret_cast<int32_t*>(g.c.st + oMeta);
    const uint8_t* record = reinterpret_cast<const uint8_t*>(g.c.st + oRecord);
    uint32_t n_nodes = hdr->n_nodes, status = hdr->status;
    int root_black = hdr->root_black;
    uint32_t fresh_mode = hdr->fresh;
    // The persistent self-play loop (gmk_trad_selfplay_run, mode 1): this wavefront plays whole games at its own pace -- one turn of the loop
    // below = Policy::prepare + one search + MCTS::stepForward()'s move + the end-of-game check, a finished game's slot takes the next
    // unstarted game from a global counter (its evaluator starts from the empty board, so a game's record does not depend on the slot it
    // landed in) -- instead of every slot waiting at every move for the slowest search of the batch.  Otherwise: one turn.
    int sp_game = kSelfPlay ? prm.sp.slot_game[game] : -1, cur_len = prm.lens[game];
    uint8_t* const slot_moves = prm.moves + static_cast<size_t>(game) * 225;
    unsigned long long playouts_run = hdr->playouts_done;
    unsigned long long prof_sel = 0, prof_sim = 0, prof_back = 0, prof_t0 = 0, prof_all = (gmk::kProfileBuild && prm.profile) ? __builtin_amdgcn_s_memtime() : 0ull;
    for (;;) {
    const bool fresh = fresh_mode == 1u;

    // The evaluator's work of one step is a script: take `n_revert` moves back, then apply script[0 .. n_apply) (bytes in LDS).
    // It runs at ONE place in the loop below (evaluator_step is ~20 KB of code); iteration -1 of that loop is
    // Policy::prepare + TraditionalPolicy::prepare, i.e. Evaluator::syncWithBoard (Pattern.cpp:356-368).
    uint8_t* script = g.record_copy;
    int n_revert = 0, n_apply = 0;
    bool rebuild = false;
    const int n_position = fresh_mode != 0u ? cur_len : 0;
    if (fresh_mode != 0u) {
        // syncWithBoard: back to the first move that differs (or to the position's length), then the rest of the position
        const uint8_t* mv = slot_moves;
        const int have = meta[0], common = min(have, n_position);
        int i0 = common;
        for (int i = lane; i < common; i += 64) if (record[i] != mv[i]) i0 = min(i0, i);
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) i0 = min(i0, __shfl_xor(i0, sft));
        n_revert = have - i0;
        n_apply = n_position - i0;
        for (int j = lane; j < n_apply; j += 64) script[j] = mv[i0 + j];
        g.updates += static_cast<unsigned long long>(n_revert + n_apply);
        g.init = n_position;
        root_black = n_position & 1;                            // the player of the last move
    } else {
        g.init = static_cast<int>(hdr->init_acts);
    }
    g.cached = g.init;

    // path[0 .. valid] is known to be the chain of first children from the root (node id | cell << 24, child range)
    int valid = 0;
    if (lane == 0) { g.path_node[0] = 0u; g.set_link(0, fresh ? 0u : g.link[0]); }
    wave_phase_fence();

    struct Level {                                              // what backup needs of one path node: its statistics and its children
        uint2 ns, cs[4], ci[4];
        uint32_t cl[4], co[4];
    };
    auto load_level = [&](int d) {
        Level L;
        const uint32_t nd = g.path_node[d] & 0xFFFFFFu, lk = g.link_at(d), first = lk & 0xFFFFFFu, n = lk >> 24;
        L.ns = g.stat[nd];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = lane + 64 * k, id = first + i;
            if (i < n) { L.cs[k] = g.stat[id]; L.ci[k] = g.info[id]; L.cl[k] = g.link[id]; L.co[k] = g.ord[id]; }
            else { L.cs[k] = make_uint2(0u, 0u); L.ci[k] = make_uint2(0u, 0u); L.cl[k] = 0u; L.co[k] = 0xFFFFFFFFu; }
        }
        return L;
    };

    for (int it = fresh_mode != 0u ? -1 : 0; it < prm.playouts && !(status & 1u); ++it) {
        if (gmk::kProfileBuild && prm.profile) prof_t0 = __builtin_amdgcn_s_memtime();
        int depth = 0;
        uint32_t node = 0, link = 0;
        if (it >= 0) {
            // ---- select: always the first child in the current order (RAVE::Select): the tree walk needs no evaluator ----
            depth = valid;
            node = g.path_node[depth] & 0xFFFFFFu;
            link = g.link_at(depth);
            while (link >> 24) {
                const uint2 rec = g.front[node];
                node = rec.x & 0xFFFFFFu;
                link = rec.y;
                ++depth;
                if (lane == 0) { g.path_node[depth] = rec.x; g.set_link(depth, link); }
            }
            wave_phase_fence();
            // ---- Heuristic::CachedApplyMove (Heuristic.hpp:165-189) for the moves of the path: the ones the evaluator's record
            //      already holds are skipped; at the first one it does not hold, it is rolled back to there (or rebuilt from the
            //      empty board when less than half of it would survive), and everything from there on is applied ----
            const int nrec = meta[0];
            int d0 = depth + 1;
            for (int d = 1 + lane; d <= depth; d += 64) {
                const int at = g.init + d - 1;
                if (at >= nrec || record[at] != (g.path_node[d] >> 24)) d0 = min(d0, d);
            }
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) d0 = min(d0, __shfl_xor(d0, sft));
            n_revert = n_apply = 0;
            rebuild = false;
            if (d0 <= depth) {
                const int c0 = g.init + d0 - 1, tail = depth - d0 + 1;
                if (nrec - c0 > c0) {                           // too little is cached: rebuild from the empty board
                    rebuild = true;
                    for (int j = lane; j < c0; j += 64) script[j] = record[j];
                    for (int j = lane; j < tail; j += 64) script[c0 + j] = static_cast<uint8_t>(g.path_node[d0 + j] >> 24);
                    n_apply = c0 + tail;
                    g.updates += static_cast<unsigned long long>(c0);
                } else {
                    n_revert = nrec - c0;
                    for (int j = lane; j < tail; j += 64) script[j] = static_cast<uint8_t>(g.path_node[d0 + j] >> 24);
                    n_apply = tail;
                    g.updates += static_cast<unsigned long long>(n_revert);
                }
                g.updates += static_cast<unsigned long long>(tail);
            } else {
                g.cached = g.init + depth;
            }
            wave_phase_fence();
        }
        // ---- the evaluator follows: the one place where moves are applied and taken back ----
        if (rebuild) reset_state(g.c);
        for (int i = 0; i < n_revert + n_apply; ++i) {
            evaluator_step(g.c, i < n_revert ? kRevert : static_cast<int>(script[i - n_revert]));
            wave_phase_fence();
        }
        if (it < 0) {                                           // the prologue ends here: a fresh root where one was asked for
            if (fresh) {
                if (lane == 0) {
                    g.stat[0] = make_uint2(0u, 0u);
                    g.info[0] = make_uint2(kNoParent | ((n_position ? slot_moves[n_position - 1] : 255u) << 24), __float_as_uint(1.0f));
                    g.link[0] = 0u;
                    g.set_link(0, 0u);
                }
                n_nodes = 1;
                status = 0;
            }
            n_revert = n_apply = 0;
            wave_phase_fence();
            continue;
        }
        if (n_revert + n_apply > 0 || rebuild) g.cached = meta[0];      // from the first move it did not hold on, the record follows the path
        if (gmk::kProfileBuild && prm.profile) { const unsigned long long t = __builtin_amdgcn_s_memtime(); prof_sel += t - prof_t0; prof_t0 = t; }
        int path_len = depth;                                   // deepest level with a known node

        // ---- TraditionalPolicy::checkGameEnd -> Evaluator::checkGameEnd (Pattern.cpp:344-354) ----
        bool ended = meta[1] == 0;
        if (!ended && meta[0] == kCells) {
            if (lane == 0) { meta[1] = 0; meta[2] = 0; }
            wave_phase_fence();
            ended = true;
        }
        float value;                                            // for the player of `node`
        if (!ended) {
            // ---- hybridSimulate (Traditional.h:48-69) ----
            const int cur_black = meta[1] > 0;
            Cells probs;
            const Cells dw_self = density_weight(g.c.st, cur_black, lane), dw_rival = density_weight(g.c.st, cur_black ^ 1, lane);
            const int32_t* scores = reinterpret_cast<const int32_t*>(g.c.st + oScores);
            Cells prod_self, prod_rival;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = min(lane + 64 * j, kCells - 1);
                const float self_worthy = static_cast<float>(scores[group2(cur_black, cur_black) * kCells + q]) * dw_self.v[j];
                const float rival_anti = static_cast<float>(scores[group2(cur_black ^ 1, cur_black) * kCells + q]) * dw_rival.v[j];
                probs.v[j] = 0.6f * self_worthy + 0.4f * rival_anti;                   // EvaluationProbs (Heuristic.hpp:16-28)
                prod_self.v[j] = self_worthy;
                prod_rival.v[j] = static_cast<float>(scores[group2(cur_black ^ 1, cur_black ^ 1) * kCells + q]) * dw_rival.v[j];
            }
            if (meta[0] != 0) {
                normalize225(probs, lane);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) probs.v[j] = (lane + 64 * j == 7 * 15 + 7) ? 1.0f : 0.0f;
            }
            decisive_filter(g.c.st, cur_black, probs, lane);
            const float self_sum = sum225(prod_self, lane), rival_sum = sum225(prod_rival, lane);
            const float state_value = static_cast<float>(tanh((1.2 * self_sum - rival_sum) / 500.0f));     // EvaluationValue (:33-37)
            value = -state_value;

            // ---- Default::Expand, extraCheck = false: children in ascending cell order ----
            int total = 0, rank[4], first_cell = -1;
            bool nz[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                nz[j] = lane + 64 * j < kCells && probs.v[j] != 0.0f;
                const unsigned long long b = __ballot(nz[j]);
                rank[j] = total + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(b >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(b), 0u)));
                if (first_cell < 0 && b) first_cell = 64 * j + __ffsll(static_cast<unsigned long long>(b)) - 1;
                total += __popcll(b);
            }
            if (total > 0) {
                if (n_nodes + total > static_cast<uint32_t>(prm.cap)) {
                    status |= 1u;                               // arena full: the search of this game stops here
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (nz[j]) {
                            const uint32_t child = n_nodes + rank[j];
                            g.stat[child] = make_uint2(0u, 0u);
                            g.info[child] = make_uint2(node | (static_cast<uint32_t>(lane + 64 * j) << 24), __float_as_uint(probs.v[j]));
                            g.link[child] = 0u;
                            g.ord[child] = static_cast<uint8_t>(rank[j]);
                        }
                    link = n_nodes | (static_cast<uint32_t>(total) << 24);
                    if (lane == 0) {
                        const uint32_t front_rec = n_nodes | (static_cast<uint32_t>(first_cell) << 24);
                        g.link[node] = link;
                        g.front[node] = make_uint2(front_rec, 0u);
                        if (depth > 0) g.front[g.path_node[depth - 1] & 0xFFFFFFu] = make_uint2(g.path_node[depth], link);      // the parent's record of this node
                        g.set_link(depth, link);
                        g.path_node[depth + 1] = front_rec;
                        g.set_link(depth + 1, 0u);
                    }
                    path_len = depth + 1;
                    n_nodes += total;
                }
            }
        } else {
            const int node_player = ((depth & 1) ? !root_black : root_black) ? 1 : -1;
            value = static_cast<float>(node_player * meta[2]);                          // CalcScore (Game.h:34-36)
        }
        wave_phase_fence();
        if (status & 1u) break;
        if (gmk::kProfileBuild && prm.profile) { const unsigned long long t = __builtin_amdgcn_s_memtime(); prof_sim += t - prof_t0; prof_t0 = t; }

        // ---- RAVE::BackPropogate<false> (MonteCarlo.hpp:160-184), leaf to root ----
        int swap_level = -1;
        uint2 swap_rec = make_uint2(0u, 0u);
        uint32_t updated_id = 0xFFFFFFFFu;                      // the path node one level below: its statistics were just rewritten
        uint2 updated_stat = make_uint2(0u, 0u);
        Level cur = load_level(depth);
        for (int d = depth; d >= 0; --d, value = -value) {
            Level nxt;
            if (d > 0) nxt = load_level(d - 1);                 // in flight while this level is reduced
            const uint32_t nd = g.path_node[d] & 0xFFFFFFu, lk = g.link_at(d);
            const uint32_t first = lk & 0xFFFFFFu, n = lk >> 24;
            const double sqrt_n = sqrt(static_cast<double>(cur.ns.x));
            double best_score = -INFINITY;
            uint32_t best_ord = 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t i = lane + 64 * k;
                if (i >= n) continue;
                uint2 cs = cur.cs[k];
                if (first + i == updated_id) cs = updated_stat;
                const double p_i = __uint_as_float(cur.ci[k].y), n_i = static_cast<double>(cs.x + 1u);
                double score = prm.c_puct * p_i * sqrt_n / n_i;                        // Default::PUCB (:23-28)
                score += __uint_as_float(cs.y);
                // the reference scans the children in their current order and keeps the first maximum
                if (score > best_score || (score == best_score && cur.co[k] < best_ord)) { best_score = score; best_ord = cur.co[k]; }
            }
            // the best score of the wave, then the first child (in the current order) that reaches it
            const double top = wave_max(best_score);
            best_ord = wave_min(best_score == top ? best_ord : 0xFFFFFFFFu);
            if (n && best_ord != 0xFFFFFFFFu && best_ord != 0u) {                      // the best child moves to the front: it swaps places with the first one
                uint2 mine = make_uint2(0u, 0u);
                bool owner = false;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (lane + 64 * k < n && cur.co[k] == best_ord) { owner = true; mine = make_uint2((first + lane + 64 * k) | (cur.ci[k].x & 0xFF000000u), cur.cl[k]); }
                const int src = __ffsll(static_cast<unsigned long long>(__ballot(owner))) - 1;
                const uint2 rec = make_uint2(__shfl(mine.x, src), __shfl(mine.y, src));
                if (lane == 0) {
                    g.ord[rec.x & 0xFFFFFFu] = 0;
                    g.ord[g.path_node[d + 1] & 0xFFFFFFu] = static_cast<uint8_t>(best_ord);
                    g.front[nd] = rec;
                }
                swap_level = d;
                swap_rec = rec;
            }
            const uint32_t visits = cur.ns.x + 1u;
            const float q = __uint_as_float(cur.ns.y);
            updated_id = nd;
            updated_stat = make_uint2(visits, __float_as_uint(q + (value - q) / static_cast<float>(visits)));
            if (lane == 0) g.stat[nd] = updated_stat;
            cur = nxt;
        }
        wave_phase_fence();
        if (swap_level >= 0) {                                  // below the shallowest swap the chain of first children is a different one
            if (lane == 0) { g.path_node[swap_level + 1] = swap_rec.x; g.set_link(swap_level + 1, swap_rec.y); }
            valid = swap_level + 1;
        } else {
            valid = path_len;
        }
        wave_phase_fence();

        if (gmk::kProfileBuild && prm.profile) prof_back += __builtin_amdgcn_s_memtime() - prof_t0;

        // ---- Heuristic::CachedRevertMove (Heuristic.hpp:192-200) ----
        if (meta[0] != g.cached) status |= 4u;                  // the reference would take stones off the inner board only: not reproduced
        g.cached = g.init;
    }

    wave_phase_fence();
    if (meta[3]) status |= 2u;
    playouts_run = (fresh ? 0ull : playouts_run) + static_cast<unsigned long long>(prm.playouts);
    if (!kSelfPlay || sp_game < 0) break;

    // ---- the move: MCTS::stepForward()'s choice (the most visited child, first in the current order), the root's visit counts into the
    //      game's record, Board::applyMove with its victory check (Game.cpp:37-49, 88-136); as trad_advance_kernel, inside the wavefront ----
    {
        const TradSelfPlay& sp = prm.sp;
        if (lane == 0 && (status & 1u)) atomicOr(sp.overflow, 1);
        const uint32_t lk = g.link[0], first = lk & 0xFFFFFFu, n = lk >> 24;
        uint32_t best_visits = 0, best_ord = 0xFFFFFFFFu, best_id = 0;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t id = first + i, o = g.ord[id], v = g.stat[id].x + 1u;
            if (v > best_visits || (v == best_visits && o < best_ord)) { best_visits = v; best_ord = o; best_id = id; }
        }
        for (int sft = 32; sft > 0; sft >>= 1) {
            const uint32_t ov = __shfl_down(best_visits, sft), oo = __shfl_down(best_ord, sft), oi = __shfl_down(best_id, sft);
            if (ov > best_visits || (ov == best_visits && ov != 0u && oo < best_ord)) { best_visits = ov; best_ord = oo; best_id = oi; }
        }
        best_visits = __shfl(best_visits, 0);
        best_id = __shfl(best_id, 0);
        bool over = best_visits == 0u || cur_len >= 225;        // no child: nothing the policy wants to play: the game ends where it stands
        int winner = 0;
        if (!over) {
            const uint32_t cell = g.info[best_id].x >> 24;
            if (sp.rec_visits) {
                uint16_t* rv = sp.rec_visits + (static_cast<size_t>(sp_game) * 225 + static_cast<size_t>(cur_len)) * 225;
                for (int i = lane; i < 225; i += 64) rv[i] = 0;
                wave_phase_fence();
                for (uint32_t i = lane; i < n; i += 64) rv[g.info[first + i].x >> 24] = static_cast<uint16_t>(min(g.stat[first + i].x, 65535u));
            }
            uint32_t* rows = g.path_node;                       // (the path is rebuilt by the next search: sixteen words of it hold the board's rows here)
            if (lane < 16) rows[lane] = 0u;
            wave_phase_fence();
            for (int i = lane; i < cur_len; i += 64) atomicOr(&rows[slot_moves[i] / 15u], 1u << (slot_moves[i] % 15u + ((i & 1) ? 16u : 0u)));
            const int shift = (cur_len & 1) ? 16 : 0;
            if (lane == 0) atomicOr(&rows[cell / 15u], 1u << (cell % 15u + shift));
            wave_phase_fence();
            const bool five = gmk::five_through<1>(rows, static_cast<int>(cell % 15u), static_cast<int>(cell / 15u), shift);
            if (lane == 0) {
                slot_moves[cur_len] = static_cast<uint8_t>(cell);
                sp.rec_moves[static_cast<size_t>(sp_game) * 225 + cur_len] = static_cast<uint8_t>(cell);
                sp.rec_lens[sp_game] = cur_len + 1;
            }
            over = five || cur_len + 1 == 225;
            winner = five ? (shift ? -1 : 1) : 0;
            ++cur_len;
        }
        bool kept = false;
        if (!over && sp.reuse) {
            // MCTS::stepForward (MCTS.cpp:129-134): the chosen child's subtree is the next search's tree -- compacted into the slot's other arena
            const TradArena a{g.stat, g.info, g.link, g.front, g.ord, nullptr};
            use_arena(live ^ 1u);
            const TradArena b{g.stat, g.info, g.link, g.front, g.ord, nullptr};
            n_nodes = copy_subtree<true>(a, b, 0, best_id, lane);
            live ^= 1u;
            root_black ^= 1;
            kept = true;
            if (sp.noise_alpha > 0.0f) {                        // Default::AddNoise before the next search (MCTS.cpp:182); the path buffer is idle between two searches
                copy_sync<true>();
                trad_root_noise(g.info, g.link, g.path_node, lane, sp.noise_alpha, sp.noise_epsilon, sp.first_game_id + static_cast<uint32_t>(sp_game),
                                static_cast<uint32_t>(cur_len), sp.seed_lo, sp.seed_hi);
            }
        }
        if (over) {
            int next = 0;
            if (lane == 0) {
                sp.rec_winner[sp_game] = static_cast<int8_t>(winner);
                next = atomicAdd(sp.next_game, 1);
            }
            next = __shfl(next, 0);
            if (next >= sp.n_total) {                           // the games have run out: the slot is done
                if (lane == 0) sp.slot_game[game] = -1;
                sp_game = -1;
                status |= kStatusIdleSlot;
                break;
            }
            sp_game = next;
            cur_len = sp.open_lens ? sp.open_lens[next] : 0;
            for (int i = lane; i < cur_len; i += 64) slot_moves[i] = sp.open_moves[static_cast<size_t>(next) * sp.open_stride + i];
            if (lane == 0) { sp.slot_game[game] = next; sp.game_ids[game] = static_cast<uint32_t>(next); }
            reset_state(g.c);                                   // a new game on a fresh evaluator (Evaluator::reset)
            status &= ~(1u | 8u);
        }
        // the slot's move list is read again by the next turn's syncWithBoard, by other lanes than the one that wrote it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        fresh_mode = kept ? 2u : 1u;                            // 2: the tree stays, the evaluator follows the move (gmk_trad_step's state)
    }
    }
    if (lane == 0) {
        if (kSelfPlay) prm.lens[game] = cur_len;
        hdr->n_nodes = n_nodes;
        hdr->init_acts = static_cast<uint32_t>(g.init);
        hdr->status = status;
        hdr->fresh = kSelfPlay ? 1u : 0u;
        hdr->root_black = static_cast<uint32_t>(root_black);
        hdr->playouts_done = static_cast<uint32_t>(playouts_run);
        hdr->evaluator_updates += g.updates;
        if (gmk::kProfileBuild && prm.profile) {
            hdr->prof[0] = static_cast<uint32_t>(prof_sel >> 10); hdr->prof[1] = static_cast<uint32_t>(prof_sim >> 10);
            hdr->prof[2] = static_cast<uint32_t>(prof_back >> 10); hdr->prof[3] = static_cast<uint32_t>((__builtin_amdgcn_s_memtime() - prof_all) >> 10);
        }
    }
    uint4* dst = reinterpret_cast<uint4*>(prm.states + static_cast<size_t>(game) * kStateWords);
    const uint4* src = reinterpret_cast<const uint4*>(g.c.st);
    for (int i = lane; i < kStateWords / 4; i += 64) dst[i] = src[i];
}

// MCTS::stepForward() / stepForward(move) (MCTS.cpp:129-147): the chosen child's subtree becomes the tree.  The reference
// frees the siblings; here the kept subtree is copied level by level into the other arena so that node indices stay dense
// (children consecutive, the root at 0).  A copied node carries its OLD child range and OLD first-child record until the
// scan reaches it, copies its children and rewrites both.  One wavefront per game.

// The subtree of node src_root of arena a becomes the tree of arena b, level by level (see above); one wavefront.
// kWaveOnly: the caller is one wavefront of a larger workgroup (the persistent self-play loop of trad_playouts_kernel): its lanes meet at a
// wavefront barrier with workgroup-scope fences (the CU's L1 is shared by the workgroup) instead of __syncthreads().  Returns the node count.
template <bool kWaveOnly>
__device__ __forceinline__ void copy_sync() {
    if constexpr (kWaveOnly) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __syncthreads();
    }
}

template <bool kWaveOnly>
__device__ uint32_t copy_subtree(const TradArena& a, const TradArena& b, size_t base, uint32_t src_root, int lane) {
    if (lane == 0) {
        b.stat[base] = a.stat[base + src_root];
        b.info[base] = make_uint2(kNoParent | (a.info[base + src_root].x & 0xFF000000u), a.info[base + src_root].y);
        b.link[base] = a.link[base + src_root];
        b.front[base] = a.front[base + src_root];
        b.ord[base] = 0;
        if (a.amaf) b.amaf[base] = a.amaf[base + src_root];
    }
    copy_sync<kWaveOnly>();
    uint32_t next = 1;
    for (uint32_t i0 = 0, chunk = 0; i0 < next; i0 += chunk) {
        chunk = min(64u, next - i0);                            // nodes appended while this chunk is handled come after it
        const uint32_t old_link = static_cast<uint32_t>(lane) < chunk ? b.link[base + i0 + lane] : 0u;
        unsigned long long todo = __ballot((old_link >> 24) != 0u);
        while (todo) {
            const int j = __ffsll(static_cast<long long>(todo)) - 1;
            todo &= todo - 1ull;
            const uint32_t ol = __shfl(old_link, j), of = ol & 0xFFFFFFu, nk = ol >> 24, node = i0 + static_cast<uint32_t>(j);
            for (uint32_t k = lane; k < nk; k += 64) {
                const uint2 inf = a.info[base + of + k];
                b.stat[base + next + k] = a.stat[base + of + k];
                b.info[base + next + k] = make_uint2(node | (inf.x & 0xFF000000u), inf.y);
                b.link[base + next + k] = a.link[base + of + k];
                b.front[base + next + k] = a.front[base + of + k];
                b.ord[base + next + k] = a.ord[base + of + k];
                if (a.amaf) b.amaf[base + next + k] = a.amaf[base + of + k];
            }
            if (lane == 0) {
                const uint32_t new_link = next | (nk << 24);
                b.link[base + node] = new_link;
                const uint2 fr = b.front[base + node];              // still the OLD id of the first child in the current order
                b.front[base + node] = make_uint2((next + ((fr.x & 0xFFFFFFu) - of)) | (fr.x & 0xFF000000u), fr.y);
                if (node != 0u) {                                   // am I my parent's first child?  then its record of me carries my child range
                    const uint32_t p = b.info[base + node].x & 0xFFFFFFu;
                    const uint2 pf = b.front[base + p];
                    if ((pf.x & 0xFFFFFFu) == node) b.front[base + p] = make_uint2(pf.x, new_link);
                }
            }
            next += nk;
            copy_sync<kWaveOnly>();
        }
        copy_sync<kWaveOnly>();                                        // children written above are scanned below
    }
    return next;
}


__global__ __launch_bounds__(64)
void trad_step_kernel(TradArena a, TradArena b, TradHeader* hdrs, int cap, int n_games, const int16_t* forced, uint8_t* moves, int32_t* lens) {
    const int game = blockIdx.x, lane = threadIdx.x;
    if (game >= n_games) return;
    TradHeader& hdr = hdrs[game];
    const size_t base = static_cast<size_t>(game) * cap;
    uint8_t* mv = moves + static_cast<size_t>(game) * 225;
    const int len = lens[game];
    if (hdr.fresh == 1u) {                                      // the position was set but never searched: its root node does not exist yet
        if (lane == 0) {
            a.stat[base] = make_uint2(0u, 0u);
            a.info[base] = make_uint2(kNoParent | ((len ? mv[len - 1] : 255u) << 24), __float_as_uint(1.0f));
            a.link[base] = 0u;
            if (a.amaf) a.amaf[base] = make_uint2(0u, 0u);
            hdr.n_nodes = 1; hdr.status = 0; hdr.root_black = static_cast<uint32_t>(len & 1);
        }
        __syncthreads();
    }
    const uint32_t lk = a.link[base], first = lk & 0xFFFFFFu, n = lk >> 24;
    int want = forced ? forced[game] : -1;
    // the child to keep: the one of the wanted move, or the most visited one, first in the current order
    uint32_t best_visits = 0, best_ord = 0xFFFFFFFFu, best_id = 0;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t id = first + i, cell = a.info[base + id].x >> 24, o = a.ord[base + id];
        const uint32_t v = want >= 0 ? (cell == static_cast<uint32_t>(want) ? 1u : 0u) : a.stat[base + id].x + 1u;
        if (v > best_visits || (v == best_visits && v != 0u && o < best_ord)) { best_visits = v; best_ord = o; best_id = id; }
    }
    for (int s = 32; s > 0; s >>= 1) {
        const uint32_t ov = __shfl_down(best_visits, s), oo = __shfl_down(best_ord, s), oi = __shfl_down(best_id, s);
        if (ov > best_visits || (ov == best_visits && ov != 0u && oo < best_ord)) { best_visits = ov; best_ord = oo; best_id = oi; }
    }
    best_visits = __shfl(best_visits, 0);
    best_id = __shfl(best_id, 0);
    const bool found = best_visits != 0u;
    if (want < 0 && !found) {                                   // stepForward() on a childless root: nothing moves (MCTS.cpp:133)
        if (lane == 0) {
            b.stat[base] = a.stat[base]; b.info[base] = a.info[base]; b.link[base] = 0u; b.ord[base] = 0; hdr.n_nodes = 1; hdr.fresh = 2;
            if (a.amaf) b.amaf[base] = a.amaf[base];
        }
        return;
    }
    const uint32_t cell = found ? a.info[base + best_id].x >> 24 : static_cast<uint32_t>(want);
    bool legal = cell < 225u && len < 225;
    for (int i = lane; i < len; i += 64) legal &= mv[i] != cell;
    legal = __all(legal);
    if (!legal) {                                               // not a move of this game: the tree stays as it is (copied over, the arenas flip for everybody)
        if (lane == 0) hdr.status |= 8u;
        want = -2;
    }
    if (lane == 0 && want != -2) { mv[len] = static_cast<uint8_t>(cell); lens[game] = len + 1; hdr.fresh = 2; hdr.root_black ^= 1u; }
    const uint32_t src_root = want == -2 ? 0u : best_id;
    if (want != -2 && !found) {                                 // stepForward(move) without such a child: a new node (MCTS.cpp:140-145)
        if (lane == 0) {
            b.stat[base] = make_uint2(0u, 0u); b.info[base] = make_uint2(kNoParent | (cell << 24), __float_as_uint(1.0f));
            b.link[base] = 0u; b.ord[base] = 0; hdr.n_nodes = 1;
            if (a.amaf) b.amaf[base] = make_uint2(0u, 0u);
        }
        return;
    }
    const uint32_t kept = copy_subtree<false>(a, b, base, src_root, lane);
    if (lane == 0) hdr.n_nodes = kept;
}

// The step of the device-resident self-play loop (gmk_trad_selfplay_run): what play_supervisor_games does on the host after every search --
// MCTS::stepForward()'s choice (MCTS.cpp:129-134), the root visit counts into the game's record, Board::applyMove with its victory check
// (Game.cpp:37-49, 88-136) -- and the hand-over of a finished game's slot (tree arena, evaluator, wavefront) to the next unstarted game
// (network/data_helper.py:56-83 plays its games one after the other per worker; here a slot does).  One wavefront per slot.
__global__ __launch_bounds__(64)
void trad_advance_kernel(TradArena a, TradArena b, TradHeader* hdrs, int cap, int n_slots, uint8_t* moves, int32_t* lens, TradSelfPlay sp, int reuse) {
    __shared__ uint32_t s_rows[16];
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= n_slots) return;
    const int game = sp.slot_game[slot];
    if (game < 0) return;                                       // idle slot
    TradHeader& hdr = hdrs[slot];
    const size_t base = static_cast<size_t>(slot) * cap;
    uint8_t* mv = moves + static_cast<size_t>(slot) * 225;
    const int len = lens[slot];
    if (lane == 0 && (hdr.status & 1u)) atomicOr(sp.overflow, 1);
    // the child to keep: the most visited one, first in the current order (as trad_step_kernel / trad_root_stats_kernel)
    const bool searched = hdr.fresh != 1u;                      // (a position that was never searched has no root node yet)
    const uint32_t lk = searched ? a.link[base] : 0u, first = lk & 0xFFFFFFu, n = lk >> 24;
    uint32_t best_visits = 0, best_ord = 0xFFFFFFFFu, best_id = 0;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t id = first + i, o = a.ord[base + id], v = a.stat[base + id].x + 1u;
        if (v > best_visits || (v == best_visits && o < best_ord)) { best_visits = v; best_ord = o; best_id = id; }
    }
    for (int sft = 32; sft > 0; sft >>= 1) {
        const uint32_t ov = __shfl_down(best_visits, sft), oo = __shfl_down(best_ord, sft), oi = __shfl_down(best_id, sft);
        if (ov > best_visits || (ov == best_visits && ov != 0u && oo < best_ord)) { best_visits = ov; best_ord = oo; best_id = oi; }
    }
    best_visits = __shfl(best_visits, 0);
    best_id = __shfl(best_id, 0);
    bool over = best_visits == 0u || len >= 225;                // no child: nothing the policy wants to play: the game ends where it stands
    int winner = 0;
    if (!over) {
        const uint32_t cell = a.info[base + best_id].x >> 24;
        if (sp.rec_visits) {                                    // the root's visit counts by cell, the searched ply's row of the game's record
            uint16_t* rv = sp.rec_visits + (static_cast<size_t>(game) * 225 + static_cast<size_t>(len)) * 225;
            for (int i = lane; i < 225; i += 64) rv[i] = 0;
            __syncthreads();
            for (uint32_t i = lane; i < n; i += 64) rv[a.info[base + first + i].x >> 24] = static_cast<uint16_t>(min(a.stat[base + first + i].x, 65535u));
        }
        // the board after the move (move i is black's when i is even), five or more through it wins, a full board is a tie
        if (lane < 16) s_rows[lane] = 0u;
        __syncthreads();
        for (int i = lane; i < len; i += 64) atomicOr(&s_rows[mv[i] / 15u], 1u << (mv[i] % 15u + ((i & 1) ? 16u : 0u)));
        const int shift = (len & 1) ? 16 : 0;
        if (lane == 0) atomicOr(&s_rows[cell / 15u], 1u << (cell % 15u + shift));
        __syncthreads();
        const bool five = gmk::five_through<1>(s_rows, static_cast<int>(cell % 15u), static_cast<int>(cell / 15u), shift);
        if (lane == 0) {
            mv[len] = static_cast<uint8_t>(cell);
            lens[slot] = len + 1;
            sp.rec_moves[static_cast<size_t>(game) * 225 + len] = static_cast<uint8_t>(cell);
            sp.rec_lens[game] = len + 1;
        }
        over = five || len + 1 == 225;
        winner = five ? (shift ? -1 : 1) : 0;
        if (!over) {
            if (lane == 0) {
                atomicAdd(sp.unfinished, 1);
                if (!reuse) { hdr.fresh = 1; hdr.playouts_done = 0; }          // a new root at the next search (gmk_trad_set_positions)
                else { hdr.fresh = 2; hdr.root_black ^= 1u; }                  // the subtree is kept (gmk_trad_step)
            }
            if (reuse) {
                const uint32_t kept = copy_subtree<false>(a, b, base, best_id, lane);
                if (lane == 0) hdr.n_nodes = kept;
            }
            return;
        }
    }
    // the game is over: its winner; the slot waits for trad_refill_kernel
    if (lane == 0) {
        sp.rec_winner[game] = static_cast<int8_t>(winner);
        sp.slot_game[slot] = -2;
    }
}

// Finished games hand their slots to the next unstarted games, in ascending slot order (the order play_supervisor_games' host loop uses:
// which slot a game lands in decides which evaluator history it inherits, so the order is part of the result); the opening of the new
// game is the slot's new position.  Slots left without a game go idle.  One wavefront for all slots.
__global__ __launch_bounds__(64)
void trad_refill_kernel(TradHeader* hdrs, int n_slots, uint8_t* moves, int32_t* lens, TradSelfPlay sp) {
    const int lane = threadIdx.x;
    int next0 = *sp.next_game;
    for (int s0 = 0; s0 < n_slots; s0 += 64) {
        const int slot = s0 + lane;
        const bool finished = slot < n_slots && sp.slot_game[slot] == -2;
        const unsigned long long mask = __ballot(finished);
        if (finished) {
            const int next = next0 + static_cast<int>(__popcll(mask & ((1ull << lane) - 1ull)));
            TradHeader& hdr = hdrs[slot];
            if (next < sp.n_total) {
                uint8_t* mv = moves + static_cast<size_t>(slot) * 225;
                const int olen = sp.open_lens ? sp.open_lens[next] : 0;
                for (int i = 0; i < olen; ++i) mv[i] = sp.open_moves[static_cast<size_t>(next) * sp.open_stride + i];     // (the game's record holds its opening already)
                lens[slot] = olen;
                sp.slot_game[slot] = next;
                sp.game_ids[slot] = static_cast<uint32_t>(next);
                hdr.fresh = 1; hdr.playouts_done = 0;
                hdr.status &= ~(1u | 8u);
                atomicAdd(sp.unfinished, 1);
            } else {
                sp.slot_game[slot] = -1;
                hdr.status |= kStatusIdleSlot;
            }
        }
        next0 = min(sp.n_total, next0 + static_cast<int>(__popcll(mask)));
    }
    if (lane == 0) *sp.next_game = next0;
}

// Default::AddNoise (MonteCarlo.hpp:97-108): the root children's priors, by cell, as the host computed them
__global__ __launch_bounds__(64)
void trad_set_root_priors_kernel(TradArena a, const TradHeader* hdrs, int cap, const float* priors) {
    const int game = blockIdx.x, lane = threadIdx.x;
    const size_t base = static_cast<size_t>(game) * cap;
    if (hdrs[game].fresh == 1u) return;                         // no root node yet: nothing to mix noise into
    const uint32_t lk = a.link[base], first = lk & 0xFFFFFFu, n = lk >> 24;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint2 inf = a.info[base + first + i];
        a.info[base + first + i] = make_uint2(inf.x, __float_as_uint(priors[static_cast<size_t>(game) * 225 + (inf.x >> 24)]));
    }
}

// Default::AddNoise with the counter-based sampler, one wavefront per game (the lock-step form of what the persistent loop does inside its launch)
__global__ __launch_bounds__(64)
void trad_root_noise_kernel(TradArena a, const TradHeader* hdrs, int cap, const int32_t* lens, const uint32_t* game_ids, uint32_t first_game_id,
                            float alpha, float epsilon, uint32_t seed_lo, uint32_t seed_hi) {
    __shared__ uint32_t s_cells[kCells];
    const int game = blockIdx.x, lane = threadIdx.x;
    if (hdrs[game].fresh == 1u || (hdrs[game].status & kStatusIdleSlot)) return;       // no root node yet / no game: nothing to mix noise into
    const size_t base = static_cast<size_t>(game) * cap;
    trad_root_noise(a.info + base, a.link + base, s_cells, lane, alpha, epsilon, first_game_id + game_ids[game], static_cast<uint32_t>(lens[game]), seed_lo, seed_hi);
}

// root statistics by cell and the child MCTS::stepForward would pick (most visited, first in the CURRENT order)
__global__ __launch_bounds__(64)
void trad_root_stats_kernel(const uint2* stat, const uint2* info, const uint32_t* link, const uint8_t* ord, const TradHeader* hdrs, int cap,
                            uint32_t* visits, float* values, float* priors, int32_t* best, uint32_t* root_visits, float* root_value) {
    const int game = blockIdx.x, lane = threadIdx.x;
    const size_t arena = static_cast<size_t>(game) * cap;
    const bool no_root = hdrs[game].fresh == 1u;                // the position was set but never searched
    const uint32_t lk = no_root ? 0u : link[arena], first = lk & 0xFFFFFFu, n = lk >> 24;
    uint32_t best_visits = 0, best_ord = 0xFFFFFFFFu, best_cell = 0;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint2 cs = stat[arena + first + i], ci = info[arena + first + i];
        const uint32_t cell = ci.x >> 24, o = ord[arena + first + i];
        if (visits) visits[static_cast<size_t>(game) * 225 + cell] = cs.x;
        if (values) values[static_cast<size_t>(game) * 225 + cell] = __uint_as_float(cs.y);
        if (priors) priors[static_cast<size_t>(game) * 225 + cell] = __uint_as_float(ci.y);
        if (best_ord == 0xFFFFFFFFu || cs.x > best_visits || (cs.x == best_visits && o < best_ord)) { best_visits = cs.x; best_ord = o; best_cell = cell; }
    }
    for (int s = 32; s > 0; s >>= 1) {
        const uint32_t ov = __shfl_down(best_visits, s), oo = __shfl_down(best_ord, s), oc = __shfl_down(best_cell, s);
        if (oo != 0xFFFFFFFFu && (best_ord == 0xFFFFFFFFu || ov > best_visits || (ov == best_visits && oo < best_ord))) { best_visits = ov; best_ord = oo; best_cell = oc; }
    }
    if (lane == 0) {
        if (best) best[game] = best_ord == 0xFFFFFFFFu ? -1 : static_cast<int32_t>(best_cell);
        if (root_visits) root_visits[game] = no_root ? 0u : stat[arena].x;
        if (root_value) root_value[game] = no_root ? 0.0f : __uint_as_float(stat[arena].y);
    }
}

}  // namespace


extern "C" int gmk_trad_destroy(gmk_trad* t) {
    if (!t) return GMK_OK;
    if (t->paired) {                                            // both arenas are halves of the five blocks
        (void)gmk::device_free(t->block_stat); (void)gmk::device_free(t->block_info); (void)gmk::device_free(t->block_link); (void)gmk::device_free(t->block_front); (void)gmk::device_free(t->block_ord);
        t->d_stat = t->d_stat2 = t->d_info = t->d_info2 = t->d_front = t->d_front2 = nullptr; t->d_link = t->d_link2 = nullptr; t->d_ord = t->d_ord2 = nullptr;
    }
    (void)gmk::device_free(t->d_states); (void)gmk::device_free(t->d_stat); (void)gmk::device_free(t->d_info); (void)gmk::device_free(t->d_link);
    (void)gmk::device_free(t->d_front); (void)gmk::device_free(t->d_ord); (void)gmk::device_free(t->d_stat2); (void)gmk::device_free(t->d_info2); (void)gmk::device_free(t->d_front2);
    (void)gmk::device_free(t->d_link2); (void)gmk::device_free(t->d_ord2); (void)gmk::device_free(t->d_amaf); (void)gmk::device_free(t->d_amaf2); (void)gmk::device_free(t->d_forced); (void)gmk::device_free(t->d_priors); (void)gmk::device_free(t->d_hdr); (void)gmk::device_free(t->d_moves); (void)gmk::device_free(t->d_lens); (void)gmk::device_free(t->d_game_ids); (void)gmk::device_free(t->d_path_spill);
    delete t;
    return GMK_OK;
}

extern "C" int gmk_trad_reset_evaluators(gmk_trad* t) {
    if (!t) return GMK_ERR_ARG;
    std::vector<uint32_t> all(static_cast<size_t>(t->n_games) * kStateWords, 0u);
    for (int g = 0; g < t->n_games; ++g) fill_initial_state(all.data() + static_cast<size_t>(g) * kStateWords);
    GMK_HIP_CHECK(hipMemcpy(t->d_states, all.data(), all.size() * 4, hipMemcpyHostToDevice));
    return GMK_OK;
}

extern "C" int gmk_trad_create(int n_games, int node_capacity, gmk_trad** out) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (!out || n_games <= 0 || node_capacity < 256 || node_capacity >= (1 << 24)) { gmk::set_error("gmk_trad_create: bad arguments (256 <= node_capacity < 2^24)"); return GMK_ERR_ARG; }
    gmk_trad* t = new gmk_trad;
    t->n_games = n_games;
    t->cap = node_capacity;
    const size_t nodes = static_cast<size_t>(n_games) * node_capacity;
    bool ok = gmk::device_malloc(&t->d_states, static_cast<size_t>(n_games) * kStateWords * 4) == hipSuccess &&
              gmk::device_malloc(&t->d_stat, nodes * 8) == hipSuccess && gmk::device_malloc(&t->d_info, nodes * 8) == hipSuccess &&
              gmk::device_malloc(&t->d_link, nodes * 4) == hipSuccess && gmk::device_malloc(&t->d_front, nodes * 8) == hipSuccess && gmk::device_malloc(&t->d_ord, nodes) == hipSuccess &&
              gmk::device_malloc(&t->d_hdr, static_cast<size_t>(n_games) * sizeof(TradHeader)) == hipSuccess &&
              gmk::device_malloc(&t->d_moves, static_cast<size_t>(n_games) * 225) == hipSuccess &&
              gmk::device_malloc(&t->d_lens, static_cast<size_t>(n_games) * 4) == hipSuccess &&
              gmk::device_malloc(&t->d_game_ids, static_cast<size_t>(n_games) * 4) == hipSuccess &&
              gmk::device_malloc(&t->d_path_spill, static_cast<size_t>(n_games) * kPathSpill * 4) == hipSuccess;
    if (ok) ok = hipMemset(t->d_hdr, 0, static_cast<size_t>(n_games) * sizeof(TradHeader)) == hipSuccess;
    if (ok) {
        t->game_ids.resize(static_cast<size_t>(n_games));
        for (int g = 0; g < n_games; ++g) t->game_ids[static_cast<size_t>(g)] = static_cast<uint32_t>(g);
        ok = hipMemcpy(t->d_game_ids, t->game_ids.data(), static_cast<size_t>(n_games) * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok || gmk_trad_reset_evaluators(t) != GMK_OK) { gmk_trad_destroy(t); gmk::set_error("gmk_trad_create: device allocation failed"); return GMK_ERR_HIP; }
    *out = t;
    return GMK_OK;
}

extern "C" int gmk_trad_set_option(gmk_trad* t, int option, int value) {
    if (!t) { gmk::set_error("gmk_trad_set_option: bad arguments"); return GMK_ERR_ARG; }
    if (option == GMK_OPT_NOISE_SAMPLER && (value == GMK_NOISE_SAMPLER_STD || value == GMK_NOISE_SAMPLER_COUNTER)) { t->noise_sampler = value; return GMK_OK; }
    if (option == GMK_OPT_LOCKSTEP && (value == 0 || value == 1)) { t->lockstep = value; return GMK_OK; }
    gmk::set_error("gmk_trad_set_option: unknown option %d or value %d", option, value);
    return GMK_ERR_ARG;
}

// Both arenas of every slot as the two halves of ONE block per array (the persistent loop with kept subtrees reaches a slot's other arena by a
// fixed node stride).  The trees are lost (the evaluators are not): the caller positions the games afterwards.
static int pair_arenas(gmk_trad* t) {
    const size_t nodes = static_cast<size_t>(t->n_games) * static_cast<size_t>(t->cap);
    auto point = [&]() {
        t->d_stat = t->block_stat; t->d_info = t->block_info; t->d_link = t->block_link; t->d_front = t->block_front; t->d_ord = t->block_ord;
        t->d_stat2 = t->block_stat + nodes; t->d_info2 = t->block_info + nodes; t->d_link2 = t->block_link + nodes; t->d_front2 = t->block_front + nodes; t->d_ord2 = t->block_ord + nodes;
    };
    if (t->paired) { point(); return GMK_OK; }                  // (the lock-step loop may have left the halves swapped)
    if (t->d_amaf) { gmk::set_error("gmk_trad_selfplay_run: the persistent loop keeps subtrees for TraditionalPolicy handles only"); return GMK_ERR_STATE; }
    uint2 *bs = nullptr, *bi = nullptr, *bf = nullptr;
    uint32_t* bl = nullptr;
    uint8_t* bo = nullptr;
    (void)gmk::device_free(t->d_stat); (void)gmk::device_free(t->d_info); (void)gmk::device_free(t->d_link); (void)gmk::device_free(t->d_front); (void)gmk::device_free(t->d_ord);
    (void)gmk::device_free(t->d_stat2); (void)gmk::device_free(t->d_info2); (void)gmk::device_free(t->d_link2); (void)gmk::device_free(t->d_front2); (void)gmk::device_free(t->d_ord2);
    t->d_stat = t->d_stat2 = t->d_info = t->d_info2 = t->d_front = t->d_front2 = nullptr; t->d_link = t->d_link2 = nullptr; t->d_ord = t->d_ord2 = nullptr;
    t->second_arena = false;
    t->positioned = false;
    const bool ok = gmk::device_malloc(&bs, 2 * nodes * 8) == hipSuccess && gmk::device_malloc(&bi, 2 * nodes * 8) == hipSuccess && gmk::device_malloc(&bl, 2 * nodes * 4) == hipSuccess &&
                    gmk::device_malloc(&bf, 2 * nodes * 8) == hipSuccess && gmk::device_malloc(&bo, 2 * nodes) == hipSuccess;
    if (!ok) {
        (void)gmk::device_free(bs); (void)gmk::device_free(bi); (void)gmk::device_free(bl); (void)gmk::device_free(bf); (void)gmk::device_free(bo);
        (void)hipGetLastError();
        // the handle must stay usable: one arena, as gmk_trad_create leaves it
        if (gmk::device_malloc(&t->d_stat, nodes * 8) != hipSuccess || gmk::device_malloc(&t->d_info, nodes * 8) != hipSuccess || gmk::device_malloc(&t->d_link, nodes * 4) != hipSuccess ||
            gmk::device_malloc(&t->d_front, nodes * 8) != hipSuccess || gmk::device_malloc(&t->d_ord, nodes) != hipSuccess)
            (void)hipGetLastError();
        gmk::set_error("gmk_trad_selfplay_run: hipMalloc of two arenas per slot (%zu nodes, %.1f GB) failed", 2 * nodes, 2 * nodes * 29.0 / 1e9);
        return GMK_ERR_HIP;
    }
    t->block_stat = bs; t->block_info = bi; t->block_link = bl; t->block_front = bf; t->block_ord = bo;
    t->paired = true;
    t->second_arena = true;
    if (!t->d_forced && gmk::device_malloc(&t->d_forced, static_cast<size_t>(t->n_games) * 2) != hipSuccess) { (void)hipGetLastError(); gmk::set_error("gmk_trad_selfplay_run: hipMalloc failed"); return GMK_ERR_HIP; }
    point();
    return GMK_OK;
}

// gmk_mcts_reserve for a K6 handle: two_arenas != 0 makes the two arenas per slot of the persistent loop with kept subtrees now (the handle must
// be positioned again afterwards); 0 is a no-op (gmk_trad_create allocates the one arena).
extern "C" int gmk_trad_reserve(gmk_trad* t, int two_arenas) {
    if (!t) { gmk::set_error("gmk_trad_reserve: bad arguments"); return GMK_ERR_ARG; }
    return two_arenas ? pair_arenas(t) : GMK_OK;
}

extern "C" int gmk_trad_set_game_ids(gmk_trad* t, const uint32_t* h_ids) {
    if (!t || !h_ids) { gmk::set_error("gmk_trad_set_game_ids: bad arguments"); return GMK_ERR_ARG; }
    GMK_HIP_CHECK(hipDeviceSynchronize());
    t->game_ids.assign(h_ids, h_ids + t->n_games);
    GMK_HIP_CHECK(hipMemcpy(t->d_game_ids, h_ids, static_cast<size_t>(t->n_games) * 4, hipMemcpyHostToDevice));
    return GMK_OK;
}

extern "C" int gmk_trad_set_positions(gmk_trad* t, const uint8_t* h_moves, const int32_t* h_lens) {
    if (!t || !h_moves || !h_lens) { gmk::set_error("gmk_trad_set_positions: bad arguments"); return GMK_ERR_ARG; }
    bool all = true;
    for (int g = 0; g < t->n_games; ++g) {
        if (h_lens[g] > 225) { gmk::set_error("gmk_trad_set_positions: game %d has %d moves", g, h_lens[g]); return GMK_ERR_ARG; }
        all &= h_lens[g] >= 0;
    }
    if (!all && !t->positioned) { gmk::set_error("gmk_trad_set_positions: the first call must position every game (a negative length keeps a game as it is)"); return GMK_ERR_ARG; }
    const size_t n = static_cast<size_t>(t->n_games);
    GMK_HIP_CHECK(hipDeviceSynchronize());
    std::vector<TradHeader> hdr(n);
    GMK_HIP_CHECK(hipMemcpy(hdr.data(), t->d_hdr, n * sizeof(TradHeader), hipMemcpyDeviceToHost));
    if (all) {
        GMK_HIP_CHECK(hipMemcpy(t->d_moves, h_moves, n * 225, hipMemcpyHostToDevice));
        GMK_HIP_CHECK(hipMemcpy(t->d_lens, h_lens, n * 4, hipMemcpyHostToDevice));
        for (TradHeader& h : hdr) { h.fresh = 1; h.playouts_done = 0; h.status &= ~kStatusIdleSlot; }
    } else {                                                    // a negative length: that game keeps its position and its tree
        std::vector<uint8_t> moves(n * 225);
        std::vector<int32_t> lens(n);
        GMK_HIP_CHECK(hipMemcpy(moves.data(), t->d_moves, n * 225, hipMemcpyDeviceToHost));
        GMK_HIP_CHECK(hipMemcpy(lens.data(), t->d_lens, n * 4, hipMemcpyDeviceToHost));
        for (size_t g = 0; g < n; ++g)
            if (h_lens[g] >= 0) {
                std::memcpy(&moves[g * 225], h_moves + g * 225, 225);
                lens[g] = h_lens[g];
                hdr[g].fresh = 1; hdr[g].playouts_done = 0; hdr[g].status &= ~kStatusIdleSlot;
            }
        GMK_HIP_CHECK(hipMemcpy(t->d_moves, moves.data(), n * 225, hipMemcpyHostToDevice));
        GMK_HIP_CHECK(hipMemcpy(t->d_lens, lens.data(), n * 4, hipMemcpyHostToDevice));
    }
    GMK_HIP_CHECK(hipMemcpy(t->d_hdr, hdr.data(), n * sizeof(TradHeader), hipMemcpyHostToDevice));
    t->positioned = true;
    return GMK_OK;
}

extern "C" int gmk_trad_run(gmk_trad* t, int playouts, double c_puct, void* stream) {
    gmk::DeviceState& st = gmk::device_state();
    if (!t || playouts < 0) { gmk::set_error("gmk_trad_run: bad arguments"); return GMK_ERR_ARG; }
    if (!t->positioned) { gmk::set_error("gmk_trad_run: gmk_trad_set_positions has not been called"); return GMK_ERR_STATE; }
    if (t->policy == 2) { gmk::set_error("gmk_trad_run: this handle searches with gmk_trad_run_poolrave (its evaluators are not kept in step)"); return GMK_ERR_STATE; }
    t->policy = 1;
    const size_t lds = static_cast<size_t>(kGamesPerBlock * kPerGame + st.n_states * 4 + st.n_records * 4 + gmk::kPrefixWords) * 4;
    if (lds > 160u * 1024u) { gmk::set_error("gmk_trad_run: tables do not fit in LDS (%zu bytes)", lds); return GMK_ERR_CAPACITY; }
    if (!t->attr_set) {
        GMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trad_playouts_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        t->attr_set = true;
    }
    TradParams prm;
    prm.states = t->d_states; prm.stat = t->d_stat; prm.info = t->d_info; prm.link = t->d_link; prm.front = t->d_front; prm.ord = t->d_ord; prm.hdr = t->d_hdr;
    prm.moves = t->d_moves; prm.lens = t->d_lens;
    prm.g_trans = st.d_trans; prm.g_records = st.d_records; prm.trans_words = st.n_states * 4; prm.record_words = st.n_records * 4 + gmk::kPrefixWords; prm.path_spill = t->d_path_spill;
    prm.n_games = t->n_games; prm.cap = t->cap; prm.playouts = playouts; prm.c_puct = c_puct;
    prm.selfplay = 0; prm.sp = TradSelfPlay{};
    static const bool profile = gmk::profile_env("GMK_TRAD_PROFILE") != nullptr;
    prm.profile = profile ? 1 : 0;
    const int grid = (t->n_games + kGamesPerBlock - 1) / kGamesPerBlock;
    hipLaunchKernelGGL(trad_playouts_kernel<false>, dim3(grid), dim3(kThreads), lds, static_cast<hipStream_t>(stream), prm);
    GMK_HIP_CHECK(hipGetLastError());
    if (profile) {                                              // share of a search spent per stage, mean over games
        std::vector<TradHeader> hdr(static_cast<size_t>(t->n_games));
        GMK_HIP_CHECK(hipDeviceSynchronize());
        GMK_HIP_CHECK(hipMemcpy(hdr.data(), t->d_hdr, hdr.size() * sizeof(TradHeader), hipMemcpyDeviceToHost));
        double sum[4] = {};
        for (const TradHeader& h : hdr) for (int k = 0; k < 4; ++k) sum[k] += h.prof[k];
        std::fprintf(stderr, "[GMK_TRAD_PROFILE] select + evaluator moves %.1f %%, simulate + expand %.1f %%, backup %.1f %% of the kernel's clocks\n",
                     100 * sum[0] / sum[3], 100 * sum[1] / sum[3], 100 * sum[2] / sum[3]);
    }
    return GMK_OK;
}

extern "C" int gmk_trad_root_stats(gmk_trad* t, uint32_t* h_visits, float* h_values, float* h_priors, int32_t* h_best,
                                   uint32_t* h_root_visits, float* h_root_value, int32_t* h_n_nodes, int32_t* h_status,
                                   uint64_t* h_evaluator_updates);

extern "C" int gmk_trad_step(gmk_trad* t, const int16_t* h_moves) {
    if (!t) { gmk::set_error("gmk_trad_step: bad arguments"); return GMK_ERR_ARG; }
    if (!t->positioned) { gmk::set_error("gmk_trad_step: gmk_trad_set_positions has not been called"); return GMK_ERR_STATE; }
    const size_t n = static_cast<size_t>(t->n_games), nodes = n * static_cast<size_t>(t->cap);
    if (!t->second_arena) {
        const bool ok = gmk::device_malloc(&t->d_stat2, nodes * 8) == hipSuccess && gmk::device_malloc(&t->d_info2, nodes * 8) == hipSuccess &&
                        gmk::device_malloc(&t->d_link2, nodes * 4) == hipSuccess && gmk::device_malloc(&t->d_front2, nodes * 8) == hipSuccess &&
                        gmk::device_malloc(&t->d_ord2, nodes) == hipSuccess && gmk::device_malloc(&t->d_forced, n * 2) == hipSuccess;
        if (!ok) {                                                  // all or nothing: a later call must not find half an arena
            (void)gmk::device_free(t->d_stat2); (void)gmk::device_free(t->d_info2); (void)gmk::device_free(t->d_link2); (void)gmk::device_free(t->d_front2); (void)gmk::device_free(t->d_ord2); (void)gmk::device_free(t->d_forced);
            t->d_stat2 = t->d_info2 = t->d_front2 = nullptr; t->d_link2 = nullptr; t->d_ord2 = nullptr; t->d_forced = nullptr;
            (void)hipGetLastError();
            gmk::set_error("gmk_trad_step: hipMalloc of the second arena (%zu nodes) failed", nodes);
            return GMK_ERR_HIP;
        }
        t->second_arena = true;
    }
    GMK_HIP_CHECK(hipDeviceSynchronize());
    if (h_moves) GMK_HIP_CHECK(hipMemcpy(t->d_forced, h_moves, n * 2, hipMemcpyHostToDevice));
    if (t->d_amaf && !t->d_amaf2 && gmk::device_malloc(&t->d_amaf2, nodes * 8) != hipSuccess) { gmk::set_error("gmk_trad_step: hipMalloc of the second arena (%zu nodes) failed", nodes); return GMK_ERR_HIP; }
    const TradArena a = t->arena(), b = t->arena2();
    hipLaunchKernelGGL(trad_step_kernel, dim3(t->n_games), dim3(64), 0, nullptr, a, b, t->d_hdr, t->cap, t->n_games,
                       h_moves ? t->d_forced : nullptr, t->d_moves, t->d_lens);
    GMK_HIP_CHECK(hipGetLastError());
    GMK_HIP_CHECK(hipDeviceSynchronize());
    std::swap(t->d_stat, t->d_stat2); std::swap(t->d_info, t->d_info2); std::swap(t->d_link, t->d_link2);
    std::swap(t->d_front, t->d_front2); std::swap(t->d_ord, t->d_ord2); std::swap(t->d_amaf, t->d_amaf2);
    return GMK_OK;
}

// Default::AddNoise (MonteCarlo.hpp:97-108) on every root that has children (root_noise.h)
extern "C" int gmk_trad_add_root_noise(gmk_trad* t, float alpha, float epsilon, uint64_t seed, uint32_t first_game_id) {
    if (!t || !(alpha > 0.0f)) { gmk::set_error("gmk_trad_add_root_noise: bad arguments"); return GMK_ERR_ARG; }
    if (!t->positioned) { gmk::set_error("gmk_trad_add_root_noise: gmk_trad_set_positions has not been called"); return GMK_ERR_STATE; }
    const size_t n = static_cast<size_t>(t->n_games);
    if (t->noise_sampler == GMK_NOISE_SAMPLER_COUNTER) {        // drawn on the device, one wavefront per game: nothing comes back to the host
        GMK_HIP_CHECK(hipMemcpy(t->d_game_ids, t->game_ids.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(trad_root_noise_kernel, dim3(t->n_games), dim3(64), 0, nullptr, t->arena(), t->d_hdr, t->cap, t->d_lens, t->d_game_ids, first_game_id,
                           alpha, epsilon, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
        GMK_HIP_CHECK(hipGetLastError());
        GMK_HIP_CHECK(hipDeviceSynchronize());
        return GMK_OK;
    }
    std::vector<float> priors(n * 225);
    std::vector<int32_t> lens(n);
    int rc = gmk_trad_root_stats(t, nullptr, nullptr, priors.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc != GMK_OK) return rc;
    GMK_HIP_CHECK(hipMemcpy(lens.data(), t->d_lens, n * 4, hipMemcpyDeviceToHost));
    gmk::for_each_game(n, [&](size_t g) {
        float* p = &priors[g * 225];
        bool any = false;
        for (int i = 0; i < 225; ++i) any |= p[i] != 0.0f;
        if (any) gmk::mix_root_noise(p, 225, alpha, epsilon, gmk::root_noise_engine_seed(seed, first_game_id + t->game_ids[g], static_cast<uint32_t>(lens[g])));
    });
    if (!t->d_priors) GMK_HIP_CHECK(gmk::device_malloc(&t->d_priors, n * 225 * 4));
    GMK_HIP_CHECK(hipMemcpy(t->d_priors, priors.data(), n * 225 * 4, hipMemcpyHostToDevice));
    const TradArena a = t->arena();
    hipLaunchKernelGGL(trad_set_root_priors_kernel, dim3(t->n_games), dim3(64), 0, nullptr, a, t->d_hdr, t->cap, t->d_priors);
    GMK_HIP_CHECK(hipGetLastError());
    GMK_HIP_CHECK(hipDeviceSynchronize());
    return GMK_OK;
}

extern "C" int gmk_trad_root_stats(gmk_trad* t, uint32_t* h_visits, float* h_values, float* h_priors, int32_t* h_best,
                                   uint32_t* h_root_visits, float* h_root_value, int32_t* h_n_nodes, int32_t* h_status,
                                   uint64_t* h_evaluator_updates) {
    if (!t) { gmk::set_error("gmk_trad_root_stats: bad arguments"); return GMK_ERR_ARG; }
    if (!t->positioned) { gmk::set_error("gmk_trad_root_stats: gmk_trad_set_positions has not been called"); return GMK_ERR_STATE; }
    const size_t n = static_cast<size_t>(t->n_games);
    uint32_t *d_visits = nullptr, *d_root_visits = nullptr;
    float *d_values = nullptr, *d_priors = nullptr, *d_root_value = nullptr;
    int32_t* d_best = nullptr;
    auto cleanup = [&]() { (void)gmk::device_free(d_visits); (void)gmk::device_free(d_values); (void)gmk::device_free(d_priors); (void)gmk::device_free(d_best); (void)gmk::device_free(d_root_visits); (void)gmk::device_free(d_root_value); };
#define GMK_TRY(expr) do { if ((expr) != hipSuccess) { gmk::set_error("%s failed", #expr); cleanup(); return GMK_ERR_HIP; } } while (0)
    GMK_TRY(gmk::device_malloc(&d_visits, n * 225 * 4)); GMK_TRY(gmk::device_malloc(&d_values, n * 225 * 4)); GMK_TRY(gmk::device_malloc(&d_priors, n * 225 * 4));
    GMK_TRY(gmk::device_malloc(&d_best, n * 4)); GMK_TRY(gmk::device_malloc(&d_root_visits, n * 4)); GMK_TRY(gmk::device_malloc(&d_root_value, n * 4));
    GMK_TRY(hipMemset(d_visits, 0, n * 225 * 4)); GMK_TRY(hipMemset(d_values, 0, n * 225 * 4)); GMK_TRY(hipMemset(d_priors, 0, n * 225 * 4));
    hipLaunchKernelGGL(trad_root_stats_kernel, dim3(t->n_games), dim3(64), 0, nullptr, t->d_stat, t->d_info, t->d_link, t->d_ord, t->d_hdr, t->cap,
                       d_visits, d_values, d_priors, d_best, d_root_visits, d_root_value);
    GMK_TRY(hipGetLastError());
    GMK_TRY(hipDeviceSynchronize());
    if (h_visits) GMK_TRY(hipMemcpy(h_visits, d_visits, n * 225 * 4, hipMemcpyDeviceToHost));
    if (h_values) GMK_TRY(hipMemcpy(h_values, d_values, n * 225 * 4, hipMemcpyDeviceToHost));
    if (h_priors) GMK_TRY(hipMemcpy(h_priors, d_priors, n * 225 * 4, hipMemcpyDeviceToHost));
    if (h_best) GMK_TRY(hipMemcpy(h_best, d_best, n * 4, hipMemcpyDeviceToHost));
    if (h_root_visits) GMK_TRY(hipMemcpy(h_root_visits, d_root_visits, n * 4, hipMemcpyDeviceToHost));
    if (h_root_value) GMK_TRY(hipMemcpy(h_root_value, d_root_value, n * 4, hipMemcpyDeviceToHost));
    std::vector<TradHeader> hdr(n);
    GMK_TRY(hipMemcpy(hdr.data(), t->d_hdr, n * sizeof(TradHeader), hipMemcpyDeviceToHost));
#undef GMK_TRY
    for (size_t g = 0; g < n; ++g) {
        if (h_n_nodes) h_n_nodes[g] = static_cast<int32_t>(hdr[g].n_nodes);
        if (h_status) h_status[g] = static_cast<int32_t>(hdr[g].status);
        if (h_evaluator_updates) h_evaluator_updates[g] = hdr[g].evaluator_updates;
    }
    cleanup();
    return GMK_OK;
}

extern "C" int gmk_trad_read_evaluators(gmk_trad* t, int32_t* h_scores, int32_t* h_density, uint32_t* h_pattern_dist, uint32_t* h_compound_dist,
                                        int32_t* h_meta, uint8_t* h_record) {
    if (!t) return GMK_ERR_ARG;
    std::vector<uint32_t> all(static_cast<size_t>(t->n_games) * kStateWords);
    GMK_HIP_CHECK(hipDeviceSynchronize());
    GMK_HIP_CHECK(hipMemcpy(all.data(), t->d_states, all.size() * 4, hipMemcpyDeviceToHost));
    for (int g = 0; g < t->n_games; ++g) {
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

// The self-play loop of network/data_helper.py:56-83 for the pattern-guided searchers, resident on the device: n_total games through the
// handle's slots, every move = Default::AddNoise (if asked for) + one search of `playouts` playouts for every slot that has a game +
// trad_advance_kernel; the host sees four bytes per move (slots that still play) and, with root noise, the slots' game numbers.
extern "C" int gmk_trad_selfplay_run(gmk_trad* t, int poolrave, int n_total, uint32_t first_game_id, int playouts, double c_puct, uint64_t seed,
                                     int reuse_subtree, float noise_alpha, float noise_epsilon,
                                     const uint8_t* h_open_moves, int open_stride, const int32_t* h_open_lens,
                                     uint8_t* d_moves, uint16_t* d_visits, int32_t* d_lens, int8_t* d_winner, int persistent, int max_steps, int32_t* h_overflow, int32_t* h_steps, void* stream) {
    const bool noisy = noise_alpha > 0.0f && reuse_subtree;     // (a new root has no children: AddNoise is a no-op without kept subtrees)
    if (t && t->lockstep) persistent = 0;
    if (persistent && (poolrave || max_steps > 0 || (noisy && t && t->noise_sampler != GMK_NOISE_SAMPLER_COUNTER))) {
        gmk::set_error("gmk_trad_selfplay_run: the persistent loop plays TraditionalPolicy games to their end; root noise inside it comes from the counter-based sampler (GMK_OPT_NOISE_SAMPLER)");
        return GMK_ERR_ARG;
    }
    if (!t || n_total <= 0 || playouts < 0 || max_steps < 0 || !d_moves || !d_lens || !d_winner || (h_open_moves && (!h_open_lens || open_stride <= 0))) {
        gmk::set_error("gmk_trad_selfplay_run: bad arguments");
        return GMK_ERR_ARG;
    }
    const int n_slots = t->n_games;
    const size_t ns = static_cast<size_t>(n_slots), nt = static_cast<size_t>(n_total);
    std::vector<int32_t> open_lens(nt, 0);
    if (h_open_moves)
        for (size_t g = 0; g < nt; ++g) {
            if (h_open_lens[g] < 0 || h_open_lens[g] > 224 || h_open_lens[g] > open_stride) { gmk::set_error("gmk_trad_selfplay_run: opening of game %zu has %d moves", g, h_open_lens[g]); return GMK_ERR_ARG; }
            open_lens[g] = h_open_lens[g];
            for (int i = 0; i < open_lens[g]; ++i)
                if (h_open_moves[g * open_stride + i] >= 225) { gmk::set_error("gmk_trad_selfplay_run: opening of game %zu holds cell %d", g, h_open_moves[g * open_stride + i]); return GMK_ERR_ARG; }
        }
    hipStream_t s = static_cast<hipStream_t>(stream);
    GMK_HIP_CHECK(hipDeviceSynchronize());
    // the first min(n_slots, n_total) games start in the slots, from their openings; the rest wait for a slot
    const int started = std::min(n_slots, n_total);
    std::vector<uint8_t> slot_moves(ns * 225, 0), first_moves(nt * 225, 0);
    std::vector<int32_t> slot_lens(ns, 0), state(ns + 3, -1);
    std::vector<uint32_t> ids(ns, 0);
    for (size_t g = 0; g < nt; ++g)
        for (int i = 0; i < open_lens[g]; ++i) {
            first_moves[g * 225 + i] = h_open_moves[g * open_stride + i];
            if (g < ns) slot_moves[g * 225 + i] = h_open_moves[g * open_stride + i];
        }
    for (int g = 0; g < started; ++g) { slot_lens[static_cast<size_t>(g)] = open_lens[static_cast<size_t>(g)]; state[static_cast<size_t>(g)] = g; ids[static_cast<size_t>(g)] = static_cast<uint32_t>(g); }
    state[ns] = started;                                        // next_game
    state[ns + 1] = 0;                                          // unfinished
    state[ns + 2] = 0;                                          // overflow
    int rc = GMK_OK;
    if (persistent && reuse_subtree) rc = pair_arenas(t);       // a slot's two arenas, a fixed stride apart
    if (rc == GMK_OK) rc = gmk_trad_set_game_ids(t, ids.data());
    if (rc == GMK_OK) { t->positioned = true; rc = gmk_trad_set_positions(t, slot_moves.data(), slot_lens.data()); }     // (every slot is positioned anew)
    if (rc != GMK_OK) return rc;
    if (n_slots > n_total) {                                    // slots without a game: idle from the start
        std::vector<TradHeader> hdr(ns);
        GMK_HIP_CHECK(hipMemcpy(hdr.data(), t->d_hdr, ns * sizeof(TradHeader), hipMemcpyDeviceToHost));
        for (size_t g = nt; g < ns; ++g) hdr[g].status |= kStatusIdleSlot;
        GMK_HIP_CHECK(hipMemcpy(t->d_hdr, hdr.data(), ns * sizeof(TradHeader), hipMemcpyHostToDevice));
    }
    int32_t* d_state = nullptr;
    uint8_t* d_open_moves = nullptr;
    int32_t* d_open_lens = nullptr;
    auto cleanup = [&]() { (void)gmk::device_free(d_state); (void)gmk::device_free(d_open_moves); (void)gmk::device_free(d_open_lens); };
#define GMK_TRY(expr) do { if ((expr) != hipSuccess) { gmk::set_error("gmk_trad_selfplay_run: %s failed", #expr); cleanup(); return GMK_ERR_HIP; } } while (0)
    GMK_TRY(gmk::device_malloc(&d_state, state.size() * 4));
    GMK_TRY(hipMemcpy(d_state, state.data(), state.size() * 4, hipMemcpyHostToDevice));
    GMK_TRY(gmk::device_malloc(&d_open_lens, nt * 4));
    GMK_TRY(hipMemcpy(d_open_lens, open_lens.data(), nt * 4, hipMemcpyHostToDevice));
    if (h_open_moves) {
        GMK_TRY(gmk::device_malloc(&d_open_moves, nt * static_cast<size_t>(open_stride)));
        GMK_TRY(hipMemcpy(d_open_moves, h_open_moves, nt * static_cast<size_t>(open_stride), hipMemcpyHostToDevice));
    }
    GMK_TRY(hipMemcpy(d_moves, first_moves.data(), nt * 225, hipMemcpyHostToDevice));
    GMK_TRY(hipMemcpy(d_lens, open_lens.data(), nt * 4, hipMemcpyHostToDevice));
    GMK_TRY(hipMemset(d_winner, 0, nt));
    if (reuse_subtree && !t->second_arena) {                     // the arenas flip at every step (as gmk_trad_step)
        const size_t nodes = ns * static_cast<size_t>(t->cap);
        const bool ok = gmk::device_malloc(&t->d_stat2, nodes * 8) == hipSuccess && gmk::device_malloc(&t->d_info2, nodes * 8) == hipSuccess &&
                        gmk::device_malloc(&t->d_link2, nodes * 4) == hipSuccess && gmk::device_malloc(&t->d_front2, nodes * 8) == hipSuccess &&
                        gmk::device_malloc(&t->d_ord2, nodes) == hipSuccess && gmk::device_malloc(&t->d_forced, ns * 2) == hipSuccess;
        if (!ok) {
            (void)gmk::device_free(t->d_stat2); (void)gmk::device_free(t->d_info2); (void)gmk::device_free(t->d_link2); (void)gmk::device_free(t->d_front2); (void)gmk::device_free(t->d_ord2); (void)gmk::device_free(t->d_forced);
            t->d_stat2 = t->d_info2 = t->d_front2 = nullptr; t->d_link2 = nullptr; t->d_ord2 = nullptr; t->d_forced = nullptr;
            (void)hipGetLastError();
            gmk::set_error("gmk_trad_selfplay_run: hipMalloc of the second arena (%zu nodes) failed", nodes);
            cleanup();
            return GMK_ERR_HIP;
        }
        t->second_arena = true;
    }
    TradSelfPlay sp;
    sp.slot_game = d_state; sp.next_game = d_state + ns; sp.unfinished = d_state + ns + 1; sp.overflow = d_state + ns + 2;
    sp.n_total = n_total; sp.open_moves = d_open_moves; sp.open_lens = d_open_lens; sp.open_stride = open_stride;
    sp.game_ids = t->d_game_ids;
    sp.rec_moves = d_moves; sp.rec_lens = d_lens; sp.rec_visits = d_visits; sp.rec_winner = d_winner;
    sp.reuse = (persistent && reuse_subtree) ? 1 : 0; sp.noise_alpha = (persistent && noisy) ? noise_alpha : 0.0f; sp.noise_epsilon = noise_epsilon;
    sp.arena_stride = (persistent && reuse_subtree) ? t->arena_stride() : 0;
    sp.seed_lo = static_cast<uint32_t>(seed); sp.seed_hi = static_cast<uint32_t>(seed >> 32); sp.first_game_id = first_game_id;
    int32_t steps = 0;
    if (persistent) {
        // ONE launch: every wavefront plays game after game at its own pace (trad_playouts_kernel, prm.selfplay)
        gmk::DeviceState& st = gmk::device_state();
        if (t->policy == 2) { gmk::set_error("gmk_trad_selfplay_run: this handle searches with PoolRAVEPolicy"); cleanup(); return GMK_ERR_STATE; }
        t->policy = 1;
        const size_t lds = static_cast<size_t>(kGamesPerBlock * kPerGame + st.n_states * 4 + st.n_records * 4 + gmk::kPrefixWords) * 4;
        if (!t->attr_set_selfplay) {
            GMK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(trad_playouts_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            t->attr_set_selfplay = true;
        }
        TradParams prm;
        prm.states = t->d_states; prm.stat = t->d_stat; prm.info = t->d_info; prm.link = t->d_link; prm.front = t->d_front; prm.ord = t->d_ord; prm.hdr = t->d_hdr;
        prm.moves = t->d_moves; prm.lens = t->d_lens;
        prm.g_trans = st.d_trans; prm.g_records = st.d_records; prm.trans_words = st.n_states * 4; prm.record_words = st.n_records * 4 + gmk::kPrefixWords; prm.path_spill = t->d_path_spill;
        prm.n_games = n_slots; prm.cap = t->cap; prm.playouts = playouts; prm.c_puct = c_puct;
        prm.profile = 0; prm.selfplay = 1; prm.sp = sp;
        hipLaunchKernelGGL(trad_playouts_kernel<true>, dim3((n_slots + kGamesPerBlock - 1) / kGamesPerBlock), dim3(kThreads), lds, s, prm);
        GMK_TRY(hipGetLastError());
        GMK_TRY(hipStreamSynchronize(s));
        steps = 1;
    }
    std::vector<int32_t> slot_game(ns);
    const long long step_limit = max_steps > 0 ? max_steps : 226ll * (n_total / n_slots + 2);
    for (long long step = 0; step < step_limit && !persistent; ++step) {
        if (noise_alpha > 0.0f && t->noise_sampler == GMK_NOISE_SAMPLER_COUNTER) {     // ... drawn on the device, keyed by the game a slot plays (d_game_ids: the refill kernel keeps it)
            hipLaunchKernelGGL(trad_root_noise_kernel, dim3(n_slots), dim3(64), 0, s, t->arena(), t->d_hdr, t->cap, t->d_lens, t->d_game_ids, first_game_id,
                               noise_alpha, noise_epsilon, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
            GMK_TRY(hipGetLastError());
        } else if (noise_alpha > 0.0f) {                        // Default::AddNoise at the start of every search (MCTS.cpp:182), keyed by the GAME a slot plays
            GMK_TRY(hipMemcpy(slot_game.data(), d_state, ns * 4, hipMemcpyDeviceToHost));
            for (size_t g = 0; g < ns; ++g) t->game_ids[g] = slot_game[g] >= 0 ? static_cast<uint32_t>(slot_game[g]) : 0u;
            rc = gmk_trad_add_root_noise(t, noise_alpha, noise_epsilon, seed, first_game_id);
            if (rc != GMK_OK) break;
        }
        rc = poolrave ? gmk_trad_run_poolrave(t, playouts, c_puct, seed, first_game_id, s) : gmk_trad_run(t, playouts, c_puct, s);
        if (rc != GMK_OK) break;
        if (reuse_subtree && t->d_amaf && !t->d_amaf2) GMK_TRY(gmk::device_malloc(&t->d_amaf2, ns * static_cast<size_t>(t->cap) * 8));
        GMK_TRY(hipMemsetAsync(sp.unfinished, 0, 4, s));
        hipLaunchKernelGGL(trad_advance_kernel, dim3(n_slots), dim3(64), 0, s, t->arena(), t->arena2(), t->d_hdr, t->cap, n_slots, t->d_moves, t->d_lens, sp, reuse_subtree ? 1 : 0);
        hipLaunchKernelGGL(trad_refill_kernel, dim3(1), dim3(64), 0, s, t->d_hdr, n_slots, t->d_moves, t->d_lens, sp);
        GMK_TRY(hipGetLastError());
        ++steps;
        int32_t unfinished = 0;
        GMK_TRY(hipMemcpyAsync(&unfinished, sp.unfinished, 4, hipMemcpyDeviceToHost, s));
        GMK_TRY(hipStreamSynchronize(s));
        if (reuse_subtree) {
            std::swap(t->d_stat, t->d_stat2); std::swap(t->d_info, t->d_info2); std::swap(t->d_link, t->d_link2);
            std::swap(t->d_front, t->d_front2); std::swap(t->d_ord, t->d_ord2); std::swap(t->d_amaf, t->d_amaf2);
        }
        if (unfinished == 0) break;
    }
    int32_t overflow = 0;
    if (rc == GMK_OK) GMK_TRY(hipMemcpy(&overflow, sp.overflow, 4, hipMemcpyDeviceToHost));
#undef GMK_TRY
    cleanup();
    if (h_overflow) *h_overflow = overflow;
    if (h_steps) *h_steps = steps;
    // the handle is left with idle slots: position it again (gmk_trad_set_positions) before any other use
    t->positioned = false;
    return rc;
}
