// rave_kernel.hip -- K8: the PoolRAVE tree search on the device.
//
// PoolRAVEPolicy (core/lib/include/policies/PoolRAVE.h:7-52) = MCTS::playout (core/lib/src/MCTS.cpp:158-177) with
//   select    RAVE::Select: the first child (core/lib/include/algorithms/MonteCarlo.hpp:149-152)
//   simulate  defaultSimulate: Default::UniformProbs of the leaf, then ONE Default::RandomRollout whose moves stay on
//             the board (PoolRAVE.h:29-48, MonteCarlo.hpp:37-47, 50-55)
//   expand    Default::Expand without the legality check: one AMAFNode per empty cell (MonteCarlo.hpp:71-80, 113-122)
//   backup    RAVE::BackPropogate<true> on the FINISHED board: every child of a path node whose move its player made
//             anywhere later in the game takes the result into its all-moves-as-first statistics; the child with the
//             best PUCB + HandSelect-weighted value moves to the front (MonteCarlo.hpp:124-184)
// It shares the first-child tree of K6 (trad_tree.h: same arenas, same re-rooting, noise and statistics entry points,
// plus 8 B of AMAF statistics per node) and the rollout of K3 (rollout_device.h, same Philox counter layout with
// rollout number 0).  One wavefront per game, four games per workgroup, no workgroup barrier: the position as 92 line
// words, the root position and the current path live in LDS (2.6 KB per game), the tree in HBM.  The rollout is a serial
// chain and runs on one lane; everything that touches children (expand, the per-level reductions of backup) runs 64
// children at a time, the next level's loads in flight while the current one is reduced.  PUCB and the weighting are
// evaluated in double like the reference, the running means in float.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "capi_common.h"
#include "rollout_device.h"
#include "trad_tree.h"

namespace {

using namespace gmk::tree;
using namespace gmk::rollout;

constexpr int kWaves = 4;                        // games per workgroup
constexpr int kPathCap = 228;                    // a path has at most 226 nodes
constexpr int kLinePad = 96;
constexpr int kPerGame = 2 * kLinePad + 2 * kPathCap;

struct RaveParams {
    TradArena a;
    TradHeader* hdr;
    const uint8_t* moves;                        // [n_games][225] the position of the root
    const int32_t* lens;
    int n_games, cap, playouts;
    uint32_t seed_lo, seed_hi, first_game_id;
    const uint32_t* game_ids;                    // [n_games] the game a slot is playing, relative to first_game_id (gmk_trad_set_game_ids)
    double c_puct;
    int profile;                                 // GMK_RAVE_PROFILE: shader clocks per stage into TradHeader::prof (diagnostic runs only)
};

__device__ __forceinline__ void wave_phase_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one stone into the four line words through its cell (the layout of rollout_device.h)
__device__ __forceinline__ void place_stone(uint32_t* lines, uint32_t cell, uint32_t shift) {
    const uint32_t y = cell / 15u, x = cell - 15u * y;
    atomicOr(&lines[y], 1u << (x + shift));
    atomicOr(&lines[kColBase + x], 1u << (y + shift));
    atomicOr(&lines[kDiagBase + x - y + 14], 1u << (x + shift));
    atomicOr(&lines[kAntiBase + x + y], 1u << (x + shift));
}

__global__ __launch_bounds__(64 * kWaves, 4)
void rave_playouts_kernel(RaveParams prm) {
    __shared__ uint32_t s_mem[kWaves][kPerGame];
    __shared__ uint2 s_cells[kWaves][30];                       // per game and playout: the rollout's cell draws, eight plies per entry
    __shared__ int s_ply[kWaves], s_winner[kWaves];             // per game and playout: stones at the leaf (-1: no rollout), the rollout's winner
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool in_range = blockIdx.x * kWaves + wave < prm.n_games;
    const int game = in_range ? blockIdx.x * kWaves + wave : prm.n_games - 1;   // a surplus wave shadows the last game read-only and only keeps the barriers
    const bool exists = in_range && !(prm.hdr[game].status & gmk::tree::kStatusIdleSlot);     // ... and so does a slot whose games have run out (continuous batching)

    uint32_t* lines = s_mem[wave];
    uint32_t* root_lines = lines + kLinePad;
    uint32_t* path_node = root_lines + kLinePad;                // node id | cell << 24
    uint32_t* path_link = path_node + kPathCap;                 // its child range
    const size_t arena = static_cast<size_t>(game) * prm.cap;
    uint2* stat = prm.a.stat + arena;
    uint2* info = prm.a.info + arena;
    uint32_t* link_of = prm.a.link + arena;
    uint2* front = prm.a.front + arena;
    uint8_t* ord = prm.a.ord + arena;
    uint2* amaf = prm.a.amaf + arena;
    TradHeader* hdr = prm.hdr + game;
    uint32_t n_nodes = hdr->n_nodes, status = exists ? hdr->status : 1u;
    const uint32_t fresh_mode = hdr->fresh;
    const bool fresh = fresh_mode == 1u;
    const uint8_t* mv = prm.moves + static_cast<size_t>(game) * 225;
    const int init = prm.lens[game];                            // Policy::m_initActs: stones on the root board
    const int root_black = init & 1;                            // the player of the last move
    const uint32_t root_last = init ? mv[init - 1] : 255u;
    const uint32_t playout0 = fresh_mode != 0u ? 0u : hdr->playouts_done;      // the rollout counter restarts with a new root

    // the root position as line words: move i is black's when i is even
    for (int w = lane; w < kLinePad; w += 64) root_lines[w] = 0u;
    wave_phase_fence();
    for (int i = lane; i < init; i += 64) place_stone(root_lines, mv[i], (i & 1) ? 16u : 0u);
    if (fresh) {                                                // MCTS::reset / a new search: the root node alone
        if (lane == 0 && exists) {
            stat[0] = make_uint2(0u, 0u);
            info[0] = make_uint2(kNoParent | (root_last << 24), __float_as_uint(1.0f));
            link_of[0] = 0u;
            amaf[0] = make_uint2(0u, 0u);
        }
        n_nodes = 1;
        status = exists ? 0u : 1u;
    }
    // path[0 .. valid] is known to be the chain of first children from the root
    int valid = 0;
    if (lane == 0) { path_node[0] = root_last << 24; path_link[0] = fresh ? 0u : link_of[0]; }
    wave_phase_fence();

    struct Level {                                              // what backup needs of one path node: its statistics and its children
        uint2 ns, cs[4], ci[4], ca[4];
        uint32_t cl[4], co[4];
    };
    auto load_level = [&](int d) {
        Level L;
        const uint32_t nd = path_node[d] & 0xFFFFFFu, lk = path_link[d], first = lk & 0xFFFFFFu, n = lk >> 24;
        L.ns = stat[nd];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = lane + 64 * k, id = first + i;
            if (i < n) { L.cs[k] = stat[id]; L.ci[k] = info[id]; L.ca[k] = amaf[id]; L.cl[k] = link_of[id]; L.co[k] = ord[id]; }
            else { L.cs[k] = make_uint2(0u, 0u); L.ci[k] = make_uint2(0u, 0u); L.ca[k] = make_uint2(0u, 0u); L.cl[k] = 0u; L.co[k] = 0xFFFFFFFFu; }
        }
        return L;
    };

    unsigned long long prof_sel = 0, prof_roll = 0, prof_back = 0, prof_t0 = 0, prof_all = (gmk::kProfileBuild && prm.profile) ? __builtin_amdgcn_s_memtime() : 0ull;
    // The rollout is a serial chain per game that keeps ONE lane busy: the four games of a workgroup meet at a barrier and one
    // wavefront plays their four rollouts side by side (a quarter of the vector issue slots four separate one-lane loops would
    // take), the others wait at the next barrier for free.  Every wave makes every iteration, a stopped game just passes through.
    const int roll_wave = blockIdx.x % kWaves;                  // spread the rollout waves over the SIMDs
    for (int it = 0; it < prm.playouts; ++it) {
        if (gmk::kProfileBuild && prm.profile) prof_t0 = __builtin_amdgcn_s_memtime();
        const bool act = !(status & 1u);
        int depth = 0, ply = 0, path_len = 0;
        uint32_t node = 0;
        bool five = false, rollout = false;
        if (act) {
            // ---- select: always the first child in the current order (RAVE::Select) ----
            depth = valid;
            node = path_node[depth] & 0xFFFFFFu;
            uint32_t link = path_link[depth];
            while (link >> 24) {
                const uint2 rec = front[node];
                node = rec.x & 0xFFFFFFu;
                link = rec.y;
                ++depth;
                if (lane == 0) { path_node[depth] = rec.x; path_link[depth] = link; }
            }
            wave_phase_fence();
            // ---- the leaf position: the root's line words plus the moves of the path (Policy::applyMove, no victory check) ----
            for (int w = lane; w < kLinePad; w += 64) lines[w] = root_lines[w];
            wave_phase_fence();
            for (int d = 1 + lane; d <= depth; d += 64) place_stone(lines, path_node[d] >> 24, ((init + d - 1) & 1) ? 16u : 0u);
            wave_phase_fence();
            ply = init + depth;
            const uint32_t last = depth ? path_node[depth] >> 24 : root_last;
            // ---- Policy::checkGameEnd -> Board::checkGameEnd (Game.cpp:88-136): five through the last move, or a full board ----
            if (ply > 0 && last < 225u) five = five_on_lines<1>(lines, static_cast<int>(last % 15u), static_cast<int>(last / 15u), (ply & 1) ? 0 : 16);
            path_len = depth;                                   // deepest level with a known node
            if (!five && ply != 225) {
                // ---- Default::UniformProbs of the leaf and Default::Expand (extraCheck = false, children in ascending cell order):
                //      both read the leaf position only, so the children are written before the rollout changes the board ----
                int total = 0, rank[4], first_cell = -1;
                bool open[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t cell = lane + 64 * j, y = min(cell / 15u, 14u), x = cell - 15u * (cell / 15u);
                    const uint32_t rw = lines[y];
                    open[j] = cell < 225u && (((rw | (rw >> 16)) >> x) & 1u) == 0u;
                    const unsigned long long b = __ballot(open[j]);
                    rank[j] = total + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(b >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(b), 0u)));
                    if (first_cell < 0 && b) first_cell = 64 * j + __ffsll(static_cast<unsigned long long>(b)) - 1;
                    total += __popcll(b);
                }
                if (n_nodes + static_cast<uint32_t>(total) > static_cast<uint32_t>(prm.cap)) {
                    status |= 1u;                               // arena full: the search of this game stops here
                } else {
                    const float prior = 1.0f / static_cast<float>(total);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (open[j]) {
                            const uint32_t child = n_nodes + rank[j];
                            stat[child] = make_uint2(0u, 0u);
                            info[child] = make_uint2(node | (static_cast<uint32_t>(lane + 64 * j) << 24), __float_as_uint(prior));
                            link_of[child] = 0u;
                            ord[child] = static_cast<uint8_t>(rank[j]);
                            amaf[child] = make_uint2(0u, 0u);
                        }
                    link = n_nodes | (static_cast<uint32_t>(total) << 24);
                    if (lane == 0) {
                        const uint32_t front_rec = n_nodes | (static_cast<uint32_t>(first_cell) << 24);
                        link_of[node] = link;
                        front[node] = make_uint2(front_rec, 0u);
                        if (depth > 0) front[path_node[depth - 1] & 0xFFFFFFu] = make_uint2(path_node[depth], link);      // the parent's record of this node
                        path_link[depth] = link;
                        path_node[depth + 1] = front_rec;
                        path_link[depth + 1] = 0u;
                    }
                    path_len = depth + 1;
                    n_nodes += total;
                    rollout = true;
                    // the rollout's random cells, one Philox block per lane: off the serial chain of the game itself
                    if (lane * 8 < 225 - ply)
                        s_cells[wave][lane] = rollout_cells(prm.first_game_id + prm.game_ids[game], playout0 + static_cast<uint32_t>(it),
                                                            static_cast<uint32_t>(init) << 8, static_cast<uint32_t>(lane), prm.seed_lo, prm.seed_hi);
                }
            }
        }
        if (lane == 0) s_ply[wave] = rollout ? ply : -1;
        if (gmk::kProfileBuild && prm.profile) { const unsigned long long t = __builtin_amdgcn_s_memtime(); prof_sel += t - prof_t0; prof_t0 = t; }
        __syncthreads();
        // ---- Default::RandomRollout for the games of the workgroup, four lanes each; a finished game stays in its `lines` ----
        if (wave == roll_wave) {
            int no_tie_before = 224;                            // the first ply at which one of the boards can fill up (wave-uniform)
            for (int g = 0; g < kWaves; ++g)
                if (s_ply[g] >= 0) no_tie_before = min(no_tie_before, 224 - s_ply[g]);
            no_tie_before = __builtin_amdgcn_readfirstlane(no_tie_before);
            if (lane < 4 * kWaves) {                            // four lanes per rollout, one line direction each (random_rollout_quads)
                const int g = lane >> 2, stones = s_ply[g];
                if (stones >= 0) {
                    const int winner = random_rollout_quads(s_mem[g], 1u, (stones & 1) ? -1 : 1, stones, no_tie_before, [&](uint32_t b) { return s_cells[g][b]; });
                    if ((lane & 3) == 0) s_winner[g] = winner;
                }
            }
        }
        __syncthreads();
        if (gmk::kProfileBuild && prm.profile) { const unsigned long long t = __builtin_amdgcn_s_memtime(); prof_roll += t - prof_t0; prof_t0 = t; }
        if (!act || (status & 1u)) continue;
        float value;                                            // for the player of `node`
        if (rollout) {
            const int to_move = (ply & 1) ? -1 : 1;
            value = -static_cast<float>(to_move * s_winner[wave]);      // CalcScore(init_player, winner), seen from the player of `node`
        } else {
            value = five ? 1.0f : 0.0f;                         // CalcScore(node->player, winner): the mover won, or a tie
        }

        // ---- RAVE::BackPropogate<true> (MonteCarlo.hpp:154-184), leaf to root, on the finished board ----
        int swap_level = -1;
        uint2 swap_rec = make_uint2(0u, 0u);
        uint32_t updated_id = 0xFFFFFFFFu;                      // the path node one level below: its statistics were just rewritten
        uint2 updated_stat = make_uint2(0u, 0u);
        Level cur = load_level(depth);
        for (int d = depth; d >= 0; --d, value = -value) {
            Level nxt;
            if (d > 0) nxt = load_level(d - 1);                 // in flight while this level is reduced
            const uint32_t nd = path_node[d] & 0xFFFFFFu, lk = path_link[d];
            const uint32_t first = lk & 0xFFFFFFu, n = lk >> 24;
            const double sqrt_n = sqrt(static_cast<double>(cur.ns.x));
            const uint32_t child_shift = (((d + 1) & 1) ? !root_black : root_black) ? 0u : 16u;       // the children's player: black in the low halves
            double best_score = -INFINITY;
            uint32_t best_ord = 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t i = lane + 64 * k;
                if (i >= n) continue;
                uint2 cs = cur.cs[k];
                if (first + i == updated_id) cs = updated_stat;
                const uint32_t cell = cur.ci[k].x >> 24, y = cell / 15u, x = cell - 15u * y;
                float aq = __uint_as_float(cur.ca[k].y);
                if ((lines[y] >> (x + child_shift)) & 1u) {     // board.moveState(child.player, child.position): all moves as first
                    const uint32_t av = cur.ca[k].x + 1u;
                    aq += (-value - aq) / static_cast<float>(av);
                    amaf[first + i] = make_uint2(av, __float_as_uint(aq));
                }
                const double p_i = __uint_as_float(cur.ci[k].y), n_i = static_cast<double>(cs.x + 1u);
                double score = prm.c_puct * p_i * sqrt_n / n_i;                        // Default::PUCB (:23-28)
                const double visits = static_cast<double>(cs.x), eqv = 800.0;
                const double weight = sqrt(eqv / (3 * visits + eqv));                  // RAVE::HandSelect (:124-128)
                score += (1 - weight) * __uint_as_float(cs.y) + weight * aq;            // RAVE::WeightedValue (:138-142)
                // the reference scans the children in their current order and keeps the first maximum
                if (score > best_score || (score == best_score && cur.co[k] < best_ord)) { best_score = score; best_ord = cur.co[k]; }
            }
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
                    ord[rec.x & 0xFFFFFFu] = 0;
                    ord[path_node[d + 1] & 0xFFFFFFu] = static_cast<uint8_t>(best_ord);
                    front[nd] = rec;
                }
                swap_level = d;
                swap_rec = rec;
            }
            const uint32_t visits = cur.ns.x + 1u;
            const float q = __uint_as_float(cur.ns.y);
            updated_id = nd;
            updated_stat = make_uint2(visits, __float_as_uint(q + (value - q) / static_cast<float>(visits)));
            if (lane == 0) stat[nd] = updated_stat;
            cur = nxt;
        }
        wave_phase_fence();
        if (swap_level >= 0) {                                  // below the shallowest swap the chain of first children is a different one
            if (lane == 0) { path_node[swap_level + 1] = swap_rec.x; path_link[swap_level + 1] = swap_rec.y; }
            valid = swap_level + 1;
        } else {
            valid = path_len;
        }
        wave_phase_fence();
        if (gmk::kProfileBuild && prm.profile) prof_back += __builtin_amdgcn_s_memtime() - prof_t0;
    }

    if (lane == 0 && exists) {
        hdr->n_nodes = n_nodes;
        hdr->init_acts = static_cast<uint32_t>(init);
        hdr->status = status;
        hdr->fresh = 0;
        hdr->root_black = static_cast<uint32_t>(root_black);
        hdr->playouts_done = playout0 + static_cast<uint32_t>(prm.playouts);
        if (gmk::kProfileBuild && prm.profile) {
            hdr->prof[0] = static_cast<uint32_t>(prof_sel >> 10); hdr->prof[1] = static_cast<uint32_t>(prof_roll >> 10);
            hdr->prof[2] = static_cast<uint32_t>(prof_back >> 10); hdr->prof[3] = static_cast<uint32_t>((__builtin_amdgcn_s_memtime() - prof_all) >> 10);
        }
    }
}

// the root children's all-moves-as-first statistics by cell
__global__ __launch_bounds__(64)
void rave_root_amaf_kernel(TradArena a, const TradHeader* hdrs, int cap, uint32_t* amaf_visits, float* amaf_values) {
    const int game = blockIdx.x, lane = threadIdx.x;
    const size_t arena = static_cast<size_t>(game) * cap;
    if (hdrs[game].fresh == 1u) return;
    const uint32_t lk = a.link[arena], first = lk & 0xFFFFFFu, n = lk >> 24;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t cell = a.info[arena + first + i].x >> 24;
        const uint2 am = a.amaf[arena + first + i];
        amaf_visits[static_cast<size_t>(game) * 225 + cell] = am.x;
        amaf_values[static_cast<size_t>(game) * 225 + cell] = __uint_as_float(am.y);
    }
}

}  // namespace

extern "C" int gmk_trad_run_poolrave(gmk_trad* t, int playouts, double c_puct, uint64_t seed, uint32_t first_game_id, void* stream) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (!t || playouts < 0) { gmk::set_error("gmk_trad_run_poolrave: bad arguments"); return GMK_ERR_ARG; }
    if (!t->positioned) { gmk::set_error("gmk_trad_run_poolrave: gmk_trad_set_positions has not been called"); return GMK_ERR_STATE; }
    if (t->policy == 1) { gmk::set_error("gmk_trad_run_poolrave: this handle searches with gmk_trad_run (its nodes carry no AMAF statistics)"); return GMK_ERR_STATE; }
    t->policy = 2;
    if (!t->d_amaf) {
        const size_t nodes = static_cast<size_t>(t->n_games) * static_cast<size_t>(t->cap);
        GMK_HIP_CHECK(gmk::device_malloc(&t->d_amaf, nodes * 8));
        GMK_HIP_CHECK(hipMemset(t->d_amaf, 0, nodes * 8));
    }
    RaveParams prm;
    prm.a = t->arena();
    prm.hdr = t->d_hdr; prm.moves = t->d_moves; prm.lens = t->d_lens;
    prm.n_games = t->n_games; prm.cap = t->cap; prm.playouts = playouts;
    prm.seed_lo = static_cast<uint32_t>(seed); prm.seed_hi = static_cast<uint32_t>(seed >> 32); prm.first_game_id = first_game_id; prm.game_ids = t->d_game_ids;
    prm.c_puct = c_puct;
    static const bool profile = gmk::profile_env("GMK_RAVE_PROFILE") != nullptr;
    prm.profile = profile ? 1 : 0;
    const int grid = (t->n_games + kWaves - 1) / kWaves;
    hipLaunchKernelGGL(rave_playouts_kernel, dim3(grid), dim3(64 * kWaves), 0, static_cast<hipStream_t>(stream), prm);
    GMK_HIP_CHECK(hipGetLastError());
    if (profile) {                                              // share of a search spent per stage, mean over games
        std::vector<TradHeader> hdr(static_cast<size_t>(t->n_games));
        GMK_HIP_CHECK(hipDeviceSynchronize());
        GMK_HIP_CHECK(hipMemcpy(hdr.data(), t->d_hdr, hdr.size() * sizeof(TradHeader), hipMemcpyDeviceToHost));
        double sum[4] = {};
        for (const TradHeader& h : hdr) for (int k = 0; k < 4; ++k) sum[k] += h.prof[k];
        std::fprintf(stderr, "[GMK_RAVE_PROFILE] select + leaf position + expand %.1f %%, rollout %.1f %%, backup %.1f %% of the kernel's clocks\n",
                     100 * sum[0] / sum[3], 100 * sum[1] / sum[3], 100 * sum[2] / sum[3]);
    }
    return GMK_OK;
}

extern "C" int gmk_trad_root_amaf(gmk_trad* t, uint32_t* h_amaf_visits, float* h_amaf_values) {
    if (!t || !h_amaf_visits || !h_amaf_values) { gmk::set_error("gmk_trad_root_amaf: bad arguments"); return GMK_ERR_ARG; }
    if (!t->d_amaf) { gmk::set_error("gmk_trad_root_amaf: gmk_trad_run_poolrave has not been called on this handle"); return GMK_ERR_STATE; }
    const size_t n = static_cast<size_t>(t->n_games);
    uint32_t* d_visits = nullptr;
    float* d_values = nullptr;
    auto cleanup = [&]() { (void)gmk::device_free(d_visits); (void)gmk::device_free(d_values); };
#define GMK_TRY(expr) do { if ((expr) != hipSuccess) { gmk::set_error("%s failed", #expr); cleanup(); return GMK_ERR_HIP; } } while (0)
    GMK_TRY(gmk::device_malloc(&d_visits, n * 225 * 4)); GMK_TRY(gmk::device_malloc(&d_values, n * 225 * 4));
    GMK_TRY(hipMemset(d_visits, 0, n * 225 * 4)); GMK_TRY(hipMemset(d_values, 0, n * 225 * 4));
    hipLaunchKernelGGL(rave_root_amaf_kernel, dim3(t->n_games), dim3(64), 0, nullptr, t->arena(), t->d_hdr, t->cap, d_visits, d_values);
    GMK_TRY(hipGetLastError());
    GMK_TRY(hipDeviceSynchronize());
    GMK_TRY(hipMemcpy(h_amaf_visits, d_visits, n * 225 * 4, hipMemcpyDeviceToHost));
    GMK_TRY(hipMemcpy(h_amaf_values, d_values, n * 225 * 4, hipMemcpyDeviceToHost));
#undef GMK_TRY
    cleanup();
    return GMK_OK;
}
