// az_kernel.hip -- K7: network-guided tree search, many games per step (BASELINE.json configs[4]).
//
// The reference's AlphaZero-style agent is MCTS with Policy(eval_state = network.eval_state, c_puct)
// (agents/alphazero.py:5-9): Default::Select, a network call at every new leaf instead of rollouts, Default::Expand with
// the returned probabilities and the legality check, Default::BackPropogate (core/lib/include/algorithms/MonteCarlo.hpp:
// 57-95, core/lib/src/MCTS.cpp:158-177).  One game evaluates ONE position per playout, so a GPU batch comes from running
// many games in lock step; a playout of every game is
//     az_select_kernel      descend to a leaf; finished games at the leaf are backed up at once, the others write the
//                           leaf's feature planes (Board.encoded_states, core/py_ext/src/game_ext.hpp:87-104) as one
//                           row of the batch [n_games][6][15][15]
//     the network           any callable on that batch (PyTorch-ROCm PolicyValueNetwork, network/model_tf.py:28-66)
//     az_expand_kernel      children with the returned priors, value backed up to the root
// all on one HIP stream (the three steps can be captured in a hipGraph and replayed per playout).
// Mapping: one wavefront per game, lanes over the (<= 225) children; the tree is an SoA arena per game in HBM.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>
#include <vector>

#include "board_device.h"
#include "capi_common.h"
#include "philox.h"
#include "rollout_device.h"
#include "root_noise.h"
#include "noise_device.h"

namespace {

using gmk::five_through;

constexpr int kCells = 225;
constexpr uint32_t kNoNode = 0xFFFFFFFFu;
constexpr uint32_t kStatusOver = 1u;

struct AzHeader {                                // 256 B per game, in HBM
    uint32_t rows[16];                           // root position: black | white << 16 per row
    uint32_t leaf_rows[16];                      // position of the pending leaf
    uint32_t n_nodes, stones, last_move, last_move2;           // cells 0..224, 255 = none
    uint32_t status;                             // bit 0: the game is over (az_advance_kernel), bit 1: node arena full, bit 2: a forced step was not a move of the game
    uint32_t leaf, leaf_pending;                 // node waiting for the network's answer (leaf_pending = 1)
    uint32_t leaf_stones;
    uint32_t pad[24];
};
static_assert(sizeof(AzHeader) == 256, "AzHeader layout");

struct AzTree {
    AzHeader* hdr;
    uint2* stat;                                 // [n_games][cap] {visits, value bits}
    uint2* kids;                                 // [n_games][cap] {first child, children | cell << 8}
    float* prior;                                // [n_games][cap]
    uint32_t* parent;                            // [n_games][cap]
    int cap, n_games;
    double c_puct;
    const int32_t* row_of;                       // [n_games] the row of the leaf batch a game writes to and is answered from: the games still
                                                 // played, in slot order (az_compact_kernel); finished games have none
};

// Default::BackPropogate (MonteCarlo.hpp:90-95): one lane walks the parent chain
__device__ __forceinline__ void backup(uint2* stat, const uint32_t* parent, uint32_t node, float value) {
    for (; node != kNoNode; node = parent[node], value = -value) {
        const uint2 s = stat[node];
        const uint32_t visits = s.x + 1u;
        const float q = __uint_as_float(s.y);
        stat[node] = make_uint2(visits, __float_as_uint(q + (value - q) / static_cast<float>(visits)));
    }
}

__global__ __launch_bounds__(64)
void az_select_kernel(AzTree t, float* __restrict__ out_states) {
    __shared__ uint32_t s_rows[16];
    const int game = blockIdx.x, lane = threadIdx.x;
    if (game >= t.n_games) return;
    AzHeader* hdr = t.hdr + game;
    if (hdr->status & kStatusOver) return;                       // the game is over (az_advance_kernel): nothing to search, leaf_pending stays 0
    const size_t row = static_cast<size_t>(t.row_of[game]);
    const size_t arena = static_cast<size_t>(game) * t.cap;
    uint2* stat = t.stat + arena;
    const uint2* kids = t.kids + arena;
    const float* prior = t.prior + arena;
    if (lane < 16) s_rows[lane] = hdr->rows[lane];
    __syncthreads();
    uint32_t node = 0;
    int stones = static_cast<int>(hdr->stones);
    uint32_t last = hdr->last_move, last2 = hdr->last_move2;

    // ---- Default::Select (MonteCarlo.hpp:57-68) down to a leaf ----
    for (;;) {
        const uint2 k = kids[node];
        const uint32_t first = k.x, n = k.y & 0xFFu;
        if (n == 0) break;
        const double sqrt_n = sqrt(static_cast<double>(stat[node].x));
        double best_score = -1.0;
        uint32_t best_i = 0xFFFFFFFFu;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint2 cs = stat[first + i];
            const double p_i = prior[first + i], n_i = static_cast<double>(cs.x + 1u);
            const double score = static_cast<double>(__uint_as_float(cs.y)) + t.c_puct * p_i * sqrt_n / n_i;       // Q + PUCB (:23-28)
            if (score > best_score) { best_score = score; best_i = i; }
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {                                             // first maximum, strict '>'
            const double os = __shfl_down(best_score, s);
            const uint32_t oi = __shfl_down(best_i, s);
            if (os > best_score || (os == best_score && oi < best_i)) { best_score = os; best_i = oi; }
        }
        best_i = __shfl(best_i, 0);
        if (best_i == 0xFFFFFFFFu) best_i = 0;                                          // nothing beat the initial -1.0: the first child
        node = first + best_i;
        const uint32_t cell = (kids[node].y >> 8) & 0xFFu;
        if (lane == 0) s_rows[cell / 15] |= 1u << (cell % 15 + ((stones & 1) ? 16 : 0));                            // Policy::applyMove: black moves on even counts
        __syncthreads();
        ++stones;
        last2 = last;
        last = cell;
    }

    // ---- Board::checkGameEnd (Game.cpp:88-136): five through the last stone, or a full board ----
    bool ended = false;
    int winner = 0;
    if (stones > 0 && last < 225u) {
        const int mover_white = (stones - 1) & 1;
        if (five_through<1>(s_rows, last % 15, last / 15, mover_white ? 16 : 0)) { ended = true; winner = mover_white ? -1 : 1; }
    }
    if (!ended && stones == kCells) ended = true;
    if (ended) {
        // CalcScore(node.player, winner) (Game.h:34-36): the node's player is the one who made the last move
        const int node_player = (stones & 1) ? 1 : -1;
        if (lane == 0) {
            backup(stat, t.parent + arena, node, static_cast<float>(node_player * winner));
            hdr->leaf_pending = 0;
        }
        if (out_states)
            for (int i = lane; i < 6 * kCells; i += 64) out_states[row * 6 * kCells + i] = 0.0f;
        return;
    }
    if (lane == 0) { hdr->leaf = node; hdr->leaf_pending = 1; hdr->leaf_stones = static_cast<uint32_t>(stones); }
    if (lane < 16) hdr->leaf_rows[lane] = s_rows[lane];
    // ---- Board.encoded_states (game_ext.hpp:87-104): own stones, opponent's, empties, last move, the one before, colour to move ----
    const int cur_white = stones & 1;
    float* out = out_states + row * 6 * kCells;
    for (int i = lane; i < kCells; i += 64) {
        const uint32_t row = s_rows[i / 15] >> (i % 15);
        const bool black = row & 1u, white = (row >> 16) & 1u;
        out[0 * kCells + i] = (cur_white ? white : black) ? 1.0f : 0.0f;
        out[1 * kCells + i] = (cur_white ? black : white) ? 1.0f : 0.0f;
        out[2 * kCells + i] = (!black && !white) ? 1.0f : 0.0f;
        out[3 * kCells + i] = static_cast<uint32_t>(i) == last ? 1.0f : 0.0f;
        out[4 * kCells + i] = static_cast<uint32_t>(i) == last2 ? 1.0f : 0.0f;
        out[5 * kCells + i] = cur_white ? 0.0f : 1.0f;
    }
}

__global__ __launch_bounds__(64)
void az_expand_kernel(AzTree t, const float* __restrict__ values, const float* __restrict__ probs, int stages /* bit 0: expand, bit 1: back up */) {
    const int game = blockIdx.x, lane = threadIdx.x;
    if (game >= t.n_games) return;
    AzHeader* hdr = t.hdr + game;
    if (!hdr->leaf_pending) return;
    const size_t row = static_cast<size_t>(t.row_of[game]);
    const size_t arena = static_cast<size_t>(game) * t.cap;
    const uint32_t leaf = hdr->leaf;
    uint32_t n_nodes = hdr->n_nodes;
    // ---- Default::Expand with extraCheck (MonteCarlo.hpp:71-80): probability not 0 and the cell is free; ascending cell id ----
    const float* p = probs + row * kCells;
    float pv[4];
    bool take[4];
    int rank[4], total = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        pv[j] = i < kCells ? p[i] : 0.0f;
        const uint32_t row = i < kCells ? hdr->leaf_rows[i / 15] >> (i % 15) : 0x10001u;
        take[j] = i < kCells && pv[j] != 0.0f && !(row & 0x10001u);
        const unsigned long long b = __ballot(take[j]);
        rank[j] = total + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(b >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(b), 0u)));
        total += __popcll(b);
    }
    if (total > 0 && (stages & 1)) {
        if (n_nodes + total > static_cast<uint32_t>(t.cap)) {
            if (lane == 0) { hdr->status |= 2u; hdr->leaf_pending = 0; }               // arena full: this playout is dropped
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (take[j]) {
                const uint32_t child = n_nodes + rank[j];
                t.stat[arena + child] = make_uint2(0u, 0u);
                t.kids[arena + child] = make_uint2(0u, static_cast<uint32_t>(lane + 64 * j) << 8);
                t.prior[arena + child] = pv[j];
                t.parent[arena + child] = leaf;
            }
        if (lane == 0) {
            const uint32_t cell_bits = t.kids[arena + leaf].y & 0xFF00u;
            t.kids[arena + leaf] = make_uint2(n_nodes, static_cast<uint32_t>(total) | cell_bits);
            hdr->n_nodes = n_nodes + total;
        }
    }
    __threadfence_block();
    if (lane == 0) {
        if (stages & 2) backup(t.stat + arena, t.parent + arena, leaf, -values[row]);  // node_value = -state_value (MCTS.cpp:166-168)
        hdr->leaf_pending = 0;
    }
}

// MCTS::stepForward() / stepForward(move) (MCTS.cpp:129-147): the chosen child's subtree becomes the tree, copied level by
// level into the other arena (root at 0, children consecutive); a move without a child starts a new node.  The move is
// played on the root position.  One wavefront per game.
struct AzArena { uint2* stat; uint2* kids; float* prior; uint32_t* parent; };
// Continuous batching (gmk_az_set_slots): the handle's games are SLOTS that play n_total games between them; slot g plays game slot_game[g]
// (-1: none left), its record goes to that game's rows, and when the game ends the slot takes the next game nobody has started.
struct AzSlots {
    int32_t* slot_game;              // [n_slots], null: slot g plays game g and is not refilled
    int32_t* next_game;              // [1]
    int n_total;
    const uint8_t* open_moves;       // [n_total][open_stride] the moves that lead to a game's first root (black first), may be null
    const int32_t* open_lens;        // [n_total]
    int open_stride;
};

// The subtree of node `src` of arena t becomes the tree of arena b: root at 0, children consecutive, level by level.  All 64 lanes of the
// game's workgroup call it.
__device__ void az_keep_subtree(const AzTree& t, const AzArena& b, size_t base, uint32_t src, int lane, AzHeader& hdr) {
    if (lane == 0) { b.stat[base] = t.stat[base + src]; b.kids[base] = t.kids[base + src]; b.prior[base] = t.prior[base + src]; b.parent[base] = kNoNode; }
    __syncthreads();
    uint32_t next = 1;
    for (uint32_t i0 = 0, chunk = 0; i0 < next; i0 += chunk) {
        chunk = min(64u, next - i0);                            // nodes appended while this chunk is handled come after it
        const uint2 old = static_cast<uint32_t>(lane) < chunk ? b.kids[base + i0 + lane] : make_uint2(0u, 0u);
        unsigned long long todo = __ballot((old.y & 0xFFu) != 0u);
        while (todo) {
            const int j = __ffsll(static_cast<long long>(todo)) - 1;
            todo &= todo - 1ull;
            const uint32_t of = __shfl(old.x, j), oy = __shfl(old.y, j), nk = oy & 0xFFu, node = i0 + static_cast<uint32_t>(j);
            for (uint32_t k = lane; k < nk; k += 64) {
                b.stat[base + next + k] = t.stat[base + of + k];
                b.kids[base + next + k] = t.kids[base + of + k];       // still the OLD child range: rewritten when the scan gets there
                b.prior[base + next + k] = t.prior[base + of + k];
                b.parent[base + next + k] = node;
            }
            if (lane == 0) b.kids[base + node] = make_uint2(next, oy);
            next += nk;
            __syncthreads();
        }
        __syncthreads();
    }
    if (lane == 0) hdr.n_nodes = next;
}

__global__ __launch_bounds__(64)
void az_step_kernel(AzTree t, AzArena b, const int16_t* forced) {
    const int game = blockIdx.x, lane = threadIdx.x;
    if (game >= t.n_games) return;
    AzHeader& hdr = t.hdr[game];
    const size_t base = static_cast<size_t>(game) * t.cap;
    const uint2 rk = t.kids[base];
    const uint32_t first = rk.x, n = rk.y & 0xFFu;
    const int want = forced ? forced[game] : -1;
    uint32_t best_v = 0, best_i = 0xFFFFFFFFu;                   // visits + 1 (or "is the wanted cell"), first maximum in child order
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t cell = (t.kids[base + first + i].y >> 8) & 0xFFu;
        const uint32_t v = want >= 0 ? (cell == static_cast<uint32_t>(want) ? 1u : 0u) : t.stat[base + first + i].x + 1u;
        if (v > best_v) { best_v = v; best_i = i; }
    }
    for (int s = 32; s > 0; s >>= 1) {
        const uint32_t ov = __shfl_down(best_v, s), oi = __shfl_down(best_i, s);
        if (ov > best_v || (ov == best_v && ov != 0u && oi < best_i)) { best_v = ov; best_i = oi; }
    }
    best_v = __shfl(best_v, 0);
    best_i = __shfl(best_i, 0);
    const bool found = best_v != 0u;
    const uint32_t stones = hdr.stones;
    const uint32_t cell = found ? (t.kids[base + first + best_i].y >> 8) & 0xFFu : static_cast<uint32_t>(want);
    const uint32_t row = cell < 225u ? hdr.rows[cell / 15] >> (cell % 15) : 0x10001u;
    const bool legal = (want >= 0 || found) && cell < 225u && !(row & 0x10001u);
    if (!legal) {                                               // nothing to play (childless root) or not a move of this game: the tree is carried over as it is
        if (want >= 0 && lane == 0) hdr.status |= 4u;
    } else if (lane == 0) {
        hdr.rows[cell / 15] |= 1u << (cell % 15 + ((stones & 1u) ? 16 : 0));
        hdr.stones = stones + 1;
        hdr.last_move2 = hdr.last_move;
        hdr.last_move = cell;
        hdr.leaf_pending = 0;
    }
    const uint32_t src = !legal ? 0u : (found ? first + best_i : kNoNode);
    if (src == kNoNode) {                                       // stepForward(move) without such a child: a new node (MCTS.cpp:140-145)
        if (lane == 0) { b.stat[base] = make_uint2(0u, 0u); b.kids[base] = make_uint2(0u, cell << 8); b.prior[base] = 1.0f; b.parent[base] = kNoNode; hdr.n_nodes = 1; }
        return;
    }
    az_keep_subtree(t, b, base, src, lane, hdr);
}

// One self-play move for every game that is still played, on the device (what selfplay.play_network_games did with numpy boards, a
// root_stats copy down and a move list up every ply): the most visited child of the root -- first maximum in child order, i.e. ascending
// cells, as MCTS::stepForward()'s max_element (MCTS.cpp:129-134) -- goes into the game's record with the root's visit counts
// (MCTSAgent.eval_state's pi is made from them), Board::applyMove with its victory check (Game.cpp:37-49, 88-136) decides whether the
// game goes on, and the tree is re-rooted: the child's subtree into the other arena (reuse) or a new root.  A root without a visited
// child ends the game where it stands.  Finished games leave a childless root behind in both arenas, so that no kernel ever walks a
// stale tree.  One wavefront per game.
__global__ __launch_bounds__(64)
void az_advance_kernel(AzTree t, AzArena b, uint8_t* __restrict__ rec_moves, uint16_t* __restrict__ rec_visits, int32_t* __restrict__ rec_lens,
                       int8_t* __restrict__ rec_winner, int32_t* __restrict__ unfinished, int reuse, AzSlots sl) {
    __shared__ uint32_t s_rows[16];
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= t.n_games) return;
    AzHeader& hdr = t.hdr[slot];
    const size_t base = static_cast<size_t>(slot) * t.cap;
    const int game = sl.slot_game ? sl.slot_game[slot] : slot;      // the rows of the records
    if ((hdr.status & kStatusOver) || game < 0) {
        if (reuse && lane == 0) { b.stat[base] = make_uint2(0u, 0u); b.kids[base] = make_uint2(0u, 0u); b.prior[base] = 1.0f; b.parent[base] = kNoNode; }
        return;
    }
    const uint2 rk = t.kids[base];
    const uint32_t first = rk.x, n = rk.y & 0xFFu;
    uint32_t best_v = 0, best_i = 0xFFFFFFFFu;                   // visits, first maximum in child order
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t v = t.stat[base + first + i].x;
        if (v > best_v) { best_v = v; best_i = i; }
    }
    for (int s = 32; s > 0; s >>= 1) {
        const uint32_t ov = __shfl_down(best_v, s), oi = __shfl_down(best_i, s);
        if (ov > best_v || (ov == best_v && ov != 0u && oi < best_i)) { best_v = ov; best_i = oi; }
    }
    best_v = __shfl(best_v, 0);
    best_i = __shfl(best_i, 0);
    const uint32_t stones = hdr.stones;
    const int len = rec_lens[game];
    bool over = best_v == 0u || stones >= 225u || len >= 225;    // nothing the search wants to play
    int winner = 0;
    uint32_t cell = 255u;
    if (!over) {
        cell = (t.kids[base + first + best_i].y >> 8) & 0xFFu;
        if (rec_visits) {
            uint16_t* rv = rec_visits + (static_cast<size_t>(game) * kCells + static_cast<size_t>(len)) * kCells;
            for (int i = lane; i < kCells; i += 64) rv[i] = 0;
            __syncthreads();
            for (uint32_t i = lane; i < n; i += 64) rv[(t.kids[base + first + i].y >> 8) & 0xFFu] = static_cast<uint16_t>(min(t.stat[base + first + i].x, 65535u));
        }
        const int shift = (stones & 1u) ? 16 : 0;                // black moves on even stone counts
        if (lane < 16) s_rows[lane] = hdr.rows[lane] | ((lane == static_cast<int>(cell / 15u)) ? 1u << (cell % 15u + shift) : 0u);
        __syncthreads();
        const bool five = five_through<1>(s_rows, static_cast<int>(cell % 15u), static_cast<int>(cell / 15u), shift);
        if (lane == 0) {
            hdr.rows[cell / 15u] = s_rows[cell / 15u];
            hdr.stones = stones + 1u;
            hdr.last_move2 = hdr.last_move;
            hdr.last_move = cell;
            hdr.leaf_pending = 0;
            rec_moves[static_cast<size_t>(game) * kCells + len] = static_cast<uint8_t>(cell);
            rec_lens[game] = len + 1;
        }
        winner = five ? (shift ? -1 : 1) : 0;
        over = five || len + 1 == 225;
    }
    if (over) {
        if (lane == 0) {
            rec_winner[game] = static_cast<int8_t>(winner);
            const int next = sl.slot_game ? atomicAdd(sl.next_game, 1) : 0x7FFFFFFF;
            uint32_t root_cell = 0u;
            if (next < sl.n_total) {                            // the slot goes on with the next unstarted game: its opening is the root position
                const int n_open = sl.open_moves ? sl.open_lens[next] : 0;
                for (int y = 0; y < 16; ++y) hdr.rows[y] = 0u;
                uint32_t last = 255u, last2 = 255u;
                for (int i = 0; i < n_open; ++i) {
                    const uint32_t c = sl.open_moves[static_cast<size_t>(next) * sl.open_stride + i];
                    hdr.rows[c / 15u] |= 1u << (c % 15u + ((i & 1) ? 16 : 0));
                    last2 = last;
                    last = c;
                }
                hdr.stones = static_cast<uint32_t>(n_open);
                hdr.last_move = last;
                hdr.last_move2 = last2;
                sl.slot_game[slot] = next;
                root_cell = (last & 0xFFu) << 8;                // as az_init_roots_kernel
                atomicAdd(unfinished, 1);
            } else {
                if (sl.slot_game) sl.slot_game[slot] = -1;
                hdr.status |= kStatusOver;
            }
            hdr.leaf_pending = 0;
            hdr.n_nodes = 1;
            t.stat[base] = make_uint2(0u, 0u); t.kids[base] = make_uint2(0u, root_cell); t.prior[base] = 1.0f; t.parent[base] = kNoNode;
            if (reuse) { b.stat[base] = make_uint2(0u, 0u); b.kids[base] = make_uint2(0u, root_cell); b.prior[base] = 1.0f; b.parent[base] = kNoNode; }
        }
        return;
    }
    if (lane == 0) atomicAdd(unfinished, 1);
    if (reuse) {
        az_keep_subtree(t, b, base, first + best_i, lane, hdr);
    } else if (lane == 0) {                                      // a new root, as gmk_az_set_roots makes it
        t.stat[base] = make_uint2(0u, 0u); t.kids[base] = make_uint2(0u, cell << 8); t.prior[base] = 1.0f; t.parent[base] = kNoNode;
        hdr.n_nodes = 1;
    }
}

// The leaf batch holds the games that are still played, and nothing else: row_of[g] = the number of live games before g.  The network's
// work per playout then follows the number of live games instead of the number of slots (the end of a batch of games is long: lengths
// run from 9 to 225 plies).  One workgroup.
__global__ __launch_bounds__(1024)
void az_compact_kernel(const AzHeader* __restrict__ hdr, int n_games, int32_t* __restrict__ row_of, int32_t* __restrict__ n_live) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int g0 = 0; g0 < n_games; g0 += 1024) {
        const int g = g0 + tid;
        const bool live = g < n_games && !(hdr[g].status & kStatusOver);
        const unsigned long long b = __ballot(live);
        const int before = static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(b >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(b), 0u)));
        if ((tid & 63) == 0) s_wave[wave] = __popcll(b);
        __syncthreads();
        int waves_before = 0, total = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) waves_before += s_wave[w]; total += s_wave[w]; }
        if (g < n_games) row_of[g] = live ? s_base + waves_before + before : -1;
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) *n_live = s_base;
}

// Default::AddNoise (MonteCarlo.hpp:97-108): the root children's priors, by cell, as the host computed them
__global__ __launch_bounds__(64)
void az_set_root_priors_kernel(AzTree t, const float* priors) {
    const int game = blockIdx.x, lane = threadIdx.x;
    const size_t base = static_cast<size_t>(game) * t.cap;
    const uint2 rk = t.kids[base];
    for (uint32_t i = lane; i < (rk.y & 0xFFu); i += 64)
        t.prior[base + rk.x + i] = priors[static_cast<size_t>(game) * kCells + ((t.kids[base + rk.x + i].y >> 8) & 0xFFu)];
}

// Default::AddNoise with the counter-based sampler (include/gomoku_noise.h), one wavefront per game, on the device: the root children's priors travel
// through 225 words of LDS into by-cell order, lane l mixes the cells l + 64 j (noise_device.h), and back.  The stream belongs to the GAME a slot
// plays (slot_game under continuous batching, game_ids otherwise) and to the stones on its root board.
__global__ __launch_bounds__(64)
void az_root_noise_kernel(AzTree t, const int32_t* __restrict__ slot_game, const uint32_t* __restrict__ game_ids, uint32_t first_game_id,
                          float alpha, float epsilon, uint32_t seed_lo, uint32_t seed_hi) {
    __shared__ uint32_t s_cells[kCells];
    const int game = blockIdx.x, lane = threadIdx.x;
    const AzHeader& hdr = t.hdr[game];
    if (hdr.status & kStatusOver) return;
    const size_t base = static_cast<size_t>(game) * t.cap;
    const uint2 rk = t.kids[base];
    const uint32_t first = rk.x, n = rk.y & 0xFFu;
    if (n == 0u) return;                                         // a root without children takes no noise (the loop over node->children is empty)
    const uint32_t gid = first_game_id + (slot_game ? static_cast<uint32_t>(max(slot_game[game], 0)) : game_ids[game]);
    for (int i = lane; i < kCells; i += 64) s_cells[i] = 0u;
    __syncthreads();
    uint32_t cell[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = lane + 64 * k;
        cell[k] = i < n ? (t.kids[base + first + i].y >> 8) & 0xFFu : 0u;
        if (i < n) s_cells[cell[k]] = __float_as_uint(t.prior[base + first + i]);
    }
    __syncthreads();
    float p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = lane + 64 * j < kCells ? __uint_as_float(s_cells[lane + 64 * j]) : 0.0f;
    gmk::noise::mix_root_priors(p, lane, alpha, epsilon, gid, hdr.stones, seed_lo, seed_hi);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) if (lane + 64 * j < kCells) s_cells[lane + 64 * j] = __float_as_uint(p[j]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = lane + 64 * k;
        if (i < n) t.prior[base + first + i] = __uint_as_float(s_cells[cell[k]]);
    }
}

__global__ void az_init_roots_kernel(AzTree t) {
    const int game = blockIdx.x * blockDim.x + threadIdx.x;
    if (game >= t.n_games) return;
    const size_t arena = static_cast<size_t>(game) * t.cap;
    t.stat[arena] = make_uint2(0u, 0u);
    t.kids[arena] = make_uint2(0u, (t.hdr[game].last_move & 0xFFu) << 8);
    t.prior[arena] = 1.0f;
    t.parent[arena] = kNoNode;
}

__global__ __launch_bounds__(64)
void az_root_stats_kernel(AzTree t, uint32_t* visits, float* values, float* priors, uint32_t* root_visits, float* root_value) {
    const int game = blockIdx.x, lane = threadIdx.x;
    const size_t arena = static_cast<size_t>(game) * t.cap;
    const uint2 k = t.kids[arena];
    for (uint32_t i = lane; i < (k.y & 0xFFu); i += 64) {
        const uint32_t child = k.x + i, cell = (t.kids[arena + child].y >> 8) & 0xFFu;
        const uint2 cs = t.stat[arena + child];
        if (visits) visits[static_cast<size_t>(game) * kCells + cell] = cs.x;
        if (values) values[static_cast<size_t>(game) * kCells + cell] = __uint_as_float(cs.y);
        if (priors) priors[static_cast<size_t>(game) * kCells + cell] = t.prior[arena + child];
    }
    if (lane == 0) {
        if (root_visits) root_visits[game] = t.stat[arena].x;
        if (root_value) root_value[game] = __uint_as_float(t.stat[arena].y);
    }
}

}  // namespace

struct gmk_az {
    AzTree t{};
    bool rooted = false;
    AzArena other{};                                             // second arena, allocated by the first gmk_az_step
    int16_t* d_forced = nullptr;
    float* d_noise_priors = nullptr;
    int32_t* d_unfinished = nullptr;                             // gmk_az_advance's counter
    int32_t* d_row_of = nullptr;                                 // [n_games] + the number of live games (AzTree::row_of)
    int n_live = 0;
    AzSlots slots{};                                             // gmk_az_set_slots (device memory owned by the handle)
    uint8_t* d_open_moves = nullptr;
    int32_t* d_open_lens = nullptr;
    std::vector<uint32_t> game_ids;                              // the game a slot is playing, relative to the callers' first_game_id (default: the slot number)
    bool second_arena = false;
    int noise_sampler = GMK_NOISE_SAMPLER_STD;                   // gmk_az_set_option
    uint32_t* d_game_ids = nullptr;                              // the device copy of game_ids for az_root_noise_kernel
    // device scratch of the host-driven form (gmk_az_select_host / gmk_az_expand_host)
    float *h_states = nullptr, *h_values = nullptr, *h_probs = nullptr;
    int16_t* h_paths = nullptr;
    int32_t* h_lens = nullptr;
};

extern "C" int gmk_az_destroy(gmk_az* a) {
    if (!a) return GMK_OK;
    (void)gmk::device_free(a->t.hdr); (void)gmk::device_free(a->t.stat); (void)gmk::device_free(a->t.kids); (void)gmk::device_free(a->t.prior); (void)gmk::device_free(a->t.parent);
    (void)gmk::device_free(a->other.stat); (void)gmk::device_free(a->other.kids); (void)gmk::device_free(a->other.prior); (void)gmk::device_free(a->other.parent);
    (void)gmk::device_free(a->d_forced); (void)gmk::device_free(a->d_game_ids); (void)gmk::device_free(a->d_noise_priors); (void)gmk::device_free(a->d_unfinished); (void)gmk::device_free(a->d_row_of);
    (void)gmk::device_free(a->slots.slot_game); (void)gmk::device_free(a->d_open_moves); (void)gmk::device_free(a->d_open_lens);
    (void)gmk::device_free(a->h_states); (void)gmk::device_free(a->h_values); (void)gmk::device_free(a->h_probs); (void)gmk::device_free(a->h_paths); (void)gmk::device_free(a->h_lens);
    delete a;
    return GMK_OK;
}

extern "C" int gmk_az_create(int n_games, int node_capacity, double c_puct, gmk_az** out) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (!out || n_games <= 0 || node_capacity < 256) { gmk::set_error("gmk_az_create: bad arguments"); return GMK_ERR_ARG; }
    gmk_az* a = new gmk_az;
    a->t.n_games = n_games; a->t.cap = node_capacity; a->t.c_puct = c_puct;
    const size_t nodes = static_cast<size_t>(n_games) * node_capacity;
    const bool ok = gmk::device_malloc(&a->t.hdr, static_cast<size_t>(n_games) * sizeof(AzHeader)) == hipSuccess &&
                    gmk::device_malloc(&a->t.stat, nodes * 8) == hipSuccess && gmk::device_malloc(&a->t.kids, nodes * 8) == hipSuccess &&
                    gmk::device_malloc(&a->t.prior, nodes * 4) == hipSuccess && gmk::device_malloc(&a->t.parent, nodes * 4) == hipSuccess;
    if (!ok || gmk::device_malloc(&a->d_row_of, (static_cast<size_t>(n_games) + 1) * 4) != hipSuccess) { gmk_az_destroy(a); gmk::set_error("gmk_az_create: device allocation failed"); return GMK_ERR_HIP; }
    a->t.row_of = a->d_row_of;
    a->game_ids.resize(static_cast<size_t>(n_games));
    for (int g = 0; g < n_games; ++g) a->game_ids[static_cast<size_t>(g)] = static_cast<uint32_t>(g);
    *out = a;
    return GMK_OK;
}

// row_of and the number of live games, after anything that changes which games are played
static int az_compact(gmk_az* a, hipStream_t s) {
    hipLaunchKernelGGL(az_compact_kernel, dim3(1), dim3(1024), 0, s, a->t.hdr, a->t.n_games, a->d_row_of, a->d_row_of + a->t.n_games);
    GMK_HIP_CHECK(hipGetLastError());
    int32_t n = 0;
    GMK_HIP_CHECK(hipMemcpyAsync(&n, a->d_row_of + a->t.n_games, 4, hipMemcpyDeviceToHost, s));
    GMK_HIP_CHECK(hipStreamSynchronize(s));
    a->n_live = n;
    return GMK_OK;
}

extern "C" int gmk_az_live_games(gmk_az* a, int32_t* n_live) {
    if (!a || !n_live) { gmk::set_error("gmk_az_live_games: bad arguments"); return GMK_ERR_ARG; }
    *n_live = a->n_live;
    return GMK_OK;
}

extern "C" int gmk_az_set_game_ids(gmk_az* a, const uint32_t* h_ids) {
    if (!a || !h_ids) { gmk::set_error("gmk_az_set_game_ids: bad arguments"); return GMK_ERR_ARG; }
    a->game_ids.assign(h_ids, h_ids + a->t.n_games);
    return GMK_OK;
}

extern "C" int gmk_az_set_roots(gmk_az* a, const uint16_t* h_planes, const int16_t* h_last_moves) {
    if (!a || !h_planes || !h_last_moves) { gmk::set_error("gmk_az_set_roots: bad arguments"); return GMK_ERR_ARG; }
    std::vector<AzHeader> hdr(a->t.n_games);
    for (int g = 0; g < a->t.n_games; ++g) {
        AzHeader& h = hdr[g];
        std::memset(&h, 0, sizeof h);
        const uint16_t* p = h_planes + static_cast<size_t>(g) * 32;
        int stones = 0;
        for (int y = 0; y < 15; ++y) {
            if ((p[y] & p[16 + y]) || ((p[y] | p[16 + y]) & 0x8000u)) { gmk::set_error("gmk_az_set_roots: game %d has an invalid position", g); return GMK_ERR_ARG; }
            h.rows[y] = static_cast<uint32_t>(p[y]) | (static_cast<uint32_t>(p[16 + y]) << 16);
            stones += __builtin_popcount(h.rows[y]);
        }
        h.n_nodes = 1;
        h.stones = static_cast<uint32_t>(stones);
        const int16_t l1 = h_last_moves[2 * g], l2 = h_last_moves[2 * g + 1];
        h.last_move = l1 >= 0 && l1 < 225 ? static_cast<uint32_t>(l1) : 255u;
        h.last_move2 = l2 >= 0 && l2 < 225 ? static_cast<uint32_t>(l2) : 255u;
    }
    GMK_HIP_CHECK(hipDeviceSynchronize());
    GMK_HIP_CHECK(hipMemcpy(a->t.hdr, hdr.data(), hdr.size() * sizeof(AzHeader), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(az_init_roots_kernel, dim3((a->t.n_games + 255) / 256), dim3(256), 0, nullptr, a->t);
    GMK_HIP_CHECK(hipGetLastError());
    GMK_HIP_CHECK(hipDeviceSynchronize());
    a->rooted = true;
    if (const int rc = az_compact(a, nullptr); rc != GMK_OK) return rc;
    (void)gmk::device_free(a->slots.slot_game); (void)gmk::device_free(a->d_open_moves); (void)gmk::device_free(a->d_open_lens);       // slot g plays game g again
    a->slots = AzSlots{}; a->d_open_moves = nullptr; a->d_open_lens = nullptr;
    return GMK_OK;
}

// Continuous batching for whole-game self-play: the handle's n_games slots play n_total games between them.  The first min(n_games,
// n_total) games start in the slots, from their openings (this call makes the roots: it takes gmk_az_set_roots's place); gmk_az_advance
// then writes a slot's move into the record rows of the GAME it plays and hands a finished game's slot to the next unstarted game.
extern "C" int gmk_az_set_slots(gmk_az* a, int n_total, const uint8_t* h_open_moves, int open_stride, const int32_t* h_open_lens) {
    if (!a || n_total <= 0 || (h_open_moves && (!h_open_lens || open_stride <= 0))) { gmk::set_error("gmk_az_set_slots: bad arguments"); return GMK_ERR_ARG; }
    const int n_slots = a->t.n_games, started = std::min(n_slots, n_total);
    const size_t ns = static_cast<size_t>(n_slots), nt = static_cast<size_t>(n_total);
    std::vector<int32_t> open_lens(nt, 0);
    if (h_open_moves)
        for (size_t g = 0; g < nt; ++g) {
            if (h_open_lens[g] < 0 || h_open_lens[g] > 8 || h_open_lens[g] > open_stride) { gmk::set_error("gmk_az_set_slots: opening of game %zu has %d moves (at most 8: an opening cannot be a finished game)", g, h_open_lens[g]); return GMK_ERR_ARG; }
            open_lens[g] = h_open_lens[g];
            uint32_t seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // every opening is checked here: the device plays the later ones unseen
            for (int i = 0; i < open_lens[g]; ++i) {
                const uint32_t c = h_open_moves[g * static_cast<size_t>(open_stride) + i];
                if (c >= 225u || ((seen[c >> 5] >> (c & 31u)) & 1u)) { gmk::set_error("gmk_az_set_slots: opening of game %zu is not a sequence of moves", g); return GMK_ERR_ARG; }
                seen[c >> 5] |= 1u << (c & 31u);
            }
        }
    std::vector<AzHeader> hdr(ns);
    std::vector<int32_t> state(ns + 1, -1);
    for (size_t g = 0; g < ns; ++g) {
        AzHeader& h = hdr[g];
        std::memset(&h, 0, sizeof h);
        h.n_nodes = 1;
        h.last_move = h.last_move2 = 255u;
        if (g >= static_cast<size_t>(started)) { h.status = kStatusOver; continue; }
        state[g] = static_cast<int32_t>(g);
        for (int i = 0; i < open_lens[g]; ++i) {
            const uint32_t c = h_open_moves[g * static_cast<size_t>(open_stride) + i];
            if (c >= 225u || ((h.rows[c / 15u] | (h.rows[c / 15u] >> 16)) >> (c % 15u)) & 1u) { gmk::set_error("gmk_az_set_slots: opening of game %zu is not a sequence of moves", g); return GMK_ERR_ARG; }
            h.rows[c / 15u] |= 1u << (c % 15u + ((i & 1) ? 16 : 0));
            h.last_move2 = h.last_move;
            h.last_move = c;
        }
        h.stones = static_cast<uint32_t>(open_lens[g]);
    }
    state[ns] = started;                                         // next_game
    GMK_HIP_CHECK(hipDeviceSynchronize());
    (void)gmk::device_free(a->slots.slot_game); (void)gmk::device_free(a->d_open_moves); (void)gmk::device_free(a->d_open_lens);
    a->slots = AzSlots{}; a->d_open_moves = nullptr; a->d_open_lens = nullptr;
    GMK_HIP_CHECK(gmk::device_malloc(&a->slots.slot_game, state.size() * 4));
    GMK_HIP_CHECK(hipMemcpy(a->slots.slot_game, state.data(), state.size() * 4, hipMemcpyHostToDevice));
    GMK_HIP_CHECK(gmk::device_malloc(&a->d_open_lens, nt * 4));
    GMK_HIP_CHECK(hipMemcpy(a->d_open_lens, open_lens.data(), nt * 4, hipMemcpyHostToDevice));
    if (h_open_moves) {
        GMK_HIP_CHECK(gmk::device_malloc(&a->d_open_moves, nt * static_cast<size_t>(open_stride)));
        GMK_HIP_CHECK(hipMemcpy(a->d_open_moves, h_open_moves, nt * static_cast<size_t>(open_stride), hipMemcpyHostToDevice));
    }
    a->slots.next_game = a->slots.slot_game + ns;
    a->slots.n_total = n_total;
    a->slots.open_moves = a->d_open_moves; a->slots.open_lens = a->d_open_lens; a->slots.open_stride = open_stride;
    GMK_HIP_CHECK(hipMemcpy(a->t.hdr, hdr.data(), hdr.size() * sizeof(AzHeader), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(az_init_roots_kernel, dim3((a->t.n_games + 255) / 256), dim3(256), 0, nullptr, a->t);
    GMK_HIP_CHECK(hipGetLastError());
    GMK_HIP_CHECK(hipDeviceSynchronize());
    for (size_t g = 0; g < ns; ++g) a->game_ids[g] = state[g] >= 0 ? static_cast<uint32_t>(state[g]) : 0u;
    a->rooted = true;
    return az_compact(a, nullptr);
}

extern "C" int gmk_az_select(gmk_az* a, float* d_states, void* stream) {
    if (!a || !d_states) { gmk::set_error("gmk_az_select: bad arguments"); return GMK_ERR_ARG; }
    if (!a->rooted) { gmk::set_error("gmk_az_select: gmk_az_set_roots has not been called"); return GMK_ERR_STATE; }
    hipLaunchKernelGGL(az_select_kernel, dim3(a->t.n_games), dim3(64), 0, static_cast<hipStream_t>(stream), a->t, d_states);
    GMK_HIP_CHECK(hipGetLastError());
    return GMK_OK;
}

extern "C" int gmk_az_expand(gmk_az* a, const float* d_values, const float* d_probs, void* stream) {
    if (!a || !d_values || !d_probs) { gmk::set_error("gmk_az_expand: bad arguments"); return GMK_ERR_ARG; }
    if (!a->rooted) { gmk::set_error("gmk_az_expand: gmk_az_set_roots has not been called"); return GMK_ERR_STATE; }
    hipLaunchKernelGGL(az_expand_kernel, dim3(a->t.n_games), dim3(64), 0, static_cast<hipStream_t>(stream), a->t, d_values, d_probs, 3);
    GMK_HIP_CHECK(hipGetLastError());
    return GMK_OK;
}

extern "C" int gmk_az_root_stats(gmk_az* a, uint32_t* h_visits, float* h_values, float* h_priors, uint32_t* h_root_visits,
                                 float* h_root_value, int32_t* h_n_nodes, int32_t* h_status);

static int az_second_arena(gmk_az* a) {
    if (a->second_arena) return GMK_OK;
    const size_t n = static_cast<size_t>(a->t.n_games), nodes = n * static_cast<size_t>(a->t.cap);
    const bool ok = gmk::device_malloc(&a->other.stat, nodes * 8) == hipSuccess && gmk::device_malloc(&a->other.kids, nodes * 8) == hipSuccess &&
                    gmk::device_malloc(&a->other.prior, nodes * 4) == hipSuccess && gmk::device_malloc(&a->other.parent, nodes * 4) == hipSuccess &&
                    gmk::device_malloc(&a->d_forced, n * 2) == hipSuccess;
    if (!ok) {                                                      // all or nothing: a later call must not find half an arena
        (void)gmk::device_free(a->other.stat); (void)gmk::device_free(a->other.kids); (void)gmk::device_free(a->other.prior); (void)gmk::device_free(a->other.parent); (void)gmk::device_free(a->d_forced);
        a->other = AzArena{}; a->d_forced = nullptr;
        (void)hipGetLastError();
        gmk::set_error("gmk_az: hipMalloc of the second arena (%zu nodes) failed", nodes);
        return GMK_ERR_HIP;
    }
    a->second_arena = true;
    return GMK_OK;
}

extern "C" int gmk_az_step(gmk_az* a, const int16_t* h_moves) {
    if (!a) { gmk::set_error("gmk_az_step: bad arguments"); return GMK_ERR_ARG; }
    if (!a->rooted) { gmk::set_error("gmk_az_step: gmk_az_set_roots has not been called"); return GMK_ERR_STATE; }
    const size_t n = static_cast<size_t>(a->t.n_games);
    if (const int rc = az_second_arena(a); rc != GMK_OK) return rc;
    GMK_HIP_CHECK(hipDeviceSynchronize());
    if (h_moves) GMK_HIP_CHECK(hipMemcpy(a->d_forced, h_moves, n * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(az_step_kernel, dim3(a->t.n_games), dim3(64), 0, nullptr, a->t, a->other, h_moves ? a->d_forced : nullptr);
    GMK_HIP_CHECK(hipGetLastError());
    GMK_HIP_CHECK(hipDeviceSynchronize());
    std::swap(a->t.stat, a->other.stat); std::swap(a->t.kids, a->other.kids); std::swap(a->t.prior, a->other.prior); std::swap(a->t.parent, a->other.parent);
    return GMK_OK;
}

// One self-play move of every game still played (az_advance_kernel).  h_unfinished: the games that go on.
extern "C" int gmk_az_advance(gmk_az* a, uint8_t* d_moves, uint16_t* d_visits, int32_t* d_lens, int8_t* d_winner, int reuse_subtree, int32_t* h_unfinished,
                              void* stream) {
    if (!a || !d_moves || !d_lens || !d_winner) { gmk::set_error("gmk_az_advance: bad arguments"); return GMK_ERR_ARG; }
    if (!a->rooted) { gmk::set_error("gmk_az_advance: gmk_az_set_roots has not been called"); return GMK_ERR_STATE; }
    if (reuse_subtree)
        if (const int rc = az_second_arena(a); rc != GMK_OK) return rc;
    if (!a->d_unfinished) GMK_HIP_CHECK(gmk::device_malloc(&a->d_unfinished, 4));
    hipStream_t s = static_cast<hipStream_t>(stream);
    GMK_HIP_CHECK(hipMemsetAsync(a->d_unfinished, 0, 4, s));
    hipLaunchKernelGGL(az_advance_kernel, dim3(a->t.n_games), dim3(64), 0, s, a->t, a->other, d_moves, d_visits, d_lens, d_winner, a->d_unfinished, reuse_subtree ? 1 : 0, a->slots);
    GMK_HIP_CHECK(hipGetLastError());
    int32_t unfinished = 0;
    GMK_HIP_CHECK(hipMemcpyAsync(&unfinished, a->d_unfinished, 4, hipMemcpyDeviceToHost, s));
    if (const int rc = az_compact(a, s); rc != GMK_OK) return rc;   // (synchronises: both numbers are here now, and they agree)
    if (reuse_subtree) { std::swap(a->t.stat, a->other.stat); std::swap(a->t.kids, a->other.kids); std::swap(a->t.prior, a->other.prior); std::swap(a->t.parent, a->other.parent); }
    if (h_unfinished) *h_unfinished = unfinished;
    return GMK_OK;
}

extern "C" int gmk_az_set_option(gmk_az* a, int option, int value) {
    if (a && option == GMK_OPT_NOISE_SAMPLER && (value == GMK_NOISE_SAMPLER_STD || value == GMK_NOISE_SAMPLER_COUNTER)) { a->noise_sampler = value; return GMK_OK; }
    gmk::set_error("gmk_az_set_option: bad handle, unknown option %d or value %d", option, value);
    return GMK_ERR_ARG;
}

// Default::AddNoise on every root with children, seeded like gmk_mcts_add_root_noise / gmk_trad_add_root_noise
extern "C" int gmk_az_add_root_noise(gmk_az* a, float alpha, float epsilon, uint64_t seed, uint32_t first_game_id) {
    if (!a || !(alpha > 0.0f)) { gmk::set_error("gmk_az_add_root_noise: bad arguments"); return GMK_ERR_ARG; }
    if (!a->rooted) { gmk::set_error("gmk_az_add_root_noise: gmk_az_set_roots has not been called"); return GMK_ERR_STATE; }
    const size_t n = static_cast<size_t>(a->t.n_games);
    if (a->noise_sampler == GMK_NOISE_SAMPLER_COUNTER) {         // drawn on the device, one wavefront per game: nothing comes back to the host
        if (!a->slots.slot_game) {
            if (!a->d_game_ids) GMK_HIP_CHECK(gmk::device_malloc(&a->d_game_ids, n * 4));
            GMK_HIP_CHECK(hipMemcpy(a->d_game_ids, a->game_ids.data(), n * 4, hipMemcpyHostToDevice));
        }
        hipLaunchKernelGGL(az_root_noise_kernel, dim3(a->t.n_games), dim3(64), 0, nullptr, a->t, a->slots.slot_game, a->d_game_ids, first_game_id, alpha, epsilon,
                           static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
        GMK_HIP_CHECK(hipGetLastError());
        return GMK_OK;
    }
    if (a->slots.slot_game) {                                    // continuous batching: the random stream belongs to the GAME a slot plays
        std::vector<int32_t> slot_game(n);
        GMK_HIP_CHECK(hipMemcpy(slot_game.data(), a->slots.slot_game, n * 4, hipMemcpyDeviceToHost));
        for (size_t g = 0; g < n; ++g) a->game_ids[g] = slot_game[g] >= 0 ? static_cast<uint32_t>(slot_game[g]) : 0u;
    }
    std::vector<float> priors(n * 225);
    int rc = gmk_az_root_stats(a, nullptr, nullptr, priors.data(), nullptr, nullptr, nullptr, nullptr);
    if (rc != GMK_OK) return rc;
    std::vector<AzHeader> hdr(n);
    GMK_HIP_CHECK(hipMemcpy(hdr.data(), a->t.hdr, n * sizeof(AzHeader), hipMemcpyDeviceToHost));
    gmk::for_each_game(n, [&](size_t g) {
        float* p = &priors[g * 225];
        bool any = false;
        for (int i = 0; i < 225; ++i) any |= p[i] != 0.0f;
        if (any) gmk::mix_root_noise(p, 225, alpha, epsilon, gmk::root_noise_engine_seed(seed, first_game_id + a->game_ids[g], hdr[g].stones));
    });
    if (!a->d_noise_priors) GMK_HIP_CHECK(gmk::device_malloc(&a->d_noise_priors, n * 225 * 4));
    GMK_HIP_CHECK(hipMemcpy(a->d_noise_priors, priors.data(), n * 225 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(az_set_root_priors_kernel, dim3(a->t.n_games), dim3(64), 0, nullptr, a->t, a->d_noise_priors);
    GMK_HIP_CHECK(hipGetLastError());
    GMK_HIP_CHECK(hipDeviceSynchronize());
    return GMK_OK;
}

// ---- host-driven form for callers that evaluate leaves on the host (one game behind CorePyExt: the evaluator is a Python
// callable that wants a Board): the moves from the root to each pending leaf, and value / probabilities from host memory ----
namespace {
__global__ void az_leaf_path_kernel(AzTree t, int16_t* paths, int32_t* lens) {
    const int game = blockIdx.x * blockDim.x + threadIdx.x;
    if (game >= t.n_games) return;
    const AzHeader& hdr = t.hdr[game];
    int16_t* out = paths + static_cast<size_t>(game) * 226;
    if (!hdr.leaf_pending) { lens[game] = -1; return; }
    const size_t arena = static_cast<size_t>(game) * t.cap;
    int depth = 0;
    for (uint32_t node = hdr.leaf; node != 0u && node != kNoNode; node = t.parent[arena + node]) ++depth;
    lens[game] = depth;
    int i = depth;
    for (uint32_t node = hdr.leaf; node != 0u && node != kNoNode; node = t.parent[arena + node])
        out[--i] = static_cast<int16_t>((t.kids[arena + node].y >> 8) & 0xFFu);
}
}  // namespace

static bool az_host_scratch(gmk_az* a) {
    if (a->h_states) return true;
    const size_t n = static_cast<size_t>(a->t.n_games);
    return gmk::device_malloc(&a->h_states, n * 6 * 225 * 4) == hipSuccess && gmk::device_malloc(&a->h_values, n * 4) == hipSuccess &&
           gmk::device_malloc(&a->h_probs, n * 225 * 4) == hipSuccess && gmk::device_malloc(&a->h_paths, n * 226 * 2) == hipSuccess &&
           gmk::device_malloc(&a->h_lens, n * 4) == hipSuccess;
}

// gmk_az_select, then for every game the moves from the root to its pending leaf: h_paths int16[n][226], h_lens int32[n]
// (-1 = the playout ended at a finished game and is already backed up: nothing to evaluate)
extern "C" int gmk_az_select_host(gmk_az* a, int16_t* h_paths, int32_t* h_lens) {
    if (!a || !h_paths || !h_lens) { gmk::set_error("gmk_az_select_host: bad arguments"); return GMK_ERR_ARG; }
    if (!az_host_scratch(a)) { gmk::set_error("gmk_az_select_host: device allocation failed"); return GMK_ERR_HIP; }
    const int rc = gmk_az_select(a, a->h_states, nullptr);
    if (rc != GMK_OK) return rc;
    hipLaunchKernelGGL(az_leaf_path_kernel, dim3((a->t.n_games + 63) / 64), dim3(64), 0, nullptr, a->t, a->h_paths, a->h_lens);
    GMK_HIP_CHECK(hipGetLastError());
    GMK_HIP_CHECK(hipMemcpy(h_paths, a->h_paths, static_cast<size_t>(a->t.n_games) * 226 * 2, hipMemcpyDeviceToHost));
    GMK_HIP_CHECK(hipMemcpy(h_lens, a->h_lens, static_cast<size_t>(a->t.n_games) * 4, hipMemcpyDeviceToHost));
    return GMK_OK;
}

extern "C" int gmk_az_expand_host(gmk_az* a, const float* h_values, const float* h_probs) {
    if (!a || !h_values || !h_probs) { gmk::set_error("gmk_az_expand_host: bad arguments"); return GMK_ERR_ARG; }
    if (!az_host_scratch(a)) { gmk::set_error("gmk_az_expand_host: device allocation failed"); return GMK_ERR_HIP; }
    GMK_HIP_CHECK(hipMemcpy(a->h_values, h_values, static_cast<size_t>(a->t.n_games) * 4, hipMemcpyHostToDevice));
    GMK_HIP_CHECK(hipMemcpy(a->h_probs, h_probs, static_cast<size_t>(a->t.n_games) * 225 * 4, hipMemcpyHostToDevice));
    const int rc = gmk_az_expand(a, a->h_values, a->h_probs, nullptr);
    if (rc != GMK_OK) return rc;
    GMK_HIP_CHECK(hipDeviceSynchronize());
    return GMK_OK;
}

// ---- Host-driven stages: Policy(select=, expand=, back_prop=) hands Python callables to MCTS::playout
// (core/py_ext/src/mcts_ext.hpp:43-61, core/lib/include/MCTS.h:74-101).  A Python callable runs on the host by definition; the tree stays on the
// device and the host reads the node it is asked about, decides, and tells the device: a node and its children out, the chosen leaf in, expand
// and / or back up as separate steps, statistics of a path written back, and Default::Simulate's random rollout (MonteCarlo.hpp:83-88) from the
// pending leaf as a one-wavefront kernel on the rollout code of K3.  One game of the handle at a time; synchronous; for CorePyExt's one-game MCTS.
namespace {
__global__ __launch_bounds__(64)
void az_leaf_rollout_kernel(AzTree t, int game, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t k0, uint32_t k1, int32_t* winner) {
    using namespace gmk::rollout;
    __shared__ uint32_t s_lines[kLineWords];
    __shared__ uint2 s_cells[30];
    const int lane = threadIdx.x;
    const AzHeader& hdr = t.hdr[game];
    const int stones = static_cast<int>(hdr.leaf_stones);
    for (int w = lane; w < kLineWords; w += 64) s_lines[w] = 0u;
    __syncthreads();
    if (lane < 15) {
        const uint32_t row = hdr.leaf_rows[lane];
        s_lines[lane] = row;
        for (uint32_t m = (row | (row >> 16)) & 0x7FFFu; m; m &= m - 1u) {
            const int x = __ffs(m) - 1, y = lane;
            const uint32_t cb = ((row >> x) & 1u) ? 0u : 16u;
            atomicOr(&s_lines[kColBase + x], 1u << (y + cb));
            atomicOr(&s_lines[kDiagBase + x - y + 14], 1u << (x + cb));
            atomicOr(&s_lines[kAntiBase + x + y], 1u << (x + cb));
        }
    }
    if (lane < 30 && 8 * lane < 225 - stones) s_cells[lane] = rollout_cells(c0, c1, c2, static_cast<uint32_t>(lane), k0, k1);
    __syncthreads();
    if (lane == 0) *winner = random_rollout_blocks(s_lines, 1u, (stones & 1) ? -1 : 1, stones, 0, [&](uint32_t b) { return s_cells[b]; });
}
}  // namespace

extern "C" int gmk_az_read_node_host(gmk_az* a, int game, uint32_t node, uint32_t* h_visits, float* h_value, float* h_prior, int32_t* h_cell, uint32_t* h_parent,
                                     uint32_t* h_first_child, int32_t* h_n_children) {
    if (!a || game < 0 || game >= a->t.n_games || node >= static_cast<uint32_t>(a->t.cap)) { gmk::set_error("gmk_az_read_node_host: bad arguments"); return GMK_ERR_ARG; }
    if (!a->rooted) { gmk::set_error("gmk_az_read_node_host: gmk_az_set_roots has not been called"); return GMK_ERR_STATE; }
    const size_t at = static_cast<size_t>(game) * a->t.cap + node;
    uint2 st, kd;
    float pr;
    uint32_t par;
    GMK_HIP_CHECK(hipDeviceSynchronize());
    GMK_HIP_CHECK(hipMemcpy(&st, a->t.stat + at, 8, hipMemcpyDeviceToHost));
    GMK_HIP_CHECK(hipMemcpy(&kd, a->t.kids + at, 8, hipMemcpyDeviceToHost));
    GMK_HIP_CHECK(hipMemcpy(&pr, a->t.prior + at, 4, hipMemcpyDeviceToHost));
    GMK_HIP_CHECK(hipMemcpy(&par, a->t.parent + at, 4, hipMemcpyDeviceToHost));
    if (h_visits) *h_visits = st.x;
    if (h_value) std::memcpy(h_value, &st.y, 4);
    if (h_prior) *h_prior = pr;
    if (h_cell) *h_cell = static_cast<int32_t>((kd.y >> 8) & 0xFFu);
    if (h_parent) *h_parent = par;
    if (h_first_child) *h_first_child = kd.x;
    if (h_n_children) *h_n_children = static_cast<int32_t>(kd.y & 0xFFu);
    return GMK_OK;
}

extern "C" int gmk_az_read_children_host(gmk_az* a, int game, uint32_t first_child, int n, int16_t* h_cells, uint32_t* h_visits, float* h_values, float* h_priors,
                                         int32_t* h_n_children) {
    if (!a || game < 0 || game >= a->t.n_games || n < 0 || n > 225 || static_cast<size_t>(first_child) + n > static_cast<size_t>(a->t.cap)) { gmk::set_error("gmk_az_read_children_host: bad arguments"); return GMK_ERR_ARG; }
    if (n == 0) return GMK_OK;
    const size_t at = static_cast<size_t>(game) * a->t.cap + first_child;
    uint2 st[225], kd[225];
    GMK_HIP_CHECK(hipDeviceSynchronize());
    GMK_HIP_CHECK(hipMemcpy(st, a->t.stat + at, static_cast<size_t>(n) * 8, hipMemcpyDeviceToHost));
    GMK_HIP_CHECK(hipMemcpy(kd, a->t.kids + at, static_cast<size_t>(n) * 8, hipMemcpyDeviceToHost));
    if (h_priors) GMK_HIP_CHECK(hipMemcpy(h_priors, a->t.prior + at, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        if (h_cells) h_cells[i] = static_cast<int16_t>((kd[i].y >> 8) & 0xFFu);
        if (h_visits) h_visits[i] = st[i].x;
        if (h_values) std::memcpy(&h_values[i], &st[i].y, 4);
        if (h_n_children) h_n_children[i] = static_cast<int32_t>(kd[i].y & 0xFFu);
    }
    return GMK_OK;
}

// the leaf the host's descent ended at: node `leaf`, reached from the root over h_path[0 .. depth): it becomes the pending leaf of the next
// gmk_az_expand_stages_host / gmk_az_rollout_host (what az_select_kernel leaves behind for a device-side descent)
extern "C" int gmk_az_set_leaf_host(gmk_az* a, int game, uint32_t leaf, const int16_t* h_path, int depth) {
    if (!a || game < 0 || game >= a->t.n_games || depth < 0 || depth > 225 || (depth > 0 && !h_path) || leaf >= static_cast<uint32_t>(a->t.cap)) { gmk::set_error("gmk_az_set_leaf_host: bad arguments"); return GMK_ERR_ARG; }
    if (!a->rooted) { gmk::set_error("gmk_az_set_leaf_host: gmk_az_set_roots has not been called"); return GMK_ERR_STATE; }
    AzHeader h;
    GMK_HIP_CHECK(hipDeviceSynchronize());
    GMK_HIP_CHECK(hipMemcpy(&h, a->t.hdr + game, sizeof h, hipMemcpyDeviceToHost));
    for (int y = 0; y < 16; ++y) h.leaf_rows[y] = h.rows[y];
    uint32_t stones = h.stones;
    for (int i = 0; i < depth; ++i, ++stones) {
        const int c = h_path[i];
        if (c < 0 || c >= 225 || ((h.leaf_rows[c / 15] | (h.leaf_rows[c / 15] >> 16)) >> (c % 15)) & 1u) { gmk::set_error("gmk_az_set_leaf_host: the path is not a sequence of moves"); return GMK_ERR_ARG; }
        h.leaf_rows[c / 15] |= 1u << (c % 15 + ((stones & 1u) ? 16 : 0));
    }
    h.leaf = leaf; h.leaf_pending = 1; h.leaf_stones = stones;
    GMK_HIP_CHECK(hipMemcpy(a->t.hdr + game, &h, sizeof h, hipMemcpyHostToDevice));
    return GMK_OK;
}

// gmk_az_expand_host with the two halves of az_expand_kernel switched separately: expand (children with h_probs, Default::Expand) and / or
// back up (-h_values along the parent chain, Default::BackPropogate); the pending leaves are done either way
extern "C" int gmk_az_expand_stages_host(gmk_az* a, const float* h_values, const float* h_probs, int do_expand, int do_backup) {
    if (!a || (do_backup && !h_values) || (do_expand && !h_probs)) { gmk::set_error("gmk_az_expand_stages_host: bad arguments"); return GMK_ERR_ARG; }
    if (!a->rooted) { gmk::set_error("gmk_az_expand_stages_host: gmk_az_set_roots has not been called"); return GMK_ERR_STATE; }
    if (!az_host_scratch(a)) { gmk::set_error("gmk_az_expand_stages_host: device allocation failed"); return GMK_ERR_HIP; }
    if (h_values) GMK_HIP_CHECK(hipMemcpy(a->h_values, h_values, static_cast<size_t>(a->t.n_games) * 4, hipMemcpyHostToDevice));
    if (h_probs) GMK_HIP_CHECK(hipMemcpy(a->h_probs, h_probs, static_cast<size_t>(a->t.n_games) * 225 * 4, hipMemcpyHostToDevice));
    else GMK_HIP_CHECK(hipMemset(a->h_probs, 0, static_cast<size_t>(a->t.n_games) * 225 * 4));
    hipLaunchKernelGGL(az_expand_kernel, dim3(a->t.n_games), dim3(64), 0, nullptr, a->t, a->h_values, a->h_probs, (do_expand ? 1 : 0) | (do_backup ? 2 : 0));
    GMK_HIP_CHECK(hipGetLastError());
    GMK_HIP_CHECK(hipDeviceSynchronize());
    return GMK_OK;
}

// node statistics as a Python back_prop left them: h_nodes[i] gets {h_visits[i], h_values[i]}
extern "C" int gmk_az_write_stats_host(gmk_az* a, int game, const uint32_t* h_nodes, const uint32_t* h_visits, const float* h_values, int n) {
    if (!a || game < 0 || game >= a->t.n_games || n < 0 || (n > 0 && (!h_nodes || !h_visits || !h_values))) { gmk::set_error("gmk_az_write_stats_host: bad arguments"); return GMK_ERR_ARG; }
    GMK_HIP_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < n; ++i) {
        if (h_nodes[i] >= static_cast<uint32_t>(a->t.cap)) { gmk::set_error("gmk_az_write_stats_host: node %u outside the arena", h_nodes[i]); return GMK_ERR_ARG; }
        uint2 st;
        st.x = h_visits[i];
        std::memcpy(&st.y, &h_values[i], 4);
        GMK_HIP_CHECK(hipMemcpy(a->t.stat + static_cast<size_t>(game) * a->t.cap + h_nodes[i], &st, 8, hipMemcpyHostToDevice));
    }
    return GMK_OK;
}

// Default::Simulate's random rollout from the pending leaf of `game` (one wavefront, the rollout of K3): the winner (+1 black, -1 white, 0 tie).
// The cell draws are Philox(seed; counter0, counter1, counter2, block of eight plies): with (global game id, playout number, root stones << 8)
// they are the draws of K3's first rollout lane, i.e. MCTS(RandomPolicy(c_puct, 1)) of gmk_mcts_*.
extern "C" int gmk_az_rollout_host(gmk_az* a, int game, uint64_t seed, uint32_t counter0, uint32_t counter1, uint32_t counter2, int32_t* h_winner) {
    if (!a || game < 0 || game >= a->t.n_games || !h_winner) { gmk::set_error("gmk_az_rollout_host: bad arguments"); return GMK_ERR_ARG; }
    if (!a->rooted) { gmk::set_error("gmk_az_rollout_host: gmk_az_set_roots has not been called"); return GMK_ERR_STATE; }
    if (!a->d_unfinished) GMK_HIP_CHECK(gmk::device_malloc(&a->d_unfinished, 4));
    hipLaunchKernelGGL(az_leaf_rollout_kernel, dim3(1), dim3(64), 0, nullptr, a->t, game, counter0, counter1, counter2, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), a->d_unfinished);
    GMK_HIP_CHECK(hipGetLastError());
    GMK_HIP_CHECK(hipMemcpy(h_winner, a->d_unfinished, 4, hipMemcpyDeviceToHost));
    return GMK_OK;
}

extern "C" int gmk_az_root_stats(gmk_az* a, uint32_t* h_visits, float* h_values, float* h_priors, uint32_t* h_root_visits,
                                 float* h_root_value, int32_t* h_n_nodes, int32_t* h_status) {
    if (!a) { gmk::set_error("gmk_az_root_stats: bad arguments"); return GMK_ERR_ARG; }
    if (!a->rooted) { gmk::set_error("gmk_az_root_stats: gmk_az_set_roots has not been called"); return GMK_ERR_STATE; }
    const size_t n = static_cast<size_t>(a->t.n_games);
    uint32_t *d_visits = nullptr, *d_root_visits = nullptr;
    float *d_values = nullptr, *d_priors = nullptr, *d_root_value = nullptr;
    auto cleanup = [&]() { (void)gmk::device_free(d_visits); (void)gmk::device_free(d_values); (void)gmk::device_free(d_priors); (void)gmk::device_free(d_root_visits); (void)gmk::device_free(d_root_value); };
#define GMK_TRY(expr) do { if ((expr) != hipSuccess) { gmk::set_error("%s failed", #expr); cleanup(); return GMK_ERR_HIP; } } while (0)
    GMK_TRY(gmk::device_malloc(&d_visits, n * 225 * 4)); GMK_TRY(gmk::device_malloc(&d_values, n * 225 * 4)); GMK_TRY(gmk::device_malloc(&d_priors, n * 225 * 4));
    GMK_TRY(gmk::device_malloc(&d_root_visits, n * 4)); GMK_TRY(gmk::device_malloc(&d_root_value, n * 4));
    GMK_TRY(hipMemset(d_visits, 0, n * 225 * 4)); GMK_TRY(hipMemset(d_values, 0, n * 225 * 4)); GMK_TRY(hipMemset(d_priors, 0, n * 225 * 4));
    GMK_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL(az_root_stats_kernel, dim3(a->t.n_games), dim3(64), 0, nullptr, a->t, d_visits, d_values, d_priors, d_root_visits, d_root_value);
    GMK_TRY(hipGetLastError());
    GMK_TRY(hipDeviceSynchronize());
    if (h_visits) GMK_TRY(hipMemcpy(h_visits, d_visits, n * 225 * 4, hipMemcpyDeviceToHost));
    if (h_values) GMK_TRY(hipMemcpy(h_values, d_values, n * 225 * 4, hipMemcpyDeviceToHost));
    if (h_priors) GMK_TRY(hipMemcpy(h_priors, d_priors, n * 225 * 4, hipMemcpyDeviceToHost));
    if (h_root_visits) GMK_TRY(hipMemcpy(h_root_visits, d_root_visits, n * 4, hipMemcpyDeviceToHost));
    if (h_root_value) GMK_TRY(hipMemcpy(h_root_value, d_root_value, n * 4, hipMemcpyDeviceToHost));
    std::vector<AzHeader> hdr(n);
    GMK_TRY(hipMemcpy(hdr.data(), a->t.hdr, n * sizeof(AzHeader), hipMemcpyDeviceToHost));
#undef GMK_TRY
    for (size_t g = 0; g < n; ++g) {
        if (h_n_nodes) h_n_nodes[g] = static_cast<int32_t>(hdr[g].n_nodes);
        if (h_status) h_status[g] = static_cast<int32_t>(hdr[g].status);
    }
    cleanup();
    return GMK_OK;
}
