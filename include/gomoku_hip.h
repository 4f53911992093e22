/*
 * gomoku_hip.h -- C-ABI of libgomoku_hip.so, the MI355X (gfx950) implementation of the GomokuAI
 * self-play hot path: the Aho-Corasick line-pattern evaluator and the MCTS playout loop.
 *
 * This is the drop-in boundary underneath the reference's pybind11 module `CorePyExt`
 * (core/py_ext/src/module.cpp:7-13): the C++ binding layer (gomokuai_amd/csrc/core_pyext.cpp, or the
 * stub shown in INTEGRATION.md for the reference tree) is the only code that touches Python objects;
 * everything below is plain pointers and sizes.  Paths in comments are relative to the reference root.
 *
 * Conventions
 *   - every function returns 0 on success, a negative gmk_status otherwise; gmk_last_error() has text;
 *   - `d_` pointers are DEVICE (HBM) pointers, `h_` pointers are host pointers;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are asynchronous on it;
 *   - there is no CPU fallback: without a usable HIP device every compute entry returns GMK_ERR_NO_DEVICE.
 *
 * Board encoding (replaces Board::m_moveStates, core/lib/include/Game.h:146-150)
 *   planes: uint16_t[n][2][16]   plane 0 = black stones, plane 1 = white stones,
 *           word y = row y, bit x = column x (x,y in 0..14), word 15 and bit 15 are zero.  64 B per board.
 *   Position id = y*15 + x (Game.h:45-56).  Player: -1 white, 0 none, +1 black (Game.h:19-21).
 */
#ifndef GOMOKU_HIP_H_
#define GOMOKU_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GMK_OK = 0,
    GMK_ERR_NO_DEVICE = -1,     /* no HIP device / HIP runtime error at init */
    GMK_ERR_HIP = -2,           /* a HIP call failed */
    GMK_ERR_ARG = -3,           /* invalid argument */
    GMK_ERR_STATE = -4,         /* library not initialised / handle invalid */
    GMK_ERR_CAPACITY = -5       /* a fixed capacity (tree arena, queue) was exceeded */
} gmk_status;

enum { GMK_BOARD_CELLS = 225, GMK_PLANE_WORDS = 16, GMK_TOTALS = 11 };

/* ---- library ---- */
int gmk_init(int device);                 /* builds the pattern automaton on the host and uploads it */
int gmk_shutdown(void);
/* Large device blocks (the tree arenas: tens of GB per handle) that a destroyed handle gives up are kept by the library for the next handle
 * (the driver clears memory before it hands it out again: seconds per 24 GB); at most 224 GB idle (the longest-idle blocks go first), and never more than three quarters of what the
 * device could hand out if the pool gave everything back (the idle blocks are invisible to other allocators of the process).  This returns the idle ones to the
 * driver, e.g. before another allocator in the process needs the memory; gmk_shutdown does it too. */
int gmk_pool_release(void);
/* Diagnostic switch of that pool (off by default): a block that is handed out again is first filled with 0xA5, so that a kernel which reads a
 * node nobody wrote since meets garbage instead of the previous handle's (plausible) tree.  The pool never clears a block otherwise: the driver's
 * clear is 1.5 s per 24 GB, and no kernel reads what it has not written -- tests/test_pool_gpu.py searches on poisoned blocks to hold that. */
int gmk_pool_poison(int on);
const char *gmk_last_error(void);
int gmk_device_info(int *cu_count, size_t *hbm_bytes, char *name, int name_cap);

/* ---- pattern tables (host side; usable without a GPU) ----
 * Replaces the static `Evaluator::Patterns` (core/lib/src/Pattern.cpp:554-596) and its builder
 * (core/lib/src/utils/ACAutomata.cpp:15-274). */
typedef struct {
    int32_t n_patterns;        /* 294 */
    int32_t n_states;          /* dense DFA states */
    int32_t dat_size;          /* length of base/check/fail (1024) */
    int32_t max_emissions;     /* longest emission list of one transition */
    int32_t trans_words;       /* n_states*4 */
    int32_t emit_words;        /* uint16 entries */
    int32_t invariants[5];
} gmk_table_info;
int gmk_tables_info(gmk_table_info *info);
/* pattern i: rich string (<=7 chars + NUL), favour (+1/-1), type (Pattern::Type, Pattern.h:33-39), score */
int gmk_tables_pattern(int i, char str[8], int *favour, int *type, int *score);
/* copies of the GPU tables (see gomokuai_amd/csrc/pattern_tables.h for the bit layout) */
int gmk_tables_copy(uint32_t *trans, uint16_t *emit_lists, uint32_t *pattern_info);
/* copies of the reference-shaped double array (PatternSearch::m_base/m_check/m_fail, Pattern.h:89-92) */
int gmk_tables_copy_dat(int32_t *base, int32_t *check, int32_t *fail);
/* Table self-check: runs the flattened DFA on the host over codes[n] (1=x 2=o 3=? 4=blank) and writes the
 * (pattern, end offset) stream; returns the number of matches.  Not used by any compute entry. */
int gmk_tables_scan(const uint8_t *codes, int n, int32_t *patterns, int32_t *offsets, int cap);

/* ---- synthetic workloads (host side) ----
 * kind 0 "random-opening": L = 8 + u32 % 53 plies, each ply r = u32 % 225 then the reference probe rule
 *        (core/lib/src/Game.cpp:68-72); stops early on five-in-row.
 * kind 1 "clustered": 90 % of plies land within Chebyshev distance 2 of a random earlier stone.
 * Philox4x32-10, key = seed, counter = (first_board + i, draw, kind, 0).
 * moves: uint8[n][stride] (stride >= 64), lens: int32[n], planes (optional): uint16[n][2][16]. */
int gmk_synth_boards(uint64_t seed, uint32_t first_board, int n, int kind,
                     uint8_t *h_moves, int stride, int32_t *h_lens, uint16_t *h_planes);
/* move list -> planes (black moves first, alternating), host side */
int gmk_moves_to_planes(const uint8_t *h_moves, int stride, const int32_t *h_lens, int n, uint16_t *h_planes);

/* ---- K1: batched position evaluation ----
 * Computes, from the stones alone, what the reference's incrementally maintained Evaluator
 * (core/lib/src/Pattern.cpp:111-386) holds after those stones were played:
 *   scores  int32[n][4][225]     Evaluator::m_scores, group = (favour==black)<<1 | (perspective==black)   (Pattern.h:159-161,219)
 *   density int32[n][2][2][225]  Evaluator::m_density [white,black][count,weight]; occupied cells hold -v-1 (Pattern.cpp:236-272)
 *   totals  uint32[n][11]        m_patternDist[225][0..7] then m_compoundDist[225][0..2]: white count in the low,
 *                                black in the high 16 bits (Pattern.cpp:390-393)
 *   status  int32[n]             bit0 game over, bit1 evaluator error (the reference would read out of bounds,
 *                                Pattern.cpp:484-485, or overflow a queue), bits 8..15 winner, bits 16..23 player to move
 * Any output pointer may be NULL.  All pointers are device pointers; n boards; asynchronous on stream. */
int gmk_eval_batch(const uint16_t *d_planes, int n,
                   int32_t *d_scores, int32_t *d_density, uint32_t *d_totals, int32_t *d_status,
                   void *stream);
/* same with host buffers (allocates, copies in, runs, copies out, synchronises) */
int gmk_eval_batch_host(const uint16_t *h_planes, int n,
                        int32_t *h_scores, int32_t *h_density, uint32_t *h_totals, int32_t *h_status);
/* launch geometry the library chose for gmk_eval_batch (for profiling reports) */
int gmk_eval_launch_info(int n, int *grid, int *block, int *lds_bytes);

/* ---- K2: incrementally maintained evaluator states ----
 * One handle = n_games Evaluator objects (core/lib/include/Pattern.h:142-220) living in HBM, 17 792 B each.  Unlike K1 it keeps
 * everything the reference keeps, including the per-cell 2-bit flag words of m_patternDist / m_compoundDist, which are
 * order-dependent shift registers (Pattern.cpp:395-400) and can only be reproduced by replaying the update rule.
 * gmk_evalstate_update applies moves_per_game entries per game in one launch, entry m of game g at d_moves[g*moves_per_game+m]:
 *   >= 0  Evaluator::applyMove(cell)   (Pattern.cpp:310-335; an illegal or post-game move is ignored, as there)
 *   -1    nothing
 *   -2    Evaluator::revertMove(1)     (Pattern.cpp:337-342)
 * gmk_evalstate_read copies the members out: scores int32[n][4][225], density int32[n][2][2][225], pattern_dist uint32[n][226][8]
 * (row 225 = totals), compound_dist uint32[n][226][3], meta int32[n][4] = {moves played, player to move, winner, error bits},
 * record uint8[n][228] (m_moveRecord).  Any pointer may be NULL. */
typedef struct gmk_evalstate gmk_evalstate;
int gmk_evalstate_create(int n_games, gmk_evalstate **out);
int gmk_evalstate_destroy(gmk_evalstate *e);
int gmk_evalstate_reset(gmk_evalstate *e);                                   /* Evaluator::reset (Pattern.cpp:371-386) */
int gmk_evalstate_update(gmk_evalstate *e, const int16_t *d_moves, int moves_per_game, void *stream);
int gmk_evalstate_update_host(gmk_evalstate *e, const int16_t *h_moves, int moves_per_game);
int gmk_evalstate_read(gmk_evalstate *e, int32_t *h_scores, int32_t *h_density, uint32_t *h_pattern_dist, uint32_t *h_compound_dist,
                       int32_t *h_meta, uint8_t *h_record);

/* ---- K3: batched MCTS with the reference's default RandomPolicy ----
 * One handle = n_games independent searches, each with its own tree arena in HBM.  Replaces, per game,
 * Gomoku::MCTS + Policies::RandomPolicy (core/lib/include/MCTS.h:135-180, core/lib/src/MCTS.cpp:99-198,
 * core/lib/include/algorithms/MonteCarlo.hpp:13-110, core/lib/include/policies/Random.h:22-35):
 *   select   argmax_i Q_i + c_puct * P_i * sqrt(N) / (n_i + 1) in double, first maximum wins
 *   expand   one child per empty cell in ascending id, prior 1/float(#empty)
 *   simulate c_rollouts uniform-probe random games (Game.cpp:64-73), value = float(sum / c_rollouts)
 *   backup   visits += 1; value += (v - value) / float(visits); v = -v, up to the root
 * The reference's random_device-seeded mt19937 is replaced by Philox4x32-10 with
 *   key = seed, counter = (global game id, playout index, (stones on the root board << 8) | rollout, ply >> 3);
 *   ply p uses the 16-bit half (p & 1) of output word (p >> 1) & 3, cell draw = (half * 225) >> 16
 * so results do not depend on how games are spread over GPUs.
 * Node capacity per game: at most 225 - stones new nodes per playout; exceeding it sets bit 1 of status. */
typedef struct gmk_mcts gmk_mcts;
int gmk_mcts_create(int n_games, int node_capacity, double c_puct, int c_rollouts, uint64_t seed, gmk_mcts **out);
int gmk_mcts_destroy(gmk_mcts *m);
/* Fresh roots (MCTS::reset + syncWithBoard on a tree without the position, MCTS.cpp:119-125,149-156):
 * h_planes uint16[n][2][16]; h_last_move int16[n] (-1 for an empty board).  first_game_id = global id of game 0. */
int gmk_mcts_set_roots(gmk_mcts *m, const uint16_t *h_planes, const int16_t *h_last_move, uint32_t first_game_id);
/* The global id of every game (uint32[n], host), overriding first_game_id + g of the last gmk_mcts_set_roots: for callers whose
 * handle plays a changing or non-contiguous set of games (slots handed from finished games to new ones, the groups of a match).
 * The id is word 0 of every random-number counter of the game (rollouts, root noise). */
int gmk_mcts_set_game_ids(gmk_mcts *m, const uint32_t *h_ids);
/* MCTS::runPlayouts with the iteration constraint (MCTS.cpp:179-198): `playouts` playouts for every game, one launch. */
int gmk_mcts_run(gmk_mcts *m, int playouts, void *stream);
/* One self-play move for every unfinished game, to be called after gmk_mcts_run (replaces the loop body of
 * agents/utils.py:29-47 around MCTSAgent.eval_state, agents/mcts.py:17-21):
 *   the most visited root child (first maximum: MCTS::stepForward, MCTS.cpp:129-134) is played on the root position,
 *   (move, root child visit counts by cell) is appended to the game record, the game is closed when the move makes
 *   five or fills the board (Board::checkGameEnd, Game.cpp:88-136), and the tree is re-rooted: reuse_subtree = 0
 *   starts the next search from a fresh one-node tree (MCTS::reset), 1 keeps the subtree of the move.
 * Device buffers: d_moves uint8[n][225], d_visits uint16[n][225][225] (may be NULL), d_lens int32[n] (zero before the
 * first move), d_winner int8[n] (valid once the game is over), d_unfinished int32[1] = games still running afterwards.
 * Finished games are skipped by later gmk_mcts_run calls. */
int gmk_mcts_advance(gmk_mcts *m, uint8_t *d_moves, uint16_t *d_visits, int32_t *d_lens, int8_t *d_winner,
                     int32_t *d_unfinished, int reuse_subtree, void *stream);
/* Whole self-play games, resident on the device with CONTINUOUS BATCHING (replaces the data generation loop of
 * network/data_helper.py:58-83 around agents/utils.py:29-63 for MCTS(RandomPolicy) on both sides): the handle's n_games are slots
 * that play n_total games between them.  Every move of the slots is one gmk_mcts_run (`playouts` playouts) and one gmk_mcts_advance;
 * a slot whose game ends takes the next game nobody has started -- its opening becomes the slot's root, its global id
 * first_game_id + index the slot's random-number key -- so the searches stay full until fewer games than slots remain, and game g's
 * record is the same whichever slot played it and however many slots there are.
 * h_open_moves uint8[n_total][open_stride] / h_open_lens int32[n_total]: opening moves per game, black first, 0 .. 8 of them (NULL: empty boards).
 * Device outputs, indexed by GAME: d_moves uint8[n_total][225] (openings included), d_visits uint16[n_total][225][225] or NULL,
 * d_lens int32[n_total], d_winner int8[n_total].  noise_alpha > 0 with reuse_subtree applies Default::AddNoise before every search.
 * ONE persistent launch plays all games (every wavefront searches and steps its slots' games turn by turn at its own pace; with
 * reuse_subtree the chosen child's subtree is compacted into the game's second arena inside the launch) unless the noise has to be drawn
 * on the host (GMK_NOISE_SAMPLER_STD with noise_alpha > 0) or GMK_OPT_LOCKSTEP is set; the records are the same bytes either way.
 * playouts >= 1 (a root that was never searched has no child to play).  A slot whose game can never move -- a node capacity too small for one
 * expansion -- stops and is reported through status bit 1 (arena full) instead of keeping the launch alive.
 * Synchronous; *h_steps (optional) = search launches it took (lock step: each one move for every busy slot; persistent: 1). */
int gmk_selfplay_run(gmk_mcts *m, int n_total, uint32_t first_game_id, int playouts, int reuse_subtree, float noise_alpha, float noise_epsilon,
                     const uint8_t *h_open_moves, int open_stride, const int32_t *h_open_lens,
                     uint8_t *d_moves, uint16_t *d_visits, int32_t *d_lens, int8_t *d_winner, int32_t *h_steps, void *stream);
/* The same step with the move given: MCTS::stepForward(next_move) (core/lib/src/MCTS.cpp:136-147), e.g. the opponent's
 * reply.  d_forced_moves int16[n] (device): the cell to step to, or -1 for the most visited child (= gmk_mcts_advance).
 * A root that was never expanded simply moves on; with reuse_subtree the child's subtree is kept.  An illegal cell
 * sets status bit 2 of that game and plays nothing. */
int gmk_mcts_step(gmk_mcts* m, const int16_t* d_forced_moves, uint8_t* d_moves, uint16_t* d_visits, int32_t* d_lens,
                  int8_t* d_winner, int32_t* d_unfinished, int reuse_subtree, void* stream);
/* the same from host memory, synchronous, without game records: h_moves int16[n] */
int gmk_mcts_step_host(gmk_mcts* m, const int16_t* h_moves, int reuse_subtree);
/* Default::AddNoise (MonteCarlo.hpp:97-108, Statistical.hpp:29-34) on every unfinished game whose root has children:
 * P <- (1-epsilon) P + epsilon * normalized(gamma(alpha,1)); the reference calls it at the start of every runPlayouts
 * (MCTS.cpp:182) with alpha 0.05, epsilon 0.25.  No-op for childless (fresh) roots.  The priors stay in force until the
 * next gmk_mcts_advance / gmk_mcts_set_roots.  Synchronises `stream`. */
int gmk_mcts_add_root_noise(gmk_mcts *m, float alpha, float epsilon, void *stream);
/* Handle options (K3 gmk_mcts_set_option, K6 / K8 gmk_trad_set_option).
 * GMK_OPT_NOISE_SAMPLER: where Default::AddNoise draws its gamma variates from (the reference: std::gamma_distribution<float> over a
 *   random_device-seeded std::mt19937, Statistical.hpp:22-34 -- a stream without a seed API, unpinned by construction):
 *     GMK_NOISE_SAMPLER_STD (default)  the toolchain's std::gamma_distribution<float> over std::mt19937, seeded per (game, stones) through
 *                                      Philox; drawn on the HOST, so the self-play loops run in lock step (search, step, noise, search ...);
 *     GMK_NOISE_SAMPLER_COUNTER        the counter-based sampler of include/gomoku_noise.h (Philox-keyed Marsaglia-Tsang, one stream per
 *                                      (game, stones, cell)), drawn by the searching wavefront itself: the reference agent's per-move
 *                                      semantics -- kept subtree + noise before every search -- then run inside ONE persistent launch.
 * GMK_OPT_LOCKSTEP: 1 = gmk_selfplay_run / gmk_trad_selfplay_run alternate search and step launches even where one persistent launch could play
 *   the games (the second form the tests hold the persistent one to); 0 (default) = persistent wherever the configuration allows. */
enum { GMK_OPT_NOISE_SAMPLER = 1, GMK_OPT_LOCKSTEP = 2 };
enum { GMK_NOISE_SAMPLER_STD = 0, GMK_NOISE_SAMPLER_COUNTER = 1 };
int gmk_mcts_set_option(gmk_mcts *m, int option, int value);
/* The handle's tree arenas, now: one arena per game (what the first gmk_mcts_set_roots allocates) or, two_arenas != 0, the two arenas per game of
 * the persistent loop with kept subtrees (what gmk_selfplay_run allocates).  Tens of GB, and the driver clears memory it has handed out before
 * (seconds per 24 GB): for callers that want that outside a region they time, or want the blocks in the library's pool before a batch starts
 * (create, reserve, destroy: the next handle of that shape finds them there). */
int gmk_mcts_reserve(gmk_mcts *m, int two_arenas);
/* Root statistics after a run (synchronises the stream used by the last run):
 *   h_visits uint32[n][225] child visit counts by cell (MCTS::evalState, MCTS.cpp:104-110),
 *   h_root_value float[n], h_root_visits uint32[n], h_nodes uint32[n] (MCTS::m_size), h_status int32[n] (bit1: arena full). */
int gmk_mcts_root_stats(gmk_mcts *m, uint32_t *h_visits, float *h_root_value, uint32_t *h_root_visits,
                        uint32_t *h_nodes, int32_t *h_status);
/* algorithmic tree bytes moved by the last run, summed over games (select 8 B/child, expand 16 B/node, backup 16 B/level) */
int gmk_mcts_alg_bytes(gmk_mcts *m, uint64_t *bytes);
int gmk_mcts_launch_info(gmk_mcts *m, int *grid, int *block, int *lds_bytes);
/* pi from visit counts exactly as MCTS::evalState does (MCTS.cpp:112-116, Statistical.hpp:37-42); host side.
 * visits uint32[225], stones = moves on the board (temperature 1 below 15 stones, else 0.01). */
int gmk_visits_to_pi(const uint32_t *visits, int stones, float *pi);

/* ---- K4 + K5: game records -> training tuples, on the device ----
 * Sample s = move d_sample_move[s] of game d_sample_game[s] of the records written by gmk_mcts_advance:
 *   d_states uint8[S][6][225]  Board::encoded_states (core/py_ext/src/game_ext.hpp:87-104) of the position BEFORE that move,
 *   d_values float[S]          Player::calc_score(player to move, winner) (agents/utils.py:55-59),
 *   d_pi     float[S][225]     MCTS::evalState's action probabilities from the recorded visit counts (MCTS.cpp:104-117).
 * augment != 0 writes the eight symmetric copies of every sample (network/data_helper.py:36-55: rot90^i, then fliplr of it):
 * the outputs then hold 8*S samples, copy a of sample s at index 8*s + a.  All pointers are device pointers. */
int gmk_samples_from_records(const uint8_t *d_moves, const int32_t *d_lens, const uint16_t *d_visits, const int8_t *d_winner,
                             const int32_t *d_sample_game, const int32_t *d_sample_move, int n_samples, int augment,
                             uint8_t *d_states, float *d_values, float *d_pi, void *stream);

/* ---- K6: pattern-guided tree search, the reference's self-play supervisor ("traditional_mcts", config.py:9-12) ----
 * Replaces MCTS(policy = TraditionalPolicy(c_puct)) : core/lib/include/policies/Traditional.h:17-69 on top of
 * Heuristic (core/lib/include/algorithms/Heuristic.hpp:16-45, 94-200), RAVE::Select / BackPropogate<false>
 * (core/lib/include/algorithms/MonteCarlo.hpp:149-184), Default::Expand (:71-80), MCTS::playout (core/lib/src/MCTS.cpp:158-177).
 * One search per game and launch, every game with its own tree and its own persistent Evaluator (the policy object's
 * m_evaluator): gmk_trad_set_positions = MCTS(c_iterations, last_move, last_player) + Policy::prepare, gmk_trad_run =
 * that many MCTS::playout iterations (further calls continue the same tree), gmk_trad_root_stats = what
 * MCTS::stepForward / evalState read from the root. */
typedef struct gmk_trad gmk_trad;
int gmk_trad_create(int n_games, int node_capacity /* nodes per game, 256 .. 2^24-1 */, gmk_trad** out);
int gmk_trad_destroy(gmk_trad* t);
int gmk_trad_reset_evaluators(gmk_trad* t);                       /* Evaluator::reset for every game */
/* The game each slot of the handle is playing, relative to the first_game_id the noise / PoolRAVE entry points take (uint32[n], host;
 * default: the slot number).  A caller that hands the slot of a finished game to a new one sets the new game's number here, so that
 * the game's random streams (root noise, PoolRAVE rollouts) belong to the GAME, not to the slot it happens to run in. */
int gmk_trad_set_game_ids(gmk_trad* t, const uint32_t* h_ids);
int gmk_trad_set_positions(gmk_trad* t, const uint8_t* h_moves /* [n][225] */, const int32_t* h_lens /* [n]; < 0: this game keeps its position and tree */);
int gmk_trad_run(gmk_trad* t, int playouts, double c_puct, void* stream);
/* host outputs, any may be NULL: per-cell root child visits / values / priors [n][225], the move stepForward() would
 * play (-1 without children), root visits and value, nodes in the tree, status (bit 0 node capacity reached, bit 1
 * evaluator error, bit 2 unsupported board-only revert), evaluator updates (applied + reverted moves) so far */
int gmk_trad_root_stats(gmk_trad* t, uint32_t* h_visits, float* h_values, float* h_priors, int32_t* h_best,
                        uint32_t* h_root_visits, float* h_root_value, int32_t* h_n_nodes, int32_t* h_status,
                        uint64_t* h_evaluator_updates);
/* MCTS::stepForward() / stepForward(move) (core/lib/src/MCTS.cpp:129-147) for every game: h_moves int16[n] = the cell to step
 * to, or -1 for the most visited child (first in the current child order); h_moves == NULL = -1 for all.  The child's
 * subtree is kept (compacted into a second arena), a move without a child starts a new node; the move is appended to the
 * game's position and the next gmk_trad_run synchronises the evaluator (Policy::prepare).  Status bit 3 = not a legal move. */
int gmk_trad_step(gmk_trad* t, const int16_t* h_moves);
/* Default::AddNoise on every root with children (the reference does this at the start of every search, MCTS.cpp:182) */
int gmk_trad_add_root_noise(gmk_trad* t, float alpha, float epsilon, uint64_t seed, uint32_t first_game_id);
/* GMK_OPT_NOISE_SAMPLER / GMK_OPT_LOCKSTEP for a K6 / K8 handle (see gmk_mcts_set_option) */
int gmk_trad_set_option(gmk_trad* t, int option, int value);
/* gmk_mcts_reserve for a K6 / K8 handle (two_arenas = 0: nothing to do, gmk_trad_create allocates the one arena) */
int gmk_trad_reserve(gmk_trad* t, int two_arenas);
/* The self-play loop of the pattern-guided searchers, resident on the device (replaces the host loop of network/data_helper.py:56-83
 * around agents/mcts.py:17-21 for config.py:9-12's supervisor): n_total games (global ids first_game_id ..) are played through the
 * handle's n_games SLOTS with continuous batching -- every move = Default::AddNoise (noise_alpha > 0; MCTS.cpp:182) + one search of
 * `playouts` playouts (poolrave = 0: TraditionalPolicy as gmk_trad_run, 1: PoolRAVEPolicy as gmk_trad_run_poolrave) + a step kernel
 * that plays MCTS::stepForward()'s choice, checks the end of the game (Game.cpp:88-136) and hands a finished game's slot to the next
 * unstarted game (reuse_subtree = 1 keeps the chosen child's subtree as gmk_trad_step does, 0 starts every search from a new root as
 * gmk_trad_set_positions does).  Records by GAME, on the device: d_moves uint8[n_total][225], d_lens int32[n_total], d_winner
 * int8[n_total], d_visits uint16[n_total][225][225] (may be NULL).  h_open_moves / h_open_lens: the games' openings, or NULL.
 * persistent = 1 (TraditionalPolicy, whole games): ONE launch in which every slot's wavefront plays game after game at its own pace -- a
 * search no longer waits for the slowest one of the batch -- taking the next unstarted game from a counter when its game ends; a game then
 * starts on a fresh evaluator (Evaluator::reset), so its record does not depend on the slot it landed in and equals the one the
 * all-games-at-once loop plays.  With reuse_subtree the chosen child's subtree is compacted into the slot's second arena inside the launch
 * (MCTS::stepForward, MCTS.cpp:129-134); root noise inside the launch is drawn by the wavefront from the counter-based sampler
 * (gmk_trad_set_option(GMK_OPT_NOISE_SAMPLER, GMK_NOISE_SAMPLER_COUNTER); with the host-drawn std sampler the call is refused).
 * persistent = 0, or GMK_OPT_LOCKSTEP: the lock-step loop described above (a slot's evaluator carries over from game to game, as the
 * reference's policy object does within a worker).
 * max_steps > 0 ends the loop after that many moves per slot (games still running keep the moves they have, winner 0): what a
 * throughput measurement with every slot busy needs; 0 = play every game to its end.
 * *h_overflow != 0: some search stopped at its node capacity.  Afterwards the handle must be positioned again before other use. */
int gmk_trad_selfplay_run(gmk_trad* t, int poolrave, int n_total, uint32_t first_game_id, int playouts, double c_puct, uint64_t seed,
                          int reuse_subtree, float noise_alpha, float noise_epsilon,
                          const uint8_t* h_open_moves, int open_stride, const int32_t* h_open_lens,
                          uint8_t* d_moves, uint16_t* d_visits, int32_t* d_lens, int8_t* d_winner, int persistent, int max_steps, int32_t* h_overflow, int32_t* h_steps, void* stream);
/* the games' evaluator states, laid out as gmk_evalstate_read */
int gmk_trad_read_evaluators(gmk_trad* t, int32_t* h_scores, int32_t* h_density, uint32_t* h_pattern_dist,
                             uint32_t* h_compound_dist, int32_t* h_meta, uint8_t* h_record);

/* ---- K8: PoolRAVE tree search on the K6 handle ----
 * Replaces MCTS(policy = PoolRAVEPolicy(c_puct, c_bias)) : core/lib/include/policies/PoolRAVE.h:7-52 = RAVE::Select,
 * Default::Expand with AMAFNodes, one Default::RandomRollout per playout that stays on the board, and
 * RAVE::BackPropogate<true> (core/lib/include/algorithms/MonteCarlo.hpp:113-184).  The tree is the one of K6: create,
 * set_positions, step, add_root_noise and root_stats are the gmk_trad_* entry points above (the handle's evaluators are
 * not used); this call runs `playouts` MCTS::playout iterations per game with PoolRAVE's stages.  Rollout draws: Philox4x32-10,
 * key = seed, counter = (first_game_id + game, playout since the root last changed, stones on the root board << 8, ply >> 3),
 * as K3 with rollout number 0.  c_bias only reaches RAVE::MinMSE, which the reference leaves unused (:130-139): no argument.
 * A handle searches with ONE policy: mixing gmk_trad_run and gmk_trad_run_poolrave on it returns GMK_ERR_STATE. */
int gmk_trad_run_poolrave(gmk_trad* t, int playouts, double c_puct, uint64_t seed, uint32_t first_game_id, void* stream);
/* the root children's all-moves-as-first statistics by cell (AMAFNode::amaf_visits / amaf_value), host [n][225] each */
int gmk_trad_root_amaf(gmk_trad* t, uint32_t* h_amaf_visits, float* h_amaf_values);

/* ---- K7: network-guided tree search, many games in lock step (BASELINE.json configs[4]) ----
 * Replaces MCTS(policy = Policy(eval_state = network.eval_state, c_puct)) (agents/alphazero.py:5-9): Default::Select
 * (core/lib/include/algorithms/MonteCarlo.hpp:57-68), the evaluator call at a new leaf (core/lib/src/MCTS.cpp:164-168),
 * Default::Expand with extraCheck = true (:71-80), Default::BackPropogate (:90-95).  One playout of every game =
 * gmk_az_select (writes the leaves' Board.encoded_states planes, core/py_ext/src/game_ext.hpp:87-104, as float32
 * [n][6][15][15]; games that are over at the leaf are backed up at once and get a zero row), the caller's network on that
 * batch, gmk_az_expand with its value [n] and probabilities [n][225] (device pointers, same stream).
 * The batch holds the games that are still PLAYED, in slot order: n = gmk_az_live_games, which is n_games until gmk_az_advance ends a
 * game (or gmk_az_set_slots leaves slots idle); a finished game has no row, so the network's work follows the live games. */
typedef struct gmk_az gmk_az;
int gmk_az_create(int n_games, int node_capacity, double c_puct, gmk_az** out);
int gmk_az_destroy(gmk_az* a);
/* fresh roots; h_planes uint16[n][2][16] as gmk_eval_batch, h_last_moves int16[n][2] = {last move, the one before} or -1 */
int gmk_az_set_roots(gmk_az* a, const uint16_t* h_planes, const int16_t* h_last_moves);
int gmk_az_live_games(gmk_az* a, int32_t* n_live);            /* rows of the leaf batch; changed by gmk_az_set_roots / _set_slots / _advance only */
int gmk_az_select(gmk_az* a, float* d_states, void* stream);
int gmk_az_expand(gmk_az* a, const float* d_values, const float* d_probs, void* stream);
/* MCTS::stepForward() / stepForward(move) (core/lib/src/MCTS.cpp:129-147) for every game, subtree kept (as gmk_trad_step): h_moves
 * int16[n] = the cell to step to, -1 = the most visited child, NULL = -1 for all; status bit 2 = not a legal move. */
int gmk_az_step(gmk_az* a, const int16_t* h_moves);
/* One self-play move for every game still played, on the device (replaces the per-ply host work of the reference's self-play loop,
 * network/data_helper.py:56-83 with agents/alphazero.py:5-9 on both sides, as gmk_mcts_advance does for K3): the most visited child of
 * the root (first maximum in cell order, MCTS.cpp:129-134) is appended to the game's record with the root's visit counts, played on the
 * root position with Board::applyMove's victory check (Game.cpp:37-49, 88-136), and the tree is re-rooted (reuse_subtree: the child's
 * subtree is kept as gmk_az_step keeps it; otherwise a new root).  A root without a visited child ends its game where it stands.
 * Finished games (status bit 0) are skipped by gmk_az_select / gmk_az_expand from then on.
 * Records on the device, row g = game g: d_moves uint8[n][225] and d_lens int32[n] must hold the moves that led to the roots (the
 * openings), d_winner int8[n] zero; d_visits uint16[n][225][225] may be NULL.  h_unfinished: the games that go on.  Synchronises `stream`. */
/* Continuous batching for whole-game self-play (as gmk_selfplay_run / gmk_trad_selfplay_run have it): the handle's n_games slots play
 * n_total games between them.  This call takes gmk_az_set_roots's place: the first min(n_games, n_total) games start in the slots from
 * their openings (h_open_moves uint8[n_total][open_stride], h_open_lens int32[n_total], at most 8 moves each; NULL: empty boards);
 * gmk_az_advance then writes a slot's move to the record rows of the GAME it plays (rows = n_total) and hands a finished game's slot to
 * the next unstarted game; gmk_az_add_root_noise keys a slot's draws by that game.  gmk_az_set_roots ends the mode. */
int gmk_az_set_slots(gmk_az* a, int n_total, const uint8_t* h_open_moves, int open_stride, const int32_t* h_open_lens);
int gmk_az_advance(gmk_az* a, uint8_t* d_moves, uint16_t* d_visits, int32_t* d_lens, int8_t* d_winner, int reuse_subtree, int32_t* h_unfinished,
                   void* stream);
/* Default::AddNoise on every root with children (core/lib/include/algorithms/MonteCarlo.hpp:97-108); the stream of slot g is keyed
 * by first_game_id + ids[g], ids as set by gmk_az_set_game_ids (uint32[n], host; default: the slot number) */
int gmk_az_set_game_ids(gmk_az* a, const uint32_t* h_ids);
int gmk_az_add_root_noise(gmk_az* a, float alpha, float epsilon, uint64_t seed, uint32_t first_game_id);
/* GMK_OPT_NOISE_SAMPLER for a K7 handle (see gmk_mcts_set_option): GMK_NOISE_SAMPLER_COUNTER draws the noise on the device, one wavefront per game */
int gmk_az_set_option(gmk_az* a, int option, int value);
/* The same two steps for an evaluator that runs on the host and wants positions, not planes (the Python callable of
 * Policy(eval_state=...)): select, then the moves from the root to every pending leaf (h_paths int16[n][226], h_lens int32[n],
 * -1 = nothing to evaluate); expand from host memory.  Synchronous. */
int gmk_az_select_host(gmk_az* a, int16_t* h_paths, int32_t* h_lens);
int gmk_az_expand_host(gmk_az* a, const float* h_values, const float* h_probs);
/* Host-driven STAGES: `Policy(select=, expand=, eval_state=, back_prop=)` hands Python callables to the four stages of
 * MCTS::playout (core/py_ext/src/mcts_ext.hpp:43-61, core/lib/include/MCTS.h:74-101).  The callables run on the host; the tree stays on the
 * device, and the host reads what it is asked about and tells the device what was decided.  One game (`game`) of the handle; synchronous.
 *   gmk_az_read_node_host / _read_children_host   a node (visits, value, prior, cell, parent, child range) and the nodes of a child range
 *   gmk_az_set_leaf_host            the leaf a host-side descent ended at (node, the moves from the root): it becomes the pending leaf
 *   gmk_az_rollout_host             Default::Simulate's random rollout (MonteCarlo.hpp:37-47, 83-88) from the pending leaf, on the device;
 *                                   counters (global game id, playout number, root stones << 8) = the draws of gmk_mcts_*'s first rollout
 *   gmk_az_expand_stages_host       gmk_az_expand_host with Default::Expand and Default::BackPropogate switched separately
 *   gmk_az_write_stats_host         {visits, value} of nodes as a Python back_prop left them */
int gmk_az_read_node_host(gmk_az* a, int game, uint32_t node, uint32_t* h_visits, float* h_value, float* h_prior, int32_t* h_cell, uint32_t* h_parent,
                          uint32_t* h_first_child, int32_t* h_n_children);
int gmk_az_read_children_host(gmk_az* a, int game, uint32_t first_child, int n, int16_t* h_cells, uint32_t* h_visits, float* h_values, float* h_priors,
                              int32_t* h_n_children);
int gmk_az_set_leaf_host(gmk_az* a, int game, uint32_t leaf, const int16_t* h_path, int depth);
int gmk_az_rollout_host(gmk_az* a, int game, uint64_t seed, uint32_t counter0, uint32_t counter1, uint32_t counter2, int32_t* h_winner);
int gmk_az_expand_stages_host(gmk_az* a, const float* h_values, const float* h_probs, int do_expand, int do_backup);
int gmk_az_write_stats_host(gmk_az* a, int game, const uint32_t* h_nodes, const uint32_t* h_visits, const float* h_values, int n);
/* host outputs, any may be NULL; status bit 0 = the game is over (gmk_az_advance), bit 1 = node arena full (playouts of that game were dropped) */
int gmk_az_root_stats(gmk_az* a, uint32_t* h_visits, float* h_values, float* h_priors, uint32_t* h_root_visits,
                      float* h_root_value, int32_t* h_n_nodes, int32_t* h_status);

/* ---- K9: the convolutional trunk of the policy-value network (the evaluator K7 calls at every leaf) as one fused kernel ----
 * Replaces the convolution layers of PolicyValueNetwork (network/model_tf.py:28-66: conv3x3 6->32->64->128 with ReLU, the 1x1
 * policy head 128->4 and the 1x1 value head 128->2, both with ReLU) for a batch of positions, in float32 on the f32 matrix cores.
 * Weights are host arrays in PyTorch's conv layout [cout][cin][3][3] ([cout][cin] for the 1x1 heads), packed once at creation.
 * gmk_pvnet_forward: d_states float32 [n][6][225] (Board.encoded_states(), game_ext.hpp:87-104) ->
 *   d_pflat float32 [n][900] = relu(policy conv) flattened (pixel, channel), d_vflat float32 [n][450] likewise for the value head:
 * the inputs of the network's dense layers (tf.layers.flatten of the NHWC tensors).
 * gmk_pvnet_set_dense: the three dense layers behind them (network/model_tf.py:53-54 policy_logits / policy_output, :64-66 value_hidden /
 *   value_logits / value_output), host arrays in [out][in] order over the (pixel, channel) flattening: w_policy [225][900], b_policy [225],
 *   w_hidden [64][450], b_hidden [64], w_out [64], b_out; packed once, may be called again with new weights (a blocking copy: not while a
 *   gmk_pvnet_evaluate of this handle is in flight on another stream).
 * gmk_pvnet_evaluate: the whole PolicyValueNetwork.eval_state forward (network/model_tf.py:136-145) for a batch, two kernels on `stream`:
 *   d_states float32 [n][6][225] -> d_value float32 [n] = tanh(...), d_probs float32 [n][225] = softmax(...).  The head activations between
 *   the kernels live in the handle (grown on demand, which synchronises `stream`: let a batch size's first call happen outside a stream
 *   capture; one handle serves one stream at a time).  GMK_ERR_STATE before gmk_pvnet_set_dense. */
typedef struct gmk_pvnet gmk_pvnet;
int gmk_pvnet_create(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                     const float* w_policy, const float* b_policy, const float* w_value, const float* b_value, gmk_pvnet** out);
int gmk_pvnet_destroy(gmk_pvnet* net);
int gmk_pvnet_forward(gmk_pvnet* net, const float* d_states, int n, float* d_pflat, float* d_vflat, void* stream);
int gmk_pvnet_set_dense(gmk_pvnet* net, const float* w_policy, const float* b_policy, const float* w_hidden, const float* b_hidden,
                        const float* w_out, float b_out);
int gmk_pvnet_evaluate(gmk_pvnet* net, const float* d_states, int n, float* d_value, float* d_probs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GOMOKU_HIP_H_ */
