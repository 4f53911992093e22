/*
 * gomoku_oracle.h -- CPU restatement of the GomokuAI self-play hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gomokuai_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker.
 *
 * Every function restates, line by line, the reference algorithm it cites
 * (paths relative to /root/reference).  The reference itself cannot be built in
 * this image (every core header includes <Eigen/Dense>, which is absent, and no
 * stand-in headers may be written), so parity is pinned by the reference's own
 * test vectors (tests/test_oracle_golden.py).
 *
 * Plain C99 (plus go_stdsort.cpp, which only calls the toolchain's std::sort; see there).
 * Build: make -C oracle
 */
#ifndef GOMOKU_ORACLE_H_
#define GOMOKU_ORACLE_H_
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Game.h:12-15, Pattern.h:11-15 ---- */
enum { GO_W = 15, GO_H = 15, GO_N = 225, GO_RENJU = 5 };
enum { GO_MAX_PAT_LEN = 7, GO_BLOCK = 7, GO_TARGET_LEN = 13 };
enum { GO_NLINES = 88, GO_LINE_CAP = 28 };
/* Player (Game.h:19-21) */
enum { GO_WHITE = -1, GO_NONE = 0, GO_BLACK = 1 };
/* Direction (Mapping.h:9-11) */
enum { GO_DIR_H = 0, GO_DIR_V = 1, GO_DIR_LD = 2, GO_DIR_RD = 3 };
/* Pattern::Type (Pattern.h:33-39) */
enum { GO_DEAD1, GO_LIVE1, GO_DEAD2, GO_LIVE2, GO_DEAD3, GO_LIVE3, GO_DEAD4, GO_LIVE4, GO_FIVE, GO_PT_SIZE };
/* Compound::Type (Pattern.h:119) */
enum { GO_DOUBLE3, GO_FOUR3, GO_DOUBLE4, GO_CT_SIZE };

/* ---------------- Board (Game.h:58-151, Game.cpp:37-146) ---------------- */
typedef struct {
    int8_t  cur_player;           /* m_curPlayer */
    int8_t  winner;               /* m_winner */
    uint8_t states[3][GO_N];      /* m_moveStates[player+1] */
    int32_t counts[3];            /* m_moveCounts[player+1] */
    int16_t record[GO_N];         /* m_moveRecord */
    int32_t nrec;
} go_board;

void go_board_reset(go_board *b);
int  go_board_apply(go_board *b, int move, int check_victory);  /* returns next player */
int  go_board_revert(go_board *b, int count);
int  go_board_check_move(const go_board *b, int move);
int  go_board_check_end(go_board *b);
/* Game.cpp:64-73 with the mt19937 draw replaced by the caller's r in [0,224];
   returns -1 where the reference throws overflow_error (full board). */
int  go_board_random_move(const go_board *b, unsigned r);
/* game_ext.hpp:87-104 : uint8[6][15][15] feature planes */
void go_board_encoded_states(const go_board *b, uint8_t *out);
/* test aid: n recorded games replayed through go_board_apply (see go_board.c) */
void go_board_replay_games(const uint8_t *moves, const int32_t *lens, int stride, int n, int8_t *legal, int32_t *end_ply, int8_t *winner);

/* ---------------- Patterns + AC automaton ---------------- */
typedef struct {
    char   str[GO_MAX_PAT_LEN + 1];
    int8_t len;
    int8_t favour;
    int8_t type;
    int32_t score;
} go_pattern;

enum { GO_MAX_PATTERNS = 512, GO_MAX_DAT = 8192 };

typedef struct {
    int        n_patterns;
    go_pattern patterns[GO_MAX_PATTERNS];
    int        size;                       /* m_base.size() */
    int32_t    base[GO_MAX_DAT], check[GO_MAX_DAT], fail[GO_MAX_DAT];
    int32_t    invariants[5];
    int        sort_ties;                  /* #equal sort keys met (0 => std::sort order is pinned) */
} go_ac;

int  go_encode_char(char ch);                              /* Mapping.h:40-48 */
/* protos: strings like "-~_ooo_~" ('+' black / '-' white first char). */
int  go_ac_build(go_ac *ac, const char *const *protos, const int *types, const int *scores, int n);
int  go_ac_build_default(go_ac *ac);                       /* Pattern.cpp:554-596 */
const go_ac *go_default_ac(void);
/* stages exposed for the reference's white-box tests (patternsearch_unittest.cpp) */
int  go_ac_augment(go_pattern *pats, int n, int stage);    /* stage 1=reverse 2=flip 3=boundary; returns new n */
/* match stream (Pattern.cpp:33-62): returns number of matches */
int  go_ac_match(const go_ac *ac, const uint8_t *codes, int n, int32_t *pat_idx, int32_t *offsets, int cap);
int  go_ac_used_slots(const go_ac *ac);
void go_ac_trie_dump_begin(int32_t *out, int cap);   /* test hook, see go_ac.c */
int  go_ac_trie_dump_count(void);

/* ---------------- Evaluator (Pattern.h:142-220, Pattern.cpp:76-550) ---------------- */
typedef struct go_evaluator go_evaluator;
go_evaluator *go_eval_new(void);
void go_eval_free(go_evaluator *ev);
void go_eval_reset(go_evaluator *ev);
/* returns next player; *err (optional) gets 1 if the reference's self-check (Pattern.cpp:314-333)
   would throw, 2 if Compound::locate would index out of bounds (Pattern.cpp:484-485). */
int  go_eval_apply(go_evaluator *ev, int move, int *err);
int  go_eval_revert(go_evaluator *ev, int count);
int  go_eval_check_end(go_evaluator *ev);
const go_board *go_eval_board(const go_evaluator *ev);
void go_eval_get_scores(const go_evaluator *ev, int32_t *out /*[4][225]*/);
void go_eval_get_density(const go_evaluator *ev, int32_t *out /*[2][2][225]*/);
void go_eval_get_pattern_dist(const go_evaluator *ev, uint32_t *out /*[226][8]*/);
void go_eval_get_compound_dist(const go_evaluator *ev, uint32_t *out /*[226][3]*/);
void go_eval_line_view(const go_evaluator *ev, int pos, int dir, uint8_t *out13);

/* Batch helper: replays each move list through a fresh Evaluator and emits the
   position outputs in the layout of include/gomoku_hip.h gmk_eval_batch.
   moves: u8[n][stride], lens: i32[n].
   totals u32[n][11] = 8 pattern types then 3 compound types, White low16 | Black high16.
   status i32[n]: bit0 end, bits 8..15 winner (signed), bits 16..23 cur_player (signed), bit1 error. */
void go_eval_replay_batch(const uint8_t *moves, const int32_t *lens, int stride, int n,
                          int32_t *scores, int32_t *density, uint32_t *totals, int32_t *status);

/* From-scratch formulation of the same outputs (go_scratch.c): model of the GPU kernel's algorithm.
   cell: int8[225] (-1 white, 0 empty, +1 black); lead/trail = '?' symbols around each line (reference 6/6). */
int  go_scratch_eval(const int8_t *cell, int lead, int trail, int32_t *scores, int32_t *density,
                     uint32_t *totals, int32_t *status);
void go_scratch_eval_batch(const uint8_t *moves, const int32_t *lens, int stride, int n, int lead, int trail,
                           int32_t *scores, int32_t *density, uint32_t *totals, int32_t *status);

/* ---------------- Philox4x32-10 (counter-based RNG shared with the GPU path) ---------------- */
void go_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* ---------------- MCTS (MCTS.h/.cpp, MonteCarlo.hpp:13-110, Random.h:22-35) ---------------- */
typedef struct go_mcts go_mcts;
/* Policy(eval_state = f) (agents/alphazero.py:5-9): f(encoded states u8[6][15][15]) -> value, probs f32[225] */
typedef void (*go_eval_state_fn)(const uint8_t *states, float *value, float *probs, void *user);
go_mcts *go_mcts_new(uint64_t c_iterations, double c_puct, int c_rollouts,
                     uint64_t seed, uint32_t game_id);
void go_mcts_free(go_mcts *m);
void go_mcts_reset(go_mcts *m);
void go_mcts_sync_with_board(go_mcts *m, const go_board *b);
void go_mcts_run_playouts(go_mcts *m, go_board *b);
int  go_mcts_get_action(go_mcts *m, go_board *b);
/* evalState: returns root value; probs f32[225] (pi), visits u32[225] raw child visits */
float go_mcts_eval_state(go_mcts *m, go_board *b, float *probs, uint32_t *visits);
int  go_mcts_step_forward(go_mcts *m);            /* returns new root position */
int  go_mcts_step_forward_move(go_mcts *m, int move);
uint64_t go_mcts_size(const go_mcts *m);
int  go_mcts_root_position(const go_mcts *m);
int  go_mcts_root_player(const go_mcts *m);
uint64_t go_mcts_root_visits(const go_mcts *m);
float go_mcts_root_value(const go_mcts *m);
void go_mcts_root_children(const go_mcts *m, uint32_t *visits /*[225]*/, float *values /*[225]*/, float *priors /*[225]*/);
/* instrumentation for the roofline: algorithmic bytes touched by tree ops since reset */
uint64_t go_mcts_alg_bytes(const go_mcts *m);
/* Default::AddNoise at the start of every search (MCTS.cpp:182); alpha = 0 (default) disables it. */
void go_mcts_set_noise(go_mcts *m, float alpha, float epsilon);
/* which stream AddNoise draws from: 0 (default) std::gamma_distribution<float> over std::mt19937, seeded by Philox(game, stones, 'nois');
   1 the counter-based sampler of include/gomoku_noise.h (what the device-resident self-play loops draw from, inside their kernels) */
void go_mcts_set_noise_sampler(go_mcts *m, int sampler);
float go_noise_gamma(float alpha, uint32_t game_id, uint32_t stones, uint32_t cell, uint64_t seed);
void go_noise_mix225(float *p, float alpha, float epsilon, uint32_t game_id, uint32_t stones, uint64_t seed);
double go_noise_log(double x);
double go_noise_exp(double x);
/* the search then calls fn at every new leaf instead of rolling out, and expands with Default::Expand(extraCheck = true) */
void go_mcts_set_evaluator(go_mcts *m, go_eval_state_fn fn, void *user);
/* KAT hook: draw rollout moves sequentially from std::mt19937(seed) (id = eng() % 225) instead of Philox,
   to replay the search recorded in SURVEY.md Appendix B.2. */
void go_mcts_use_mt19937(go_mcts *m, uint32_t seed);
/* MT19937 itself (Matsumoto & Nishimura 1998), for the replay-hash KAT */
typedef struct { uint32_t mt[624]; int idx; } go_mt19937;
void go_mt_seed(go_mt19937 *g, uint32_t seed);
uint32_t go_mt_next(go_mt19937 *g);

/* Stats::TempBasedProbs on normalized visits (MCTS.cpp:104-117, Statistical.hpp:37-42) */
void go_visits_to_pi(const uint32_t *visits, int n_moves_on_board, float *pi);

/* ---------------- Pattern-guided search (Traditional.h:17-69, Heuristic.hpp, MonteCarlo.hpp:149-184; go_trad.c) ---------------- */
typedef struct go_trad go_trad;
go_trad *go_trad_new(double c_puct);
void go_trad_free(go_trad *t);
/* fresh root at the position after `moves`, `playouts` iterations; the policy's evaluator persists across calls */
void go_trad_search(go_trad *t, const uint8_t *moves, int n_moves, uint64_t playouts);
/* the same inside a persistent MCTS object: the tree is kept from call to call (syncWithBoard / stepForward(move)), noise is
   mixed into the root priors before every search when alpha > 0 (Default::AddNoise), go_trad_step_forward plays the best child */
void go_trad_set_noise(go_trad *t, float alpha, float epsilon, uint64_t seed, uint32_t game_id);
void go_trad_set_noise_sampler(go_trad *t, int sampler);
void go_trad_run(go_trad *t, const uint8_t *moves, int n_moves, uint64_t playouts);
int  go_trad_step_forward(go_trad *t);
/* per-cell root child statistics; returns the move MCTS::stepForward() would play (-1 without children) */
int  go_trad_root_children(const go_trad *t, uint32_t *visits, float *values, float *priors);
uint64_t go_trad_root_visits(const go_trad *t);
float    go_trad_root_value(const go_trad *t);
int      go_trad_n_nodes(const go_trad *t);
uint64_t go_trad_evaluator_updates(const go_trad *t);
const go_evaluator *go_trad_evaluator(const go_trad *t);
/* EvaluationProbs + DecisiveFilter + EvaluationValue on a fresh in-order replay; returns the value */
float go_trad_heuristic(const uint8_t *moves, int n_moves, float *probs /*[225]*/);

/* ---------------- PoolRAVE search (PoolRAVE.h:7-52, MonteCarlo.hpp:113-184; go_rave.c) ---------------- */
typedef struct go_rave go_rave;
go_rave *go_rave_new(double c_puct, double c_bias, uint64_t seed, uint32_t game_id);
void go_rave_free(go_rave *t);
void go_rave_set_noise(go_rave *t, float alpha, float epsilon);
/* MCTS::runPlayouts on the kept tree from the position reached by `moves` */
void go_rave_run(go_rave *t, const uint8_t *moves, int n_moves, uint64_t playouts);
int  go_rave_step_forward(go_rave *t);
int  go_rave_root_children(const go_rave *t, uint32_t *visits, float *values, float *priors, uint32_t *amaf_visits, float *amaf_values);
uint64_t go_rave_root_visits(const go_rave *t);
float    go_rave_root_value(const go_rave *t);
uint64_t go_rave_size(const go_rave *t);

#ifdef __cplusplus
}
#endif
#endif
