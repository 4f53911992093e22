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

#ifdef __cplusplus
}
#endif
#endif /* GOMOKU_HIP_H_ */
