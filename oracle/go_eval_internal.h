/*
 * go_eval_internal.h -- the Evaluator's state as go_eval.c keeps it, shared with go_trad.c (which reads scores,
 * density and the two distributions the way Heuristic.hpp does).  TEST INFRASTRUCTURE, see gomoku_oracle.h.
 */
#ifndef GO_EVAL_INTERNAL_H_
#define GO_EVAL_INTERNAL_H_
#include "gomoku_oracle.h"

/* generator internals shared with go_ac.c */
typedef struct { const uint8_t *t; int n; int pos; int offset; int state; } go_gen;
void go__gen_init(go_gen *g, const uint8_t *t, int n);
int  go__gen_next(const go_ac *ac, go_gen *g);
int  go__gen_pattern(const go_ac *ac, const go_gen *g);

enum { MAX_RESULTS = 32, MAX_COMPOUNDS = 64, MAX_COMPONENTS = 8 };

typedef struct { int pat; int offset; } go_entry;

typedef struct {
    int position, favour;
    int ncomp;
    struct { int dir, type; } comps[MAX_COMPONENTS];
    int type;
    go_gen generator;
    int gen_dir;
    int count, l3_count, triple_cross;
} go_compound;

struct go_evaluator {
    const go_ac *ac;
    go_board board;                               /* BoardMap::m_board */
    uint8_t  lines[GO_NLINES][GO_LINE_CAP];       /* BoardMap::m_lineMap */
    uint32_t pattern_dist[GO_N + 1][GO_PT_SIZE - 1];
    uint32_t compound_dist[GO_N + 1][GO_CT_SIZE];
    int32_t  density[2][2][GO_N];                 /* [White,Black][count,weight] */
    int32_t  scores[4][GO_N];
    /* Updater */
    int delta, move, player;
    go_entry results[2][4][MAX_RESULTS];
    int nresults[2][4];
    int ncompounds;
    struct { int pos, player; } compound_keys[MAX_COMPOUNDS];
    go_compound compounds[MAX_COMPOUNDS];
    int err;
};

#endif
