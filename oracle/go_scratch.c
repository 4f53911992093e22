/*
 * go_scratch.c -- from-scratch ("position-pure") formulation of the Evaluator outputs.
 * TEST INFRASTRUCTURE, see gomoku_oracle.h.
 *
 * The reference maintains scores / density / totals incrementally, move by move
 * (core/lib/src/Pattern.cpp:274-302).  The GPU kernel K1 gets only the stones, so it needs a
 * formulation of the same outputs as a function of the position.  This file
 * states that formulation on the CPU with the oracle's own matcher, so that
 *   tests/test_formulation.py: go_scratch_eval(position) == replay of the moves through go_eval_apply
 * can run on 10^5 boards without a GPU.  It is a model of the kernel's ALGORITHM, not of its code.
 */
#include "gomoku_oracle.h"
#include <string.h>

static const int k_stride[4] = { 1, GO_W, GO_W + 1, GO_W - 1 };
static const int k_dx[4] = { 1, 0, 1, -1 };
static const int k_dy[4] = { 0, 1, 1, 1 };
static const int k_w[7][7] = {                   /* Pattern.cpp:601-607 */
    { 2, 0, 0, 1, 0, 0, 2 }, { 0, 4, 3, 3, 3, 4, 0 }, { 0, 3, 5, 4, 5, 3, 0 }, { 1, 3, 4, 0, 4, 3, 1 },
    { 0, 3, 5, 4, 5, 3, 0 }, { 0, 4, 3, 3, 3, 4, 0 }, { 2, 0, 0, 1, 0, 0, 2 },
};

static int grp(int favour, int perspective) { return ((favour == GO_BLACK) << 1) | (perspective == GO_BLACK); }

/* symbol of cell (x,y): 1 black, 2 white, 4 blank, 3 off-board */
static int sym(const int8_t *cell, int x, int y) {
    if (x < 0 || x >= GO_W || y < 0 || y >= GO_H) return 3;
    return cell[y * GO_W + x] == GO_BLACK ? 1 : cell[y * GO_W + x] == GO_WHITE ? 2 : 4;
}

/* lead / trail: number of '?' symbols put before / after the cells of every line.  The reference uses 6/6
   (Mapping.cpp:61-77); the kernel uses 1/2, which this model lets the tests prove equivalent. */
int go_scratch_eval(const int8_t *cell /*[225] -1/0/+1*/, int lead, int trail,
                    int32_t *scores /*[4][225]*/, int32_t *density /*[2][2][225]*/,
                    uint32_t *totals /*[11]*/, int32_t *status) {
    const go_ac *ac = go_default_ac();
    static int cnt[GO_N][3][2][4];                 /* '_' pieces of L3 / D3 / L2 per cell, colour, direction */
    int winner = GO_NONE, err = 0, n_black = 0, n_white = 0;
    memset(scores, 0, sizeof(int32_t) * 4 * GO_N);
    memset(density, 0, sizeof(int32_t) * 4 * GO_N);
    memset(totals, 0, sizeof(uint32_t) * 11);
    memset(cnt, 0, sizeof cnt);
    for (int i = 0; i < GO_N; ++i) { n_black += cell[i] == GO_BLACK; n_white += cell[i] == GO_WHITE; }

    /* 1+2: scan every line, deposit every non-Five match (A.8 steps 1-2; Pattern.cpp:138-165) */
    for (int dir = 0; dir < 4; ++dir) {
        for (int line = 0; line < (dir < 2 ? 15 : 29); ++line) {
            int x0, y0, len;
            if (dir == GO_DIR_H) { x0 = 0; y0 = line; len = 15; }
            else if (dir == GO_DIR_V) { x0 = line; y0 = 0; len = 15; }
            else if (dir == GO_DIR_LD) { int d = line - 14; x0 = d > 0 ? d : 0; y0 = d > 0 ? 0 : -d; len = 15 - (d > 0 ? d : -d); }
            else { int k = line; x0 = k < 14 ? k : 14; y0 = k - x0; len = (k < 14 ? k : 28 - k) + 1; }
            uint8_t codes[32];
            int n = 0;
            for (int i = 0; i < lead; ++i) codes[n++] = 3;
            for (int i = 0; i < len; ++i) codes[n++] = (uint8_t)sym(cell, x0 + i * k_dx[dir], y0 + i * k_dy[dir]);
            for (int i = 0; i < trail; ++i) codes[n++] = 3;
            int32_t pat[64], off[64];
            int m = go_ac_match(ac, codes, n, pat, off, 64);
            if (m > 64) { err = 1; m = 64; }
            for (int k = 0; k < m; ++k) {
                const go_pattern *p = &ac->patterns[pat[k]];
                if (p->type == GO_FIVE) { winner = p->favour; continue; }
                int end = (y0 * GO_W + x0) + (off[k] - lead) * k_stride[dir];
                totals[p->type] += (p->favour == GO_BLACK) ? 0x10000u : 1u;
                int s = (dir >= 2) ? (int)(1.2 * p->score) : p->score;
                for (int j = 0; j < p->len; ++j) {
                    char piece = p->str[p->len - 1 - j];
                    int c = end - j * k_stride[dir];
                    if (piece == '_') {
                        scores[grp(p->favour, p->favour) * GO_N + c] += s;
                        if (p->type == GO_LIVE3 || p->type == GO_DEAD3 || p->type == GO_LIVE2)
                            cnt[c][p->type == GO_LIVE3 ? 0 : p->type == GO_DEAD3 ? 1 : 2][p->favour == GO_BLACK][dir]++;
                    }
                    if (piece == '_' || piece == '^') scores[grp(p->favour, -p->favour) * GO_N + c] += s;
                }
            }
        }
    }

    /* 3: density (A.8 step 3; Pattern.cpp:236-272) */
    for (int q = 0; q < GO_N; ++q) {
        int qx = q % GO_W, qy = q / GO_W;
        for (int c = 0; c < 2; ++c) {
            int colour = c ? GO_BLACK : GO_WHITE, count = 0, weight = 0;
            for (int dy = -3; dy <= 3; ++dy) for (int dx = -3; dx <= 3; ++dx) {
                int x = qx + dx, y = qy + dy;
                if (x < 0 || x >= GO_W || y < 0 || y >= GO_H || cell[y * GO_W + x] != colour) continue;
                weight += k_w[dy + 3][dx + 3];
                count += k_w[dy + 3][dx + 3] > 0;
            }
            if (cell[q] != GO_NONE) { count = -count - 1; weight = -weight - 1; }
            else if (weight > 0) scores[grp(colour, colour) * GO_N + q] += 160;
            density[(c * 2 + 0) * GO_N + q] = count;
            density[(c * 2 + 1) * GO_N + q] = weight;
        }
    }

    /* 4: compounds (A.8 step 4; Pattern.cpp:167-197, 420-550) */
    for (int q = 0; q < GO_N; ++q) {
        if (cell[q] != GO_NONE) continue;
        for (int c = 0; c < 2; ++c) {
            int colour = c ? GO_BLACK : GO_WHITE;
            int k[3][4], bits = 0;
            for (int t = 0; t < 3; ++t) for (int d = 0; d < 4; ++d) {
                k[t][d] = cnt[q][t][c][d] > 2 ? 2 : cnt[q][t][c][d];
                bits |= (k[t][d] == 1 ? 1 : k[t][d] == 2 ? 3 : 0) << (2 * d);
            }
            if (!(bits & (bits - 1))) continue;
            if (density[(c * 2 + 0) * GO_N + q] < 2) continue;
            enum { S0, L2, LD3, To33, To43, To44 };
            int state = S0, l3 = 0, triple = 0, ncomp = 0, cdir[8], ctype[8];
            for (int d = 0; d < 4; ++d) {
                int t = k[0][d] ? 0 : k[1][d] ? 1 : k[2][d] ? 2 : -1;
                if (t < 0) continue;
                int cond = t == 2 ? L2 : LD3;
                if (t == 0) l3++;
                for (int r = 0; r < k[t][d]; ++r) {
                    cdir[ncomp] = d; ctype[ncomp] = t == 0 ? GO_LIVE3 : t == 1 ? GO_DEAD3 : GO_LIVE2; ncomp++;
                    if (state == S0) state += cond;
                    else if (state == L2 || state == LD3) state += cond + 1;
                    else { triple = 1; state += cond + (state == To44 ? -cond : -1); }
                }
            }
            int type = state - To33;
            if (type < 0 || type > 2) { err = 1; continue; }      /* reference: out-of-bounds read */
            totals[8 + type] += c ? 0x10000u : 1u;
            for (int i = 0; i < ncomp; ++i) {
                scores[grp(colour, colour) * GO_N + q] += 600;
                scores[grp(colour, -colour) * GO_N + q] += 600;
                if (triple || l3) continue;
                /* first match, in stream order over the 13-symbol window centred on q, of the component's
                   type that covers q with '_' on it: +600 on its other '_' / '^' cells (opponent's view) */
                int d = cdir[i], qx = q % GO_W, qy = q / GO_W;
                uint8_t win[13];
                for (int w = 0; w < 13; ++w) win[w] = (uint8_t)sym(cell, qx + (w - 6) * k_dx[d], qy + (w - 6) * k_dy[d]);
                int32_t pat[64], off[64];
                int m = go_ac_match(ac, win, 13, pat, off, 64);
                for (int e = 0; e < m && e < 64; ++e) {
                    const go_pattern *p = &ac->patterns[pat[e]];
                    if (p->type != ctype[i]) continue;
                    if (!((unsigned)off[e] - 6u < (unsigned)p->len)) continue;
                    if (p->str[p->len - 1 - (off[e] - 6)] != '_') continue;
                    int cur = q + (off[e] - 6) * k_stride[d];
                    for (int j = 0; j < p->len; ++j, cur -= k_stride[d]) {
                        char piece = p->str[p->len - 1 - j];
                        if ((piece == '_' || piece == '^') && cur != q) scores[grp(colour, -colour) * GO_N + cur] += 600;
                    }
                    break;
                }
            }
        }
    }
    int end = winner != GO_NONE || n_black + n_white == GO_N;
    int cur = end ? GO_NONE : (n_black == n_white ? GO_BLACK : GO_WHITE);
    if (status) *status = (end ? 1 : 0) | (err ? 2 : 0) | ((int)(uint8_t)(int8_t)winner << 8) | ((int)(uint8_t)(int8_t)cur << 16);
    return err;
}

void go_scratch_eval_batch(const uint8_t *moves, const int32_t *lens, int stride, int n, int lead, int trail,
                           int32_t *scores, int32_t *density, uint32_t *totals, int32_t *status) {
    for (int b = 0; b < n; ++b) {
        int8_t cell[GO_N];
        memset(cell, 0, sizeof cell);
        for (int i = 0; i < lens[b]; ++i) cell[moves[(size_t)b * stride + i]] = (i & 1) ? GO_WHITE : GO_BLACK;
        go_scratch_eval(cell, lead, trail, scores + (size_t)b * 4 * GO_N, density + (size_t)b * 4 * GO_N,
                        totals + (size_t)b * 11, status + b);
    }
}
