/*
 * go_eval.c -- restatement of BoardMap + Evaluator (incremental pattern / compound / density
 * evaluator).  TEST INFRASTRUCTURE, see gomoku_oracle.h.
 * Follows core/lib/src/Mapping.cpp:11-77 and core/lib/src/Pattern.cpp:76-550, 598-611.
 */
#include "go_eval_internal.h"
#include <string.h>
#include <stdlib.h>

/* Mapping.h:14-27 : id stride of one step along a direction */
static const int k_stride[4] = { 1, GO_W, GO_W + 1, GO_W - 1 };

/* Pattern.h:153-161 */
static int group1(int player) { return player == GO_BLACK; }
static int group2(int favour, int perspective) { return ((favour == GO_BLACK) << 1) | (perspective == GO_BLACK); }

/* Mapping.cpp:11-25 */
static void parse_index(int pos, int dir, int *index, int *offset) {
    int x = pos % GO_W, y = pos / GO_W, off = GO_MAX_PAT_LEN - 1;
    switch (dir) {
        case GO_DIR_H:  *index = y; *offset = off + x; break;
        case GO_DIR_V:  *index = GO_H + x; *offset = off + y; break;
        case GO_DIR_LD: *index = GO_W + 2 * GO_H - 1 + x - y; *offset = off + (x < y ? x : y); break;
        default:        *index = 2 * (GO_W + GO_H) - 1 + x + y; *offset = off + ((GO_W - 1 - x) < y ? (GO_W - 1 - x) : y); break;
    }
}

/* Mapping.cpp:31-34 */
static const uint8_t *line_view(const go_evaluator *ev, int pos, int dir) {
    int index, offset;
    parse_index(pos, dir, &index, &offset);
    return &ev->lines[index][offset - GO_TARGET_LEN / 2];
}

void go_eval_line_view(const go_evaluator *ev, int pos, int dir, uint8_t *out13) {
    memcpy(out13, line_view(ev, pos, dir), GO_TARGET_LEN);
}

/* Mapping.cpp:37-45 */
static int boardmap_apply(go_evaluator *ev, int move) {
    for (int dir = 0; dir < 4; ++dir) {
        int index, offset;
        parse_index(move, dir, &index, &offset);
        ev->lines[index][offset] = (uint8_t)go_encode_char(ev->board.cur_player == GO_BLACK ? 'x' : 'o');
    }
    return go_board_apply(&ev->board, move, 0);
}

/* Mapping.cpp:47-59 */
static int boardmap_revert(go_evaluator *ev, int count) {
    for (int i = 0; i < count; ++i) {
        int move = ev->board.record[ev->board.nrec - 1];
        for (int dir = 0; dir < 4; ++dir) {
            int index, offset;
            parse_index(move, dir, &index, &offset);
            ev->lines[index][offset] = (uint8_t)go_encode_char('-');
        }
        go_board_revert(&ev->board, 1);
    }
    return ev->board.cur_player;
}

/* Mapping.cpp:61-77 */
static void boardmap_reset(go_evaluator *ev) {
    int len[GO_NLINES];
    go_board_reset(&ev->board);
    memset(ev->lines, 0, sizeof ev->lines);
    for (int l = 0; l < GO_NLINES; ++l) { memset(ev->lines[l], go_encode_char('?'), GO_MAX_PAT_LEN - 1); len[l] = GO_MAX_PAT_LEN - 1; }
    for (int i = 0; i < GO_N; ++i)
        for (int dir = 0; dir < 4; ++dir) {
            int index, offset;
            parse_index(i, dir, &index, &offset);
            ev->lines[index][len[index]++] = (uint8_t)go_encode_char('-');
        }
    for (int l = 0; l < GO_NLINES; ++l) { memset(ev->lines[l] + len[l], go_encode_char('?'), GO_MAX_PAT_LEN - 1); }
}

/* ---- Evaluator::Record (Pattern.cpp:390-416) ---- */
static void rec_set_total(uint32_t *field, int delta, int player) {
    unsigned offset = 4 * (unsigned)sizeof(*field) * (unsigned)group1(player);
    *field += (uint32_t)delta << offset;
}
static void rec_set_flag(uint32_t *field, int delta, int favour, int perspective, int dir) {
    unsigned group = (unsigned)group2(favour, perspective), offset = (4 * group + (unsigned)dir) * 2;
    uint32_t lower = 1u << offset, higher = lower << 1, mask = higher | lower;
    uint32_t value = (delta == 1 ? (*field << 1) | lower : (*field >> 1) & ~higher);
    *field = (*field & ~mask) | (value & mask);
}
static unsigned rec_get_flag(uint32_t field, int favour, int perspective, int dir) {
    unsigned offset = (4 * (unsigned)group2(favour, perspective) + (unsigned)dir) * 2;
    return (field >> offset) & 3u;
}
static unsigned rec_get_group(uint32_t field, int favour, int perspective) {
    return (field >> (8 * (unsigned)group2(favour, perspective))) & 0xffu;
}

/* Pattern.cpp:22-25 (unsigned wrap of offset - pose) */
static int has_covered(const go_pattern *p, int offset, unsigned pose) {
    return (unsigned)offset - pose < (unsigned)p->len;
}

static int popcount8(unsigned v) { int c = 0; for (v &= 0xff; v; v &= v - 1) ++c; return c; }

/* ---- Compound (Pattern.cpp:420-550) ---- */
static const int k_comp_types[3] = { GO_LIVE3, GO_DEAD3, GO_LIVE2 };
static const int k_base_score = 600;                      /* Pattern.cpp:611 */

/* Pattern.cpp:424-433 */
static int compound_test(const go_evaluator *ev, int pose, int player) {
    unsigned bits = 0;
    for (int k = 0; k < 3; ++k) bits |= rec_get_group(ev->pattern_dist[pose][k_comp_types[k]], player, player);
    return (bits & (bits - 1)) != 0;
}

/* Pattern.cpp:440-486 */
static void compound_locate(go_evaluator *ev, go_compound *c) {
    enum { S0, L2, LD3, To33, To43, To44 };
    int state = S0;
    for (int dir = 0; dir < 4; ++dir) {
        int cond = S0, count = 0;
        for (int k = 0; k < 3; ++k) {
            int type = k_comp_types[k];
            switch (rec_get_flag(ev->pattern_dist[c->position][type], c->favour, c->favour, dir)) {
                case 0: count = 0; break;
                case 1: count = 1; break;
                case 3: count = 2; break;
                default: break;                  /* 0b10 is not handled by the reference: count keeps its value */
            }
            if (count != 0) {
                if (type == GO_LIVE3) { c->l3_count += 1; cond = LD3; }
                else if (type == GO_DEAD3) cond = LD3;
                else cond = L2;
                for (int r = 0; r < count && c->ncomp < MAX_COMPONENTS; ++r) { c->comps[c->ncomp].dir = dir; c->comps[c->ncomp].type = type; c->ncomp++; }
                break;
            }
        }
        if (cond == S0) continue;
        for (int i = 0; i < count; ++i) {
            int offset;
            switch (state) {
                case S0: offset = 0; break;
                case L2: case LD3: offset = 1; break;
                default: c->triple_cross = 1; offset = (state == To44 ? -cond : -1); break;
            }
            state = state + cond + offset;
        }
    }
    c->type = state - To33;
    if (c->type < 0 || c->type >= GO_CT_SIZE) {  /* reference would index out of bounds here (UB) */
        ev->err |= 2;
        c->type = 0; c->ncomp = 0; c->count = 0;
        return;
    }
    c->count = popcount8(rec_get_group(ev->compound_dist[c->position][c->type], c->favour, c->favour));
}

/* Pattern.cpp:545-550 */
static void compound_update_pose(go_evaluator *ev, go_compound *c, int delta, int pose, int comp_dir, int perspective) {
    rec_set_flag(&ev->compound_dist[pose][c->type], delta, c->favour, perspective, comp_dir);
    ev->scores[group2(c->favour, perspective)][pose] += delta * k_base_score;
}

/* Pattern.cpp:520-543 */
static void compound_update_antis(go_evaluator *ev, go_compound *c, int delta, int comp_dir, int comp_type) {
    if (c->gen_dir != comp_dir) {
        go__gen_init(&c->generator, line_view(ev, c->position, comp_dir), GO_TARGET_LEN);
        c->gen_dir = comp_dir;
    }
    /* range-for: begin() advances the member in place only while its state is 0, then iterates a copy */
    if (c->generator.state == 0) go__gen_next(ev->ac, &c->generator);
    go_gen it = c->generator;
    while (!(it.pos >= it.n && it.state == 0)) {
        const go_pattern *p = &ev->ac->patterns[go__gen_pattern(ev->ac, &it)];
        int offset = it.offset;
        if (p->type == comp_type && has_covered(p, offset, GO_TARGET_LEN / 2) &&
            p->str[p->len - 1 - (offset - GO_TARGET_LEN / 2)] == '_') {
            int current = c->position + (offset - GO_TARGET_LEN / 2) * k_stride[c->gen_dir];
            for (int i = 0; i < p->len; ++i, current -= k_stride[c->gen_dir]) {
                char piece = p->str[p->len - 1 - i];
                if ((piece == '_' || piece == '^') && current != c->position)
                    compound_update_pose(ev, c, delta, current, comp_dir, -c->favour);
            }
            break;
        }
        go__gen_next(ev->ac, &it);
    }
}

/* Pattern.cpp:488-518 */
static void compound_update(go_evaluator *ev, go_compound *c, int delta) {
    for (int k = 0; k < c->ncomp; ++k) {
        int dir = c->comps[k].dir, type = c->comps[k].type;
        if (2 * c->count + delta == -1) return;
        compound_update_pose(ev, c, delta, c->position, dir, c->favour);       /* updateCritical */
        compound_update_pose(ev, c, delta, c->position, dir, -c->favour);
        if (!c->triple_cross && c->l3_count == 0) compound_update_antis(ev, c, delta, dir, type);
        if (2 * c->count + delta == 3) rec_set_total(&ev->compound_dist[GO_N][c->type], delta, c->favour);
        c->count += delta;
    }
}

/* Pattern.cpp:435-438 */
static void compound_init(go_evaluator *ev, go_compound *c, int pose, int favour) {
    memset(c, 0, sizeof *c);
    c->position = pose; c->favour = favour; c->gen_dir = -1;
    go__gen_init(&c->generator, NULL, 0);
    compound_locate(ev, c);
}

/* ---- Updater (Pattern.cpp:111-302) ---- */
static void upd_reset(go_evaluator *ev, int delta, int move, int player) {
    ev->delta = delta; ev->move = move; ev->player = player; ev->ncompounds = 0;
}

static int find_compound(const go_evaluator *ev, int pose, int player) {
    for (int i = 0; i < ev->ncompounds; ++i)
        if (ev->compound_keys[i].pos == pose && ev->compound_keys[i].player == player) return i;
    return -1;
}

/* Pattern.cpp:128-136 */
static void match_patterns(go_evaluator *ev, int dir) {
    int slot = ev->delta == 1;
    ev->nresults[slot][dir] = 0;
    go_gen g;
    go__gen_init(&g, line_view(ev, ev->move, dir), GO_TARGET_LEN);
    while (go__gen_next(ev->ac, &g)) {
        int pat = go__gen_pattern(ev->ac, &g);
        if (has_covered(&ev->ac->patterns[pat], g.offset, GO_TARGET_LEN / 2)) {
            if (ev->nresults[slot][dir] >= MAX_RESULTS) { ev->err |= 4; break; }
            go_entry *e = &ev->results[slot][dir][ev->nresults[slot][dir]++];
            e->pat = pat; e->offset = g.offset;
        }
    }
}

/* Pattern.cpp:138-165 */
static void update_patterns(go_evaluator *ev, int dir) {
    int slot = ev->delta == 1, delta = ev->delta;
    for (int r = 0; r < ev->nresults[slot][dir]; ++r) {
        const go_pattern *p = &ev->ac->patterns[ev->results[slot][dir][r].pat];
        int offset = ev->results[slot][dir][r].offset;
        if (p->type == GO_FIVE) {
            ev->board.cur_player = GO_NONE;
            ev->board.winner = p->favour;
            continue;
        }
        int current = ev->move + (offset - GO_TARGET_LEN / 2) * k_stride[dir];
        rec_set_total(&ev->pattern_dist[GO_N][p->type], delta, p->favour);
        for (int i = 0; i < p->len; ++i, current -= k_stride[dir]) {
            char piece = p->str[p->len - 1 - i];
            if (piece == '_' || piece == '^') {
                const double multiplier = (dir == GO_DIR_LD || dir == GO_DIR_RD ? 1.2 : 1);
                const int score = (int)(delta * multiplier * p->score);
                if (piece == '_') {
                    rec_set_flag(&ev->pattern_dist[current][p->type], delta, p->favour, p->favour, dir);
                    ev->scores[group2(p->favour, p->favour)][current] += score;
                }
                rec_set_flag(&ev->pattern_dist[current][p->type], delta, p->favour, -p->favour, dir);
                ev->scores[group2(p->favour, -p->favour)][current] += score;
            }
        }
    }
}

/* Pattern.cpp:167-197 (the `#if true` branch) */
static void update_compound(go_evaluator *ev, int dir) {
    const uint8_t *view = line_view(ev, ev->move, dir);
    static const int players[2] = { GO_WHITE, GO_BLACK };
    for (int pi = 0; pi < 2; ++pi) {
        int player = players[pi], current = -1, offset = 0;
        for (int i = 0; i < GO_TARGET_LEN; ++i) {
            if (view[i] != 4) continue;
            else if (current == -1) current = ev->move + (i - GO_TARGET_LEN / 2) * k_stride[dir];
            else current += (i - offset) * k_stride[dir];
            offset = i;
            if (ev->density[group1(player)][0][current] < 2) continue;
            if (find_compound(ev, current, player) >= 0) continue;
            if (compound_test(ev, current, player)) {
                if (ev->ncompounds >= MAX_COMPOUNDS) { ev->err |= 4; continue; }
                int k = ev->ncompounds++;
                ev->compound_keys[k].pos = current; ev->compound_keys[k].player = player;
                compound_init(ev, &ev->compounds[k], current, player);
                compound_update(ev, &ev->compounds[k], ev->delta);
            }
        }
    }
}

/* Pattern.cpp:598-609 */
static const int k_block_weights[GO_BLOCK][GO_BLOCK] = {
    { 2, 0, 0, 1, 0, 0, 2 },
    { 0, 4, 3, 3, 3, 4, 0 },
    { 0, 3, 5, 4, 5, 3, 0 },
    { 1, 3, 4, 0, 4, 3, 1 },
    { 0, 3, 5, 4, 5, 3, 0 },
    { 0, 4, 3, 3, 3, 4, 0 },
    { 2, 0, 0, 1, 0, 0, 2 },
};
static const int k_block_score = 160;

/* Pattern.cpp:236-272 (block bounds: :94-109) */
static void update_block(go_evaluator *ev, int delta, int src_player) {
    int move = ev->move, mx = move % GO_W, my = move / GO_W;
    int left = mx - 3 > 0 ? mx - 3 : 0, right = mx + 3 < GO_W - 1 ? mx + 3 : GO_W - 1;
    int up = my - 3 > 0 ? my - 3 : 0, down = my + 3 < GO_H - 1 ? my + 3 : GO_H - 1;
    int32_t *count_arr = ev->density[group1(src_player)][0];
    int32_t *weight_arr = ev->density[group1(src_player)][1];
    int32_t *score_arr = ev->scores[group2(src_player, src_player)];
    int mask_before[GO_BLOCK][GO_BLOCK];
    for (int y = up; y <= down; ++y) for (int x = left; x <= right; ++x)
        mask_before[y - up][x - left] = weight_arr[y * GO_W + x] > 0;            /* mask_block (.eval()) */
    for (int y = up; y <= down; ++y) for (int x = left; x <= right; ++x) {       /* weight_block += sign * delta * W */
        int q = y * GO_W + x, w = k_block_weights[y - my + 3][x - mx + 3];
        weight_arr[q] += (weight_arr[q] < 0 ? -1 : 1) * delta * w;
    }
    for (int y = up; y <= down; ++y) for (int x = left; x <= right; ++x) {       /* sign_block is lazy: re-read */
        int q = y * GO_W + x, w = k_block_weights[y - my + 3][x - mx + 3];
        count_arr[q] += (weight_arr[q] < 0 ? -1 : 1) * delta * (w > 0 ? 1 : 0);
    }
    for (int pp = 0; pp < 2; ++pp) for (int cw = 0; cw < 2; ++cw) {              /* {Black, White} x {count, weight} */
        int32_t *value = &ev->density[pp == 0 ? 1 : 0][cw][move];
        if (delta == 1) { *value *= -1; *value -= 1; }
        else if (delta == -1) { *value += 1; *value *= -1; }
    }
    for (int y = up; y <= down; ++y) for (int x = left; x <= right; ++x) {
        int q = y * GO_W + x;
        score_arr[q] += k_block_score * ((weight_arr[q] > 0) - mask_before[y - up][x - left]);
    }
    int count = ev->density[group1(-src_player)][0][move];
    if (count != 0 && count != -1) ev->scores[group2(-src_player, -src_player)][move] -= delta * k_block_score;
}

/* Pattern.cpp:274-302 */
static void update_move(go_evaluator *ev, int move, int src_player) {
    upd_reset(ev, -1, move, src_player);
    for (int dir = 0; dir < 4; ++dir) match_patterns(ev, dir);
    for (int dir = 0; dir < 4; ++dir) update_compound(ev, dir);
    for (int dir = 0; dir < 4; ++dir) update_patterns(ev, dir);
    if (src_player != GO_NONE) {
        boardmap_apply(ev, move);
        update_block(ev, 1, src_player);
    } else {
        boardmap_revert(ev, 1);
        update_block(ev, -1, ev->board.cur_player);
    }
    upd_reset(ev, 1, move, src_player);
    for (int dir = 0; dir < 4; ++dir) match_patterns(ev, dir);
    for (int dir = 0; dir < 4; ++dir) update_patterns(ev, dir);
    for (int dir = 0; dir < 4; ++dir) update_compound(ev, dir);
}

/* ---- Evaluator facade (Pattern.cpp:306-386) ---- */
void go_eval_reset(go_evaluator *ev) {
    boardmap_reset(ev);
    memset(ev->scores, 0, sizeof ev->scores);
    memset(ev->density, 0, sizeof ev->density);
    memset(ev->pattern_dist, 0, sizeof ev->pattern_dist);
    memset(ev->compound_dist, 0, sizeof ev->compound_dist);
    ev->err = 0;
}

go_evaluator *go_eval_new(void) {
    go_evaluator *ev = (go_evaluator *)calloc(1, sizeof *ev);
    ev->ac = go_default_ac();
    go_eval_reset(ev);
    return ev;
}

void go_eval_free(go_evaluator *ev) { free(ev); }

/* Pattern.cpp:310-335 */
int go_eval_apply(go_evaluator *ev, int move, int *err) {
    if (ev->board.cur_player != GO_NONE && go_board_check_move(&ev->board, move))
        update_move(ev, move, ev->board.cur_player);
    for (int i = 0; i < GO_N; ++i) {                       /* the reference throws here */
        if (!ev->board.states[GO_NONE + 1][i]) {
            for (int j = 0; j < 4; ++j) if (ev->scores[j][i] != 0) ev->err |= 1;
        } else {
            for (int j = 0; j < 4; ++j) if (ev->scores[j][i] < 0) ev->err |= 1;
        }
    }
    if (err) *err = ev->err;
    return ev->board.cur_player;
}

/* Pattern.cpp:337-342 */
int go_eval_revert(go_evaluator *ev, int count) {
    for (int i = 0; i < count && ev->board.nrec > 0; ++i)
        update_move(ev, ev->board.record[ev->board.nrec - 1], GO_NONE);
    return ev->board.cur_player;
}

/* Pattern.cpp:344-354 */
int go_eval_check_end(go_evaluator *ev) {
    if (ev->board.cur_player == GO_NONE) return 1;
    if (ev->board.counts[GO_NONE + 1] == 0) { ev->board.winner = GO_NONE; ev->board.cur_player = GO_NONE; return 1; }
    return 0;
}

const go_board *go_eval_board(const go_evaluator *ev) { return &ev->board; }
void go_eval_get_scores(const go_evaluator *ev, int32_t *out) { memcpy(out, ev->scores, sizeof ev->scores); }
void go_eval_get_density(const go_evaluator *ev, int32_t *out) { memcpy(out, ev->density, sizeof ev->density); }
void go_eval_get_pattern_dist(const go_evaluator *ev, uint32_t *out) { memcpy(out, ev->pattern_dist, sizeof ev->pattern_dist); }
void go_eval_get_compound_dist(const go_evaluator *ev, uint32_t *out) { memcpy(out, ev->compound_dist, sizeof ev->compound_dist); }

void go_eval_replay_batch(const uint8_t *moves, const int32_t *lens, int stride, int n,
                          int32_t *scores, int32_t *density, uint32_t *totals, int32_t *status) {
    go_evaluator *ev = go_eval_new();
    for (int b = 0; b < n; ++b) {
        go_eval_reset(ev);
        for (int i = 0; i < lens[b]; ++i) go_eval_apply(ev, moves[(size_t)b * stride + i], NULL);
        go_eval_check_end(ev);
        if (scores)  memcpy(scores + (size_t)b * 4 * GO_N, ev->scores, sizeof ev->scores);
        if (density) memcpy(density + (size_t)b * 4 * GO_N, ev->density, sizeof ev->density);
        if (totals) {
            for (int t = 0; t < GO_PT_SIZE - 1; ++t) totals[(size_t)b * 11 + t] = ev->pattern_dist[GO_N][t];
            for (int t = 0; t < GO_CT_SIZE; ++t) totals[(size_t)b * 11 + 8 + t] = ev->compound_dist[GO_N][t];
        }
        if (status) {
            int end = ev->board.cur_player == GO_NONE;
            status[b] = (end ? 1 : 0) | (ev->err ? 2 : 0) |
                        ((int)(uint8_t)ev->board.winner << 8) | ((int)(uint8_t)ev->board.cur_player << 16);
        }
    }
    go_eval_free(ev);
}
