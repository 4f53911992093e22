/*
 * go_trad.c -- restatement of the pattern-guided search: TraditionalPolicy (core/lib/include/policies/Traditional.h:17-69)
 * on top of Heuristic (core/lib/include/algorithms/Heuristic.hpp:16-45, 94-200), RAVE::Select / BackPropogate<false>
 * (core/lib/include/algorithms/MonteCarlo.hpp:149-184), Default::Expand (:71-80) and MCTS::playout / runPlayouts /
 * stepForward (core/lib/src/MCTS.cpp:129-198).  TEST INFRASTRUCTURE, see gomoku_oracle.h.
 *
 * The search is deterministic (no random numbers).  Its float reductions go through Eigen in the reference
 * (normalized(), dot()), whose summation order is not part of any contract: this file fixes ONE order,
 * sum225() below, and the GPU kernel uses the same one, so that oracle and kernel agree bit for bit; against the real
 * reference the float part is "parity unpinned" (no reference test covers it and the reference cannot be built here).
 * normalized() follows Eigen 3.3+ (a zero vector stays zero).  Expressions keep the reference's types: float
 * products and sums, the 0.6 / 0.4 literals taken as float (Eigen's promote_scalar_arg), PUCB and tanh in double.
 */
#include "go_eval_internal.h"
#include "../include/gomoku_noise.h"   /* the counter-based Dirichlet sampler, written once for the kernels and for this file */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t  parent;
    int16_t  pos;
    int8_t   player;
    float    value, prior;          /* state_value, action_prob */
    uint64_t visits;
    int32_t  first, n;              /* children: kids[first .. first + n), in the order the reference's vector holds them */
} tnode;

struct go_trad {
    go_evaluator *ev;               /* TraditionalPolicy::m_evaluator: lives across searches (its flag words keep history) */
    double   c_puct;
    size_t   cached, init;          /* m_cachedActs, m_initActs */
    tnode   *nodes; int n_nodes, cap_nodes;
    int32_t *kids;  int n_kids, cap_kids;
    int      root;
    uint64_t evaluator_updates;     /* instrumentation: applied + reverted moves */
    /* the MCTS object around the policy (MCTS.h:135-180) when the tree is kept from search to search */
    int      have_tree;
    float    noise_alpha, noise_epsilon;
    int      noise_sampler;         /* 0: std::gamma_distribution over std::mt19937; 1: include/gomoku_noise.h */
    uint64_t seed; uint32_t game_id;
};

void go__gamma_draws(unsigned seed, float alpha, int n, float *out);     /* go_stdsort.cpp */

static int group1(int player) { return player == GO_BLACK; }
static int group2(int favour, int perspective) { return ((favour == GO_BLACK) << 1) | (perspective == GO_BLACK); }

/* the one summation order of this file and of the kernel: 64 strided partial sums; a binary tree inside every group of
   16 (offsets 8, 4, 2, 1: one DPP row on the GPU); then (group 0 + group 1) + (group 2 + group 3) */
static float sum225(const float *x) {
    float p[64];
    for (int l = 0; l < 64; ++l) {
        p[l] = x[l];
        for (int k = 1; l + 64 * k < GO_N; ++k) p[l] += x[l + 64 * k];
    }
    for (int r = 0; r < 4; ++r)
        for (int s = 8; s > 0; s >>= 1)
            for (int l = 0; l < s; ++l) p[16 * r + l] += p[16 * r + l + s];
    return (p[0] + p[16]) + (p[32] + p[48]);
}

/* MatrixBase::normalized() / normalize(), Eigen 3.3+ */
static void normalize225(float *x) {
    float sq[GO_N];
    for (int i = 0; i < GO_N; ++i) sq[i] = x[i] * x[i];
    float z = sum225(sq);
    if (z > 0.0f) {
        float n = sqrtf(z);
        for (int i = 0; i < GO_N; ++i) x[i] = x[i] / n;
    }
}

/* Heuristic::DensityWeight (Heuristic.hpp:39-45) */
static void density_weight(const go_evaluator *ev, int player, float *out) {
    const int32_t *counts = ev->density[group1(player)][0], *weights = ev->density[group1(player)][1];
    for (int i = 0; i < GO_N; ++i) {
        float N = (float)(counts[i] > 0 ? counts[i] : 0), W = (float)(weights[i] > 0 ? weights[i] : 0);
        out[i] = (3.0f * W) / (1.0f + 2.0f * N);
    }
    normalize225(out);
}

/* Heuristic::EvaluationProbs (Heuristic.hpp:16-28) */
static void evaluation_probs(const go_evaluator *ev, int player, float *probs) {
    if (ev->board.nrec != 0) {
        float dw_self[GO_N], dw_rival[GO_N];
        density_weight(ev, player, dw_self);
        density_weight(ev, -player, dw_rival);
        const int32_t *s_self = ev->scores[group2(player, player)], *s_rival = ev->scores[group2(-player, player)];
        for (int i = 0; i < GO_N; ++i) {
            float self_worthy = (float)s_self[i] * dw_self[i], rival_anti = (float)s_rival[i] * dw_rival[i];
            probs[i] = 0.6f * self_worthy + 0.4f * rival_anti;
        }
        normalize225(probs);
    } else {
        memset(probs, 0, GO_N * sizeof *probs);
        probs[(GO_H / 2) * GO_W + GO_W / 2] = 1.0f;
    }
}

/* Heuristic::EvaluationValue (Heuristic.hpp:33-37) */
static float evaluation_value(const go_evaluator *ev, int player) {
    float dw[GO_N], prod[GO_N];
    density_weight(ev, player, dw);
    for (int i = 0; i < GO_N; ++i) prod[i] = (float)ev->scores[group2(player, player)][i] * dw[i];
    float self_worthy = sum225(prod);
    density_weight(ev, -player, dw);
    for (int i = 0; i < GO_N; ++i) prod[i] = (float)ev->scores[group2(-player, -player)][i] * dw[i];
    float rival_worthy = sum225(prod);
    return (float)tanh((1.2 * self_worthy - rival_worthy) / 500.0f);
}

/* Evaluator::Record getters (Pattern.cpp:402-416) */
static unsigned rec_group(uint32_t field, int favour, int perspective) { return (field >> (8 * (unsigned)group2(favour, perspective))) & 0xffu; }
static unsigned rec_total(uint32_t field, int player) { return (field >> (16 * (unsigned)group1(player))) & 0xffffu; }

/* Heuristic::DecisiveFilter (Heuristic.hpp:94-161).  A candidate is (pattern, player): pattern < GO_PT_SIZE is a
   Pattern::Type, otherwise GO_PT_SIZE + Compound::Type. */
static void decisive_filter(const go_evaluator *ev, float *probs) {
    enum { S4, SL3, STo44, STo43, STo33, SEnd };
    static const struct { int state, anti; } table[2][6] = {
        { {S4, 1},  {STo44, 0}, {SL3, 1},   {STo43, 1}, {STo33, 1}, {SEnd, 0} },
        { {SL3, 0}, {STo44, 1}, {STo43, 0}, {STo33, 0}, {SEnd, 0},  {SEnd, 1} } };
    struct { int pattern, player; } cand[8];
    int head = 0, tail = 0;                      /* the std::deque */
    int state = S4, anti = 0;
    const int cur_player = ev->board.cur_player;
    while (state != SEnd) {
        const int player = anti ? -cur_player : cur_player;
        switch (state) {
            case S4:  cand[tail].pattern = GO_LIVE4; cand[tail++].player = player;
                      cand[tail].pattern = GO_DEAD4; cand[tail++].player = player; break;
            case SL3: cand[tail].pattern = GO_LIVE3; cand[tail++].player = player; break;
            default:  cand[tail].pattern = GO_PT_SIZE + (STo33 - state); cand[tail++].player = player; break;
        }
        while (head != tail) {
            const int pattern = cand[head].pattern, pl = cand[head].player;
            unsigned count = pattern < GO_PT_SIZE ? rec_total(ev->pattern_dist[GO_N][pattern], pl)
                                                  : rec_total(ev->compound_dist[GO_N][pattern - GO_PT_SIZE], pl);
            if (count != 0) {
                if (anti && state != S4) { cand[tail].pattern = GO_DEAD3; cand[tail++].player = -pl; }
                break;
            }
            ++head;
        }
        if (head != tail) {
            for (int i = 0; i < GO_N; ++i) {
                int keep = 0;
                for (int k = head; k < tail && !keep; ++k) {
                    const int pattern = cand[k].pattern, pl = cand[k].player;
                    keep = pattern < GO_PT_SIZE ? rec_group(ev->pattern_dist[i][pattern], pl, cur_player) != 0
                                                : rec_group(ev->compound_dist[i][pattern % GO_PT_SIZE], pl, cur_player) != 0;
                }
                if (!keep) probs[i] = 0.0f;
            }
            normalize225(probs);
            head = tail = 0;
            state = SEnd;
        } else {
            const int s = table[anti][state].state, a = table[anti][state].anti;
            state = s; anti = a;
            head = tail = 0;
        }
    }
}

/* ---- tree ---- */
static int new_node(go_trad *t, int parent, int pos, int player, float value, float prior) {
    if (t->n_nodes == t->cap_nodes) { t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 1 << 16; t->nodes = (tnode *)realloc(t->nodes, (size_t)t->cap_nodes * sizeof(tnode)); }
    tnode *nd = &t->nodes[t->n_nodes];
    nd->parent = parent; nd->pos = (int16_t)pos; nd->player = (int8_t)player; nd->value = value; nd->prior = prior;
    nd->visits = 0; nd->first = 0; nd->n = 0;
    return t->n_nodes++;
}

/* Default::Expand with extraCheck = false (MonteCarlo.hpp:71-80, Traditional.h:20) */
static void expand(go_trad *t, int node, const float *probs) {
    int count = 0;
    for (int i = 0; i < GO_N; ++i) count += probs[i] != 0.0f;
    if (t->n_kids + count > t->cap_kids) { while (t->n_kids + count > t->cap_kids) t->cap_kids = t->cap_kids ? 2 * t->cap_kids : 1 << 16; t->kids = (int32_t *)realloc(t->kids, (size_t)t->cap_kids * sizeof(int32_t)); }
    t->nodes[node].first = t->n_kids;
    for (int i = 0; i < GO_N; ++i)
        if (probs[i] != 0.0f) {
            int child = new_node(t, node, i, -t->nodes[node].player, 0.0f, probs[i]);
            t->kids[t->n_kids++] = child;
        }
    t->nodes[node].n = count;
}

/* Heuristic::CachedApplyMove (Heuristic.hpp:165-189); the outer board's virtual state is not restated (nothing reads it) */
static void cached_apply_move(go_trad *t, int move) {
    go_evaluator *ev = t->ev;
    go_board *ref = &ev->board;
    if (t->cached == (size_t)ref->nrec || ref->record[t->cached] != move) {
        if ((size_t)ref->nrec - t->cached > t->cached - 0) {
            int16_t cached_record[GO_N];
            memcpy(cached_record, ref->record, sizeof cached_record);
            go_eval_reset(ev);
            for (size_t i = 0; i < t->cached; ++i) { go_eval_apply(ev, cached_record[i], NULL); ++t->evaluator_updates; }
        } else {
            t->evaluator_updates += (size_t)ref->nrec - t->cached;
            go_eval_revert(ev, (int)((size_t)ref->nrec - t->cached));
        }
        go_eval_apply(ev, move, NULL);
        ++t->evaluator_updates;
        if (t->cached < (size_t)ref->nrec) ++t->cached;
    } else {
        ++t->cached;
    }
}

/* Heuristic::CachedRevertMove (Heuristic.hpp:192-200): only the inner BOARD goes back to the cached length */
static void cached_revert_move(go_trad *t) {
    go_board *ref = &t->ev->board;
    go_board_revert(ref, (int)((size_t)ref->nrec - t->cached));
    t->cached = t->init;
}

/* RAVE::BackPropogate<false> (MonteCarlo.hpp:160-184) */
static void back_propagate(go_trad *t, int node, float value) {
    for (; node >= 0; node = t->nodes[node].parent, value = -value) {
        tnode *nd = &t->nodes[node];
        int max_index = 0;
        double max_score = -INFINITY;
        for (int i = 0; i < nd->n; ++i) {
            const tnode *ch = &t->nodes[t->kids[nd->first + i]];
            const double P_i = ch->prior, N = (double)nd->visits, n_i = (double)(ch->visits + 1);       /* Default::PUCB (:23-28) */
            double score = t->c_puct * P_i * sqrt(N) / n_i;
            score += ch->value;
            if (score > max_score) { max_score = score; max_index = i; }
        }
        if (nd->n) { int32_t tmp = t->kids[nd->first]; t->kids[nd->first] = t->kids[nd->first + max_index]; t->kids[nd->first + max_index] = tmp; }
        nd->visits += 1;
        nd->value += (value - nd->value) / (float)nd->visits;
    }
}

/* MCTS::playout (MCTS.cpp:158-177) with TraditionalPolicy's functions */
static void playout(go_trad *t) {
    int node = t->root;
    while (t->nodes[node].n) {                                   /* RAVE::Select: the first child (MonteCarlo.hpp:149-152) */
        node = t->kids[t->nodes[node].first];
        cached_apply_move(t, t->nodes[node].pos);
    }
    double node_value;
    if (!go_eval_check_end(t->ev)) {                             /* TraditionalPolicy::checkGameEnd (Traditional.h:41-46) */
        float probs[GO_N];                                       /* hybridSimulate (Traditional.h:48-69); report.level stays None */
        const int init_player = t->ev->board.cur_player;
        evaluation_probs(t->ev, init_player, probs);
        decisive_filter(t->ev, probs);
        float value = evaluation_value(t->ev, init_player);
        expand(t, node, probs);
        node_value = -value;
    } else {
        node_value = (double)(t->nodes[node].player * t->ev->board.winner);      /* CalcScore (Game.h:34-36) */
    }
    back_propagate(t, node, (float)node_value);
    cached_revert_move(t);
}

/* Evaluator::syncWithBoard (Pattern.cpp:356-368) */
static void eval_sync(go_trad *t, const uint8_t *moves, int n) {
    go_evaluator *ev = t->ev;
    int i = 0;
    for (; i < n; ++i) {
        if (i < ev->board.nrec) {
            if (ev->board.record[i] == moves[i]) continue;
            t->evaluator_updates += (uint64_t)(ev->board.nrec - i);
            go_eval_revert(ev, ev->board.nrec - i);
        }
        go_eval_apply(ev, moves[i], NULL);
        ++t->evaluator_updates;
    }
    t->evaluator_updates += (uint64_t)(ev->board.nrec - i);
    go_eval_revert(ev, ev->board.nrec - i);
}

go_trad *go_trad_new(double c_puct) {
    go_trad *t = (go_trad *)calloc(1, sizeof *t);
    t->ev = go_eval_new();
    t->c_puct = c_puct;
    return t;
}

void go_trad_free(go_trad *t) { if (t) { go_eval_free(t->ev); free(t->nodes); free(t->kids); free(t); } }

/* One search from the position reached by `moves` with a fresh root (MCTS(c_iterations, last_move, last_player) then
   runPlayouts, MCTS.cpp:179-198; Default::AddNoise is a no-op on a childless root).  The evaluator is NOT reset: like
   the reference's policy object it is synchronised with the new position from wherever the last search left it. */
void go_trad_search(go_trad *t, const uint8_t *moves, int n_moves, uint64_t playouts) {
    t->n_nodes = 0; t->n_kids = 0;
    t->root = new_node(t, -1, n_moves ? moves[n_moves - 1] : -1, (n_moves & 1) ? GO_BLACK : GO_WHITE, 0.0f, 1.0f);
    t->init = (size_t)n_moves;                                   /* Policy::prepare (MCTS.cpp:40-42) */
    eval_sync(t, moves, n_moves);                                /* TraditionalPolicy::prepare (Traditional.h:27-31) */
    t->cached = t->init;
    for (uint64_t i = 0; i < playouts; ++i) playout(t);
}

/* root statistics by cell, and the child MCTS::stepForward would pick (most visited, first maximum in the CURRENT
   child order, MCTS.cpp:131-136); returns that move or -1 */
int go_trad_root_children(const go_trad *t, uint32_t *visits, float *values, float *priors) {
    const tnode *r = &t->nodes[t->root];
    int best = -1; uint64_t best_visits = 0;
    if (visits) memset(visits, 0, GO_N * sizeof *visits);
    if (values) memset(values, 0, GO_N * sizeof *values);
    if (priors) memset(priors, 0, GO_N * sizeof *priors);
    for (int i = 0; i < r->n; ++i) {
        const tnode *ch = &t->nodes[t->kids[r->first + i]];
        if (visits) visits[ch->pos] = (uint32_t)ch->visits;
        if (values) values[ch->pos] = ch->value;
        if (priors) priors[ch->pos] = ch->prior;
        if (best < 0 || ch->visits > best_visits) { best = ch->pos; best_visits = ch->visits; }
    }
    return best;
}

/* ---- the same search inside a persistent MCTS object: MCTS::syncWithBoard / stepForward / runPlayouts (MCTS.cpp:119-198) ---- */
void go_trad_set_noise(go_trad *t, float alpha, float epsilon, uint64_t seed, uint32_t game_id) {
    t->noise_alpha = alpha; t->noise_epsilon = epsilon; t->seed = seed; t->game_id = game_id;
}
void go_trad_set_noise_sampler(go_trad *t, int sampler) { t->noise_sampler = sampler; }

/* MCTS::stepForward(next_move) (MCTS.cpp:136-147): the child of that move becomes the root, or a new node does */
static void step_forward_move(go_trad *t, int move) {
    tnode *r = &t->nodes[t->root];
    int next = -1;
    for (int i = 0; i < r->n && next < 0; ++i)
        if (t->nodes[t->kids[r->first + i]].pos == move) next = t->kids[r->first + i];
    if (next < 0) next = new_node(t, -1, move, -t->nodes[t->root].player, 0.0f, 1.0f);
    t->nodes[next].parent = -1;
    t->root = next;
}

/* MCTS::stepForward() (MCTS.cpp:129-134): the most visited child, first maximum in the current order; returns its move */
int go_trad_step_forward(go_trad *t) {
    const tnode *r = &t->nodes[t->root];
    int best = -1; uint64_t best_visits = 0;
    for (int i = 0; i < r->n; ++i) {
        const tnode *ch = &t->nodes[t->kids[r->first + i]];
        if (best < 0 || ch->visits > best_visits) { best = t->kids[r->first + i]; best_visits = ch->visits; }
    }
    if (best >= 0) { t->nodes[best].parent = -1; t->root = best; }
    return t->nodes[t->root].pos;
}

/* Default::AddNoise (MonteCarlo.hpp:97-108) + Stats::DirichletNoise (Statistical.hpp:29-34), seeded like go_mcts.c:
   std::mt19937(Philox(game, stones, 'nois'; seed).word0), one gamma draw per child in ascending cell order */
static void add_noise(go_trad *t, int stones) {
    tnode *root = &t->nodes[t->root];
    if (!(t->noise_alpha > 0.0f) || root->n == 0) return;
    float prior[GO_N], noise[GO_N], draws[GO_N], sq[GO_N];
    uint32_t ctr[4] = { t->game_id, (uint32_t)stones, 0x6E6F6973u, 0u }, key[2] = { (uint32_t)t->seed, (uint32_t)(t->seed >> 32) }, w[4];
    int k = 0;
    for (int i = 0; i < GO_N; ++i) prior[i] = 0.0f;
    for (int i = 0; i < root->n; ++i) prior[t->nodes[t->kids[root->first + i]].pos] = t->nodes[t->kids[root->first + i]].prior;
    if (t->noise_sampler == 1) {                                  /* the stream the device-resident loops draw from */
        gmk_noise_mix225(prior, t->noise_alpha, t->noise_epsilon, t->game_id, (uint32_t)stones, key[0], key[1]);
    } else {
        for (int i = 0; i < GO_N; ++i) prior[i] *= 1 - t->noise_epsilon;
        go_philox4x32(ctr, key, w);
        go__gamma_draws(w[0], t->noise_alpha, root->n, draws);
        for (int i = 0; i < GO_N; ++i) { noise[i] = prior[i] ? draws[k++] : 0.0f; sq[i] = noise[i] * noise[i]; }
        float z = 0.0f;
        for (int i = 0; i < GO_N; ++i) z += sq[i];               /* sequential, like go_mcts.c and the product's host code */
        if (z > 0.0f) { float nrm = sqrtf(z); for (int i = 0; i < GO_N; ++i) noise[i] = noise[i] / nrm; }
        for (int i = 0; i < GO_N; ++i) prior[i] += t->noise_epsilon * noise[i];
    }
    for (int i = 0; i < root->n; ++i) t->nodes[t->kids[root->first + i]].prior = prior[t->nodes[t->kids[root->first + i]].pos];
}

/* MCTS::runPlayouts (MCTS.cpp:179-198) on the kept tree: syncWithBoard, AddNoise, Policy::prepare, the playouts */
void go_trad_run(go_trad *t, const uint8_t *moves, int n_moves, uint64_t playouts) {
    if (!t->have_tree) {                                          /* MCTS(c_iterations): root = (Position(-1), White) */
        t->n_nodes = 0; t->n_kids = 0;
        t->root = new_node(t, -1, -1, GO_WHITE, 0.0f, 1.0f);
        t->have_tree = 1;
    }
    int i = 0;                                                    /* MCTS::syncWithBoard (MCTS.cpp:119-125) */
    while (i < n_moves && moves[i] != t->nodes[t->root].pos) ++i;
    i = (i == n_moves) ? 0 : i + 1;
    for (; i < n_moves; ++i) step_forward_move(t, moves[i]);
    add_noise(t, n_moves);
    t->init = (size_t)n_moves;
    eval_sync(t, moves, n_moves);
    t->cached = t->init;
    for (uint64_t k = 0; k < playouts; ++k) playout(t);
}

uint64_t go_trad_root_visits(const go_trad *t) { return t->nodes[t->root].visits; }
float    go_trad_root_value(const go_trad *t) { return t->nodes[t->root].value; }
int      go_trad_n_nodes(const go_trad *t) { return t->n_nodes; }
uint64_t go_trad_evaluator_updates(const go_trad *t) { return t->evaluator_updates; }
const go_evaluator *go_trad_evaluator(const go_trad *t) { return t->ev; }

/* the policy head alone, on a fresh evaluator replayed in order: probs after DecisiveFilter and the value */
float go_trad_heuristic(const uint8_t *moves, int n_moves, float *probs) {
    go_evaluator *ev = go_eval_new();
    for (int i = 0; i < n_moves; ++i) go_eval_apply(ev, moves[i], NULL);
    float value = 0.0f;
    if (!go_eval_check_end(ev)) {
        evaluation_probs(ev, ev->board.cur_player, probs);
        decisive_filter(ev, probs);
        value = evaluation_value(ev, ev->board.cur_player);
    } else {
        memset(probs, 0, GO_N * sizeof *probs);
    }
    go_eval_free(ev);
    return value;
}
