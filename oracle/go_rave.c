/*
 * go_rave.c -- restatement of PoolRAVEPolicy (core/lib/include/policies/PoolRAVE.h:7-52): MCTS::playout
 * (core/lib/src/MCTS.cpp:158-177) with
 *   select    RAVE::Select: the first child (core/lib/include/algorithms/MonteCarlo.hpp:149-152)
 *   simulate  defaultSimulate: UniformProbs of the leaf, then ONE Default::RandomRollout that is NOT taken back
 *             (PoolRAVE.h:29-48, MonteCarlo.hpp:37-47, 50-55)
 *   expand    Default::Expand without the legality check (MonteCarlo.hpp:71-80), nodes are AMAFNodes (:113-122)
 *   backup    RAVE::BackPropogate<true> on the finished board: all-moves-as-first statistics of every child whose
 *             move its player made later in the game, HandSelect weighting, best child to the front (:124-184)
 * inside a persistent MCTS object (syncWithBoard / stepForward / runPlayouts, MCTS.cpp:119-198).
 * TEST INFRASTRUCTURE, see gomoku_oracle.h.
 *
 * Random numbers: the reference seeds std::mt19937 from random_device (Game.cpp:11-13); like go_mcts.c this file draws
 * from Philox4x32-10 instead, counter = (game id, playout, stones on the root board << 8, ply >> 3), eight 16-bit draws per
 * block, cell = (draw * 225) >> 16, then the reference's probe rule (Game.cpp:68-72).  The playout number restarts at 0
 * whenever the root changes and continues across runs from the same root.
 */
#include "gomoku_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t  parent;
    int16_t  pos;
    int8_t   player;
    float    value, prior;          /* state_value, action_prob */
    uint64_t visits;
    float    amaf_value;            /* AMAFNode (MonteCarlo.hpp:113-122) */
    uint64_t amaf_visits;
    int32_t  first, n;              /* children: kids[first .. first + n), in the order the reference's vector holds them */
} rnode;

struct go_rave {
    double   c_puct, c_bias;
    rnode   *nodes; int n_nodes, cap_nodes;
    int32_t *kids;  int n_kids, cap_kids;
    int      root;
    int      init;                  /* Policy::m_initActs */
    go_board board;
    uint64_t seed; uint32_t game_id;
    uint32_t playouts_done;
    float    noise_alpha, noise_epsilon;
    uint64_t size;                  /* MCTS::m_size */
};

void go__gamma_draws(unsigned seed, float alpha, int n, float *out);     /* go_stdsort.cpp */

static int new_node(go_rave *t, int parent, int pos, int player, float value, float prior) {
    if (t->n_nodes == t->cap_nodes) { t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 1 << 16; t->nodes = (rnode *)realloc(t->nodes, (size_t)t->cap_nodes * sizeof(rnode)); }
    rnode *nd = &t->nodes[t->n_nodes];
    nd->parent = parent; nd->pos = (int16_t)pos; nd->player = (int8_t)player; nd->value = value; nd->prior = prior;
    nd->visits = 0; nd->amaf_value = 0.0f; nd->amaf_visits = 0; nd->first = 0; nd->n = 0;
    return t->n_nodes++;
}

go_rave *go_rave_new(double c_puct, double c_bias, uint64_t seed, uint32_t game_id) {
    go_rave *t = (go_rave *)calloc(1, sizeof *t);
    t->c_puct = c_puct; t->c_bias = c_bias; t->seed = seed; t->game_id = game_id;
    t->root = new_node(t, -1, -1, GO_WHITE, 0.0f, 1.0f);         /* MCTS(c_iterations): root = (Position(-1), White) (MCTS.cpp:84-97) */
    t->size = 1;
    go_board_reset(&t->board);
    return t;
}

void go_rave_free(go_rave *t) { if (t) { free(t->nodes); free(t->kids); free(t); } }

void go_rave_set_noise(go_rave *t, float alpha, float epsilon) { t->noise_alpha = alpha; t->noise_epsilon = epsilon; }

/* Default::Expand with extraCheck = false (MonteCarlo.hpp:71-80, PoolRAVE.h:16) */
static size_t expand(go_rave *t, int node, const float *probs) {
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
    return (size_t)count;
}

/* RAVE::HandSelect / WeightedValue (MonteCarlo.hpp:124-142); MinMSE and c_bias are dead code there */
static double weighted_value(const rnode *nd) {
    const double n = (double)nd->visits, k = (double)(size_t)800;
    const double weight = sqrt(k / (3 * n + k));
    return (1 - weight) * nd->value + weight * nd->amaf_value;
}

/* RAVE::BackPropogate<true> (MonteCarlo.hpp:154-184); the board holds the finished game */
static void back_propagate(go_rave *t, int node, float value) {
    const go_board *b = &t->board;
    for (; node >= 0; node = t->nodes[node].parent, value = -value) {
        rnode *nd = &t->nodes[node];
        int max_index = 0;
        double max_score = -INFINITY;
        for (int i = 0; i < nd->n; ++i) {
            rnode *ch = &t->nodes[t->kids[nd->first + i]];
            const double P_i = ch->prior, N = (double)nd->visits, n_i = (double)(ch->visits + 1);       /* Default::PUCB (:23-28) */
            double score = t->c_puct * P_i * sqrt(N) / n_i;
            if (b->states[ch->player + 1][ch->pos]) {            /* the same player made this move, here or later */
                ch->amaf_visits += 1;
                ch->amaf_value += (-value - ch->amaf_value) / (float)ch->amaf_visits;
            }
            score += weighted_value(ch);
            if (score > max_score) { max_score = score; max_index = i; }
        }
        if (nd->n) { int32_t tmp = t->kids[nd->first]; t->kids[nd->first] = t->kids[nd->first + max_index]; t->kids[nd->first + max_index] = tmp; }
        nd->visits += 1;
        nd->value += (value - nd->value) / (float)nd->visits;
    }
}

/* MCTS::playout (MCTS.cpp:158-177) with PoolRAVEPolicy's functions */
static size_t playout(go_rave *t, uint32_t idx) {
    go_board *b = &t->board;
    int node = t->root;
    while (t->nodes[node].n) {                                   /* RAVE::Select: the first child */
        node = t->kids[t->nodes[node].first];
        go_board_apply(b, t->nodes[node].pos, 0);                /* Policy::applyMove: no victory check */
    }
    double node_value;
    size_t expand_size = 0;
    if (!go_board_check_end(b)) {
        float probs[GO_N];                                       /* Default::UniformProbs (MonteCarlo.hpp:50-55), before the rollout */
        const float empties = (float)b->counts[GO_NONE + 1];
        for (int i = 0; i < GO_N; ++i) probs[i] = (b->states[GO_NONE + 1][i] ? 1.0f : 0.0f) / empties;
        const int init_player = b->cur_player;
        const uint32_t key[2] = { (uint32_t)t->seed, (uint32_t)(t->seed >> 32) };
        uint32_t words[4] = { 0, 0, 0, 0 };
        int total_moves = 0;                                     /* Default::RandomRollout, the board keeps the finished game */
        for (int result = b->cur_player; result != GO_NONE; ++total_moves) {
            if ((total_moves & 7) == 0) {
                const uint32_t ctr[4] = { t->game_id, idx, (uint32_t)t->init << 8, (uint32_t)total_moves >> 3 };
                go_philox4x32(ctr, key, words);
            }
            const uint32_t half = (words[(total_moves >> 1) & 3] >> (16 * (total_moves & 1))) & 0xFFFFu;
            result = go_board_apply(b, go_board_random_move(b, (half * 225u) >> 16), 1);
        }
        const float state_value = (float)init_player * (float)b->winner;        /* CalcScore (Game.h:34-36) */
        expand_size = expand(t, node, probs);
        node_value = -state_value;
    } else {
        node_value = (float)t->nodes[node].player * (float)b->winner;
    }
    back_propagate(t, node, (float)node_value);
    go_board_revert(b, b->nrec - t->init);
    return expand_size;
}

/* MCTS::stepForward(next_move) (MCTS.cpp:136-147) */
static void step_forward_move(go_rave *t, int move) {
    rnode *r = &t->nodes[t->root];
    int next = -1;
    for (int i = 0; i < r->n && next < 0; ++i)
        if (t->nodes[t->kids[r->first + i]].pos == move) next = t->kids[r->first + i];
    if (next < 0) next = new_node(t, -1, move, -t->nodes[t->root].player, 0.0f, 1.0f);
    t->nodes[next].parent = -1;
    t->root = next;
    t->playouts_done = 0;
}

/* MCTS::stepForward() (MCTS.cpp:129-134): the most visited child, first maximum in the current order; returns its move */
int go_rave_step_forward(go_rave *t) {
    const rnode *r = &t->nodes[t->root];
    int best = -1; uint64_t best_visits = 0;
    for (int i = 0; i < r->n; ++i) {
        const rnode *ch = &t->nodes[t->kids[r->first + i]];
        if (best < 0 || ch->visits > best_visits) { best = t->kids[r->first + i]; best_visits = ch->visits; }
    }
    if (best >= 0) { t->nodes[best].parent = -1; t->root = best; t->playouts_done = 0; }
    return t->nodes[t->root].pos;
}

/* Default::AddNoise (MonteCarlo.hpp:97-108) + Stats::DirichletNoise (Statistical.hpp:29-34), seeded like go_mcts.c / go_trad.c */
static void add_noise(go_rave *t, int stones) {
    rnode *root = &t->nodes[t->root];
    if (!(t->noise_alpha > 0.0f) || root->n == 0) return;
    float prior[GO_N], noise[GO_N], draws[GO_N], z = 0.0f;
    uint32_t ctr[4] = { t->game_id, (uint32_t)stones, 0x6E6F6973u, 0u }, key[2] = { (uint32_t)t->seed, (uint32_t)(t->seed >> 32) }, w[4];
    int k = 0;
    for (int i = 0; i < GO_N; ++i) prior[i] = 0.0f;
    for (int i = 0; i < root->n; ++i) prior[t->nodes[t->kids[root->first + i]].pos] = t->nodes[t->kids[root->first + i]].prior;
    for (int i = 0; i < GO_N; ++i) prior[i] *= 1 - t->noise_epsilon;
    go_philox4x32(ctr, key, w);
    go__gamma_draws(w[0], t->noise_alpha, root->n, draws);
    for (int i = 0; i < GO_N; ++i) { noise[i] = prior[i] ? draws[k++] : 0.0f; z += noise[i] * noise[i]; }
    if (z > 0.0f) { float nrm = sqrtf(z); for (int i = 0; i < GO_N; ++i) noise[i] = noise[i] / nrm; }
    for (int i = 0; i < GO_N; ++i) prior[i] += t->noise_epsilon * noise[i];
    for (int i = 0; i < root->n; ++i) t->nodes[t->kids[root->first + i]].prior = prior[t->nodes[t->kids[root->first + i]].pos];
}

/* MCTS::runPlayouts (MCTS.cpp:179-198): syncWithBoard, AddNoise, Policy::prepare, the playouts, Policy::cleanup */
void go_rave_run(go_rave *t, const uint8_t *moves, int n_moves, uint64_t playouts) {
    int i = 0;                                                    /* MCTS::syncWithBoard (MCTS.cpp:119-125) */
    while (i < n_moves && moves[i] != t->nodes[t->root].pos) ++i;
    i = (i == n_moves) ? 0 : i + 1;
    for (; i < n_moves; ++i) step_forward_move(t, moves[i]);
    add_noise(t, n_moves);
    go_board_reset(&t->board);
    for (int k = 0; k < n_moves; ++k) go_board_apply(&t->board, moves[k], 1);
    t->init = n_moves;
    for (uint64_t k = 0; k < playouts; ++k) t->size += playout(t, t->playouts_done + (uint32_t)k);
    t->playouts_done += (uint32_t)playouts;
}

/* root statistics by cell and the move MCTS::stepForward() would make (or -1) */
int go_rave_root_children(const go_rave *t, uint32_t *visits, float *values, float *priors, uint32_t *amaf_visits, float *amaf_values) {
    const rnode *r = &t->nodes[t->root];
    int best = -1; uint64_t best_visits = 0;
    if (visits) memset(visits, 0, GO_N * sizeof *visits);
    if (values) memset(values, 0, GO_N * sizeof *values);
    if (priors) memset(priors, 0, GO_N * sizeof *priors);
    if (amaf_visits) memset(amaf_visits, 0, GO_N * sizeof *amaf_visits);
    if (amaf_values) memset(amaf_values, 0, GO_N * sizeof *amaf_values);
    for (int i = 0; i < r->n; ++i) {
        const rnode *ch = &t->nodes[t->kids[r->first + i]];
        if (visits) visits[ch->pos] = (uint32_t)ch->visits;
        if (values) values[ch->pos] = ch->value;
        if (priors) priors[ch->pos] = ch->prior;
        if (amaf_visits) amaf_visits[ch->pos] = (uint32_t)ch->amaf_visits;
        if (amaf_values) amaf_values[ch->pos] = ch->amaf_value;
        if (best < 0 || ch->visits > best_visits) { best = ch->pos; best_visits = ch->visits; }
    }
    return best;
}

uint64_t go_rave_root_visits(const go_rave *t) { return t->nodes[t->root].visits; }
float    go_rave_root_value(const go_rave *t) { return t->nodes[t->root].value; }
uint64_t go_rave_size(const go_rave *t) { return t->size; }
