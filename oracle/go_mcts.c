/*
 * go_mcts.c -- restatement of MCTS + Default policy algorithms + RandomPolicy, with the
 * reference's std::mt19937 replaced by the counter-based Philox4x32-10 stream that the GPU
 * path uses.  TEST INFRASTRUCTURE, see gomoku_oracle.h.
 * Follows core/lib/src/MCTS.cpp:99-198, core/lib/include/algorithms/MonteCarlo.hpp:13-110,
 * core/lib/include/policies/Random.h:22-35, core/lib/include/algorithms/Statistical.hpp:29-44.
 *
 * Compile with -ffp-contract=off: PUCB is f64 from f32 operands, the running mean is f32.
 */
#include "gomoku_oracle.h"
#include "../include/gomoku_noise.h"   /* the counter-based Dirichlet sampler, written once for the kernels and for this file */
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <float.h>

/* ---- Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants) ---- */
void go_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ---- MT19937 (Matsumoto & Nishimura 1998) == std::mt19937, used only by the KAT hooks ---- */
void go_mt_seed(go_mt19937 *g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
uint32_t go_mt_next(go_mt19937 *g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

/* ---- Node store (MCTS.h:25-66).  Children of one node are contiguous, ascending cell id. ---- */
typedef struct {
    int32_t  parent;
    int16_t  position;
    int8_t   player;
    float    state_value, action_prob;
    uint64_t node_visits;
    int32_t  first_child, n_children;
} mnode;

struct go_mcts {
    mnode   *nodes;
    int32_t  n, cap, root;
    uint64_t size, iterations;
    double   c_puct;
    int      c_rollouts;
    uint64_t seed;
    uint32_t game_id;
    int32_t  init_acts;
    uint64_t alg_bytes;
    int      use_mt;
    go_mt19937 mt;
    float    noise_alpha, noise_epsilon;      /* alpha == 0: AddNoise disabled */
    int      noise_sampler;                   /* 0: std::gamma_distribution over std::mt19937 (go_stdsort.cpp); 1: gomoku_noise.h */
    go_eval_state_fn eval_state;              /* Policy(eval_state = ...) (agents/alphazero.py:5-9): replaces the rollouts */
    void    *eval_user;
};

void go__gamma_draws(unsigned seed, float alpha, int n, float *out);     /* go_stdsort.cpp */

static int32_t new_node(go_mcts *m, int parent, int pos, int player, float value, float prob) {
    if (m->n == m->cap) { m->cap = m->cap ? 2 * m->cap : 4096; m->nodes = (mnode *)realloc(m->nodes, sizeof(mnode) * (size_t)m->cap); }
    mnode *nd = &m->nodes[m->n];
    nd->parent = parent; nd->position = (int16_t)pos; nd->player = (int8_t)player;
    nd->state_value = value; nd->action_prob = prob; nd->node_visits = 0;
    nd->first_child = -1; nd->n_children = 0;
    return m->n++;
}

/* MCTS.cpp:84-97 (iterations constructor; last_move = -1, last_player = White) */
go_mcts *go_mcts_new(uint64_t c_iterations, double c_puct, int c_rollouts, uint64_t seed, uint32_t game_id) {
    go_mcts *m = (go_mcts *)calloc(1, sizeof *m);
    m->iterations = c_iterations; m->c_puct = c_puct; m->c_rollouts = c_rollouts;
    m->seed = seed; m->game_id = game_id;
    m->root = new_node(m, -1, -1, GO_WHITE, 0.0f, 1.0f);
    m->size = 1;
    return m;
}

void go_mcts_free(go_mcts *m) { if (m) { free(m->nodes); free(m); } }

/* MCTS.cpp:149-156 */
void go_mcts_reset(go_mcts *m) {
    m->n = 0;
    m->root = new_node(m, -1, -1, GO_WHITE, 0.0f, 1.0f);
    m->size = 1;
}

/* MCTS.cpp:129-134 : std::max_element => first child with maximal visits */
int go_mcts_step_forward(go_mcts *m) {
    mnode *root = &m->nodes[m->root];
    if (root->n_children > 0) {
        int best = root->first_child;
        for (int i = 1; i < root->n_children; ++i)
            if (m->nodes[best].node_visits < m->nodes[root->first_child + i].node_visits) best = root->first_child + i;
        m->root = best;
        m->nodes[best].parent = -1;
    }
    return m->nodes[m->root].position;
}

/* MCTS.cpp:136-147 */
int go_mcts_step_forward_move(go_mcts *m, int move) {
    mnode *root = &m->nodes[m->root];
    int found = -1;
    for (int i = 0; i < root->n_children; ++i)
        if (m->nodes[root->first_child + i].position == move) { found = root->first_child + i; break; }
    if (found < 0) {
        int player = -root->player;
        found = new_node(m, -1, move, player, 0.0f, 1.0f);
    }
    m->root = found;
    m->nodes[found].parent = -1;
    return m->nodes[found].position;
}

/* MCTS.cpp:119-125 */
void go_mcts_sync_with_board(go_mcts *m, const go_board *b) {
    int rootpos = m->nodes[m->root].position, i = 0;
    while (i < b->nrec && b->record[i] != rootpos) ++i;
    i = (i == b->nrec) ? 0 : i + 1;
    for (; i < b->nrec; ++i) go_mcts_step_forward_move(m, b->record[i]);
}

/* MonteCarlo.hpp:23-28, 57-68 */
static int32_t default_select(go_mcts *m, int32_t ni) {
    const mnode *node = &m->nodes[ni];
    size_t max_index = 0;
    double max_score = -1.0;
    for (int i = 0; i < node->n_children; ++i) {
        const mnode *child = &m->nodes[node->first_child + i];
        const double P_i = child->action_prob;
        const double N = (double)node->node_visits;
        const double n_i = (double)(child->node_visits + 1);
        double score = child->state_value + m->c_puct * P_i * sqrt(N) / n_i;
        if (score > max_score) { max_score = score; max_index = (size_t)i; }
    }
    m->alg_bytes += (uint64_t)node->n_children * 8;
    return node->first_child + (int32_t)max_index;
}

/* MonteCarlo.hpp:71-80 with UniformProbs (:50-55): prior = 1/float(#empties) on every empty cell */
static size_t default_expand(go_mcts *m, int32_t ni, const go_board *b) {
    float prob = 1.0f / (float)b->counts[GO_NONE + 1];
    int first = -1, count = 0;
    int player = -m->nodes[ni].player;
    for (int i = 0; i < GO_N; ++i) {
        float p = b->states[GO_NONE + 1][i] ? prob : 0.0f;
        if (p != 0.0f && go_board_check_move(b, i)) {
            int32_t c = new_node(m, ni, i, player, 0.0f, p);
            if (first < 0) first = c;
            ++count;
        }
    }
    m->nodes[ni].first_child = first;
    m->nodes[ni].n_children = count;
    m->alg_bytes += (uint64_t)count * 16;
    return (size_t)count;
}

/* Default::Expand with the probabilities an evaluator returned and extraCheck = true (MonteCarlo.hpp:71-80): one child
   per cell whose probability is not 0 AND which is a legal move */
static size_t probs_expand(go_mcts *m, int32_t ni, const go_board *b, const float *probs) {
    int first = -1, count = 0;
    int player = -m->nodes[ni].player;
    for (int i = 0; i < GO_N; ++i)
        if (probs[i] != 0.0f && go_board_check_move(b, i)) {
            int32_t c = new_node(m, ni, i, player, 0.0f, probs[i]);
            if (first < 0) first = c;
            ++count;
        }
    m->nodes[ni].first_child = first;
    m->nodes[ni].n_children = count;
    m->alg_bytes += (uint64_t)count * 16;
    return (size_t)count;
}

/* Random.h:22-35 + MonteCarlo.hpp:37-47.  RNG: Philox counter = (game, playout, search<<8 | rollout, ply>>3), key = seed;
   ply p takes the 16-bit half (p&1) of word (p>>1)&3; cell draw = (half * 225) >> 16, then the probe rule (Game.cpp:68-72).
   (The KAT hook keeps the survey's recipe: id = mt19937() % 225.) */
static float averaged_simulate(go_mcts *m, go_board *b, uint32_t playout) {
    int init_player = b->cur_player;
    double score = 0;
    uint32_t key[2] = { (uint32_t)m->seed, (uint32_t)(m->seed >> 32) };
    for (int i = 0; i < m->c_rollouts; ++i) {
        int total_moves = 0;
        uint32_t words[4] = { 0, 0, 0, 0 };
        for (int result = b->cur_player; result != GO_NONE; ++total_moves) {
            unsigned draw;
            if (m->use_mt) {
                draw = go_mt_next(&m->mt) % GO_N;
            } else {
                if ((total_moves & 7) == 0) {
                    uint32_t ctr[4] = { m->game_id, playout, ((uint32_t)m->init_acts << 8) | (uint32_t)i, (uint32_t)total_moves >> 3 };
                    go_philox4x32(ctr, key, words);
                }
                uint32_t half = (words[(total_moves >> 1) & 3] >> (16 * (total_moves & 1))) & 0xFFFFu;
                draw = (half * 225u) >> 16;
            }
            result = go_board_apply(b, go_board_random_move(b, draw), 1);
        }
        score += (double)((float)init_player * (float)b->winner);   /* CalcScore (Game.h:34-36) */
        go_board_revert(b, total_moves);
    }
    score /= (double)m->c_rollouts;
    return (float)score;
}

/* MonteCarlo.hpp:90-95 */
static void default_backprop(go_mcts *m, int32_t ni, float value) {
    for (; ni >= 0; ni = m->nodes[ni].parent, value = -value) {
        mnode *nd = &m->nodes[ni];
        nd->node_visits += 1;
        nd->state_value += (value - nd->state_value) / (float)nd->node_visits;
        m->alg_bytes += 16;
    }
}

/* MCTS.cpp:158-177 */
static size_t playout(go_mcts *m, go_board *b, uint32_t idx) {
    int32_t ni = m->root;
    while (m->nodes[ni].n_children != 0) {
        ni = default_select(m, ni);
        go_board_apply(b, m->nodes[ni].position, 0);             /* Policy::applyMove: no victory check */
    }
    double node_value;
    size_t expand_size;
    if (!go_board_check_end(b)) {
        float state_value;
        if (m->eval_state) {                                     /* policy->simulate = the callback (MCTS.cpp:18-33, policy_ext / mcts_ext bindings) */
            uint8_t states[6 * GO_N];
            float probs[GO_N];
            go_board_encoded_states(b, states);
            state_value = 0.0f;
            memset(probs, 0, sizeof probs);
            m->eval_state(states, &state_value, probs, m->eval_user);
            expand_size = probs_expand(m, ni, b, probs);
        } else {
            state_value = averaged_simulate(m, b, idx);
            expand_size = default_expand(m, ni, b);
        }
        node_value = -state_value;
    } else {
        expand_size = 0;
        node_value = (float)m->nodes[ni].player * (float)b->winner;
    }
    default_backprop(m, ni, (float)node_value);
    go_board_revert(b, b->nrec - m->init_acts);
    return expand_size;
}

/* MCTS.cpp:179-198 (iteration constraint). */
/* Default::AddNoise (MonteCarlo.hpp:97-108) + Stats::DirichletNoise (Statistical.hpp:29-34).  The engine is
   std::mt19937 seeded with Philox(game, stones, 'nois') instead of random_device. */
static void add_noise(go_mcts *m, const go_board *b) {
    mnode *root = &m->nodes[m->root];
    if (!(m->noise_alpha > 0.0f) || root->n_children == 0) return;
    float prior[GO_N], noise[GO_N], draws[GO_N], sq = 0.0f;
    uint32_t ctr[4] = { m->game_id, (uint32_t)b->nrec, 0x6E6F6973u, 0u }, key[2] = { (uint32_t)m->seed, (uint32_t)(m->seed >> 32) }, w[4];
    int k = 0;
    for (int i = 0; i < GO_N; ++i) prior[i] = 0.0f;
    for (int i = 0; i < root->n_children; ++i) prior[m->nodes[root->first_child + i].position] = m->nodes[root->first_child + i].action_prob;
    if (m->noise_sampler == 1) {                                 /* the stream the device-resident loops draw from (include/gomoku_noise.h) */
        gmk_noise_mix225(prior, m->noise_alpha, m->noise_epsilon, m->game_id, (uint32_t)b->nrec, key[0], key[1]);
    } else {
        for (int i = 0; i < GO_N; ++i) prior[i] *= 1 - m->noise_epsilon;
        go_philox4x32(ctr, key, w);
        go__gamma_draws(w[0], m->noise_alpha, root->n_children, draws);
        for (int i = 0; i < GO_N; ++i) { noise[i] = prior[i] ? draws[k++] : 0.0f; sq += noise[i] * noise[i]; }
        if (sq > 0.0f) { float nrm = sqrtf(sq); for (int i = 0; i < GO_N; ++i) noise[i] = noise[i] / nrm; }
        for (int i = 0; i < GO_N; ++i) prior[i] += m->noise_epsilon * noise[i];
    }
    for (int i = 0; i < root->n_children; ++i) m->nodes[root->first_child + i].action_prob = prior[m->nodes[root->first_child + i].position];
}

void go_mcts_set_noise(go_mcts *m, float alpha, float epsilon) { m->noise_alpha = alpha; m->noise_epsilon = epsilon; }
void go_mcts_set_noise_sampler(go_mcts *m, int sampler) { m->noise_sampler = sampler; }
/* one draw of the counter-based sampler / the mix on a vector of priors by cell: what the KATs of tests/test_noise.py pin */
float go_noise_gamma(float alpha, uint32_t game_id, uint32_t stones, uint32_t cell, uint64_t seed) { return gmk_noise_gamma(alpha, game_id, stones, cell, (uint32_t)seed, (uint32_t)(seed >> 32)); }
void go_noise_mix225(float *p, float alpha, float epsilon, uint32_t game_id, uint32_t stones, uint64_t seed) { gmk_noise_mix225(p, alpha, epsilon, game_id, stones, (uint32_t)seed, (uint32_t)(seed >> 32)); }
double go_noise_log(double x) { return gmk_noise_log(x); }
double go_noise_exp(double x) { return gmk_noise_exp(x); }
void go_mcts_set_evaluator(go_mcts *m, go_eval_state_fn fn, void *user) { m->eval_state = fn; m->eval_user = user; }

void go_mcts_run_playouts(go_mcts *m, go_board *b) {
    go_mcts_sync_with_board(m, b);
    add_noise(m, b);                                             /* MCTS.cpp:182 */
    m->init_acts = b->nrec;                                      /* Policy::prepare */
    for (uint64_t i = 0; i < m->iterations; ++i) m->size += playout(m, b, (uint32_t)i);
    go_board_revert(b, b->nrec - m->init_acts);                  /* Policy::cleanup */
}

/* MCTS.cpp:99-102 */
int go_mcts_get_action(go_mcts *m, go_board *b) {
    go_mcts_run_playouts(m, b);
    return go_mcts_step_forward(m);
}

/* MCTS.cpp:104-117 + Statistical.hpp:37-42.  Float reductions follow a fixed sequential order;
   Eigen's packet order is unpinned, so tests compare pi with a tolerance. */
void go_visits_to_pi(const uint32_t *visits, int n_moves_on_board, float *pi) {
    float v[GO_N], sq = 0.0f;
    const float eps = FLT_EPSILON;
    for (int i = 0; i < GO_N; ++i) { v[i] = (float)visits[i]; sq += v[i] * v[i]; }
    if (sq > 0.0f) { float nrm = sqrtf(sq); for (int i = 0; i < GO_N; ++i) v[i] = v[i] / nrm; }
    for (int i = 0; i < GO_N; ++i) v[i] = v[i] ? v[i] + 1 : v[i];
    float temperature = (float)(n_moves_on_board < 15 ? 1 : 1e-2);
    double e[GO_N], sum = 0.0;
    for (int i = 0; i < GO_N; ++i) { e[i] = exp((double)(logf(v[i] + eps) / temperature)); sum += e[i]; }
    for (int i = 0; i < GO_N; ++i) { float p = (float)(e[i] / sum); pi[i] = p > eps ? p : 0.0f; }
}

float go_mcts_eval_state(go_mcts *m, go_board *b, float *probs, uint32_t *visits) {
    uint32_t cv[GO_N];
    go_mcts_run_playouts(m, b);
    memset(cv, 0, sizeof cv);
    const mnode *root = &m->nodes[m->root];
    for (int i = 0; i < root->n_children; ++i) {
        const mnode *c = &m->nodes[root->first_child + i];
        cv[c->position] = (uint32_t)c->node_visits;
    }
    if (visits) memcpy(visits, cv, sizeof cv);
    if (probs) go_visits_to_pi(cv, b->nrec, probs);
    return root->state_value;
}

void go_mcts_use_mt19937(go_mcts *m, uint32_t seed) { m->use_mt = 1; go_mt_seed(&m->mt, seed); }

uint64_t go_mcts_size(const go_mcts *m) { return m->size; }
int go_mcts_root_position(const go_mcts *m) { return m->nodes[m->root].position; }
int go_mcts_root_player(const go_mcts *m) { return m->nodes[m->root].player; }
uint64_t go_mcts_root_visits(const go_mcts *m) { return m->nodes[m->root].node_visits; }
float go_mcts_root_value(const go_mcts *m) { return m->nodes[m->root].state_value; }
uint64_t go_mcts_alg_bytes(const go_mcts *m) { return m->alg_bytes; }

void go_mcts_root_children(const go_mcts *m, uint32_t *visits, float *values, float *priors) {
    const mnode *root = &m->nodes[m->root];
    for (int i = 0; i < GO_N; ++i) { if (visits) visits[i] = 0; if (values) values[i] = 0.0f; if (priors) priors[i] = 0.0f; }
    for (int i = 0; i < root->n_children; ++i) {
        const mnode *c = &m->nodes[root->first_child + i];
        if (visits) visits[c->position] = (uint32_t)c->node_visits;
        if (values) values[c->position] = c->state_value;
        if (priors) priors[c->position] = c->action_prob;
    }
}
