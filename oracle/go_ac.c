/*
 * go_ac.c -- restatement of AhoCorasickBuilder + PatternSearch::generator
 * (TEST INFRASTRUCTURE, see gomoku_oracle.h).
 * Follows core/lib/src/utils/ACAutomata.cpp:15-274 and core/lib/src/Pattern.cpp:14-74, 554-596.
 */
#include "gomoku_oracle.h"
#include <string.h>
#include <stdlib.h>
#include <math.h>

/* Mapping.h:40-48 (switch with fall-through: blanks 4, '?' 3, 'o' 2, 'x' 1, else 0) */
int go_encode_char(char ch) {
    switch (ch) {
        case '-': case '_': case '^': case '~': return 4;
        case '?': return 3;
        case 'o': return 2;
        case 'x': return 1;
        default:  return 0;
    }
}

/* Pattern.cpp:14-18 */
static void make_pattern(go_pattern *p, const char *proto, int type, int score) {
    memset(p, 0, sizeof *p);
    size_t n = strlen(proto + 1);
    memcpy(p->str, proto + 1, n);
    p->len = (int8_t)n;
    p->favour = proto[0] == '+' ? GO_BLACK : GO_WHITE;
    p->type = (int8_t)type;
    p->score = score;
}

/* ACAutomata.cpp:25-64.  stage 1: reverseAugment, 2: flipAugment, 3: boundaryAugment. */
int go_ac_augment(go_pattern *pats, int n, int stage) {
    int size = n;
    if (stage == 1) {                                       /* :25-33 */
        for (int i = 0; i < size; ++i) {
            go_pattern r = pats[i];
            for (int a = 0, b = r.len - 1; a < b; ++a, --b) { char t = r.str[a]; r.str[a] = r.str[b]; r.str[b] = t; }
            if (strcmp(r.str, pats[i].str) != 0) pats[n++] = r;
        }
    } else if (stage == 2) {                                /* :35-45 */
        for (int i = 0; i < size; ++i) {
            go_pattern f = pats[i];
            f.favour = (int8_t)-f.favour;
            for (int k = 0; k < f.len; ++k) {
                if (f.str[k] == 'x') f.str[k] = 'o';
                else if (f.str[k] == 'o') f.str[k] = 'x';
            }
            pats[n++] = f;
        }
    } else if (stage == 3) {                                /* :47-64 */
        for (int i = 0; i < size; ++i) {
            char enemy = pats[i].favour == GO_BLACK ? 'o' : 'x';
            int first = -1, last = -1;
            for (int k = 0; k < pats[i].len; ++k) if (pats[i].str[k] == enemy) { if (first < 0) first = k; last = k; }
            if (first >= 0) {
                go_pattern bnd = pats[i];
                bnd.str[first] = '?';
                pats[n++] = bnd;
                if (last != first) {
                    bnd.str[last] = '?';
                    pats[n++] = bnd;
                    bnd.str[first] = enemy;
                    pats[n++] = bnd;
                }
            }
        }
    }
    return n;
}

/* ---- trie node set keyed by (depth, first) : ACAutomata.h:11-29, .cpp:105-134 ---- */
typedef struct { int code, depth, first, last; } tnode;

typedef struct {
    tnode *nodes; int n, cap;          /* kept sorted by (depth, first): the std::set order */
} tset;

static int tkey_less(int d1, int f1, int d2, int f2) { return d1 < d2 || (d1 == d2 && f1 < f2); }

/* index of first element not less than (depth, first) */
static int tset_lower(const tset *t, int depth, int first) {
    int lo = 0, hi = t->n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (tkey_less(t->nodes[mid].depth, t->nodes[mid].first, depth, first)) lo = mid + 1; else hi = mid; }
    return lo;
}
/* index of first element greater than (depth, first) */
static int tset_upper(const tset *t, int depth, int first) {
    int lo = 0, hi = t->n;
    while (lo < hi) { int mid = (lo + hi) / 2; if (!tkey_less(depth, first, t->nodes[mid].depth, t->nodes[mid].first)) lo = mid + 1; else hi = mid; }
    return lo;
}
/* std::set::insert / emplace: no-op when the key exists.  Returns index of the (existing or new) node. */
static int tset_insert(tset *t, tnode nd) {
    int i = tset_lower(t, nd.depth, nd.first);
    if (i < t->n && t->nodes[i].depth == nd.depth && t->nodes[i].first == nd.first) return i;
    if (t->n == t->cap) { t->cap = t->cap ? 2 * t->cap : 256; t->nodes = (tnode *)realloc(t->nodes, sizeof(tnode) * (size_t)t->cap); }
    memmove(t->nodes + i + 1, t->nodes + i, sizeof(tnode) * (size_t)(t->n - i));
    t->nodes[i] = nd; t->n++;
    return i;
}
static int tset_find(const tset *t, int depth, int first) {
    int i = tset_lower(t, depth, first);
    return (i < t->n && t->nodes[i].depth == depth && t->nodes[i].first == first) ? i : -1;
}
/* ACAutomata.h:61-65 : children = nodes of depth+1 with first in [node.first, node.last-1] */
static void tset_children(const tset *t, const tnode *nd, int *first, int *last) {
    *first = tset_lower(t, nd->depth + 1, nd->first);
    *last  = tset_upper(t, nd->depth + 1, nd->last - 1);
}

/* ACAutomata.cpp:107-129 : nodes are addressed by key because inserts shift array indices. */
static void insert_pattern(tset *t, const char *suffix, int parent_depth, int parent_first) {
    int pi = tset_find(t, parent_depth, parent_first);
    if (suffix[0] == '\0') {
        t->nodes[pi].last += 1;                                   /* ++parent->last */
        tnode leaf = { 0, parent_depth + 1, parent_first, t->nodes[pi].last };
        tset_insert(t, leaf);                                     /* emplace(0, depth+1, first, last) */
    } else {
        tnode key = { go_encode_char(suffix[0]), parent_depth + 1, 0, 0 };
        int cf, cl, child = -1;
        tset_children(t, &t->nodes[pi], &cf, &cl);
        for (int i = cf; i < cl; ++i) if (t->nodes[i].code == key.code) { child = i; break; }
        int child_first;
        if (child < 0) {
            key.first = t->nodes[pi].last;
            key.last = key.first;
            child = tset_insert(t, key);
        }
        child_first = t->nodes[child].first;
        insert_pattern(t, suffix + 1, parent_depth + 1, child_first);
        pi = tset_find(t, parent_depth, parent_first);
        child = tset_find(t, parent_depth + 1, child_first);
        t->nodes[pi].last = t->nodes[child].last;                 /* parent->last = child->last */
    }
}

/* ---- double-array construction : ACAutomata.cpp:158-229 ---- */
static void dat_grow(go_ac *ac) {
    int pre = ac->size;
    ac->size = 2 * pre;
    for (int i = pre; i < ac->size; ++i) { ac->base[i] = -(i - 1); ac->check[i] = -(i + 1); }
}

static int dat_build_recursive(go_ac *ac, const tset *t, int index, int ni) {
    const tnode *node = &t->nodes[ni];
    if (node->depth > 0 && node->code == 0) {
        ac->base[index] = -node->first;                           /* leaf: -(pattern index) */
        return 0;
    }
    int cf, cl;
    tset_children(t, node, &cf, &cl);
    int begin = 0, front = 0, ok;
    do {
        front = -ac->check[front];
        begin = front - t->nodes[cf].code;
        if (begin >= 0) {                                         /* `if (begin < 0) continue;` jumps to the condition */
            while (begin + 4 + 1 >= ac->size) {
                if (2 * ac->size > GO_MAX_DAT) return -1;
                dat_grow(ac);
            }
        }
        ok = 1;
        for (int i = cf; i < cl; ++i) {
            int c_i = begin + t->nodes[i].code;
            if (!(c_i != 0 && ac->check[c_i] < 0)) { ok = 0; break; }
        }
    } while (!ok);
    for (int i = cf; i < cl; ++i) {
        int c_i = begin + t->nodes[i].code;
        ac->check[-ac->base[c_i]] = ac->check[c_i];               /* unlink from the free list */
        ac->base[-ac->check[c_i]] = ac->base[c_i];
        ac->check[c_i] = index;
    }
    ac->base[index] = begin;
    for (int i = cf; i < cl; ++i)
        if (dat_build_recursive(ac, t, begin + t->nodes[i].code, i) != 0) return -1;
    return 0;
}

/* ACAutomata.cpp:231-274 */
static void build_ac_graph(go_ac *ac) {
    for (int i = 0; i < ac->size; ++i) ac->fail[i] = 0;
    for (int i = 0; i < 5; ++i) ac->invariants[i] = 0;
    int *queue = (int *)malloc(sizeof(int) * (size_t)ac->size);
    int qh = 0, qt = 0;
    queue[qt++] = 0;
    while (qh < qt) {
        int cur = queue[qh++];
        for (int code = 1; code <= 4; ++code) {
            int child = ac->base[cur] + code;
            if (ac->check[child] == cur) queue[qt++] = child;
        }
        if (cur == 0) continue;
        int code = cur - ac->base[ac->check[cur]];
        int pre_fail = ac->check[cur];
        while (pre_fail != 0) {
            pre_fail = ac->fail[pre_fail];
            int fail_node = ac->base[pre_fail] + code;
            if (ac->check[fail_node] == pre_fail) { ac->fail[cur] = fail_node; break; }
        }
        if (ac->check[ac->base[cur] + code] != cur && ac->base[ac->fail[cur]] + code == cur)
            ac->invariants[code] = cur;
    }
    free(queue);
}

void go__std_sort_indices(const int *codes, int *indices, int n);   /* go_stdsort.cpp */

/* ACAutomata.cpp:66-90 : sort key = base-4 digits 1..4 left-aligned to MAX_PATTERN_LEN */
static int sort_code(const go_pattern *p) {
    double align = pow(4.0, (double)(GO_MAX_PAT_LEN - p->len));
    int sum = 0;
    for (int k = 0; k < p->len; ++k) { sum *= 4; sum += go_encode_char(p->str[k]); }
    return (int)(sum * align);
}

static int32_t *g_trie_dump = NULL; static int g_trie_dump_cap = 0, g_trie_dump_n = 0;

/* test hook: the next go_ac_build() copies its (code, depth, first, last) node set, in std::set order,
   into out[4*cap]; go_ac_trie_dump_count() returns how many nodes there were. */
void go_ac_trie_dump_begin(int32_t *out, int cap) { g_trie_dump = out; g_trie_dump_cap = cap; g_trie_dump_n = 0; }
int  go_ac_trie_dump_count(void) { return g_trie_dump_n; }

int go_ac_build(go_ac *ac, const char *const *protos, const int *types, const int *scores, int n) {
    memset(ac, 0, sizeof *ac);
    go_pattern *pats = ac->patterns;
    for (int i = 0; i < n; ++i) make_pattern(&pats[i], protos[i], types[i], scores[i]);
    /* ACAutomata.cpp:15-23 */
    n = go_ac_augment(pats, n, 1);
    n = go_ac_augment(pats, n, 2);
    n = go_ac_augment(pats, n, 3);
    if (n > GO_MAX_PATTERNS) return -1;
    /* sortPatterns (ACAutomata.cpp:66-90): std::sort of indices by key; equal keys exist (counted in
       sort_ties), so the toolchain's std::sort is called to get the order a g++ build gets. */
    int codes[GO_MAX_PATTERNS], indices[GO_MAX_PATTERNS];
    for (int i = 0; i < n; ++i) codes[i] = sort_code(&pats[i]);
    go__std_sort_indices(codes, indices, n);
    {
        go_pattern *medium = (go_pattern *)malloc(sizeof(go_pattern) * (size_t)n);
        for (int i = 0; i < n; ++i) medium[i] = pats[indices[i]];
        memcpy(pats, medium, sizeof(go_pattern) * (size_t)n);
        free(medium);
    }
    for (int i = 1; i < n; ++i) if (codes[indices[i]] == codes[indices[i - 1]]) ac->sort_ties++;
    ac->n_patterns = n;
    /* buildNodeBasedTrie */
    tset t = { 0, 0, 0 };
    tnode root = { 0, 0, 0, 0 };
    tset_insert(&t, root);
    for (int i = 0; i < n; ++i) insert_pattern(&t, pats[i].str, 0, 0);
    if (g_trie_dump) {
        g_trie_dump_n = t.n;
        for (int i = 0; i < t.n && i < g_trie_dump_cap; ++i) {
            g_trie_dump[4 * i] = t.nodes[i].code; g_trie_dump[4 * i + 1] = t.nodes[i].depth;
            g_trie_dump[4 * i + 2] = t.nodes[i].first; g_trie_dump[4 * i + 3] = t.nodes[i].last;
        }
        g_trie_dump = NULL;
    }
    /* buildDAT */
    ac->size = 1; ac->base[0] = 0; ac->check[0] = -1;
    int rc = dat_build_recursive(ac, &t, 0, tset_find(&t, 0, 0));
    free(t.nodes);
    if (rc != 0) return rc;
    build_ac_graph(ac);
    return 0;
}

/* Pattern.cpp:554-596 */
static const char *const k_protos[] = {
    "+xxxxx", "-_oooo_", "-xoooo_", "-o_ooo", "-oo_oo", "-~_ooo_~", "-x^ooo_~", "-~o_oo~",
    "-~o~oo_~", "-~oo~o_~", "-x_o~oo~", "-x_oo~o~", "-xooo__~", "-xoo_o_~", "-xoo__o~", "-xo_oo_~",
    "-xo__oo", "-xooo__x", "-xoo_o_x", "-xoo__ox", "-xo_oo_x", "-x_ooo_x", "-~oo__o~", "-oo__oo",
    "-o_o_o", "-~oo__~", "-~_o_o_~", "-x^o_o_^", "-^o__o^", "-xoo___", "-xo_o__", "-xo__o_",
    "-o___o", "-x_oo__x", "-x_o_o_x", "-~o___~", "-x~_o__^", "-x~__o_^", "-xo___~", "-x_o___x",
    "-x__o__x",
};
static const int k_types[] = {
    GO_FIVE, GO_LIVE4, GO_DEAD4, GO_DEAD4, GO_DEAD4, GO_LIVE3, GO_LIVE3, GO_LIVE3,
    GO_DEAD3, GO_DEAD3, GO_DEAD3, GO_DEAD3, GO_DEAD3, GO_DEAD3, GO_DEAD3, GO_DEAD3,
    GO_DEAD3, GO_DEAD3, GO_DEAD3, GO_DEAD3, GO_DEAD3, GO_DEAD3, GO_DEAD3, GO_DEAD3,
    GO_DEAD3, GO_LIVE2, GO_LIVE2, GO_LIVE2, GO_LIVE2, GO_DEAD2, GO_DEAD2, GO_DEAD2,
    GO_DEAD2, GO_DEAD2, GO_DEAD2, GO_LIVE1, GO_LIVE1, GO_LIVE1, GO_DEAD1, GO_DEAD1,
    GO_DEAD1,
};
static const int k_scores[] = {
    9999, 9000, 2500, 3000, 2600, 3000, 2900, 2800,
    1400, 1200, 1300, 1100, 510, 520, 520, 530,
    530, 500, 500, 500, 500, 500, 750, 540,
    550, 650, 600, 550, 550, 150, 160, 170,
    180, 120, 120, 150, 140, 150, 30, 40,
    50,
};

int go_ac_build_default(go_ac *ac) {
    return go_ac_build(ac, k_protos, k_types, k_scores, (int)(sizeof k_protos / sizeof k_protos[0]));
}

const go_ac *go_default_ac(void) {
    /* built once, also when several host threads ask at the same time (tools/stress_parity.py, bench.py's all-cores baseline): the pointer is
     * published only behind the finished automaton */
    static go_ac *ac = NULL;
    static int state = 0;                        /* 0 nobody has started, 1 being built, 2 ready */
    if (__atomic_load_n(&state, __ATOMIC_ACQUIRE) != 2) {
        int expected = 0;
        if (__atomic_compare_exchange_n(&state, &expected, 1, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
            go_ac *fresh = (go_ac *)malloc(sizeof(go_ac));
            go_ac_build_default(fresh);
            ac = fresh;
            __atomic_store_n(&state, 2, __ATOMIC_RELEASE);
        } else {
            while (__atomic_load_n(&state, __ATOMIC_ACQUIRE) != 2) { }
        }
    }
    return ac;
}

int go_ac_used_slots(const go_ac *ac) {
    int used = 1;                                /* the root */
    for (int i = 1; i < ac->size; ++i) if (ac->check[i] >= 0) used++;
    return used;
}

/* ---- generator (Pattern.cpp:33-62) ---- */
typedef struct { const uint8_t *t; int n; int pos; int offset; int state; } go_gen;

#define IS_TERMINAL(ac, s) ((ac)->check[(ac)->base[s]] == (s))

/* operator++ : returns 1 if positioned on a match, 0 at end. */
static int gen_next(const go_ac *ac, go_gen *g) {
    do {
        if (g->pos >= g->n) { g->state = 0; return 0; }          /* target.empty(): reset + break => end */
        int code = g->t[g->pos];
        if (g->state == ac->invariants[code]) {
            while (g->pos < g->n && g->t[g->pos] == code) { ++g->offset; ++g->pos; }
            continue;                                             /* evaluates the loop condition */
        }
        int next = ac->base[g->state] + code;
        if (ac->check[next] == g->state) {
            g->state = next;
        } else if (g->state != 0) {
            g->state = ac->fail[g->state];
            continue;                                             /* evaluates the loop condition */
        }
        ++g->offset; ++g->pos;
    } while (!IS_TERMINAL(ac, g->state));
    return 1;
}

/* operator* */
static int gen_pattern(const go_ac *ac, const go_gen *g) { return -ac->base[ac->base[g->state]]; }

int go_ac_match(const go_ac *ac, const uint8_t *codes, int n, int32_t *pat_idx, int32_t *offsets, int cap) {
    go_gen g = { codes, n, 0, -1, 0 };
    int m = 0;
    while (gen_next(ac, &g)) {
        if (m < cap) { pat_idx[m] = gen_pattern(ac, &g); offsets[m] = g.offset; }
        ++m;
    }
    return m;
}

/* exported to go_eval.c */
void go__gen_init(go_gen *g, const uint8_t *t, int n) { g->t = t; g->n = n; g->pos = 0; g->offset = -1; g->state = 0; }
int  go__gen_next(const go_ac *ac, go_gen *g) { return gen_next(ac, g); }
int  go__gen_pattern(const go_ac *ac, const go_gen *g) { return gen_pattern(ac, g); }
