// pattern_tables.cpp -- see pattern_tables.h.
//
// The construction steps are those of the reference (core/lib/src/utils/ACAutomata.cpp): augment
// (:25-64), sort by a base-4 key (:66-90), interval-labelled trie (:105-134), double array with a
// free list (:158-229), BFS fail links and "invariant" self-loop states (:231-274).  They are kept
// because the RESULT depends on them: four pairs of patterns have equal sort keys, std::sort leaves
// three of the pairs swapped (libstdc++), and the interval trie then files the longer pattern of each
// swapped pair under the wrong node, so the automaton the reference really runs recognises 291 of its
// 294 patterns and reports one pattern under another's state.  A textbook Aho-Corasick would not.
#include "pattern_tables.h"

#include <algorithm>
#include <deque>
#include <map>
#include <numeric>
#include <stdexcept>

namespace gmk {

namespace {

constexpr int kMaxLen = 7;          // PatternConfig::MAX_PATTERN_LEN (Pattern.h:12)
constexpr int kSymbols = 4;         // Codeset (Mapping.h:51)

std::vector<PatternRec> production_protos() {
    // Pattern.cpp:554-596.  First char: '+' favours black, '-' favours white.
    struct P { const char* s; int type; int score; };
    static const P table[] = {
        {"+xxxxx", Five, 9999},      {"-_oooo_", LiveFour, 9000},   {"-xoooo_", DeadFour, 2500},
        {"-o_ooo", DeadFour, 3000},  {"-oo_oo", DeadFour, 2600},    {"-~_ooo_~", LiveThree, 3000},
        {"-x^ooo_~", LiveThree, 2900}, {"-~o_oo~", LiveThree, 2800}, {"-~o~oo_~", DeadThree, 1400},
        {"-~oo~o_~", DeadThree, 1200}, {"-x_o~oo~", DeadThree, 1300}, {"-x_oo~o~", DeadThree, 1100},
        {"-xooo__~", DeadThree, 510}, {"-xoo_o_~", DeadThree, 520},  {"-xoo__o~", DeadThree, 520},
        {"-xo_oo_~", DeadThree, 530}, {"-xo__oo", DeadThree, 530},   {"-xooo__x", DeadThree, 500},
        {"-xoo_o_x", DeadThree, 500}, {"-xoo__ox", DeadThree, 500},  {"-xo_oo_x", DeadThree, 500},
        {"-x_ooo_x", DeadThree, 500}, {"-~oo__o~", DeadThree, 750},  {"-oo__oo", DeadThree, 540},
        {"-o_o_o", DeadThree, 550},   {"-~oo__~", LiveTwo, 650},     {"-~_o_o_~", LiveTwo, 600},
        {"-x^o_o_^", LiveTwo, 550},   {"-^o__o^", LiveTwo, 550},     {"-xoo___", DeadTwo, 150},
        {"-xo_o__", DeadTwo, 160},    {"-xo__o_", DeadTwo, 170},     {"-o___o", DeadTwo, 180},
        {"-x_oo__x", DeadTwo, 120},   {"-x_o_o_x", DeadTwo, 120},    {"-~o___~", LiveOne, 150},
        {"-x~_o__^", LiveOne, 140},   {"-x~__o_^", LiveOne, 150},    {"-xo___~", DeadOne, 30},
        {"-x_o___x", DeadOne, 40},    {"-x__o__x", DeadOne, 50},
    };
    std::vector<PatternRec> out;
    for (const P& p : table) out.push_back({std::string(p.s + 1), p.s[0] == '+' ? 1 : -1, p.type, p.score});
    return out;
}

}  // namespace

PatternAutomaton::PatternAutomaton() : PatternAutomaton(production_protos()) {}

PatternAutomaton::PatternAutomaton(const std::vector<PatternRec>& protos) : patterns_(protos) {
    augment();
    sort_patterns();
    build_trie();
    build_double_array();
    build_fail_links();
    flatten();
}

// Mirror images, colour swaps and board-edge variants (ACAutomata.cpp:25-64).
void PatternAutomaton::augment() {
    const size_t n0 = patterns_.size();
    for (size_t i = 0; i < n0; ++i) {
        PatternRec r = patterns_[i];
        std::reverse(r.rich.begin(), r.rich.end());
        if (r.rich != patterns_[i].rich) patterns_.push_back(r);
    }
    const size_t n1 = patterns_.size();
    for (size_t i = 0; i < n1; ++i) {
        PatternRec f = patterns_[i];
        f.favour = -f.favour;
        for (char& c : f.rich) c = (c == 'x') ? 'o' : (c == 'o') ? 'x' : c;
        patterns_.push_back(f);
    }
    const size_t n2 = patterns_.size();
    for (size_t i = 0; i < n2; ++i) {
        const char enemy = patterns_[i].favour == 1 ? 'o' : 'x';
        const size_t a = patterns_[i].rich.find_first_of(enemy), b = patterns_[i].rich.find_last_of(enemy);
        if (a == std::string::npos) continue;
        PatternRec e = patterns_[i];
        e.rich[a] = '?';
        patterns_.push_back(e);               // left enemy stone is the board edge
        if (b != a) {
            e.rich[b] = '?';
            patterns_.push_back(e);           // both
            e.rich[a] = enemy;
            patterns_.push_back(e);           // right only
        }
    }
}

// Key = digits 1..4 in base 4, left-aligned to kMaxLen (ACAutomata.cpp:66-90).  Keys can tie; the
// toolchain's std::sort decides the order of ties, as it does in a g++ build of the reference.
void PatternAutomaton::sort_patterns() {
    const int n = static_cast<int>(patterns_.size());
    std::vector<int> key(n), order(n);
    for (int i = 0; i < n; ++i) {
        int k = 0;
        for (char c : patterns_[i].rich) k = k * kSymbols + symbol_code(c);
        key[i] = k << (2 * (kMaxLen - static_cast<int>(patterns_[i].rich.size())));
    }
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&key](int l, int r) { return key[l] < key[r]; });
    std::vector<PatternRec> sorted;
    sorted.reserve(n);
    for (int i : order) sorted.push_back(patterns_[i]);
    patterns_.swap(sorted);
}

// Interval-labelled trie (ACAutomata.cpp:105-134): a node is (code, depth, [first,last)) where the
// interval is the range of pattern indices below it; nodes live in one ordered container keyed by
// (depth, first), and a node's children are the nodes one level down whose `first` falls in its
// interval.  Inserting patterns out of lexicographic order makes a new key collide with an existing
// node, and the existing node is then used -- kept on purpose (see the file comment).
void PatternAutomaton::build_trie() {
    using Key = std::pair<int, int>;
    std::map<Key, TrieNode> nodes;
    nodes[{0, 0}] = TrieNode{0, 0, 0, 0};

    auto child_range = [&nodes](const TrieNode& p) {
        return std::make_pair(nodes.lower_bound({p.depth + 1, p.first}), nodes.upper_bound({p.depth + 1, p.last - 1}));
    };

    for (const PatternRec& pat : patterns_) {
        std::vector<Key> path{{0, 0}};
        for (char ch : pat.rich) {
            TrieNode& parent = nodes[path.back()];
            const int code = symbol_code(ch);
            auto [lo, hi] = child_range(parent);
            auto it = std::find_if(lo, hi, [code](const auto& kv) { return kv.second.code == code; });
            Key k;
            if (it != hi) {
                k = it->first;
            } else {
                k = {parent.depth + 1, parent.last};
                nodes.insert({k, TrieNode{code, parent.depth + 1, parent.last, parent.last}});   // keeps an existing node
            }
            path.push_back(k);
        }
        TrieNode& tail = nodes[path.back()];
        tail.last += 1;
        nodes.insert({{tail.depth + 1, tail.first}, TrieNode{0, tail.depth + 1, tail.first, tail.last}});   // end marker
        for (size_t i = path.size() - 1; i > 0; --i) nodes[path[i - 1]].last = nodes[path[i]].last;
    }
    trie_.clear();
    for (const auto& kv : nodes) trie_.push_back(kv.second);
}

// Double-array placement with a doubly linked free list threaded through base/check
// (ACAutomata.cpp:158-229): base < 0 on a free slot is minus its predecessor, check < 0 minus its successor.
void PatternAutomaton::build_double_array() {
    auto lower = [this](int depth, int first) {
        return static_cast<int>(std::lower_bound(trie_.begin(), trie_.end(), std::make_pair(depth, first),
            [](const TrieNode& n, const std::pair<int, int>& k) { return std::make_pair(n.depth, n.first) < k; }) - trie_.begin());
    };
    auto upper = [this](int depth, int first) {
        return static_cast<int>(std::upper_bound(trie_.begin(), trie_.end(), std::make_pair(depth, first),
            [](const std::pair<int, int>& k, const TrieNode& n) { return k < std::make_pair(n.depth, n.first); }) - trie_.begin());
    };
    base_.assign(1, 0);
    check_.assign(1, -1);

    struct Frame { int slot, node; };
    // depth-first, children in container order, each child's subtree finished before its sibling's
    std::vector<Frame> stack{{0, lower(0, 0)}};
    while (!stack.empty()) {
        const Frame f = stack.back();
        stack.pop_back();
        const TrieNode& nd = trie_[f.node];
        if (nd.depth > 0 && nd.code == 0) { base_[f.slot] = -nd.first; continue; }
        const int lo = lower(nd.depth + 1, nd.first), hi = upper(nd.depth + 1, nd.last - 1);
        int begin = 0, front = 0;
        bool fits;
        do {
            front = -check_[front];
            begin = front - trie_[lo].code;
            if (begin >= 0) {
                while (begin + kSymbols + 1 >= static_cast<int>(check_.size())) {
                    const int old = static_cast<int>(base_.size());
                    base_.resize(2 * old);
                    check_.resize(2 * old);
                    for (int i = old; i < 2 * old; ++i) { base_[i] = -(i - 1); check_[i] = -(i + 1); }
                }
            }
            fits = true;
            for (int c = lo; c < hi && fits; ++c) {
                const int slot = begin + trie_[c].code;
                fits = slot != 0 && check_[slot] < 0;
            }
        } while (!fits);
        for (int c = lo; c < hi; ++c) {
            const int slot = begin + trie_[c].code;
            check_[-base_[slot]] = check_[slot];
            base_[-check_[slot]] = base_[slot];
            check_[slot] = f.slot;
        }
        base_[f.slot] = begin;
        for (int c = hi - 1; c >= lo; --c) stack.push_back({begin + trie_[c].code, c});   // reversed: LIFO
    }
}

// Fail links by BFS, plus per symbol the state that loops onto itself (ACAutomata.cpp:231-274).
void PatternAutomaton::build_fail_links() {
    fail_.assign(base_.size(), 0);
    invariants_.assign(kSymbols + 1, 0);
    std::deque<int> queue{0};
    while (!queue.empty()) {
        const int cur = queue.front();
        queue.pop_front();
        for (int code = 1; code <= kSymbols; ++code)
            if (check_[base_[cur] + code] == cur) queue.push_back(base_[cur] + code);
        if (cur == 0) continue;
        const int code = cur - base_[check_[cur]];
        for (int up = check_[cur]; up != 0;) {
            up = fail_[up];
            const int cand = base_[up] + code;
            if (check_[cand] == up) { fail_[cur] = cand; break; }
        }
        if (check_[base_[cur] + code] != cur && base_[fail_[cur]] + code == cur) invariants_[code] = cur;
    }
}

// Dense DFA + emission lists.  One transition = everything PatternSearch::generator::operator++
// (Pattern.cpp:33-56) does between two consumed symbols: fall back along fail links (a pattern-final
// state met on the way is reported with the PREVIOUS symbol's offset), then consume.
void PatternAutomaton::flatten() {
    struct Step { int next; std::vector<Emission> emits; };
    auto step = [this](int s, int code) {
        Step r{0, {}};
        int cur = s;
        for (;;) {
            if (cur == invariants_[code]) {                    // run of `code`: stays put (Pattern.cpp:40-45)
                if (terminal(cur)) r.emits.push_back({static_cast<uint16_t>(pattern_of(cur)), 0});
                r.next = cur;
                return r;
            }
            const int nxt = base_[cur] + code;
            if (check_[nxt] == cur) {
                cur = nxt;
                if (terminal(cur)) r.emits.push_back({static_cast<uint16_t>(pattern_of(cur)), 0});
                r.next = cur;
                return r;
            }
            if (cur == 0) { r.next = 0; return r; }
            cur = fail_[cur];
            if (terminal(cur)) r.emits.push_back({static_cast<uint16_t>(pattern_of(cur)), 1});
        }
    };

    std::map<int, int> dense{{0, 0}};
    std::vector<int> order{0};
    std::vector<Step> steps;
    for (size_t i = 0; i < order.size(); ++i)
        for (int code = 1; code <= kSymbols; ++code) {
            Step st = step(order[i], code);
            if (dense.emplace(st.next, static_cast<int>(order.size())).second) order.push_back(st.next);
            steps.push_back(std::move(st));
        }

    dev_ = DeviceTables{};
    dev_.n_states = static_cast<int>(order.size());
    dev_.n_patterns = static_cast<int>(patterns_.size());
    if (dev_.n_states > 1024) throw std::runtime_error("pattern automaton: too many DFA states for 10-bit ids");
    dev_.emit_lists.push_back(0);                               // index 0 = empty list
    std::map<std::vector<uint16_t>, int> list_index;
    dev_.trans.resize(static_cast<size_t>(dev_.n_states) * 4);
    for (size_t i = 0; i < steps.size(); ++i) {
        uint32_t word = static_cast<uint32_t>(dense[steps[i].next]);
        if (!steps[i].emits.empty()) {
            std::vector<uint16_t> enc;
            for (const Emission& e : steps[i].emits) enc.push_back(static_cast<uint16_t>(e.pattern | (e.back << 15)));
            auto it = list_index.find(enc);
            if (it == list_index.end()) {
                it = list_index.emplace(enc, static_cast<int>(dev_.emit_lists.size())).first;
                dev_.emit_lists.push_back(static_cast<uint16_t>(enc.size()));
                dev_.emit_lists.insert(dev_.emit_lists.end(), enc.begin(), enc.end());
            }
            if (it->second >= 1024) throw std::runtime_error("pattern automaton: emission table overflow");
            word |= static_cast<uint32_t>(it->second) << 10;
            dev_.max_emissions = std::max<int>(dev_.max_emissions, static_cast<int>(enc.size()));
        }
        dev_.trans[i] = word;
    }

    dev_.pattern_info.resize(static_cast<size_t>(dev_.n_patterns) * 2);
    for (int p = 0; p < dev_.n_patterns; ++p) {
        const PatternRec& pr = patterns_[p];
        const int len = static_cast<int>(pr.rich.size());
        uint32_t w0 = static_cast<uint32_t>(pr.type) | (pr.favour == 1 ? 16u : 0u) | (static_cast<uint32_t>(len) << 5);
        if (pr.type != Five)
            for (int j = 0; j < len; ++j) {
                const char piece = pr.rich[len - 1 - j];
                const uint32_t kind = piece == '_' ? 1u : piece == '^' ? 2u : 0u;
                w0 |= kind << (8 + 2 * j);
            }
        // Pattern.cpp:151-152: int(1.2 * score) on diagonals, evaluated in double like the reference
        const uint32_t diag = static_cast<uint32_t>(static_cast<int>(1.2 * pr.score));
        dev_.pattern_info[2 * p] = w0;
        dev_.pattern_info[2 * p + 1] = static_cast<uint32_t>(pr.score) | (diag << 16);
    }

    // device form: dense record numbers instead of list offsets, next-row byte offsets instead of state ids
    if (dev_.max_emissions > 2) throw std::runtime_error("pattern automaton: a transition reports more than two matches");
    std::map<int, int> record_of_list;                           // emit_lists offset -> record number
    dev_.dev_records.assign(4, 0u);                              // record 0 = nothing
    dev_.dev_trans.resize(dev_.trans.size());
    for (size_t i = 0; i < dev_.trans.size(); ++i) {
        const uint32_t next = dev_.trans[i] & 1023u, list = dev_.trans[i] >> 10;
        uint32_t rec = 0;
        if (list) {
            auto it = record_of_list.find(static_cast<int>(list));
            if (it == record_of_list.end()) {
                it = record_of_list.emplace(static_cast<int>(list), static_cast<int>(dev_.dev_records.size() / 4)).first;
                uint32_t words[4] = {0, 0, 0, 0};
                for (int e = 0; e < dev_.emit_lists[list]; ++e) {
                    const uint16_t v = dev_.emit_lists[list + 1 + e];
                    const PatternRec& pr = patterns_[v & 0x7FFF];
                    const int len = static_cast<int>(pr.rich.size());
                    uint32_t w0 = static_cast<uint32_t>(pr.type) | (pr.favour == 1 ? 16u : 0u) | (static_cast<uint32_t>(len) << 5);
                    int n_dep = 0;
                    if (pr.type != Five)
                        for (int j = 0; j < len; ++j) {
                            const char piece = pr.rich[len - 1 - j];
                            if (piece != '_' && piece != '^') continue;
                            if (n_dep == 4) throw std::runtime_error("pattern automaton: a pattern has more than four scored blanks");
                            w0 |= (static_cast<uint32_t>(j) | (piece == '_' ? 8u : 0u)) << (11 + 4 * n_dep);
                            ++n_dep;
                        }
                    w0 |= static_cast<uint32_t>(n_dep) << 8;
                    if (v >> 15) w0 |= 1u << 27;
                    words[2 * e] = w0;
                    words[2 * e + 1] = dev_.pattern_info[2 * (v & 0x7FFF) + 1];
                }
                dev_.dev_records.insert(dev_.dev_records.end(), words, words + 4);
            }
            rec = static_cast<uint32_t>(it->second);
        }
        if (rec >= 1024) throw std::runtime_error("pattern automaton: too many emission records");
        uint32_t kinds = 0;                                       // which compound-feeding types the record holds
        for (int e = 0; list && e < dev_.emit_lists[list]; ++e) {
            const int type = patterns_[dev_.emit_lists[list + 1 + e] & 0x7FFF].type;
            kinds |= type == LiveThree ? 1u : type == DeadThree ? 2u : type == LiveTwo ? 4u : 0u;
        }
        dev_.dev_trans[i] = (next * 16u) | (kinds << 14) | (rec << 17);
    }
    dev_.n_records = static_cast<int>(dev_.dev_records.size() / 4);
    dev_.dev_prefix4.assign(256, 0);
    for (uint32_t idx = 0; idx < 256; ++idx) {
        uint32_t row = 0;                                          // byte offset of the root's row
        for (int i = 0; i < 4; ++i) row = dev_trans_row(dev_.dev_trans[row / 4 + ((idx >> (2 * i)) & 3u)]);
        dev_.dev_prefix4[idx] = static_cast<uint16_t>(row);
    }

    // How many symbols until the automaton has forgotten where it started: the image of the full state set under every
    // string of that length is a single state.  The incremental evaluator's window matcher (evalstate_device.h) starts
    // lanes in the middle of a window and relies on this being <= 7.
    {
        const int n = dev_.n_states;
        std::vector<std::vector<uint32_t>> frontier(1), next_frontier;
        frontier[0].resize(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) frontier[0][static_cast<size_t>(i)] = static_cast<uint32_t>(i);
        int length = 0;
        while (!frontier.empty()) {
            if (++length > 12) throw std::runtime_error("pattern automaton: does not forget its start state within 12 symbols");
            next_frontier.clear();
            std::vector<char> seen(static_cast<size_t>(n));
            for (const std::vector<uint32_t>& set : frontier)
                for (uint32_t sym = 0; sym < 4; ++sym) {
                    std::fill(seen.begin(), seen.end(), 0);
                    std::vector<uint32_t> image;
                    for (uint32_t st : set) {
                        const uint32_t to = dev_.trans[st * 4 + sym] & 1023u;
                        if (!seen[to]) { seen[to] = 1; image.push_back(to); }
                    }
                    if (image.size() > 1) next_frontier.push_back(std::move(image));
                }
            frontier.swap(next_frontier);
        }
        dev_.sync_symbols = length;
        if (length > 7) throw std::runtime_error("pattern automaton: needs more than 7 symbols to forget its start state");
    }
}

std::vector<std::pair<int, int>> PatternAutomaton::scan(const uint8_t* codes, int n) const {
    std::vector<std::pair<int, int>> out;
    uint32_t state = 0;
    for (int k = 0; k < n; ++k) {
        const uint32_t w = dev_.trans[state * 4 + (codes[k] - 1)];
        state = w & 1023u;
        const uint32_t li = w >> 10;
        if (li) {
            const int cnt = dev_.emit_lists[li];
            for (int e = 0; e < cnt; ++e) {
                const uint16_t v = dev_.emit_lists[li + 1 + e];
                out.emplace_back(v & 0x7fff, k - (v >> 15));
            }
        }
    }
    return out;
}

const PatternAutomaton& production_automaton() {
    static const PatternAutomaton a;
    return a;
}

}  // namespace gmk
