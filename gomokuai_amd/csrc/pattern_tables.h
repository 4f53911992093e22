// pattern_tables.h -- host-side construction of the line-pattern automaton and its GPU form.
//
// Replaces, for the MI355X path, the reference's static `Evaluator::Patterns`
// (core/lib/src/Pattern.cpp:554-596) built by AhoCorasickBuilder
// (core/lib/src/utils/ACAutomata.cpp:15-274).  The automaton is constructed once on the host with the
// reference's algorithm (so that its quirks survive, see build_trie()), then flattened into a dense
// DFA with per-transition emission lists: one LDS lookup per symbol on the GPU, no fail-chain walks.
#pragma once
#include <cstdint>
#if defined(__HIP__)
#include <hip/hip_runtime.h>
#endif
#include <string>
#include <vector>

namespace gmk {

enum PatternType : int { DeadOne, LiveOne, DeadTwo, LiveTwo, DeadThree, LiveThree, DeadFour, LiveFour, Five, NumPatternTypes };

struct PatternRec {
    std::string rich;      // 'x','o','?','_','^','~'   (rich blanks keep their scoring role)
    int favour;            // +1 black, -1 white
    int type;
    int score;
};

// Symbol codes of the reference (Mapping.h:40-48): x=1 o=2 ?=3 blank=4.  The GPU uses code-1 (0..3).
inline int symbol_code(char ch) {
    switch (ch) { case 'x': return 1; case 'o': return 2; case '?': return 3;
                  case '-': case '_': case '^': case '~': return 4; default: return 0; }
}

// One emission of the flattened matcher: pattern id, and whether it ends on the symbol consumed by
// this transition (back = 0) or on the previous one (back = 1, produced while falling back).
struct Emission { uint16_t pattern; uint8_t back; };

// decoders of a dev_trans word (host and device)
#if defined(__HIP__)
#define GMK_TABLE_FN __host__ __device__ __forceinline__
#else
#define GMK_TABLE_FN inline
#endif
GMK_TABLE_FN uint32_t dev_trans_row(uint32_t tw) { return tw & 0x3FFFu; }
GMK_TABLE_FN uint32_t dev_trans_kinds(uint32_t tw) { return (tw >> 14) & 7u; }
GMK_TABLE_FN uint32_t dev_trans_record(uint32_t tw) { return tw >> 17; }
constexpr int kPrefixWords = 128;       // DeviceTables::dev_prefix4: 256 entries of 16 bits

struct DeviceTables {
    // trans[state*4 + sym] : bits 0..9 next state, bits 10..19 index into emit_lists (0 = nothing)
    std::vector<uint32_t> trans;
    // emit_lists: at index i: count, then count entries of (pattern | back<<15)
    std::vector<uint16_t> emit_lists;
    // per pattern, 2 words:
    //   w0: bits 0..3 type | bit 4 favour-is-black | bits 5..7 len | bits 8..21 piece kinds, 2 bits per piece
    //       counted from the LAST symbol backwards (0 none, 1 '_', 2 '^')
    //   w1: bits 0..15 score on rows/columns | bits 16..31 score on diagonals (= int(1.2*score))
    std::vector<uint32_t> pattern_info;
    // ---- what the kernels stage into LDS ----
    // dev_trans[state*4 + sym]: bits 0..13 byte offset of the next state's row (state * 16), bits 14..16 the record holds a
    //   LiveThree / DeadThree / LiveTwo match (filter for the compound rescans), bits 17..26 emission record number (0 = none),
    //   nothing above it (the record number is one shift away); decoders: dev_trans_row / _kinds / _record below
    std::vector<uint32_t> dev_trans;
    // dev_records[4*r .. 4*r+3], r >= 1: the (at most two) matches one transition reports, two words each, second pair 0 if absent:
    //   w0: bits 0..3 type | bit 4 favour-is-black | bits 5..7 len | bits 8..10 number of deposits |
    //       bits 11..26 four deposits of 4 bits (bits 0..2 piece index counted from the LAST symbol, bit 3 set for '_', clear for '^') |
    //       bit 27 the match ends one symbol before the one just consumed
    //   w1: as pattern_info w1
    std::vector<uint32_t> dev_records;
    int n_records = 0;
    // dev_prefix4[s0 | s1 << 2 | s2 << 4 | s3 << 6]: byte offset of the row of the state the ROOT reaches over the symbols s0 s1 s2 s3 (the
    //   value dev_trans_row gives after four lookups).  The incremental evaluator's window matcher starts every lane at the root and only
    //   uses what its LAST transition reports, so the first four of its dependent lookups are one lookup here (evalstate_device.h).
    //   Uploaded behind the records (kPrefixWords 32-bit words).
    std::vector<uint16_t> dev_prefix4;
    int sync_symbols = 0;                // after this many symbols the state no longer depends on the start state (7 for the production table)
    int n_states = 0;
    int n_patterns = 0;
    int max_emissions = 0;
};

class PatternAutomaton {
public:
    PatternAutomaton();                                  // production table (Pattern.cpp:554-596)
    explicit PatternAutomaton(const std::vector<PatternRec>& protos);

    const std::vector<PatternRec>& patterns() const { return patterns_; }
    const std::vector<int>& base() const { return base_; }
    const std::vector<int>& check() const { return check_; }
    const std::vector<int>& fail() const { return fail_; }
    const std::vector<int>& invariants() const { return invariants_; }
    const DeviceTables& device() const { return dev_; }

    // Walks the flattened DFA over `codes` (values 1..4) exactly as the kernel does and returns the
    // (pattern, end offset) stream ordered by end offset.  Host-side self-check of the tables only.
    std::vector<std::pair<int, int>> scan(const uint8_t* codes, int n) const;

private:
    void augment();
    void sort_patterns();
    void build_trie();
    void build_double_array();
    void build_fail_links();
    void flatten();
    bool terminal(int s) const { return check_[base_[s]] == s; }
    int pattern_of(int s) const { return -base_[base_[s]]; }

    struct TrieNode { int code, depth, first, last; };
    std::vector<TrieNode> trie_;                         // ordered by (depth, first)
    std::vector<PatternRec> patterns_;
    std::vector<int> base_, check_, fail_, invariants_;
    DeviceTables dev_;
};

const PatternAutomaton& production_automaton();

}  // namespace gmk
