// evalstate_device.h -- device functions of the incrementally maintained Evaluator (one state per game, kept in LDS,
// one wavefront per game).  Shared by K2 (evalstate_kernel.hip: apply / revert scripts) and by the pattern-guided
// search (trad_kernel.hip), which walks its tree with the same apply / revert steps.
//
// This is Evaluator::applyMove / revertMove (core/lib/src/Pattern.cpp:274-342) with EVERYTHING the reference keeps,
// including the per-cell 2-bit flag words of m_patternDist / m_compoundDist, which are saturating shift registers
// (Pattern.cpp:395-400) and therefore depend on the order moves were played in.
//
// One update = the reference's two passes
//     delta = -1 on the old position: match, compounds, patterns;  board / 7x7 block update;
//     delta = +1 on the new position: match, patterns, compounds
// where inside a pass every step is order-independent (flag shift registers, adds), so the lanes work in parallel:
// 4 lanes walk the 13-symbol windows (one per direction) through the automaton in LDS, 64 lanes = 4 directions x
// 16 result slots apply the matches, 52 lanes test the window's blanks for compounds, 49 lanes update the 7x7 block.
#pragma once
#include "capi_common.h"

namespace gmk::evs {

constexpr int kCells = 225;
// state layout, 32-bit words (the host mirrors it in gmk_evalstate_read)
constexpr int kLineWords = 96, kColBase = 20, kDiagBase = 36, kAntiBase = 65;     // line words as in eval_kernel.hip: 2 bits per cell = its DFA
                                                                                 // symbol (0 black, 1 white, 3 blank), cell p of the line at bits 2p
// The pattern distribution is [cell 0..225 (225 = the totals)][type 0..7].  The lanes of a phase read ONE type of the cells along a line
// (strides 1, 15, 16, 14 cells = 8, 120, 128, 112 words: on 64 banks a diagonal's thirteen cells share one bank, a row's or column's
// eight, an anti-diagonal's four): this is where the update's 40 % bank-conflict share comes from.  A word of skew every eight cells
// (cell * 8 + (cell >> 3) + type) halves the conflict cycles and leaves the time per update where it was, 25 index instructions per
// update dearer (profiles/r02f_k2_phases.txt): measured, not kept.
__host__ __device__ constexpr int pdist_index(int cell, int type) { return cell * 8 + type; }
constexpr int kPdistWords = pdist_index(225, 7) + 1;
// density: ONE word per (colour, cell): the count in the low half, the weight in the high half, both as int16 (|count| <= 49, |weight| <= 246: a
// 7x7 block of nibble weights; occupied cells hold -v-1).  Half the words of two int32 planes: what decides how many games fit a CU's LDS.
constexpr int oLines = 0, oScores = oLines + kLineWords, oDensity = oScores + 4 * kCells, oPdist = oDensity + 2 * kCells,
              oCdist = oPdist + kPdistWords, oRecord = oCdist + 226 * 3, oMeta = oRecord + 57;
constexpr int kStateWords = (oMeta + 4 + 3) & ~3;                                // 3996 words = 15 984 B
__host__ __device__ __forceinline__ int density_count(uint32_t w) { return static_cast<int16_t>(w & 0xFFFFu); }
__host__ __device__ __forceinline__ int density_weight_of(uint32_t w) { return static_cast<int16_t>(w >> 16); }
__host__ __device__ __forceinline__ uint32_t density_word(int count, int weight) { return (static_cast<uint32_t>(count) & 0xFFFFu) | (static_cast<uint32_t>(weight) << 16); }
// host mirror: int32 [colour][count, weight][cell] (the layout of gmk_evalstate_read) from the packed words of one state
inline void unpack_density(const uint32_t* state, int32_t* out) {
    for (int colour = 0; colour < 2; ++colour)
        for (int cell = 0; cell < kCells; ++cell) {
            const uint32_t w = state[oDensity + colour * kCells + cell];
            out[(colour * 2 + 0) * kCells + cell] = density_count(w);
            out[(colour * 2 + 1) * kCells + cell] = density_weight_of(w);
        }
}
// meta: [0] moves played, [1] player to move (+1 black, -1 white, 0 game over), [2] winner, [3] error bits
constexpr int kCompoundCap = 64;                                                 // compound components handled in one pass
constexpr int oItemCount = 0, oItems = oItemCount + 4;
constexpr int kScratchWords = oItems + kCompoundCap;                             // the component queue and its counter (the matches travel in registers)

__device__ __forceinline__ int dir_stride(int dir) { return dir == 0 ? 1 : dir == 1 ? 15 : dir == 2 ? 16 : 14; }
__device__ __forceinline__ int group2(int favour_black, int perspective_black) { return (favour_black << 1) | perspective_black; }   // Pattern.h:159-161

__device__ __forceinline__ void wave_phase_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Ctx {
    uint32_t* st;                // state in LDS
    uint32_t* scratch;           // compound component queue: count, items
    const char* trans;           // device transition words (pattern_tables.h)
    const uint4* rec;            // emission records
    const char* prefix;          // DeviceTables::dev_prefix4 (in LDS, behind the records)
    int lane;
    unsigned long long* prof = nullptr;      // profiling aid (GMK_EVS_PROFILE): cycles per phase of update_move, summed by lane 0
    int phases = 0x3F;                       // profiling aid (GMK_EVS_PHASE_MASK, K2 only): bit p runs phase p of update_move (match, compounds -, patterns -,
                                             // board + block, patterns +, compounds +); the states are wrong unless it is 0x3F
};

__device__ __forceinline__ void prof_mark(const Ctx& c, int slot, unsigned long long& t_last) {
    if (c.prof) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (c.lane == 0) c.prof[slot] += now - t_last;
        t_last = __builtin_amdgcn_s_memtime();
    }
}

// Evaluator::Record::set(delta, favour, perspective, dir) (Pattern.cpp:395-400) for one or more 2-bit fields of one LDS
// word (`lowers` = the low bit of every field to update).  The field is a saturating shift register: +1 takes
// 00 -> 01 -> 11 -> 11, -1 takes 11 -> 01 -> 00 -> 00.  Two atomics without a compare-and-swap loop do that, and
// lanes that update the same field in the same pass still compose like the reference's sequential updates:
//   +1: set the low bits; where the low bit was already set, set the high bit
//   -1: clear the high bits; where the high bit was already clear, clear the low bit
__device__ __forceinline__ uint32_t set_flags_begin(uint32_t* word, int delta, uint32_t lowers) {
    return delta == 1 ? atomicOr(word, lowers) : atomicAnd(word, ~(lowers << 1));
}
__device__ __forceinline__ void set_flags_end(uint32_t* word, int delta, uint32_t lowers, uint32_t old) {
    if (delta == 1) { const uint32_t up = (old & lowers) << 1; if (up) atomicOr(word, up); }
    else { const uint32_t down = lowers & ~(old >> 1); if (down) atomicAnd(word, ~down); }
}
__device__ __forceinline__ void set_flag(uint32_t* word, int delta, int group, int dir) {
    const uint32_t lower = 1u << (static_cast<uint32_t>(4 * group + dir) * 2u);
    set_flags_end(word, delta, lower, set_flags_begin(word, delta, lower));
}

// a line word with every cell blank
__host__ __device__ __forceinline__ uint32_t blank_line_word(int index) {
    if (index < 15 || (index >= kColBase && index < kColBase + 15)) return 0x3FFFFFFFu;
    const int d = index >= kAntiBase ? index - kAntiBase : index - kDiagBase;
    if (index < kDiagBase || d > 28) return 0u;
    const int len = 15 - (d > 14 ? d - 14 : 14 - d);
    return (1u << (2 * len)) - 1u;
}

// the 13 symbols centred on `cell` along `dir` (BoardMap::lineView, Mapping.cpp:31-34) as a 2-bit stream
__device__ __forceinline__ uint64_t window_symbols(const uint32_t* lines, int cell, int dir) {
    const int x = cell % 15, y = cell / 15, diag = x - y + 14, anti = x + y;
    const int line = dir == 0 ? y : dir == 1 ? kColBase + x : dir == 2 ? kDiagBase + diag : kAntiBase + anti;
    const int at = dir == 0 ? x : dir == 1 ? y : dir == 2 ? min(x, y) : min(14 - x, y);
    const int len = dir < 2 ? 15 : dir == 2 ? 15 - abs(diag - 14) : min(anti, 28 - anti) + 1;
    return (0xAAAull | (static_cast<uint64_t>(lines[line]) << 12) | (0xAAAull << (2 * len + 12))) >> (2 * at);       // six '?' | cells | six '?'
}

// Updater::matchPatterns (Pattern.cpp:128-136).  The reference matches twice per update, before and after the stone
// changes; the two 13-symbol windows differ in the centre symbol only and do not depend on anything else the update
// touches, so both are matched at once (window 0 = the board as it is, window 1 = `new_sym` in the centre).  Only
// transitions at window indices 6..12 can report a match that covers the centre, and the automaton forgets where it
// started after 7 symbols (checked for all 556 states x 4^7 strings when the tables are built,
// PatternAutomaton::flatten), so every such transition gets its own lane: lane = (window * 4 + direction) * 7 +
// (k - 6) starts at the root at index max(0, k - 7) and is at the right state after at most 7 lookups -- of which the first four
// (from the root, their reports unused) are ONE lookup in dev_prefix4.  A chain of 5 dependent LDS reads instead of 13.
// The matches stay in registers: a transition reports at most two, the lane keeps the first for the pattern phase of ITS window
// and hands the second to the lane of the same transition in the OTHER window (lane +- 28, two ds_bpermute), which is idle in that
// phase: every lane then has at most one match per pattern phase, and nothing goes through an LDS queue (no counter, no slots, no
// read-back).  The order in which the matches are applied does not matter (every update they feed commutes).
struct Matches {
    uint32_t w0[2], w1[2];       // [window whose pattern phase applies it]: the match (pattern_tables.h record words; w0 = 0: none),
                                 // bits 28..30 of w0 = how far the match's last symbol lies beyond the centre
};
__device__ inline Matches match_patterns_both(const Ctx& c, int move, uint32_t new_sym) {
    uint32_t first_w0 = 0, first_w1 = 0, second_w0 = 0, second_w1 = 0;
    const int wd = c.lane / 7, k = 6 + c.lane % 7, dir = wd & 3, w = (wd >> 2) & 1;
    if (c.lane < 56) {
        uint64_t syms = window_symbols(c.st + oLines, move, dir);
        if (w) syms = (syms & ~(3ull << 12)) | (static_cast<uint64_t>(new_sym) << 12);
        const int start = k > 7 ? k - 7 : 0;
        const uint32_t stream = static_cast<uint32_t>(syms >> (2 * start));
        uint32_t tw = *reinterpret_cast<const uint16_t*>(c.prefix + 2u * (stream & 0xFFu)), tw_before = 0;   // the first four symbols: one lookup (dev_prefix4)
        uint32_t cur = (stream >> 8) << 2;                        // symbol * 4 in bits 2..3, as in eval_kernel.hip
#pragma unroll
        for (int i = 0; i < 4; ++i) {                             // four more for everybody; the lane of index 6 stops after three
            if (i == 3) tw_before = tw;
            tw = *reinterpret_cast<const uint32_t*>(c.trans + (gmk::dev_trans_row(tw) | (cur & 12u)));
            cur >>= 2;
        }
        if (k == 6) tw = tw_before;
        const uint32_t rid = gmk::dev_trans_record(tw);
        if (rid) {
            const uint4 r = c.rec[rid];
            const uint32_t w0s[2] = {r.x, r.z}, w1s[2] = {r.y, r.w};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t w0 = w0s[e];
                const int back = k - static_cast<int>((w0 >> 27) & 1u) - 6, len = (w0 >> 5) & 7;      // HasCovered (Pattern.cpp:22-25)
                const bool covers = w0 != 0u && back >= 0 && back < len;
                const uint32_t kept = covers ? (w0 & 0x07FFFFFFu) | (static_cast<uint32_t>(back) << 28) : 0u;
                if (e == 0) { first_w0 = kept; first_w1 = w1s[0]; } else { second_w0 = kept; second_w1 = w1s[1]; }
            }
            if (!first_w0) { first_w0 = second_w0; first_w1 = second_w1; second_w0 = 0u; }       // (a lone second match is the lane's own)
        }
    }
    // the second match goes to the same transition's lane in the other window
    const int partner = (c.lane < 28 ? c.lane + 28 : c.lane - 28) * 4;
    const uint32_t from_w0 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(partner, static_cast<int>(second_w0)));
    const uint32_t from_w1 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(partner, static_cast<int>(second_w1)));
    Matches m;                                                    // (selects, not m.w0[w]: an array indexed per lane would live in scratch memory)
    const bool live = c.lane < 56, own_is_0 = w == 0;
    const uint32_t got_w0 = live ? from_w0 : 0u;
    m.w0[0] = own_is_0 ? first_w0 : got_w0; m.w1[0] = own_is_0 ? first_w1 : from_w1;
    m.w0[1] = own_is_0 ? got_w0 : first_w0; m.w1[1] = own_is_0 ? from_w1 : first_w1;
    return m;
}

// Updater::updatePatterns (Pattern.cpp:138-165) for the matches of window `slot`: lane = the transition that found the match (or its
// twin in the other window, for a transition's second match)
__device__ inline void update_patterns(const Ctx& c, int move, const Matches& found, int slot, int delta) {
    const uint32_t w0 = found.w0[slot], w1 = found.w1[slot];
    if (!w0) return;
    const int dir = (c.lane / 7) & 3;
    const int type = w0 & 15, fav = (w0 >> 4) & 1, back = static_cast<int>(w0 >> 28);
    int32_t* meta = reinterpret_cast<int32_t*>(c.st + oMeta);
    if (type == 8) { meta[1] = 0; meta[2] = fav ? 1 : -1; return; }                    // Five ends the game (:140-145)
    const int stride = dir_stride(dir);
    const int last_cell = move + back * stride;                                        // cell of the match's last symbol
    atomicAdd(&c.st[oPdist + pdist_index(225, type)], static_cast<uint32_t>(delta) << (16 * fav));      // Record::set(delta, favour) (:390-393)
    const int score = delta * static_cast<int>(dir >= 2 ? (w1 >> 16) : (w1 & 0xFFFFu));
    uint32_t* scores = c.st + oScores;
    const int n_dep = (w0 >> 8) & 7;
    // per scored blank: the flag fields of both views live in one word (one returning atomic starts them all), the score adds
    // need no answer; the second half of the flag updates follows once the answers are back
    // No branches (this wave has nobody to hide them behind): an unused slot (all zero bits: the match's last cell) and the
    // owner's view of a '^' piece update nothing -- OR / AND with an empty mask, add 0.
    uint32_t* words[4];
    uint32_t lowers[4], olds[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t f = (w0 >> (11 + 4 * d)) & 15u;
        const int cell = last_cell - static_cast<int>(f & 7u) * stride;
        const bool blank = (f & 8u) != 0u;
        words[d] = &c.st[oPdist + pdist_index(cell, type)];
        lowers[d] = (d < n_dep ? 1u << ((4 * group2(fav, fav ^ 1) + dir) * 2) : 0u) |  // '_' and '^': the opponent's view
                    (blank ? 1u << ((4 * group2(fav, fav) + dir) * 2) : 0u);           // '_': the owner's view too
        olds[d] = set_flags_begin(words[d], delta, lowers[d]);
        atomicAdd(&scores[group2(fav, fav) * kCells + cell], blank ? static_cast<uint32_t>(score) : 0u);
        atomicAdd(&scores[group2(fav, fav ^ 1) * kCells + cell], d < n_dep ? static_cast<uint32_t>(score) : 0u);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (delta == 1) atomicOr(words[d], (olds[d] & lowers[d]) << 1);
        else atomicAnd(words[d], ~(lowers[d] & ~(olds[d] >> 1)));
    }
}

// One compound at (cell, player): Compound::Compound / locate (Pattern.cpp:435-486) and the bookkeeping of
// Compound::update (:488-518).  update() walks the components one after the other, but what it does for component i
// only depends on i and on the count the compound had before (count_i = count + i * delta): the components are
// queued here and handled side by side by apply_compound_items().
// Item: cell | black << 8 | compound type << 9 | direction << 11 | component type (0 L3, 1 D3, 2 L2) << 13 |
//       look for counter moves << 15 | bump the compound total << 16
__device__ inline void queue_compound(const Ctx& c, int cell, int pb /* player is black */, int delta) {
    const uint32_t* pd = c.st + oPdist + pdist_index(cell, 0);
    const int g_own = group2(pb, pb);
    // locate() walks the directions in order through the state machine S0, L2, LD3, To33, To43, To44 (Pattern.cpp:440-486); per direction
    // the component is the first present of LiveThree, DeadThree, LiveTwo, taken once or twice (a 2-bit flag field counts 0, 1, 2-or-more
    // as 00, 01, 11; 10 does not occur and counts as nothing, Pattern.cpp:457-462).  Its outcome does not depend on the order: with n
    // components of which s are threes (weight 2, a LiveTwo 1) the state ends at 3 + min(2, s) for n >= 2 -- the first two transitions add
    // w1 + w2 + 1, every further one w - 1, capped at 5 -- and "triple" is n >= 3.  So: masks over the four 2-bit fields, no loops
    // (the wavefront pays this code's full length for a handful of lanes).
    const uint32_t f_l3 = (pd[5] >> (8 * g_own)) & 0xFFu, f_d3 = (pd[4] >> (8 * g_own)) & 0xFFu, f_l2 = (pd[3] >> (8 * g_own)) & 0xFFu;
    const uint32_t sel3 = f_l3 & 0x55u, seld = f_d3 & 0x55u & ~sel3, sel2 = f_l2 & 0x55u & ~(sel3 | seld);       // the direction's component: bit 2 d
    const uint32_t strong = sel3 | seld, any_dir = strong | sel2;
    const uint32_t twice_strong = ((f_l3 >> 1) & sel3) | ((f_d3 >> 1) & seld), twice = twice_strong | ((f_l2 >> 1) & sel2);
    const int n_comp = __popc(any_dir) + __popc(twice);
    const int l3 = sel3 != 0u, triple = n_comp >= 3;
    if (n_comp < 2) { c.st[oMeta + 3] |= 2u; return; }                                 // the reference indexes out of bounds here
    const int ctype = min(__popc(strong) + __popc(twice_strong), 2);
    // the components in direction order, one nibble each (dir | (0 L3, 1 D3, 2 L2) << 2), a component taken twice twice in a row
    uint32_t comps = 0;
    {
        int at = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t bit = 1u << (2 * d);
            const uint32_t code = static_cast<uint32_t>(d) | ((sel3 & bit) ? 0u : (seld & bit) ? 4u : 8u);
            const uint32_t once = (any_dir >> (2 * d)) & 1u, again = (twice >> (2 * d)) & 1u;
            comps |= (code * once) << (4 * at);
            at += static_cast<int>(once);
            comps |= (code * again) << (4 * at);
            at += static_cast<int>(again);
        }
    }
    const int count0 = __popc((c.st[oCdist + cell * 3 + ctype] >> (8 * g_own)) & 0xFFu);
    const int todo = delta == 1 ? n_comp : min(n_comp, count0);                        // update(-1) stops at "2 * count + delta == -1", i.e. at count 0
    if (todo == 0) return;
    const uint32_t slot0 = atomicAdd(&c.scratch[oItemCount], static_cast<uint32_t>(todo));
    for (int i = 0; i < todo; ++i) {
        const int count_i = count0 + i * delta;
        const uint32_t item = static_cast<uint32_t>(cell) | (static_cast<uint32_t>(pb) << 8) | (static_cast<uint32_t>(ctype) << 9) |
                              (((comps >> (4 * i)) & 15u) << 11) | ((!triple && l3 == 0) ? 1u << 15 : 0u) | ((2 * count_i + delta == 3) ? 1u << 16 : 0u);
        if (slot0 + i < static_cast<uint32_t>(kCompoundCap)) c.scratch[oItems + slot0 + i] = item;
        else c.st[oMeta + 3] |= 8u;
    }
}

// The queued components: updateCritical and the compound total by one lane per component, then updateAntis
// (Pattern.cpp:488-550).  updateAntis wants the FIRST match of the component's type that runs through the cell with a
// blank there, scanning the 13-symbol window from the left.  Such a match ends at window index 6..12, so eight lanes
// share a component: lane kk looks at the transition at index 6 + kk only (its state is right after <= 7 lookups from
// the root, see match_patterns_both), and the lowest lane with a hit applies it.
__device__ inline void apply_compound_items(const Ctx& c, int delta) {
    const int n = min(static_cast<int>(c.scratch[oItemCount]), kCompoundCap);
    uint32_t* cd = c.st + oCdist;
    uint32_t* scores = c.st + oScores;
    if (c.lane < n) {
        const uint32_t item = c.scratch[oItems + c.lane];
        const int cell = item & 255, pb = (item >> 8) & 1, ctype = (item >> 9) & 3, cdir = (item >> 11) & 3;
        const int g_own = group2(pb, pb), g_opp = group2(pb, pb ^ 1);
        // updateCritical: the cell itself, both views (the two flag fields share a word)
        const uint32_t lowers = (1u << ((4 * g_own + cdir) * 2)) | (1u << ((4 * g_opp + cdir) * 2));
        const uint32_t old = set_flags_begin(&cd[cell * 3 + ctype], delta, lowers);
        atomicAdd(&scores[g_own * kCells + cell], static_cast<uint32_t>(delta * 600));
        atomicAdd(&scores[g_opp * kCells + cell], static_cast<uint32_t>(delta * 600));
        if (item & (1u << 16)) atomicAdd(&cd[225 * 3 + ctype], static_cast<uint32_t>(delta) << (16 * pb));
        set_flags_end(&cd[cell * 3 + ctype], delta, lowers, old);
    }
    for (int m0 = 0; m0 < n; m0 += 8) {                          // eight components per pass
        const int m = m0 + (c.lane >> 3), kk = c.lane & 7, k = 6 + kk;
        const uint32_t item = m < n ? c.scratch[oItems + m] : 0u;
        const int cell = item & 255, pb = (item >> 8) & 1, ctype = (item >> 9) & 3, cdir = (item >> 11) & 3, ct = (item >> 13) & 3;
        uint32_t hit_w0 = 0;                                      // the first qualifying match of this lane's transition
        int hit_back = 0;
        if ((item & (1u << 15)) && kk < 7) {
            const int want = ct == 0 ? 5 : ct == 1 ? 4 : 3;
            const uint64_t syms = window_symbols(c.st + oLines, cell, cdir);
            const int start = k > 7 ? k - 7 : 0;
            const uint32_t stream = static_cast<uint32_t>(syms >> (2 * start));
            uint32_t tw = *reinterpret_cast<const uint16_t*>(c.prefix + 2u * (stream & 0xFFu)), tw_before = 0;   // as in match_patterns_both
            uint32_t cur = (stream >> 8) << 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i == 3) tw_before = tw;
                tw = *reinterpret_cast<const uint32_t*>(c.trans + (gmk::dev_trans_row(tw) | (cur & 12u)));
                cur >>= 2;
            }
            if (k == 6) tw = tw_before;
            if ((gmk::dev_trans_kinds(tw) >> ct) & 1u) {
                const uint4 r = c.rec[gmk::dev_trans_record(tw)];
                const uint32_t w0s[2] = {r.x, r.z};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const uint32_t w0 = w0s[e];
                    if (hit_w0 || !w0 || static_cast<int>(w0 & 15u) != want) continue;
                    const int back = k - static_cast<int>((w0 >> 27) & 1u) - 6, len = (w0 >> 5) & 7, n_dep = (w0 >> 8) & 7;
                    if (back < 0 || back >= len) continue;
                    bool on_cell = false;
                    for (int d = 0; d < n_dep; ++d) on_cell |= ((w0 >> (11 + 4 * d)) & 15u) == (8u | static_cast<uint32_t>(back));
                    if (on_cell) { hit_w0 = w0; hit_back = back; }
                }
            }
        }
        // the scan stops at the first hit: the lowest lane of the component's eight
        const unsigned long long hits = __ballot(hit_w0 != 0u);
        const uint32_t mine = static_cast<uint32_t>(hits >> (c.lane & ~7)) & 0xFFu;
        if (hit_w0 && (mine & ((1u << kk) - 1u)) == 0u) {
            const int g_opp = group2(pb, pb ^ 1), stride = dir_stride(cdir), n_dep = (hit_w0 >> 8) & 7;
            const int last_cell = cell + hit_back * stride;
            for (int d = 0; d < n_dep; ++d) {
                const int j = (hit_w0 >> (11 + 4 * d)) & 7;
                if (j == hit_back) continue;
                const int other = last_cell - j * stride;                              // updatePose(delta, other, component, -favour)
                set_flag(&cd[other * 3 + ctype], delta, g_opp, cdir);
                atomicAdd(&scores[g_opp * kCells + other], static_cast<uint32_t>(delta * 600));
            }
        }
    }
}

// Updater::updateCompound for the four directions (Pattern.cpp:167-197): lane = direction * 13 + window index finds the
// compounds, then one lane per component applies them
__device__ inline void update_compounds(const Ctx& c, int move, int delta) {
    if (c.lane == 0) c.scratch[oItemCount] = 0u;
    wave_phase_fence();
    if (c.lane < 52) {
        const int dir = c.lane / 13, i = c.lane % 13;
        const uint64_t syms = window_symbols(c.st + oLines, move, dir);
        // only blanks; the centre lies on all four lines and is handled once (findCompound)
        if (!(i == 6 && dir != 0) && ((syms >> (2 * i)) & 3u) == 3u) {
            const int cell = move + (i - 6) * dir_stride(dir);
            const uint32_t* density = c.st + oDensity;
            const uint32_t* pd = c.st + oPdist + pdist_index(cell, 0);
            const uint32_t any = pd[5] | pd[4] | pd[3];
            for (int pb = 0; pb < 2; ++pb) {                    // { White, Black }
                if (density_count(density[pb * kCells + cell]) < 2) continue;
                const uint32_t bits = (any >> (8 * group2(pb, pb))) & 0xFFu;           // Compound::Test (Pattern.cpp:424-433)
                if (!(bits & (bits - 1u))) continue;
                queue_compound(c, cell, pb, delta);
            }
        }
    }
    wave_phase_fence();
    apply_compound_items(c, delta);
}

// Updater::updateBlock (Pattern.cpp:236-272): lanes 0..48 = the 7x7 block around the move
__device__ inline void update_block(const Ctx& c, int move, int delta, int src_black) {
    if (c.lane >= 49) return;
    constexpr uint32_t kW[7] = {0x2001002u, 0x0433340u, 0x0354530u, 0x1340431u, 0x0354530u, 0x0433340u, 0x2001002u};   // rows of Pattern.cpp:601-607, one nibble per column
    const int dy = c.lane / 7 - 3, dx = c.lane % 7 - 3;
    const int x = move % 15 + dx, y = move / 15 + dy;
    if (static_cast<unsigned>(x) >= 15u || static_cast<unsigned>(y) >= 15u) return;
    const int q = y * 15 + x, w = static_cast<int>((kW[dy + 3] >> (4 * (6 - (dx + 3)))) & 15u);
    uint32_t* density = c.st + oDensity;
    int32_t* scores = reinterpret_cast<int32_t*>(c.st + oScores);
    uint32_t* mine = density + src_black * kCells + q;          // count | weight << 16 of the mover's colour at q
    int32_t* score = scores + group2(src_black, src_black) * kCells;
    const bool centre = q == move;
    uint32_t* other = density + (src_black ^ 1) * kCells + move;                       // the other colour's entry of the centre cell
    int32_t* o_score = scores + group2(src_black ^ 1, src_black ^ 1) * kCells + move;
    // everything is read first (one LDS round trip), updated in registers in the reference's order, then stored
    const uint32_t mw = *mine, ow = centre ? *other : 0u;
    int wv = density_weight_of(mw), cv = density_count(mw), sv = score[q];
    int ocv = density_count(ow), owv = density_weight_of(ow), osv = centre ? *o_score : 0;
    const int before = wv > 0;
    wv += (wv < 0 ? -1 : 1) * delta * w;
    cv += (wv < 0 ? -1 : 1) * delta * (w > 0 ? 1 : 0);          // the sign of the UPDATED weight (a lazily evaluated expression in the reference)
    if (centre) {                                               // both colours, count and weight: occupied cells hold -v-1
        wv = delta == 1 ? -wv - 1 : -(wv + 1);
        cv = delta == 1 ? -cv - 1 : -(cv + 1);
        owv = delta == 1 ? -owv - 1 : -(owv + 1);
        ocv = delta == 1 ? -ocv - 1 : -(ocv + 1);
    }
    sv += 160 * ((wv > 0) - before);
    if (centre && ocv != 0 && ocv != -1) osv -= delta * 160;
    *mine = density_word(cv, wv); score[q] = sv;
    if (centre) { *other = density_word(ocv, owv); *o_score = osv; }
}

// BoardMap::applyMove / revertMove (Mapping.cpp:37-59): lanes 0..3, one line word each.  Placing a stone turns the cell's
// blank (3) into black (0) or white (1); taking it back is the same XOR.
__device__ inline void board_set(const Ctx& c, int move, int black) {
    if (c.lane >= 4) return;
    const int x = move % 15, y = move / 15, k = c.lane;
    const int at = k == 0 ? x : k == 1 ? y : k == 2 ? min(x, y) : min(14 - x, y);
    const int idx = k == 0 ? y : k == 1 ? kColBase + x : k == 2 ? kDiagBase + x - y + 14 : kAntiBase + x + y;
    atomicXor(&c.st[oLines + idx], (black ? 3u : 2u) << (2 * at));
}

// Updater::updateMove (Pattern.cpp:274-302).  src: +1 / -1 = the mover of an applied move, 0 = revert the last move
__device__ inline void update_move(const Ctx& c, int move, int src) {
    int32_t* meta = reinterpret_cast<int32_t*>(c.st + oMeta);
    uint8_t* record = reinterpret_cast<uint8_t*>(c.st + oRecord);
    unsigned long long t_last = c.prof ? __builtin_amdgcn_s_memtime() : 0ull;
    // symbol the centre takes: the mover's stone (0 black, 1 white) when a move is applied, blank (3) when it is taken back
    Matches found{};
    if (c.phases & 1) found = match_patterns_both(c, move, src != 0 ? (src > 0 ? 0u : 1u) : 3u);
    wave_phase_fence();
    prof_mark(c, 0, t_last);
    if (c.phases & 2) update_compounds(c, move, -1);
    wave_phase_fence();
    prof_mark(c, 1, t_last);
    if (c.phases & 4) update_patterns(c, move, found, 0, -1);
    wave_phase_fence();
    prof_mark(c, 2, t_last);
    // the stone itself (line words, move record, player to move) and the 7x7 block touch different words: one phase
    int block_colour;
    if (src != 0) {
        board_set(c, move, src > 0);
        if (c.lane == 0) { record[meta[0]] = static_cast<uint8_t>(move); meta[0] += 1; meta[1] = -src; }          // Board::applyMove(move, false)
        block_colour = src > 0;
    } else {
        const int n = meta[0];
        int cur = meta[1];
        if (cur == 0) cur = (n % 2 == 0) ? 1 : -1;              // Board::revertMove: back from a finished game (Game.cpp:51-54)
        const int mover = -cur;                                 // the player whose stone is taken back
        board_set(c, move, mover > 0);
        if (c.lane == 0) { meta[0] = n - 1; meta[1] = mover; meta[2] = 0; }
        block_colour = mover > 0;
    }
    if (c.phases & 8) update_block(c, move, src != 0 ? 1 : -1, block_colour);
    wave_phase_fence();
    prof_mark(c, 3, t_last);
    if (c.phases & 16) update_patterns(c, move, found, 1, 1);
    wave_phase_fence();
    prof_mark(c, 5, t_last);
    if (c.phases & 32) update_compounds(c, move, 1);
    wave_phase_fence();
    prof_mark(c, 6, t_last);
    if (c.prof && c.lane == 0) c.prof[7] += 1;
}

// Evaluator::reset (Pattern.cpp:370-386, Game.cpp:138-146): empty board, black to move
__device__ __forceinline__ void reset_state(const Ctx& c) {
    for (int i = c.lane; i < kStateWords; i += 64) c.st[i] = i < kLineWords ? blank_line_word(i) : 0u;
    wave_phase_fence();
    if (c.lane == 0) c.st[oMeta + 1] = 1u;
    wave_phase_fence();
}

// One step of an evaluator script: Evaluator::applyMove(mv) for mv >= 0 (Pattern.cpp:310-313: ignored when the game is
// over or the cell is taken), Evaluator::revertMove(1) for mv == kRevert (Pattern.cpp:337-342), nothing otherwise.
// update_move is a lot of code (~20 KB): every kernel calls it through this one function, from ONE place in a loop, so
// that it exists once and the kernel stays inside the instruction cache.
constexpr int kRevert = -2;
__device__ __forceinline__ void evaluator_step(const Ctx& c, int mv) {
    const int32_t* meta = reinterpret_cast<const int32_t*>(c.st + oMeta);
    const uint8_t* record = reinterpret_cast<const uint8_t*>(c.st + oRecord);
    const int n = meta[0], cur = meta[1];
    int move = mv, src = cur;
    bool go;
    if (mv == kRevert) { go = n > 0; move = go ? record[n - 1] : 0; src = 0; }
    else go = mv >= 0 && mv < kCells && cur != 0 && ((c.st[oLines + mv / 15] >> (2 * (mv % 15))) & 3u) == 3u;
    if (go) update_move(c, move, src);
}

// the state of a fresh Evaluator, for the host-side resets
inline void fill_initial_state(uint32_t* st) {
    for (int i = 0; i < kStateWords; ++i) st[i] = i < kLineWords ? blank_line_word(i) : 0u;
    st[oMeta + 1] = 1u;                                          // black to move (Board::reset, Game.cpp:138-146)
}

}  // namespace gmk::evs
