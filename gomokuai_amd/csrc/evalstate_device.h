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
constexpr int kLineWords = 96, kColBase = 20, kDiagBase = 36, kAntiBase = 65;     // line words as in eval_kernel.hip
constexpr int oLines = 0, oScores = oLines + kLineWords, oDensity = oScores + 4 * kCells, oPdist = oDensity + 4 * kCells,
              oCdist = oPdist + 226 * 8, oRecord = oCdist + 226 * 3, oMeta = oRecord + 57;
constexpr int kStateWords = (oMeta + 4 + 3) & ~3;                                // 4448 words = 17 792 B
// meta: [0] moves played, [1] player to move (+1 black, -1 white, 0 game over), [2] winner, [3] error bits
constexpr int kResultCap = 16;                                                   // matches covering the centre, per direction
constexpr int kScratchWords = 2 * 4 * kResultCap * 2 + 8;                        // two result sets + counters

__device__ __forceinline__ int dir_stride(int dir) { return dir == 0 ? 1 : dir == 1 ? 15 : dir == 2 ? 16 : 14; }
__device__ __forceinline__ int group2(int favour_black, int perspective_black) { return (favour_black << 1) | perspective_black; }   // Pattern.h:159-161

__device__ __forceinline__ uint32_t spread_bits(uint32_t v) {
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

__device__ __forceinline__ void wave_phase_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Ctx {
    uint32_t* st;                // state in LDS
    uint32_t* scratch;           // results[2][4][kResultCap][2] + count[2][4]
    const char* trans;           // device transition words (pattern_tables.h)
    const uint4* rec;            // emission records
    int lane;
    unsigned long long* prof = nullptr;      // profiling aid (GMK_EVS_PROFILE): cycles per phase of update_move, summed by lane 0
};

__device__ __forceinline__ void prof_mark(const Ctx& c, int slot, unsigned long long& t_last) {
    if (c.prof) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (c.lane == 0) c.prof[slot] += now - t_last;
        t_last = __builtin_amdgcn_s_memtime();
    }
}

// Evaluator::Record::set(delta, favour, perspective, dir) (Pattern.cpp:395-400) as an atomic update of one LDS word
__device__ __forceinline__ void set_flag(uint32_t* word, int delta, int group, int dir) {
    const uint32_t offset = static_cast<uint32_t>(4 * group + dir) * 2u, lower = 1u << offset, higher = lower << 1, mask = lower | higher;
    uint32_t old = *word, assumed;
    do {
        assumed = old;
        const uint32_t value = delta == 1 ? ((assumed << 1) | lower) : ((assumed >> 1) & ~higher);
        old = atomicCAS(word, assumed, (assumed & ~mask) | (value & mask));
    } while (old != assumed);
}

// the 13 symbols centred on `cell` along `dir` (BoardMap::lineView, Mapping.cpp:31-34) as a 2-bit stream
__device__ __forceinline__ uint64_t window_symbols(const uint32_t* lines, int cell, int dir) {
    const int x = cell % 15, y = cell / 15, diag = x - y + 14, anti = x + y;
    const int line = dir == 0 ? y : dir == 1 ? kColBase + x : dir == 2 ? kDiagBase + diag : kAntiBase + anti;
    const int at = dir == 0 ? x : dir == 1 ? y : dir == 2 ? min(x, y) : min(14 - x, y);
    const int len = dir < 2 ? 15 : dir == 2 ? 15 - abs(diag - 14) : min(anti, 28 - anti) + 1;
    const uint32_t lw = lines[line], in_line = (1u << len) - 1u;
    const uint32_t black = lw & 0x7FFFu, white = lw >> 16;
    const uint64_t cells = static_cast<uint64_t>(spread_bits(~black & in_line) | (spread_bits(~(black | white) & in_line) << 1));
    return (0xAAAull | (cells << 12) | (0xAAAull << (2 * len + 12))) >> (2 * at);       // six '?' | cells | six '?'
}

// Updater::matchPatterns (Pattern.cpp:128-136) for the four directions: lanes 0..3
__device__ inline void match_patterns(const Ctx& c, int move, int slot) {
    if (c.lane < 4) {
        const int dir = c.lane;
        uint64_t syms = window_symbols(c.st + oLines, move, dir);
        uint32_t* out = c.scratch + (slot * 4 + dir) * kResultCap * 2;
        uint32_t row_off = 0;
        int n = 0;
        for (int k = 0; k < 13; ++k) {
            const uint32_t tw = *reinterpret_cast<const uint32_t*>(c.trans + row_off + ((static_cast<uint32_t>(syms) & 3u) << 2));
            syms >>= 2;
            row_off = gmk::dev_trans_row(tw);
            const uint32_t rid = gmk::dev_trans_record(tw);
            if (!rid) continue;
            const uint4 r = c.rec[rid];
            const uint32_t w0s[2] = {r.x, r.z}, w1s[2] = {r.y, r.w};
            for (int e = 0; e < 2; ++e) {
                const uint32_t w0 = w0s[e];
                if (!w0) continue;
                const int back = k - static_cast<int>((w0 >> 27) & 1u) - 6, len = (w0 >> 5) & 7;      // HasCovered (Pattern.cpp:22-25)
                if (back < 0 || back >= len) continue;
                if (n < kResultCap) { out[2 * n] = (w0 & 0x07FFFFFFu) | (static_cast<uint32_t>(back) << 28); out[2 * n + 1] = w1s[e]; ++n; }
                else c.st[oMeta + 3] |= 4u;
            }
        }
        c.scratch[2 * 4 * kResultCap * 2 + slot * 4 + dir] = static_cast<uint32_t>(n);
    }
}

// Updater::updatePatterns (Pattern.cpp:138-165): lane = direction * 16 + result slot
__device__ inline void update_patterns(const Ctx& c, int move, int slot, int delta) {
    const int dir = c.lane >> 4, r = c.lane & 15;
    if (r >= static_cast<int>(c.scratch[2 * 4 * kResultCap * 2 + slot * 4 + dir])) return;
    const uint32_t* res = c.scratch + ((slot * 4 + dir) * kResultCap + r) * 2;
    const uint32_t w0 = res[0], w1 = res[1];
    const int type = w0 & 15, fav = (w0 >> 4) & 1, back = static_cast<int>(w0 >> 28);
    int32_t* meta = reinterpret_cast<int32_t*>(c.st + oMeta);
    if (type == 8) { meta[1] = 0; meta[2] = fav ? 1 : -1; return; }                    // Five ends the game (:140-145)
    const int stride = dir_stride(dir);
    const int last_cell = move + back * stride;                                        // cell of the match's last symbol
    atomicAdd(&c.st[oPdist + 225 * 8 + type], static_cast<uint32_t>(delta) << (16 * fav));      // Record::set(delta, favour) (:390-393)
    const int score = delta * static_cast<int>(dir >= 2 ? (w1 >> 16) : (w1 & 0xFFFFu));
    uint32_t* scores = c.st + oScores;
    const int n_dep = (w0 >> 8) & 7;
    for (int d = 0; d < n_dep; ++d) {
        const uint32_t f = (w0 >> (11 + 4 * d)) & 15u;
        const int cell = last_cell - static_cast<int>(f & 7u) * stride;
        uint32_t* word = &c.st[oPdist + cell * 8 + type];
        if (f & 8u) {                                                                  // '_': the owner's view, then falls through
            set_flag(word, delta, group2(fav, fav), dir);
            atomicAdd(&scores[group2(fav, fav) * kCells + cell], static_cast<uint32_t>(score));
        }
        set_flag(word, delta, group2(fav, fav ^ 1), dir);                              // '_' and '^': the opponent's view
        atomicAdd(&scores[group2(fav, fav ^ 1) * kCells + cell], static_cast<uint32_t>(score));
    }
}

// One compound at (cell, player): Compound::Compound / locate / update (Pattern.cpp:435-550)
__device__ inline void update_one_compound(const Ctx& c, int cell, int pb /* player is black */, int delta) {
    const uint32_t* pd = c.st + oPdist + cell * 8;
    const int g_own = group2(pb, pb), g_opp = group2(pb, pb ^ 1);
    // locate(): state machine S0,L2,LD3,To33,To43,To44 = 0..5 over the directions; first present of L3, D3, L2 per direction
    int state = 0, l3 = 0, triple = 0, n_comp = 0;
    uint32_t comps = 0;                                         // 4 bits per component: dir | (0 L3, 1 D3, 2 L2) << 2
    for (int d = 0; d < 4; ++d) {
        const int shift = (4 * g_own + d) * 2;
        int count = 0, t = -1;
        const uint32_t flags[3] = {(pd[5] >> shift) & 3u, (pd[4] >> shift) & 3u, (pd[3] >> shift) & 3u};      // LiveThree, DeadThree, LiveTwo
        for (int k = 0; k < 3; ++k) {
            count = flags[k] == 1u ? 1 : flags[k] == 3u ? 2 : flags[k] == 0u ? 0 : count;
            if (count) { t = k; break; }
        }
        if (t < 0) continue;
        const int cond = t == 2 ? 1 : 2;
        if (t == 0) ++l3;
        for (int r = 0; r < count; ++r) {
            if (n_comp < 8) { comps |= static_cast<uint32_t>(d | (t << 2)) << (4 * n_comp); ++n_comp; }
            if (state == 0) state += cond;
            else if (state <= 2) state += cond + 1;
            else { triple = 1; state += (state == 5) ? 0 : cond - 1; }
        }
    }
    const int ctype = state - 3;
    if (ctype < 0 || ctype > 2) { c.st[oMeta + 3] |= 2u; return; }                     // the reference indexes out of bounds here
    uint32_t* cd = c.st + oCdist;
    uint32_t* scores = c.st + oScores;
    int count = __popc((cd[cell * 3 + ctype] >> (8 * g_own)) & 0xFFu);
    for (int i = 0; i < n_comp; ++i) {                                                 // update(delta)
        const int cdir = (comps >> (4 * i)) & 3, ct = (comps >> (4 * i + 2)) & 3;
        if (2 * count + delta == -1) return;
        set_flag(&cd[cell * 3 + ctype], delta, g_own, cdir);                           // updateCritical
        atomicAdd(&scores[g_own * kCells + cell], static_cast<uint32_t>(delta * 600));
        set_flag(&cd[cell * 3 + ctype], delta, g_opp, cdir);
        atomicAdd(&scores[g_opp * kCells + cell], static_cast<uint32_t>(delta * 600));
        if (!triple && l3 == 0) {                                                      // updateAntis: first match of the component's type through the cell
            const int want = ct == 0 ? 5 : ct == 1 ? 4 : 3, stride = dir_stride(cdir);
            uint64_t syms = window_symbols(c.st + oLines, cell, cdir);
            uint32_t row_off = 0;
            bool found = false;
            for (int k = 0; k < 13 && !found; ++k) {
                const uint32_t tw = *reinterpret_cast<const uint32_t*>(c.trans + row_off + ((static_cast<uint32_t>(syms) & 3u) << 2));
                syms >>= 2;
                row_off = gmk::dev_trans_row(tw);
                if (k < 6 || !((gmk::dev_trans_kinds(tw) >> ct) & 1u)) continue;
                const uint4 r = c.rec[gmk::dev_trans_record(tw)];
                const uint32_t w0s[2] = {r.x, r.z};
                for (int e = 0; e < 2 && !found; ++e) {
                    const uint32_t w0 = w0s[e];
                    if (!w0 || static_cast<int>(w0 & 15u) != want) continue;
                    const int back = k - static_cast<int>((w0 >> 27) & 1u) - 6, len = (w0 >> 5) & 7, n_dep = (w0 >> 8) & 7;
                    if (back < 0 || back >= len) continue;
                    bool on_cell = false;
                    for (int d = 0; d < n_dep; ++d) on_cell |= ((w0 >> (11 + 4 * d)) & 15u) == (8u | static_cast<uint32_t>(back));
                    if (!on_cell) continue;
                    found = true;
                    const int last_cell = cell + back * stride;
                    for (int d = 0; d < n_dep; ++d) {
                        const int j = (w0 >> (11 + 4 * d)) & 7;
                        if (j == back) continue;
                        const int other = last_cell - j * stride;                      // updatePose(delta, other, component, -favour)
                        set_flag(&cd[other * 3 + ctype], delta, g_opp, cdir);
                        atomicAdd(&scores[g_opp * kCells + other], static_cast<uint32_t>(delta * 600));
                    }
                }
            }
        }
        if (2 * count + delta == 3) atomicAdd(&cd[225 * 3 + ctype], static_cast<uint32_t>(delta) << (16 * pb));
        count += delta;
    }
}

// Updater::updateCompound for the four directions (Pattern.cpp:167-197): lane = direction * 13 + window index
__device__ inline void update_compounds(const Ctx& c, int move, int delta) {
    if (c.lane >= 52) return;
    const int dir = c.lane / 13, i = c.lane % 13;
    if (i == 6 && dir != 0) return;                             // the centre lies on all four lines: handled once (findCompound)
    const uint64_t syms = window_symbols(c.st + oLines, move, dir);
    if (((syms >> (2 * i)) & 3u) != 3u) return;                 // only blanks
    const int cell = move + (i - 6) * dir_stride(dir);
    const int32_t* density = reinterpret_cast<const int32_t*>(c.st + oDensity);
    const uint32_t* pd = c.st + oPdist + cell * 8;
    for (int pb = 0; pb < 2; ++pb) {                            // { White, Black }
        if (density[(pb * 2 + 0) * kCells + cell] < 2) continue;
        const int g = group2(pb, pb);
        const uint32_t bits = ((pd[5] | pd[4] | pd[3]) >> (8 * g)) & 0xFFu;           // Compound::Test (Pattern.cpp:424-433)
        if (!(bits & (bits - 1u))) continue;
        update_one_compound(c, cell, pb, delta);
    }
}

// Updater::updateBlock (Pattern.cpp:236-272): lanes 0..48 = the 7x7 block around the move
__device__ inline void update_block(const Ctx& c, int move, int delta, int src_black) {
    if (c.lane >= 49) return;
    constexpr uint32_t kW[7] = {0x2001002u, 0x0433340u, 0x0354530u, 0x1340431u, 0x0354530u, 0x0433340u, 0x2001002u};   // rows of Pattern.cpp:601-607, one nibble per column
    const int dy = c.lane / 7 - 3, dx = c.lane % 7 - 3;
    const int x = move % 15 + dx, y = move / 15 + dy;
    if (static_cast<unsigned>(x) >= 15u || static_cast<unsigned>(y) >= 15u) return;
    const int q = y * 15 + x, w = static_cast<int>((kW[dy + 3] >> (4 * (6 - (dx + 3)))) & 15u);
    int32_t* density = reinterpret_cast<int32_t*>(c.st + oDensity);
    int32_t* count = density + (src_black * 2 + 0) * kCells;
    int32_t* weight = density + (src_black * 2 + 1) * kCells;
    const int before = weight[q] > 0;
    weight[q] += (weight[q] < 0 ? -1 : 1) * delta * w;
    count[q] += (weight[q] < 0 ? -1 : 1) * delta * (w > 0 ? 1 : 0);
    if (q == move)
        for (int k = 0; k < 4; ++k) {                           // both colours, count and weight: occupied cells hold -v-1
            int32_t& v = density[k * kCells + move];
            v = delta == 1 ? -v - 1 : -(v + 1);
        }
    int32_t* scores = reinterpret_cast<int32_t*>(c.st + oScores);
    scores[group2(src_black, src_black) * kCells + q] += 160 * ((weight[q] > 0) - before);
    if (q == move) {
        const int other = density[((src_black ^ 1) * 2 + 0) * kCells + move];
        if (other != 0 && other != -1) scores[group2(src_black ^ 1, src_black ^ 1) * kCells + move] -= delta * 160;
    }
}

// BoardMap::applyMove / revertMove (Mapping.cpp:37-59) + Board bookkeeping (Game.cpp:37-62): lane 0
__device__ inline void board_set(const Ctx& c, int move, bool place, int black) {
    if (c.lane != 0) return;
    const int x = move % 15, y = move / 15, cb = black ? 0 : 16;
    uint32_t* lines = c.st + oLines;
    const uint32_t bits[4] = {1u << (x + cb), 1u << (y + cb), 1u << (min(x, y) + cb), 1u << (min(14 - x, y) + cb)};
    const int idx[4] = {y, kColBase + x, kDiagBase + x - y + 14, kAntiBase + x + y};
    for (int k = 0; k < 4; ++k) lines[idx[k]] = place ? (lines[idx[k]] | bits[k]) : (lines[idx[k]] & ~bits[k]);
}

// Updater::updateMove (Pattern.cpp:274-302).  src: +1 / -1 = the mover of an applied move, 0 = revert the last move
__device__ inline void update_move(const Ctx& c, int move, int src) {
    int32_t* meta = reinterpret_cast<int32_t*>(c.st + oMeta);
    uint8_t* record = reinterpret_cast<uint8_t*>(c.st + oRecord);
    unsigned long long t_last = c.prof ? __builtin_amdgcn_s_memtime() : 0ull;
    match_patterns(c, move, 0);
    wave_phase_fence();
    prof_mark(c, 0, t_last);
    update_compounds(c, move, -1);
    wave_phase_fence();
    prof_mark(c, 1, t_last);
    update_patterns(c, move, 0, -1);
    wave_phase_fence();
    prof_mark(c, 2, t_last);
    int block_colour;
    if (src != 0) {
        board_set(c, move, true, src > 0);
        if (c.lane == 0) { record[meta[0]] = static_cast<uint8_t>(move); meta[0] += 1; meta[1] = -src; }          // Board::applyMove(move, false)
        block_colour = src > 0;
    } else {
        const int n = meta[0];
        int cur = meta[1];
        if (cur == 0) cur = (n % 2 == 0) ? 1 : -1;              // Board::revertMove: back from a finished game (Game.cpp:51-54)
        const int mover = -cur;                                 // the player whose stone is taken back
        board_set(c, move, false, mover > 0);
        wave_phase_fence();
        if (c.lane == 0) { meta[0] = n - 1; meta[1] = mover; meta[2] = 0; }
        block_colour = mover > 0;
    }
    wave_phase_fence();
    update_block(c, move, src != 0 ? 1 : -1, block_colour);
    wave_phase_fence();
    prof_mark(c, 3, t_last);
    match_patterns(c, move, 1);
    wave_phase_fence();
    prof_mark(c, 4, t_last);
    update_patterns(c, move, 1, 1);
    wave_phase_fence();
    prof_mark(c, 5, t_last);
    update_compounds(c, move, 1);
    wave_phase_fence();
    prof_mark(c, 6, t_last);
    if (c.prof && c.lane == 0) c.prof[7] += 1;
}

// Evaluator::reset (Pattern.cpp:370-386, Game.cpp:138-146): empty board, black to move
__device__ __forceinline__ void reset_state(const Ctx& c) {
    for (int i = c.lane; i < kStateWords; i += 64) c.st[i] = 0u;
    wave_phase_fence();
    if (c.lane == 0) c.st[oMeta + 1] = 1u;
    wave_phase_fence();
}

// Evaluator::applyMove (Pattern.cpp:310-313): ignored when the game is over or the cell is taken
__device__ __forceinline__ void apply_move(const Ctx& c, int mv) {
    const int32_t* meta = reinterpret_cast<const int32_t*>(c.st + oMeta);
    const int cur = meta[1];
    const bool empty = mv >= 0 && mv < kCells && !((c.st[oLines + mv / 15] >> (mv % 15)) & 0x10001u);
    if (cur != 0 && empty) update_move(c, mv, cur);
}

// Evaluator::revertMove(1) (Pattern.cpp:337-342)
__device__ __forceinline__ void revert_move(const Ctx& c) {
    const int32_t* meta = reinterpret_cast<const int32_t*>(c.st + oMeta);
    const uint8_t* record = reinterpret_cast<const uint8_t*>(c.st + oRecord);
    if (meta[0] > 0) update_move(c, record[meta[0] - 1], 0);
}

}  // namespace gmk::evs
