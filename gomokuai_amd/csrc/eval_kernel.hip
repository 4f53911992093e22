// eval_kernel.hip -- K1: batched from-scratch position evaluation on gfx950 (MI355X).
//
// For every board (two 15x15 bit-planes, 64 B) the kernel produces what the reference's incrementally
// maintained Evaluator holds after those stones were played (core/lib/src/Pattern.cpp:111-386):
// scores[4][225], density[2][2][225], pattern / compound totals and winner.  The formulation is the one
// validated against in-order replay in tests/test_formulation.py.
//
// Mapping onto CDNA4
//   * one 64-lane wavefront per board, four boards per 256-thread workgroup, persistent grid-stride loop:
//     the automaton (dense DFA 556x4 words, emission lists, pattern records: ~12.5 KB) is staged into LDS
//     ONCE per workgroup and reused for every board the workgroup evaluates;
//   * phase 1: the 72 board lines that can hold a pattern (>= 5 cells) are spread over the 64 lanes (the 8
//     shortest lines ride behind the shortest primaries: <= 19 steps per lane); each step is one LDS
//     lookup trans[state][symbol]; matches go to a per-board LDS queue;
//   * phase 2: one lane per match scatters its score deposits with LDS atomics (ds_add_u32);
//   * phase 3: one lane per cell computes the 7x7 density stencil from the bit-planes with popcounts,
//     decides compound patterns from per-cell counters, queues their counter-move rescans;
//   * phase 4: one lane per queued compound component rescans a 13-symbol window;
//   * phase 5: the 3.6 KB score block leaves LDS as coalesced 16-byte stores.
// HBM traffic per board: 64 B in, 7 248 B out (7 312 B algorithmic); everything else stays on chip.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "capi_common.h"

namespace {

constexpr int kBoardsPerBlock = 4;
constexpr int kThreads = 64 * kBoardsPerBlock;
constexpr int kCells = 225;
constexpr int kQueueCap = 316;
constexpr int kMaxBlocksPerCu = 4;

// per-board LDS region (32-bit words)
constexpr int kScoreWords = 4 * kCells;          // 900, 16-byte aligned block
constexpr int kCntWords = 2 * kCells;            // per cell 2 words: [LiveThree | DeadThree << 16], [LiveTwo]; 2-bit fields [colour][dir]: bit0 ">= 1", bit1 ">= 2"
constexpr int kRowWords = 92;                    // line words, black | white << 16, bit = position along the line:
                                                 // rows [0,15), columns [16,31), diagonals x-y+14 at [32,61), anti-diagonals x+y at [61,90)
constexpr int kColBase = 16, kDiagBase = 32, kAntiBase = 61;
constexpr int kMiscWords = 16;                   // [0] queue count, [1] winner bits, [2] error, [3] second queue count, [4..14] totals
constexpr int kBoardWords = (kScoreWords + kCntWords + kRowWords + kQueueCap + kMiscWords + 3) & ~3;   // keeps each board's score block 16-byte aligned

// Lane -> line jobs.  A job word: bits 0..3 len, 4..7 x0, 8..11 y0, 12..13 dir, bit 14 valid, 16..22 line word index.
__constant__ uint32_t c_lane_jobs[64 * 2];
__constant__ int c_scan_steps;

__device__ __forceinline__ int dir_dx(int dir) { return dir == 1 ? 0 : dir == 3 ? -1 : 1; }
__device__ __forceinline__ int dir_dy(int dir) { return dir == 0 ? 0 : 1; }
__device__ __forceinline__ int dir_stride(int dir) { return dir == 0 ? 1 : dir == 1 ? 15 : dir == 2 ? 16 : 14; }

// symbol codes on the device: 0 black stone 'x', 1 white stone 'o', 2 off-board '?', 3 blank
__device__ __forceinline__ int cell_symbol(const uint32_t* rows, int x, int y) {
    if (static_cast<unsigned>(x) >= 15u || static_cast<unsigned>(y) >= 15u) return 2;
    const uint32_t w = rows[y] >> x;
    return (w & 1u) ? 0 : (w & 0x10000u) ? 1 : 3;
}

// weights of the 7x7 block (core/lib/src/Pattern.cpp:601-607) for one colour at one cell.
// win[k] = 7-bit window (bit i <-> column x-3+i) of row y-3+k.  Rows are symmetric in |dy|, so the two rows
// of a pair are concatenated (low byte / high byte) and counted with one popcount.
__device__ __forceinline__ void block_density(const uint32_t win[7], int& count, int& weight) {
    const uint32_t p3 = win[0] | (win[6] << 8), p2 = win[1] | (win[5] << 8), p1 = win[2] | (win[4] << 8), p0 = win[3];
    weight = 2 * __popc(p3 & 0x4141u) + __popc(p3 & 0x0808u)
           + 4 * __popc(p2 & 0x2222u) + 3 * __popc(p2 & 0x1C1Cu)
           + 3 * __popc(p1 & 0x2222u) + 5 * __popc(p1 & 0x1414u) + 4 * __popc(p1 & 0x0808u)
           + __popc(p0 & 0x41u) + 3 * __popc(p0 & 0x22u) + 4 * __popc(p0 & 0x14u);
    count = __popc(p3 & 0x4949u) + __popc(p2 & 0x3E3Eu) + __popc(p1 & 0x3E3Eu) + __popc(p0 & 0x77u);
}

// spreads the low 15 bits of v to the even bit positions
__device__ __forceinline__ uint32_t spread_bits(uint32_t v) {
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// A line word (black | white << 16, bit = position) -> the DFA's symbol stream, 2 bits per symbol, first symbol in
// the low bits: '?' (2), the len cells (0 black, 1 white, 3 blank), '?', '?'  (1 leading + 2 trailing pads).
__device__ __forceinline__ uint64_t line_symbols(uint32_t lw, int len) {
    const uint32_t in_line = (1u << len) - 1u;
    const uint32_t black = lw & 0x7FFFu, white = lw >> 16;
    const uint32_t lo = ~black & in_line, hi = ~(black | white) & in_line;      // blank 11, white 01, black 00
    const uint64_t cells = static_cast<uint64_t>(spread_bits(lo) | (spread_bits(hi) << 1));
    return 2ull | (cells << 2) | (0xAull << (2 * len + 2));
}

// queue entries carry dir << 10 | (cell + 16) << 12; the scan keeps the entry of the CURRENT symbol in a register
__device__ __forceinline__ uint32_t queue_entry_step(uint32_t job) { return static_cast<uint32_t>(dir_stride((job >> 12) & 3)) << 12; }
__device__ __forceinline__ uint32_t queue_entry_base(uint32_t job) {
    const int x0 = (job >> 4) & 15, y0 = (job >> 8) & 15, dir = (job >> 12) & 3;
    return (static_cast<uint32_t>(dir) << 10) | (static_cast<uint32_t>(y0 * 15 + x0 - dir_stride(dir) + 16) << 12);   // symbol 0 is the leading pad
}

// Phases of one board only exchange data between lanes of the SAME wavefront through LDS.  LDS instructions of
// one wave execute in issue order, so all that is needed between phases is that the compiler keeps the order:
// a wavefront-scope fence (no instruction) instead of a workgroup barrier, which would make the four
// independent boards of a block wait for each other at every phase.
__device__ __forceinline__ void wave_phase_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(kThreads)
void eval_positions_kernel(const uint16_t* __restrict__ planes, int n_boards, int iterations,
                           int32_t* __restrict__ out_scores, int32_t* __restrict__ out_density,
                           uint32_t* __restrict__ out_totals, int32_t* __restrict__ out_status,
                           const uint32_t* __restrict__ g_trans, const uint16_t* __restrict__ g_emit,
                           const uint32_t* __restrict__ g_pinfo, int trans_words, int emit_words, int pinfo_words,
                           int phase_mask /* profiling aid: bit p runs phase p; 0x3F in production */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // layout: [boards: kBoardsPerBlock * kBoardWords][trans][pinfo][emit (u16)]
    uint32_t* s_trans = lds + kBoardsPerBlock * kBoardWords;
    uint32_t* s_pinfo = s_trans + trans_words;
    uint16_t* s_emit = reinterpret_cast<uint16_t*>(s_pinfo + pinfo_words);

    for (int i = threadIdx.x; i < trans_words; i += kThreads) s_trans[i] = g_trans[i];
    for (int i = threadIdx.x; i < pinfo_words; i += kThreads) s_pinfo[i] = g_pinfo[i];
    for (int i = threadIdx.x; i < emit_words; i += kThreads) s_emit[i] = g_emit[i];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* s_scores = lds + wave * kBoardWords;           // int32 scores, accumulated with ds_add
    uint32_t* s_cnt = s_scores + kScoreWords;
    uint32_t* s_rows = s_cnt + kCntWords;
    uint32_t* s_queue = s_rows + kRowWords;
    uint32_t* s_misc = s_queue + kQueueCap;

    __syncthreads();                                         // tables staged; from here on waves never wait for each other
    const uint32_t job_a = c_lane_jobs[lane * 2], job_b = c_lane_jobs[lane * 2 + 1];
    const int scan_steps = c_scan_steps;

    for (int it = 0; it < iterations; ++it) {
        const int board = (it * gridDim.x + blockIdx.x) * kBoardsPerBlock + wave;
        const bool live = board < n_boards;

        // ---- phase 0: clear accumulators, fetch the two bit-planes (64 B), transpose them into line words ----
        for (int i = lane; i < kScoreWords + kCntWords; i += 64) s_scores[i] = 0;
        if (lane < kMiscWords) s_misc[lane] = 0;
        uint32_t my_row = 0;
        if (lane < 16 && live) my_row = static_cast<uint32_t>(planes[static_cast<size_t>(board) * 32 + lane]) |
                                        (static_cast<uint32_t>(planes[static_cast<size_t>(board) * 32 + 16 + lane]) << 16);
        for (int i = lane; i < kRowWords; i += 64) s_rows[i] = (i < 16) ? my_row : 0u;
        wave_phase_fence();
        if (lane < 15) {                                            // lane y owns row y: one OR per stone into the 3 other line words
            const int y = lane;
            for (uint32_t m = (my_row | (my_row >> 16)) & 0x7FFFu; m; m &= m - 1u) {
                const int x = __ffs(m) - 1;
                const uint32_t cb = ((my_row >> x) & 1u) ? 0u : 16u;
                atomicOr(&s_rows[kColBase + x], 1u << (y + cb));
                atomicOr(&s_rows[kDiagBase + x - y + 14], 1u << (min(x, y) + cb));
                atomicOr(&s_rows[kAntiBase + x + y], 1u << (min(14 - x, y) + cb));
            }
        }
        wave_phase_fence();

        // ---- phase 1: walk the DFA along this lane's lines; transitions that emit go to the queue ----
        // Per job the line is turned once into a stream of 2-bit DFA symbols in a 64-bit register ('?', cells, '?', '?'),
        // so a step is: take 2 bits, one LDS lookup trans[state][sym] (the word holds the next row's byte offset), done.
        // The queue slot comes from a ballot prefix (this wave is the only producer): no returning atomic in the loop;
        // emission lists are expanded in phase 2.
        int n_queued = 0;                                       // wave-uniform
        if (phase_mask & 2) {
            uint32_t job = job_a;
            bool second_done = false;
            uint64_t syms = line_symbols(s_rows[(job >> 16) & 127u], job & 15);
            int steps_left = (job & 0x4000u) ? static_cast<int>(job & 15u) + 3 : 0;
            uint32_t row_off = 0;                               // byte offset of the current state's row in trans[]
            uint32_t entry = queue_entry_base(job);             // dir << 10 | (cell of the current symbol + 16) << 12
            const char* trans_bytes = reinterpret_cast<const char*>(s_trans);
            for (int step = 0; step < scan_steps; ++step) {
                const uint32_t sym = static_cast<uint32_t>(syms) & 3u;
                syms >>= 2;
                const uint32_t tw = *reinterpret_cast<const uint32_t*>(trans_bytes + row_off + sym * 4u);
                row_off = tw & 0x3FFFu;
                const uint32_t li = steps_left > 0 ? (tw >> 14) : 0u;
                const unsigned long long emitters = __ballot(li != 0u);
                if (emitters) {
                    if (li) {
                        const int slot = n_queued + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(emitters >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(emitters), 0u)));
                        if (slot < kQueueCap) s_queue[slot] = entry | li;
                    }
                    n_queued += __popcll(emitters);
                }
                entry += queue_entry_step(job);
                if (--steps_left == 0 && !second_done) {       // the short second line of this lane, if it has one
                    second_done = true;
                    job = job_b;
                    syms = line_symbols(s_rows[(job >> 16) & 127u], job & 15);
                    steps_left = (job & 0x4000u) ? static_cast<int>(job & 15u) + 3 : 0;
                    row_off = 0;
                    entry = queue_entry_base(job);
                }
            }
            if (n_queued > kQueueCap) { s_misc[2] = 1; n_queued = kQueueCap; }
        }
        wave_phase_fence();

        // ---- phase 2: one lane per emitting transition: its 1-2 matches' score deposits (Pattern.cpp:138-165) ----
        if (phase_mask & 4) {
            for (int m = lane; m < n_queued; m += 64) {
              const uint32_t qe = s_queue[m];
              const uint32_t li = qe & 1023u;
              const int dir = (qe >> 10) & 3, cell_at = static_cast<int>(qe >> 12) - 16;
              const int stride = dir_stride(dir);
              const int n_emit = s_emit[li];
              for (int e = 0; e < n_emit; ++e) {
                const uint32_t v = s_emit[li + 1 + e];
                const int pat = v & 0x1FF, endcell = cell_at - static_cast<int>(v >> 15) * stride;   // "back" emissions end one symbol earlier
                const uint32_t w0 = s_pinfo[2 * pat], w1 = s_pinfo[2 * pat + 1];
                const int type = w0 & 15, fav = (w0 >> 4) & 1, len = (w0 >> 5) & 7;
                if (type == 8) { atomicOr(&s_misc[1], fav ? 1u : 2u); continue; }       // Five: winner only
                atomicAdd(&s_misc[4 + type], fav ? 0x10000u : 1u);
                const uint32_t score = dir >= 2 ? (w1 >> 16) : (w1 & 0xFFFFu);
                const int g_own = fav ? 3 : 0, g_opp = fav ? 2 : 1;                     // Pattern.h:159-161
                const int tslot = type == 5 ? 0 : type == 4 ? 1 : type == 3 ? 2 : -1;   // LiveThree, DeadThree, LiveTwo
                for (int j = 0; j < len; ++j) {
                    const uint32_t kind = (w0 >> (8 + 2 * j)) & 3u;
                    if (!kind) continue;
                    const int c = endcell - j * stride;
                    atomicAdd(&s_scores[g_opp * kCells + c], score);
                    if (kind == 1) {
                        atomicAdd(&s_scores[g_own * kCells + c], score);
                        if (tslot >= 0) {                                                   // saturating count 0 / 1 / >= 2
                            uint32_t* word = &s_cnt[c * 2 + (tslot >> 1)];
                            const uint32_t bit = 1u << (16 * (tslot & 1) + 2 * (fav * 4 + dir));
                            if (atomicOr(word, bit) & bit) atomicOr(word, bit << 1);
                        }
                    }
                }
              }
            }
        }
        wave_phase_fence();

        // ---- phase 3: one lane per cell: density stencil, area bonus, compound decision ----
        if (phase_mask & 8)
        for (int q = lane; q < kCells; q += 64) {
            const int x = q % 15, y = q / 15;
            uint32_t win_w[7], win_b[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int yy = y - 3 + k;
                const uint32_t w = (static_cast<unsigned>(yy) < 15u) ? s_rows[yy] : 0u;
                win_b[k] = (((w & 0x7FFFu) << 3) >> x) & 0x7Fu;
                win_w[k] = (((w >> 16) << 3) >> x) & 0x7Fu;
            }
            int cnt_c[2], wgt_c[2];                         // [0] white, [1] black (Evaluator::Group, Pattern.h:154-156)
            block_density(win_w, cnt_c[0], wgt_c[0]);
            block_density(win_b, cnt_c[1], wgt_c[1]);
            const uint32_t here = s_rows[y] >> x;
            const bool occupied = (here & 0x10001u) != 0;
            if (!occupied) {
                if (wgt_c[0] > 0) atomicAdd(&s_scores[0 * kCells + q], 160u);           // Pattern.cpp:268
                if (wgt_c[1] > 0) atomicAdd(&s_scores[3 * kCells + q], 160u);
            }
            if (live && out_density) {
                int32_t* d = out_density + static_cast<size_t>(board) * 4 * kCells + q;
                d[0 * kCells] = occupied ? -cnt_c[0] - 1 : cnt_c[0];
                d[1 * kCells] = occupied ? -wgt_c[0] - 1 : wgt_c[0];
                d[2 * kCells] = occupied ? -cnt_c[1] - 1 : cnt_c[1];
                d[3 * kCells] = occupied ? -wgt_c[1] - 1 : wgt_c[1];
            }
            if (occupied) continue;
            // compound patterns (Pattern.cpp:167-197, 420-486): per colour, counters of '_' pieces of
            // LiveThree / DeadThree / LiveTwo on this cell per direction, saturated at 2
            const uint32_t cw0 = s_cnt[q * 2], cw1 = s_cnt[q * 2 + 1];
            if (!(cw0 | cw1)) continue;
            for (int c = 0; c < 2; ++c) {
                // 2-bit fields per direction: 00 none, 01 one, 11 two or more (= the reference's flag encoding, Pattern.cpp:395-400)
                const uint32_t f_l3 = (cw0 >> (8 * c)) & 0xFFu, f_d3 = (cw0 >> (16 + 8 * c)) & 0xFFu, f_l2 = (cw1 >> (8 * c)) & 0xFFu;
                if (!(f_l3 | f_d3 | f_l2) || cnt_c[c] < 2) continue;
                if (__popc(f_l3 | f_d3 | f_l2) < 2) continue;       // Compound::Test (Pattern.cpp:424-433)
                // state machine S0,L2,LD3,To33,To43,To44 = 0..5 (Pattern.cpp:440-486)
                int state = 0, l3 = 0, triple = 0, n_comp = 0;
                uint32_t comps = 0;                         // 4 bits per component: dir | tslot << 2
                for (int d = 0; d < 4; ++d) {
                    const int k3 = __popc((f_l3 >> (2 * d)) & 3u), kd = __popc((f_d3 >> (2 * d)) & 3u), k2 = __popc((f_l2 >> (2 * d)) & 3u);
                    const int t = k3 ? 0 : kd ? 1 : k2 ? 2 : -1;
                    if (t < 0) continue;
                    const int k = t == 0 ? k3 : t == 1 ? kd : k2, cond = t == 2 ? 1 : 2;
                    if (t == 0) ++l3;
                    for (int r = 0; r < k; ++r) {
                        comps |= static_cast<uint32_t>(d | (t << 2)) << (4 * n_comp);
                        ++n_comp;
                        if (state == 0) state += cond;
                        else if (state <= 2) state += cond + 1;
                        else { triple = 1; state += (state == 5) ? 0 : cond - 1; }
                    }
                }
                const int ctype = state - 3;
                if (ctype < 0 || ctype > 2) { s_misc[2] = 1; continue; }               // reference reads out of bounds here
                atomicAdd(&s_misc[12 + ctype], c ? 0x10000u : 1u);
                const int g_own = c ? 3 : 0, g_opp = c ? 2 : 1;
                atomicAdd(&s_scores[g_own * kCells + q], 600u * n_comp);               // updateCritical, both perspectives
                atomicAdd(&s_scores[g_opp * kCells + q], 600u * n_comp);
                if (triple || l3) continue;
                for (int i = 0; i < n_comp; ++i) {                                     // queue the counter-move rescans
                    const uint32_t cd = (comps >> (4 * i)) & 15u;
                    const uint32_t slot = atomicAdd(&s_misc[3], 1u);
                    if (slot < kQueueCap) s_queue[slot] = static_cast<uint32_t>(q) | (static_cast<uint32_t>(c) << 8) | (cd << 9);
                    else s_misc[2] = 1;
                }
            }
        }
        wave_phase_fence();

        // ---- phase 4: one lane per compound component: first match of its type through the cell
        //      (Compound::updateAntis, Pattern.cpp:520-543) ----
        if (phase_mask & 16) {
            const int n_comp = min(static_cast<int>(s_misc[3]), kQueueCap);
            for (int m = lane; m < n_comp; m += 64) {
                const uint32_t ent = s_queue[m];
                const int q = ent & 255, c = (ent >> 8) & 1, dir = (ent >> 9) & 3, tslot = (ent >> 11) & 3;
                const int want = tslot == 0 ? 5 : tslot == 1 ? 4 : 3;
                const int x = q % 15, y = q / 15, stride = dir_stride(dir);
                // the line through q in this direction: its word, q's position on it, its length
                const int diag = x - y + 14, anti = x + y;
                const int line = dir == 0 ? y : dir == 1 ? kColBase + x : dir == 2 ? kDiagBase + diag : kAntiBase + anti;
                const int at = dir == 0 ? x : dir == 1 ? y : dir == 2 ? min(x, y) : min(14 - x, y);
                const int len = dir < 2 ? 15 : dir == 2 ? 15 - abs(diag - 14) : min(anti, 28 - anti) + 1;
                const uint32_t lw = s_rows[line];
                uint32_t row_off = 0;
                bool found = false;
                for (int k = 0; k < 13 && !found; ++k) {
                    const int p = at + k - 6;
                    int sym = 2;
                    if (static_cast<unsigned>(p) < static_cast<unsigned>(len)) {
                        const uint32_t t = lw >> p;
                        sym = (t & 1u) ? 0 : (t & 0x10000u) ? 1 : 3;
                    }
                    const uint32_t tw = s_trans[(row_off >> 2) + sym];
                    row_off = tw & 0x3FFFu;
                    const uint32_t li = tw >> 14;
                    if (!li || k < 6) continue;                              // a match covering q ends at window index >= 6
                    const int cnt = s_emit[li];
                    for (int e = 0; e < cnt && !found; ++e) {
                        const uint32_t v = s_emit[li + 1 + e];
                        const int off = k - static_cast<int>(v >> 15);
                        const uint32_t w0 = s_pinfo[2 * (v & 0x1FFu)];
                        const int type = w0 & 15, len = (w0 >> 5) & 7;
                        const int back = off - 6;                                      // piece index (from the end) lying on q
                        if (type != want || back < 0 || back >= len) continue;
                        if (((w0 >> (8 + 2 * back)) & 3u) != 1u) continue;             // must be '_' on q
                        found = true;
                        const int endcell = q + back * stride;
                        for (int j = 0; j < len; ++j) {
                            if (j == back || !((w0 >> (8 + 2 * j)) & 3u)) continue;
                            atomicAdd(&s_scores[(c ? 2 : 1) * kCells + endcell - j * stride], 600u);
                        }
                    }
                }
            }
        }
        wave_phase_fence();

        // ---- phase 5: results leave LDS ----
        if (live && (phase_mask & 32)) {
            if (out_scores) {
                int4* dst = reinterpret_cast<int4*>(out_scores + static_cast<size_t>(board) * kScoreWords);
                const int4* src = reinterpret_cast<const int4*>(s_scores);
                for (int i = lane; i < kScoreWords / 4; i += 64) dst[i] = src[i];
            }
            if (out_totals && lane < 11) out_totals[static_cast<size_t>(board) * 11 + lane] = s_misc[4 + lane];
            if (out_status && lane == 0) {
                const uint32_t wbits = s_misc[1];
                int stones_b = 0, stones_w = 0;
                for (int r = 0; r < 15; ++r) { stones_b += __popc(s_rows[r] & 0x7FFFu); stones_w += __popc(s_rows[r] >> 16); }
                // the side that completed five is the only one that can own a Five (the game stops there)
                const int winner = (wbits & 1u) ? 1 : (wbits & 2u) ? -1 : 0;
                const bool over = winner != 0 || stones_b + stones_w == kCells;
                const int to_move = over ? 0 : (stones_b == stones_w ? 1 : -1);
                out_status[board] = (over ? 1 : 0) | (s_misc[2] ? 2 : 0) | ((winner & 0xFF) << 8) | ((to_move & 0xFF) << 16);
            }
        }
        wave_phase_fence();
    }
}

// ---- host side ----
struct LineJob { int len, x0, y0, dir; };

int upload_lane_jobs() {
    std::vector<LineJob> lines;
    for (int i = 0; i < 15; ++i) lines.push_back({15, 0, i, 0});
    for (int i = 0; i < 15; ++i) lines.push_back({15, i, 0, 1});
    for (int d = -10; d <= 10; ++d) lines.push_back({15 - std::abs(d), d > 0 ? d : 0, d > 0 ? 0 : -d, 2});
    for (int k = 4; k <= 24; ++k) { const int x0 = std::min(k, 14); lines.push_back({std::min(k, 28 - k) + 1, x0, k - x0, 3}); }
    std::stable_sort(lines.begin(), lines.end(), [](const LineJob& a, const LineJob& b) { return a.len > b.len; });
    uint32_t jobs[128] = {};
    auto pack = [](const LineJob& l) {
        const int line = l.dir == 0 ? l.y0 : l.dir == 1 ? kColBase + l.x0 : l.dir == 2 ? kDiagBase + l.x0 - l.y0 + 14 : kAntiBase + l.x0 + l.y0;
        return static_cast<uint32_t>(l.len | (l.x0 << 4) | (l.y0 << 8) | (l.dir << 12) | 0x4000 | (line << 16));
    };
    int steps = 0;
    for (int lane = 0; lane < 64; ++lane) {
        jobs[lane * 2] = pack(lines[lane]);
        int total = lines[lane].len + 3;
        const int extra = 64 + (63 - lane);                   // shortest leftovers ride behind the shortest primaries
        if (extra < static_cast<int>(lines.size())) { jobs[lane * 2 + 1] = pack(lines[extra]); total += lines[extra].len + 3; }
        steps = std::max(steps, total);
    }
    GMK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_lane_jobs), jobs, sizeof jobs));
    if (const char* env = std::getenv("GMK_EVAL_SCAN_STEPS")) steps = std::atoi(env);      // profiling aid only: wrong results
    GMK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_scan_steps), &steps, sizeof steps));
    return GMK_OK;
}

struct Launch { int grid, iterations; size_t lds; };

Launch plan_launch(int n, const gmk::DeviceState& st) {
    const int tiles = (n + kBoardsPerBlock - 1) / kBoardsPerBlock;
    const int max_grid = std::max(1, st.cu_count * kMaxBlocksPerCu);
    Launch l;
    l.iterations = std::max(1, (tiles + max_grid - 1) / max_grid);
    l.grid = std::max(1, (tiles + l.iterations - 1) / l.iterations);
    const int pinfo_words = st.n_patterns * 2;
    l.lds = static_cast<size_t>(kBoardsPerBlock * kBoardWords + st.n_states * 4 + pinfo_words) * 4 + static_cast<size_t>((st.emit_words + 1) & ~1) * 2;
    return l;
}

bool g_jobs_uploaded = false;

}  // namespace

extern "C" int gmk_eval_batch(const uint16_t* d_planes, int n, int32_t* d_scores, int32_t* d_density,
                              uint32_t* d_totals, int32_t* d_status, void* stream) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (n < 0 || (n > 0 && !d_planes)) { gmk::set_error("gmk_eval_batch: bad arguments"); return GMK_ERR_ARG; }
    if (n == 0) return GMK_OK;
    if (!g_jobs_uploaded) {
        const int rc = upload_lane_jobs();
        if (rc != GMK_OK) return rc;
        g_jobs_uploaded = true;
    }
    const Launch l = plan_launch(n, st);
    static const int phase_mask = std::getenv("GMK_EVAL_PHASE_MASK") ? std::atoi(std::getenv("GMK_EVAL_PHASE_MASK")) : 0x3F;
    hipLaunchKernelGGL(eval_positions_kernel, dim3(l.grid), dim3(kThreads), l.lds, static_cast<hipStream_t>(stream),
                       d_planes, n, l.iterations, d_scores, d_density, d_totals, d_status,
                       st.d_trans, st.d_emit, st.d_pattern_info, st.n_states * 4, st.emit_words, st.n_patterns * 2, phase_mask);
    GMK_HIP_CHECK(hipGetLastError());
    return GMK_OK;
}

extern "C" int gmk_eval_launch_info(int n, int* grid, int* block, int* lds_bytes) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded"); return GMK_ERR_STATE; }
    const Launch l = plan_launch(std::max(n, 1), st);
    if (grid) *grid = l.grid;
    if (block) *block = kThreads;
    if (lds_bytes) *lds_bytes = static_cast<int>(l.lds);
    return GMK_OK;
}

extern "C" int gmk_eval_batch_host(const uint16_t* h_planes, int n, int32_t* h_scores, int32_t* h_density,
                                   uint32_t* h_totals, int32_t* h_status) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (n < 0 || (n > 0 && !h_planes)) { gmk::set_error("gmk_eval_batch_host: bad arguments"); return GMK_ERR_ARG; }
    if (n == 0) return GMK_OK;
    uint16_t* d_planes = nullptr;
    int32_t *d_scores = nullptr, *d_density = nullptr, *d_status = nullptr;
    uint32_t* d_totals = nullptr;
    const size_t nb = static_cast<size_t>(n);
    int rc = GMK_OK;
    auto cleanup = [&]() { (void)hipFree(d_planes); (void)hipFree(d_scores); (void)hipFree(d_density); (void)hipFree(d_totals); (void)hipFree(d_status); };
#define GMK_TRY(expr) do { if ((expr) != hipSuccess) { gmk::set_error("%s failed", #expr); cleanup(); return GMK_ERR_HIP; } } while (0)
    GMK_TRY(hipMalloc(&d_planes, nb * 64));
    GMK_TRY(hipMalloc(&d_scores, nb * 3600));
    GMK_TRY(hipMalloc(&d_density, nb * 3600));
    GMK_TRY(hipMalloc(&d_totals, nb * 44));
    GMK_TRY(hipMalloc(&d_status, nb * 4));
    GMK_TRY(hipMemcpy(d_planes, h_planes, nb * 64, hipMemcpyHostToDevice));
    rc = gmk_eval_batch(d_planes, n, d_scores, d_density, d_totals, d_status, nullptr);
    if (rc != GMK_OK) { cleanup(); return rc; }
    GMK_TRY(hipDeviceSynchronize());
    if (h_scores) GMK_TRY(hipMemcpy(h_scores, d_scores, nb * 3600, hipMemcpyDeviceToHost));
    if (h_density) GMK_TRY(hipMemcpy(h_density, d_density, nb * 3600, hipMemcpyDeviceToHost));
    if (h_totals) GMK_TRY(hipMemcpy(h_totals, d_totals, nb * 44, hipMemcpyDeviceToHost));
    if (h_status) GMK_TRY(hipMemcpy(h_status, d_status, nb * 4, hipMemcpyDeviceToHost));
#undef GMK_TRY
    cleanup();
    return GMK_OK;
}
