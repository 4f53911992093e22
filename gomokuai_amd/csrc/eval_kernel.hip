// eval_kernel.hip -- K1: batched from-scratch position evaluation on gfx950 (MI355X).
//
// For every board (two 15x15 bit-planes, 64 B) the kernel produces what the reference's incrementally
// maintained Evaluator holds after those stones were played (core/lib/src/Pattern.cpp:111-386):
// scores[4][225], density[2][2][225], pattern / compound totals and winner.  The formulation is the one
// validated against in-order replay in tests/test_formulation.py.
//
// The kernel is bound by VALU issue (PMC: ~85 % VALU busy, HBM traffic = the algorithmic bytes), so the design rule
// below is "fewest wave-instructions per board":
//   * one 64-lane wavefront per board, sixteen boards per 1024-thread workgroup (one workgroup per CU), persistent
//     grid-stride loop: the automaton (dense DFA 556x4 words + emission records, ~14 KB) is staged at LDS address 0
//     ONCE per workgroup, so a DFA step's address is just (next-row offset | symbol * 4); after that single barrier
//     the waves never wait for each other (phases of one board are ordered by wavefront-scope fences only);
//   * phase 0: the two bit-planes become 88 "line words" (rows, columns, both diagonals) that already hold the 2-bit
//     DFA symbols of their cells (one LDS XOR per stone and line), plus the rows as 4-bit digits for phase 3;
//   * phase 1: the 72 lines that can hold a pattern (>= 5 cells) are spread over the 64 lanes (the 8 shortest ride
//     behind the shortest primaries: 19 steps per lane); a lane's lines are one stream of 2-bit symbols,
//     a step is one LDS lookup; emitting transitions are queued by ballot prefix;
//   * phase 2: one lane per queued transition: one 16-byte record read gives the (<= 2) matches, each with a
//     compact list of <= 4 score deposits (ds_add_u32) and 4-bit per-(cell, colour, direction, type) counters;
//   * phase 3: one lane per cell: the 7x7 density stencil as v_dot8_u32_u4 dot products of digit windows with the
//     block's weight rows, area bonus, compound decision from the counters;
//   * phase 4: eight lanes per compound component: its counter-move cells from the 13-symbol window around it;
//   * phase 5: the 3.6 KB score block leaves LDS as coalesced 16-byte stores.
// HBM traffic per board: 64 B in, 7 248 B out (7 312 B algorithmic); everything else stays on chip.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "capi_common.h"

namespace {

constexpr int kBoardsPerBlock = 16;
constexpr int kThreads = 64 * kBoardsPerBlock;
constexpr int kCells = 225;
constexpr int kQueueCap = 448;
constexpr int kMaxBlocksPerCu = 1;

// per-board LDS region (32-bit words)
constexpr int kScoreWords = 4 * kCells;          // 900, 16-byte aligned block
constexpr int kCntWords = 3 * kCells + 1;        // [LiveThree, DeadThree, LiveTwo][cell]: eight 4-bit counters per word, field = colour * 4 + direction:
                                                 // how many '_' pieces of matches of that type lie on the cell (<= 15: at most 8 transitions x 2 matches reach a cell)
constexpr int kNibWords = 21 * 6 + 2;            // stones as 4-bit digits for v_dot8_u32_u4: [row -3 .. 17][black, white][3 words]; a row is
                                                 // 3 zero digits, 15 cells, 3 zero digits (+ 3 unused), so the 7 digits around column x start at digit x
constexpr int kZeroWords = kNibWords + kScoreWords + kCntWords;      // cleared for every board (a multiple of 4); the digit rows come first:
                                                                      // phase 3 reaches all 28 of its words from one base register with immediate offsets
constexpr int kLineWords = 96;                   // line words, 2 bits per cell = its DFA symbol (0 black, 1 white, 3 blank), cell p of the line at bits 2p:
                                                 // rows [0,15), columns [20,35), diagonals x-y+14 at [36,65), anti-diagonals x+y at [65,94)
constexpr int kColBase = 20, kDiagBase = 36, kAntiBase = 65;
constexpr int kMiscWords = 16;                   // [0] stones black | white << 16, [1] winner bits, [2] error, [3] compound queue count, [4..14] totals
constexpr int kBoardWords = kZeroWords + kLineWords + kQueueCap + kMiscWords;
static_assert(kZeroWords % 4 == 0 && kBoardWords % 4 == 0, "16-byte alignment of the per-board blocks");
constexpr int kStaticTableWords = 128 + kLineWords;   // lane jobs, initial line words

// Lane -> line jobs.  A job word: bits 0..3 len, 4..7 x0, 8..11 y0, 12..13 dir, bit 14 valid, 16..22 line word index.
__constant__ uint32_t c_lane_jobs[64 * 2];
__constant__ uint32_t c_line_init[kLineWords];   // all cells blank: (1 << 2 len) - 1
__constant__ int c_scan_steps;

__device__ __forceinline__ int dir_stride(int dir) { return dir == 0 ? 1 : dir == 1 ? 15 : dir == 2 ? 16 : 14; }

// '?' cells '?' '?' of one line as a symbol stream, first symbol in the low bits: exactly 2 * (len + 3) bits
// (1 leading + 2 trailing pads instead of the reference's 6 + 6; '?' = 2 is the off-board symbol)
__device__ __forceinline__ uint64_t line_symbols(uint32_t line_word, int len) {
    return 2ull | (static_cast<uint64_t>(line_word) << 2) | (0xAull << (2 * len + 2));
}

// Phases of one board only exchange data between lanes of the SAME wavefront through LDS.  LDS instructions of
// one wave execute in issue order, so all that is needed between phases is that the compiler keeps the order:
// a wavefront-scope fence (no instruction) instead of a workgroup barrier, which would make the independent
// boards of a block wait for each other at every phase.
__device__ __forceinline__ void wave_phase_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Score deposits of one match (Evaluator::Updater::updatePatterns, Pattern.cpp:138-165).  w0 / w1: emission
// record words (pattern_tables.h).  cell_at = cell of the symbol the transition consumed.
__device__ __forceinline__ void deposit_match(uint32_t w0, uint32_t w1, int cell_at, int dir, int stride,
                                              uint32_t* s_scores, uint32_t* s_cnt, uint32_t* s_misc) {
    const int type = w0 & 15, fav = (w0 >> 4) & 1;
    if (type == 8) { atomicOr(&s_misc[1], fav ? 1u : 2u); return; }                 // Five: winner only (Pattern.cpp:140-145)
    atomicAdd(&s_misc[4 + type], fav ? 0x10000u : 1u);                              // totals row (Pattern.cpp:147, 390-393)
    const int endcell = cell_at - static_cast<int>((w0 >> 27) & 1u) * stride;
    const uint32_t score = dir >= 2 ? (w1 >> 16) : (w1 & 0xFFFFu);                  // int(1.2 * score) on diagonals (Pattern.cpp:151-152)
    uint32_t* own = s_scores + (fav ? 3 : 0) * kCells;                              // Group(favour, favour) (Pattern.h:159-161)
    uint32_t* opp = s_scores + (fav ? 2 : 1) * kCells;                              // Group(favour, -favour)
    const int tslot = type == 5 ? 0 : type == 4 ? 1 : type == 3 ? 2 : -1;           // LiveThree, DeadThree, LiveTwo feed compounds
    uint32_t* cnt = s_cnt + (tslot < 0 ? 0 : tslot) * kCells;
    const uint32_t one = tslot < 0 ? 0u : 1u << (4 * (fav * 4 + dir));
    const int n_dep = (w0 >> 8) & 7;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d >= n_dep) break;
        const uint32_t f = (w0 >> (11 + 4 * d)) & 15u;
        const int c = endcell - static_cast<int>(f & 7u) * stride;
        atomicAdd(&opp[c], score);                                                  // '_' and '^': the opponent's view
        if (f & 8u) {                                                               // '_': the owner's view too
            atomicAdd(&own[c], score);
            if (one) atomicAdd(&cnt[c], one);
        }
    }
}

// Compound::updateAntis (Pattern.cpp:520-543) for one match met at window index k: it qualifies if it is of the wanted
// type, covers the centre cell q and has '_' there; returns the piece index (from the match's end) lying on q, or -1
__device__ __forceinline__ int counter_match(uint32_t w0, int k, int want) {
    const int type = w0 & 15, len = (w0 >> 5) & 7;
    const int back = k - static_cast<int>((w0 >> 27) & 1u) - 6;
    if (!w0 || type != want || back < 0 || back >= len) return -1;
    const int n_dep = (w0 >> 8) & 7;
    bool on_q = false;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t f = (w0 >> (11 + 4 * d)) & 15u;
        on_q |= d < n_dep && f == (8u | static_cast<uint32_t>(back));               // '_' on q
    }
    return on_q ? back : -1;
}

// ... and its other scored blanks get +600 in the opponent's view
__device__ __forceinline__ void add_counter_cells(uint32_t w0, int back, int q, int stride, uint32_t* opp) {
    const int n_dep = (w0 >> 8) & 7, endcell = q + back * stride;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t f = (w0 >> (11 + 4 * d)) & 15u;
        if (d < n_dep && static_cast<int>(f & 7u) != back) atomicAdd(&opp[endcell - static_cast<int>(f & 7u) * stride], 600u);
    }
}

__global__ __launch_bounds__(kThreads)
void eval_positions_kernel(const uint16_t* __restrict__ planes, int n_boards, int iterations,
                           int32_t* __restrict__ out_scores, int32_t* __restrict__ out_density,
                           uint32_t* __restrict__ out_totals, int32_t* __restrict__ out_status,
                           const uint32_t* __restrict__ g_trans, const uint32_t* __restrict__ g_records,
                           int trans_words, int record_words,
                           int phase_mask /* profiling aid: bit p runs phase p; 0x3F in production */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // layout: [trans (LDS address 0)][records (16-byte aligned)][lane jobs 128][initial line words 96][boards: kBoardsPerBlock * kBoardWords]
    const uint4* s_rec = reinterpret_cast<const uint4*>(lds + trans_words);
    uint32_t* s_jobs = lds + trans_words + record_words;
    const uint32_t* s_line_init = s_jobs + 128;

    for (int i = threadIdx.x; i < trans_words; i += kThreads) lds[i] = g_trans[i];
    for (int i = threadIdx.x; i < record_words; i += kThreads) lds[trans_words + i] = g_records[i];
    if (threadIdx.x < 128) s_jobs[threadIdx.x] = c_lane_jobs[threadIdx.x];
    if (threadIdx.x < kLineWords) s_jobs[128 + threadIdx.x] = c_line_init[threadIdx.x];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* s_nib = lds + trans_words + record_words + kStaticTableWords + wave * kBoardWords;
    uint32_t* s_scores = s_nib + kNibWords;                  // int32 scores, accumulated with ds_add
    uint32_t* s_cnt = s_scores + kScoreWords;
    uint32_t* s_lines = s_nib + kZeroWords;
    uint32_t* s_queue = s_lines + kLineWords;
    uint32_t* s_misc = s_queue + kQueueCap;

    __syncthreads();                                         // tables staged; from here on waves never wait for each other
    const uint32_t job_a = s_jobs[lane * 2], job_b = s_jobs[lane * 2 + 1];
    const int scan_steps = c_scan_steps;
    const char* lds_bytes = reinterpret_cast<const char*>(lds);           // the transition table starts at LDS address 0
    const uint32_t lane_tag = static_cast<uint32_t>(lane) << 10;

    // the next board's 64 B are fetched while the current board is evaluated (one dependent HBM round trip per board otherwise)
    auto fetch_row = [&](int b) -> uint32_t {
        if (lane >= 16 || b >= n_boards) return 0u;
        return static_cast<uint32_t>(planes[static_cast<size_t>(b) * 32 + lane]) | (static_cast<uint32_t>(planes[static_cast<size_t>(b) * 32 + 16 + lane]) << 16);
    };
    uint32_t next_row = fetch_row(blockIdx.x * kBoardsPerBlock + wave);

    for (int it = 0; it < iterations; ++it) {
        const int board = (it * gridDim.x + blockIdx.x) * kBoardsPerBlock + wave;
        const bool live = board < n_boards;

        // ---- phase 0: clear accumulators, take the two bit-planes (64 B), turn them into line words and digit rows ----
        {
            uint4* z = reinterpret_cast<uint4*>(s_nib) + lane;
#pragma unroll
            for (int i = 0; i < kZeroWords / 4; i += 64)
                if (i + 64 <= kZeroWords / 4 || lane < kZeroWords / 4 - i) z[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        s_lines[lane] = s_line_init[lane];
        if (lane < kLineWords - 64) s_lines[64 + lane] = s_line_init[64 + lane];
        if (lane < kMiscWords) s_misc[lane] = 0;
        const uint32_t my_row = next_row;
        next_row = fetch_row(((it + 1) * gridDim.x + blockIdx.x) * kBoardsPerBlock + wave);
        wave_phase_fence();
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {                      // the rows as 4-bit digits (phase 3's dot products)
            const int i = lane + 64 * pass, y = i / 6, part = i - 6 * y;      // part = colour * 3 + word
            const uint32_t roww = __shfl(my_row, min(y, 14));
            const uint32_t half = part >= 3 ? roww >> 16 : roww & 0x7FFFu;
            uint32_t v = ((half << 3) >> (8 * (part >= 3 ? part - 3 : part))) & 0xFFu;
            v = (v | (v << 12)) & 0x000F000Fu;
            v = (v | (v << 6)) & 0x03030303u;
            v = (v | (v << 3)) & 0x11111111u;
            if (i < 90) s_nib[18 + i] = v;
        }
        if (lane < 15) atomicAdd(&s_misc[0], static_cast<uint32_t>(__popc(my_row & 0x7FFFu)) | (static_cast<uint32_t>(__popc(my_row >> 16)) << 16));
        {
            // a stone turns its cell's blank (3) into black (0) or white (1) in the four lines through it: one XOR each.
            // Four lanes share a row (cells 0-3, 4-7, 8-11, 12-14), so the loop runs as long as the fullest quarter row.
            const int y = lane >> 2, part = lane & 3;
            const uint32_t row = __shfl(my_row, min(y, 14));
            uint32_t row_sym = 0;
            for (uint32_t m = lane < 60 ? (row | (row >> 16)) & (0xFu << (4 * part)) & 0x7FFFu : 0u; m; m &= m - 1u) {
                const int x = __ffs(m) - 1;
                const uint32_t code = ((row >> x) & 1u) ? 3u : 2u;
                row_sym |= code << (2 * x);
                atomicXor(&s_lines[kColBase + x], code << (2 * y));
                atomicXor(&s_lines[kDiagBase + x - y + 14], code << (2 * min(x, y)));
                atomicXor(&s_lines[kAntiBase + x + y], code << (2 * min(14 - x, y)));
            }
            if (row_sym) atomicXor(&s_lines[y], row_sym);
        }
        wave_phase_fence();

        // ---- phase 1: walk the DFA along this lane's lines; transitions that emit go to the queue ----
        // The lane's one or two lines become ONE stream of 2-bit DFA symbols:
        //   '?' cells '?' '?'  ['?' cells '?' '?']  '?' '?' ...
        // (after "??" the automaton sits in its '?' self-loop state, which the next line's leading '?' keeps: no reset
        // between the two lines).  The stream is kept shifted left by 2, so a step is: symbol * 4 = low word & 12, LDS
        // address = (previous word's next-row offset) | symbol * 4, one lookup.  Emitting transitions are queued raw
        // (record, lane, step); the slot is a ballot prefix (this wave is the only producer), decoding happens in phase 2.
        int n_queued = 0;                                       // wave-uniform
        if (phase_mask & 2) {
            const int len_a = job_a & 15, len_b = job_b & 15;
            uint64_t syms = line_symbols(s_lines[(job_a >> 16) & 127u], len_a);
            syms |= line_symbols(s_lines[(job_b >> 16) & 127u], len_b) << (2 * len_a + 6);
            syms |= 0xAAAAAAAAAAAAAAAAull << (2 * (len_a + len_b) + 12);
            uint32_t cur = static_cast<uint32_t>(syms) << 2;    // symbols 0..14 at bits 2..31
            const uint32_t rest = static_cast<uint32_t>(syms >> 28);      // symbols 15.. at bits 2..
            uint32_t tw = 0;                                    // the previous step's table word (row 0 = root)
            for (int step = 0; step < scan_steps; ++step) {
                if (step == 15) cur = rest;
                const uint32_t addr = (tw & 0x3FFFu) | (cur & 12u);
                cur >>= 2;
                tw = *reinterpret_cast<const uint32_t*>(lds_bytes + addr);
                const uint32_t rec = gmk::dev_trans_record(tw);
                const unsigned long long emitters = __ballot(rec != 0u);
                if (emitters) {
                    if (rec) {
                        const int slot = n_queued + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(emitters >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(emitters), 0u)));
                        if (slot < kQueueCap) s_queue[slot] = rec | lane_tag | (static_cast<uint32_t>(step) << 16);
                    }
                    n_queued += __popcll(emitters);
                }
            }
            if (n_queued > kQueueCap) { s_misc[2] = 1; n_queued = kQueueCap; }
        }
        wave_phase_fence();

        // ---- phase 2: one lane per emitting transition: the score deposits of its 1-2 matches ----
        if (phase_mask & 4) {
            for (int m = lane; m < n_queued; m += 64) {
                const uint32_t qe = s_queue[m];
                // which line of which lane, and where on it (symbol 0 of a line is its leading pad)
                const int src_lane = (qe >> 10) & 63, src_step = static_cast<int>(qe >> 16);
                const uint4 rec = s_rec[qe & 1023u];
                const uint32_t ja = s_jobs[src_lane * 2], jb = s_jobs[src_lane * 2 + 1];
                const int first_len = static_cast<int>(ja & 15u) + 3;
                const uint32_t job = src_step < first_len ? ja : jb;
                const int pos = (src_step < first_len ? src_step : src_step - first_len) - 1;
                const int dir = (job >> 12) & 3, stride = dir_stride(dir);
                const int cell_at = static_cast<int>(((job >> 8) & 15u) * 15u + ((job >> 4) & 15u)) + pos * stride;
                deposit_match(rec.x, rec.y, cell_at, dir, stride, s_scores, s_cnt, s_misc);
                if (rec.z) deposit_match(rec.z, rec.w, cell_at, dir, stride, s_scores, s_cnt, s_misc);
            }
        }
        wave_phase_fence();

        // ---- phase 3: one lane per cell: density stencil, area bonus, compound candidates ----
        int n_cand = 0;                                         // wave-uniform
        if (phase_mask & 8)
        for (int q0 = 0; q0 < kCells; q0 += 64) {
            const int q = min(q0 + lane, kCells - 1);           // the last pass has 33 cells; spare lanes redo cell 224 harmlessly
            const bool spare = q0 + lane >= kCells;
            const int x = q % 15, y = q / 15;
            // rows y-3 .. y+3 as 4-bit digits: the 7 digits around column x start at digit x of the padded row, i.e. at bit
            // 4 * (x & 7) of word x >> 3; v_alignbit takes them out of two words.  Rows y-k and y+k have the same weights
            // (Pattern.cpp:601-607), so their digits are added first (<= 2, no carry); v_dot8_u32_u4 multiplies the
            // digits with the row's weights (the eighth digit gets weight 0) and accumulates.
            uint32_t nib_at = y * 6 + (x >> 3);
            asm volatile("" : "+v"(nib_at));                    // one base register, the 28 reads use immediate offsets
            const uint32_t* nib = s_nib + nib_at;
            const uint32_t sh = (x & 7) * 4;
            uint32_t cnt_c[2], wgt_c[2];                        // [0] white, [1] black (Evaluator::Group, Pattern.h:154-156)
            uint32_t centre = 0;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                uint32_t w[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) w[k] = __builtin_amdgcn_alignbit(nib[k * 6 + c * 3 + 1], nib[k * 6 + c * 3], sh);
                const uint32_t p3 = w[0] + w[6], p2 = w[1] + w[5], p1 = w[2] + w[4], p0 = w[3];
                uint32_t wg = __builtin_amdgcn_udot8(p3, 0x2001002u, 0u, false);
                wg = __builtin_amdgcn_udot8(p2, 0x0433340u, wg, false);
                wg = __builtin_amdgcn_udot8(p1, 0x0354530u, wg, false);
                wg = __builtin_amdgcn_udot8(p0, 0x1340431u, wg, false);
                uint32_t cn = __builtin_amdgcn_udot8(p3, 0x1001001u, 0u, false);
                cn = __builtin_amdgcn_udot8(p2 + p1, 0x0111110u, cn, false);
                cn = __builtin_amdgcn_udot8(p0, 0x1110111u, cn, false);
                wgt_c[1 - c] = wg;
                cnt_c[1 - c] = cn;
                centre |= p0;
            }
            const bool occupied = (centre & 0x1000u) != 0;
            if (!occupied && !spare) {
                if (wgt_c[0] > 0) atomicAdd(&s_scores[0 * kCells + q], 160u);           // Pattern.cpp:268
                if (wgt_c[1] > 0) atomicAdd(&s_scores[3 * kCells + q], 160u);
            }
            if (live && out_density && !spare) {
                int32_t* d = out_density + static_cast<size_t>(board) * 4 * kCells + q;
                const uint32_t neg = occupied ? ~0u : 0u;       // occupied cells hold -v - 1 (Pattern.cpp:253-265)
                d[0 * kCells] = static_cast<int32_t>(cnt_c[0] ^ neg);
                d[1 * kCells] = static_cast<int32_t>(wgt_c[0] ^ neg);
                d[2 * kCells] = static_cast<int32_t>(cnt_c[1] ^ neg);
                d[3 * kCells] = static_cast<int32_t>(wgt_c[1] ^ neg);
            }
            // compound candidates (Compound::Test, Pattern.cpp:424-433, and the density gate, Pattern.cpp:182): cells whose
            // LiveThree / DeadThree / LiveTwo '_' counters, each clipped to 2 (the reference's 2-bit shift flags), OR-ed over the
            // types, sum to two or more over the directions; decided in phase 3b
            uint32_t cand = 0;
            if (!occupied && !spare) {
                const uint32_t any = s_cnt[q] | s_cnt[kCells + q] | s_cnt[2 * kCells + q];
                const uint32_t upper = (any >> 1) | (any >> 2) | (any >> 3);
                const uint32_t ge2 = upper & 0x11111111u, ge1 = (any | upper) & 0x11111111u;         // one bit per field with count >= 2 / >= 1
                if (__popc(ge1 & 0xFFFFu) + __popc(ge2 & 0xFFFFu) >= 2 && cnt_c[0] >= 2) cand |= 1u;
                if (__popc(ge1 >> 16) + __popc(ge2 >> 16) >= 2 && cnt_c[1] >= 2) cand |= 2u;
            }
            const unsigned long long pushers = __ballot(cand != 0u);
            if (pushers) {
                if (cand) {
                    const int slot = n_cand + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(pushers >> 32),
                                                               __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(pushers), 0u)));
                    if (slot < kQueueCap / 2) s_queue[slot] = static_cast<uint32_t>(q) | (cand << 8);
                }
                n_cand += __popcll(pushers);
            }
        }
        wave_phase_fence();

        // ---- phase 3b: one lane per candidate cell: compound state machine (Pattern.cpp:440-486), critical-point deposits,
        //      counter-move rescans queued in the upper half of the queue ----
        if (n_cand > kQueueCap / 2) { s_misc[2] = 1; n_cand = kQueueCap / 2; }
        if (phase_mask & 8)
        for (int m = lane; m < n_cand; m += 64) {
            const uint32_t ce = s_queue[m];
            const int q = ce & 255;
            const uint32_t cw_l3 = s_cnt[q], cw_d3 = s_cnt[kCells + q], cw_l2 = s_cnt[2 * kCells + q];
            for (int c = 0; c < 2; ++c) {
                if (!((ce >> (8 + c)) & 1u)) continue;
                // state machine S0,L2,LD3,To33,To43,To44 = 0..5; a counter counts like the reference's 2-bit shift flags: 0, 1, 2 or more (Pattern.cpp:395-400)
                int state = 0, l3 = 0, triple = 0, n_comp = 0;
                uint32_t comps = 0;                         // 4 bits per component: dir | tslot << 2
                for (int d = 0; d < 4; ++d) {
                    const int f = 4 * (c * 4 + d);
                    const int k3 = min((cw_l3 >> f) & 15u, 2u), kd = min((cw_d3 >> f) & 15u, 2u), k2 = min((cw_l2 >> f) & 15u, 2u);
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
                    if (slot < kQueueCap / 2) s_queue[kQueueCap / 2 + slot] = static_cast<uint32_t>(q) | (static_cast<uint32_t>(c) << 8) | (cd << 9);
                    else s_misc[2] = 1;
                }
            }
        }
        wave_phase_fence();

        // ---- phase 4: the counter-move cells of every compound component: the FIRST match of its type that runs through
        //      the cell with a blank there (Compound::updateAntis, Pattern.cpp:520-543), scanning the 13-symbol window
        //      centred on the cell.  Such a match ends at window index 6..12; eight lanes share a component, lane kk looks
        //      at the transition at index 6 + kk only (the automaton forgets its start state after 7 symbols, so <= 8
        //      lookups from the root bring it to the right state), and the lowest lane with a hit applies it ----
        if (phase_mask & 16) {
            const int n_comp = min(static_cast<int>(s_misc[3]), kQueueCap / 2);
            for (int m0 = 0; m0 < n_comp; m0 += 8) {
                const int m = m0 + (lane >> 3), kk = lane & 7, k = 6 + kk;
                const uint32_t ent = m < n_comp ? s_queue[kQueueCap / 2 + m] : 0u;
                const int q = ent & 255, c = (ent >> 8) & 1, dir = (ent >> 9) & 3, tslot = (ent >> 11) & 3;
                const int want = tslot == 0 ? 5 : tslot == 1 ? 4 : 3;
                const int x = q % 15, y = q / 15, stride = dir_stride(dir);
                uint32_t hit_w0 = 0;
                int hit_back = -1;
                if (m < n_comp && kk < 7) {
                    // the line through q in this direction: its word, q's position on it, its length
                    const int diag = x - y + 14, anti = x + y;
                    const int line = dir == 0 ? y : dir == 1 ? kColBase + x : dir == 2 ? kDiagBase + diag : kAntiBase + anti;
                    const int at = dir == 0 ? x : dir == 1 ? y : dir == 2 ? min(x, y) : min(14 - x, y);
                    const int len = dir < 2 ? 15 : dir == 2 ? 15 - abs(diag - 14) : min(anti, 28 - anti) + 1;
                    // six '?' | cells | six '?', then the 13 symbols starting six before q
                    const uint64_t syms = (0xAAAull | (static_cast<uint64_t>(s_lines[line]) << 12) | (0xAAAull << (2 * len + 12))) >> (2 * at);
                    const int start = k > 7 ? k - 7 : 0;
                    uint32_t cur = static_cast<uint32_t>(syms >> (2 * start)) << 2, tw = 0;      // kept shifted left by 2 as in phase 1
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (start + i <= k) tw = *reinterpret_cast<const uint32_t*>(lds_bytes + ((tw & 0x3FFFu) | (cur & 12u)));
                        cur >>= 2;
                    }
                    if ((gmk::dev_trans_kinds(tw) >> tslot) & 1u) {                     // the record holds the wanted type
                        const uint4 rec = s_rec[gmk::dev_trans_record(tw)];
                        hit_back = counter_match(rec.x, k, want);
                        hit_w0 = rec.x;
                        if (hit_back < 0) { hit_back = counter_match(rec.z, k, want); hit_w0 = rec.z; }
                    }
                }
                const unsigned long long hits = __ballot(hit_back >= 0);
                const uint32_t mine = static_cast<uint32_t>(hits >> (lane & ~7)) & 0xFFu;
                if (hit_back >= 0 && (mine & ((1u << kk) - 1u)) == 0u) add_counter_cells(hit_w0, hit_back, q, stride, s_scores + (c ? 2 : 1) * kCells);
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
                const uint32_t wbits = s_misc[1], stones = s_misc[0];
                const int stones_b = stones & 0xFFFFu, stones_w = stones >> 16;
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
    uint32_t jobs[128];
    for (uint32_t& j : jobs) j = 15u << 16;                       // "no line": length 0 on a line word that stays zero
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
    uint32_t init[kLineWords] = {};                              // every cell of every line blank (symbol 3)
    for (int i = 0; i < 15; ++i) init[i] = init[kColBase + i] = 0x3FFFFFFFu;
    for (int d = 0; d <= 28; ++d) init[kDiagBase + d] = init[kAntiBase + d] = (1u << (2 * (15 - std::abs(d - 14)))) - 1u;
    GMK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_line_init), init, sizeof init));
    GMK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_lane_jobs), jobs, sizeof jobs));
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
    l.lds = static_cast<size_t>(kBoardsPerBlock * kBoardWords + st.n_states * 4 + st.n_records * 4 + kStaticTableWords) * 4;
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
        GMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(eval_positions_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        g_jobs_uploaded = true;
    }
    const Launch l = plan_launch(n, st);
    if (l.lds > 160u * 1024u) { gmk::set_error("gmk_eval_batch: tables do not fit in LDS (%zu bytes)", l.lds); return GMK_ERR_CAPACITY; }
    static const int phase_mask = std::getenv("GMK_EVAL_PHASE_MASK") ? std::atoi(std::getenv("GMK_EVAL_PHASE_MASK")) : 0x3F;
    hipLaunchKernelGGL(eval_positions_kernel, dim3(l.grid), dim3(kThreads), l.lds, static_cast<hipStream_t>(stream),
                       d_planes, n, l.iterations, d_scores, d_density, d_totals, d_status,
                       st.d_trans, st.d_records, st.n_states * 4, st.n_records * 4, phase_mask);
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
