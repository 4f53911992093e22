// eval_kernel.hip -- K1: batched from-scratch position evaluation on gfx950 (MI355X).
//
// For every board (two 15x15 bit-planes, 64 B) the kernel produces what the reference's incrementally
// maintained Evaluator holds after those stones were played (core/lib/src/Pattern.cpp:111-386):
// scores[4][225], density[2][2][225], pattern / compound totals and winner.  The formulation is the one
// validated against in-order replay in tests/test_formulation.py.
//
// What binds it (rocprofv3 PMC of the round-4 build, profiles/r04_k1_*): 842 vector, 356 scalar and 142 LDS wave-instructions per board; a wave64
// vector instruction occupies its SIMD for 4 cycles whatever it does (tools/valu_probe.hip), so the vector units are ~75 % and the LDS pipe ~60 % busy
// (41 % of its cycles bank conflicts) with sixteen boards of LDS per CU; HBM traffic = 1.05 x the algorithmic bytes.  Neither the store path nor
// HBM bandwidth is the limit (tools/store_probe.hip): instructions and dependent LDS round trips are.  So: few wave-instructions per board, few
// dependent round trips, and the one dense piece on the matrix cores:
//   * one 64-lane wavefront per board, sixteen boards in flight per 1024-thread workgroup (one workgroup per CU); the automaton
//     (dense DFA 556x4 words + emission records, ~14 KB) is staged at LDS address 0 ONCE per workgroup, so a DFA step's address is
//     just (next-row offset | symbol * 4): one v_bfe + one v_bfi; after that single barrier the waves never wait for each other (phases of one
//     board are ordered by wavefront-scope fences only).  A workgroup owns a contiguous run of groups of SIXTEEN boards and hands its boards
//     (and the density bursts of its groups) out one at a time from counters in LDS;
//   * phase 0: the two bit-planes become 88 "line words" (rows, columns, both diagonals) that already hold the 2-bit
//     DFA symbols of their cells (one LDS XOR per stone and line); where a colour's density count is positive (the area
//     bonus) comes from the rows dilated by the three row patterns of the 7x7 mask, with DPP row shifts; the score block -- [cell][4 groups] in
//     LDS -- is written ONCE, with the bonus in it;
//   * phase 1: the 72 lines that can hold a pattern (>= 5 cells) are spread over the 64 lanes (the 8 shortest ride
//     behind the shortest primaries: 19 steps per lane, fully unrolled); a lane's lines are one stream of 2-bit symbols,
//     a step is one LDS lookup; emitting transitions are queued by ballot prefix, in the shadow of the next step's lookup (the queue's fill
//     level lives on the scalar unit, clamped there: no per-lane bounds check);
//   * phase 2: one lane per queued transition: one 16-byte record read gives the (<= 2) matches, each with a
//     compact list of <= 4 score deposits -- a colour's own and opponent view of a cell are the halves of one 64-bit word, so a deposit is ONE
//     ds_add_u64 whose value decides -- and 4-bit per-(cell, colour, direction, type) counters;
//   * phase 3: one lane per cell: compound candidates from the counters; phase 3b: one lane per (candidate, colour):
//     the density gate "count >= 2" from seven row popcounts, the compound decision in closed form (components n, threes s), +-600 deposits;
//   * phase 4: eight lanes per compound component: its counter-move cells from the 13-symbol window around it;
//   * phase 5: the 3.6 KB score block leaves LDS transposed to [group][cell]: sixteen non-temporal dword stores of 256 contiguous bytes, in
//     ADDRESS order (the two parts of a cache line that two pieces share reach the L2 back to back);
//   * phase D, once per group, in the board iteration that is the wavefront's turn within its workgroup: the density planes of the
//     sixteen boards on the matrix cores -- Out[board, colour][cell] = Stone[board, colour][cell'] * W[cell'][cell], the stones as
//     the A operand and the constant banded weight matrix (rows of a 6 KB table in LDS) as the B operand of
//     v_mfma_i32_32x32x32_i8 (14 passes of 3-5 MFMAs, small integers: exact), stored as 2 x 128 contiguous bytes per instruction.
// HBM traffic per board: 64 B in, 7 248 B out (7 312 B algorithmic); everything else stays on chip.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "capi_common.h"

namespace {

constexpr int kBoardsPerBlock = 16;               // wavefronts per workgroup, one board in flight each
constexpr int kGroupBoards = 16;                  // boards whose density planes are one unit of work (the columns of phase D's matrix product)
constexpr int kThreads = 64 * kBoardsPerBlock;
constexpr int kCells = 225;
constexpr int kQueueCap = 448;
constexpr int kMaxBlocksPerCu = 1;

// per-board LDS region (32-bit words)
constexpr int kScoreWords = 4 * kCells;          // 900, 16-byte aligned block.  In LDS the block is [cell][4 groups]: a colour's own and opponent view of a
                                                 // cell are the halves of one 64-bit word (white: groups 0, 1; black: 2, 3), so a deposit into both is ONE ds_add_u64
constexpr int kCntWords = 3 * kCells + 1;        // [LiveThree, DeadThree, LiveTwo][cell]: eight 4-bit counters per word, field = colour * 4 + direction:
                                                 // how many '_' pieces of matches of that type lie on the cell (<= 15: at most 8 transitions x 2 matches reach a cell)
constexpr int kZeroWords = kScoreWords + kCntWords;                   // cleared for every board (a multiple of 4)
constexpr int kLineWords = 96;                   // line words, 2 bits per cell = its DFA symbol (0 black, 1 white, 3 blank): rows [0,15) cell x at bits 2x,
                                                 // columns [20,35) cell y at bits 2y, diagonals x-y+14 at [36,65) at bits 2x, anti-diagonals x+y at [65,94) at bits 2y
                                                 // (a stone's shift is then 2x for its row and diagonal, 2y for its column and anti-diagonal: two shifted
                                                 // codes per stone instead of four; a reader shifts a diagonal's word down to its first cell)
constexpr int kColBase = 20, kDiagBase = 36, kAntiBase = 65;
#ifndef GMK_K1_TOTALS_COPIES
#define GMK_K1_TOTALS_COPIES 8
#endif
constexpr int kTotalsCopies = GMK_K1_TOTALS_COPIES;   // of the per-type totals, in the queue's last words (phase 0 says why); 8 x copies <= 64
constexpr int kMiscWords = 48;                   // [0] stones black | white << 16, [1] winner bits, [2] error, [3] compound queue count, [4..14] totals,
                                                 // [19..33] the rows (black | white << 16) between three zero rows on either side ([16..18], [34..36])
constexpr int kBoardWords = kZeroWords + kLineWords + kQueueCap + kMiscWords;
static_assert(kZeroWords % 4 == 0 && kBoardWords % 4 == 0, "16-byte alignment of the per-board blocks");
constexpr int kStaticTableWords = 128 + kLineWords + 512 + 1560 + 4;   // lane jobs, initial line words, the bits-to-bytes table, the weight table of phase D,
                                                                        // and the workgroup's two hand-out counters (boards, density bursts)

// Lane -> line jobs.  A job word holds what the scan and the deposits need of a line, ready to use: bits 0..4 symbols in its stream
// segment (len + 3: one leading and two trailing pads), 5..6 dir, 7..11 cell stride, 12..19 its first cell, 20..26 line word index.
__constant__ uint32_t c_lane_jobs[64 * 2];
__constant__ uint32_t c_line_init[kLineWords];   // all cells blank: (1 << 2 len) - 1
constexpr int kScanSteps = 19;                    // symbols in the longest lane stream (upload_lane_jobs checks it)

__device__ __forceinline__ int dir_stride(int dir) { return (0x0E100F01u >> (8 * dir)) & 0xFFu; }      // 1, 15, 16, 14: a shift, not three branches

// '?' cells '?' '?' of one line as a symbol stream, first symbol in the low bits: exactly 2 * (len + 3) bits
// (1 leading + 2 trailing pads instead of the reference's 6 + 6; '?' = 2 is the off-board symbol)
__device__ __forceinline__ uint64_t line_symbols(uint32_t line_word, int len) {
    return 2ull | (static_cast<uint64_t>(line_word) << 2) | (0xAull << (2 * len + 2));
}

// Phases of one board only exchange data between lanes of the SAME wavefront through LDS.  LDS instructions of
// one wave execute in issue order, so all that is needed between phases is that the compiler keeps the order:
// a wavefront-scope fence (no instruction) instead of a workgroup barrier, which would make the independent
// boards of a block wait for each other at every phase.
// the 32-bit word at a byte address of LDS (no base added: the staged automaton starts at address 0)
__device__ __forceinline__ const __attribute__((address_space(3))) uint32_t* lds_word(uint32_t byte_address) {
    return reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(static_cast<uintptr_t>(byte_address));
}

__device__ __forceinline__ __attribute__((address_space(3))) uint32_t* lds_word_rw(uint32_t byte_address) {
    return reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(static_cast<uintptr_t>(byte_address));
}

// (table word & 0x3FF3) | (four & 12) for four < 16: the lookup address of a DFA step as ONE v_bfi_b32 (written out: the compiler turns the
// expression back into a shift, a mask and an and-or)
__device__ __forceinline__ uint32_t dfa_address(uint32_t table_word, uint32_t four) {
    uint32_t addr;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(addr) : "s"(0x3FF3u), "v"(table_word), "v"(four));
    return addr;
}

// v_mul_u32_u24, written out: once the compiler has proved that only low bits of a product are used it drops the mask that made the factor 24 bits
// wide and then has to take the full 32-bit multiply, which runs at a quarter of the rate
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}

__device__ __forceinline__ void wave_phase_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Score deposits of one match (Evaluator::Updater::updatePatterns, Pattern.cpp:138-165).  w0 / w1: emission
// record words (pattern_tables.h).  cell_at = cell of the symbol the transition consumed.
// back_step = -stride: the cells of a match lie behind the symbol that ended it, one multiply-add each.
__device__ __forceinline__ void deposit_match(uint32_t w0, uint32_t w1, int cell_at, int dir, int back_step,
                                              uint32_t* s_scores, uint32_t* s_cnt, uint32_t* s_misc, uint32_t* totals /* this lane's copy of the eight per-type totals */) {
    const int type = w0 & 15, fav = (w0 >> 4) & 1;
    if (type == 8) { atomicOr(&s_misc[1], fav ? 1u : 2u); return; }                 // Five: winner only (Pattern.cpp:140-145)
    atomicAdd(&totals[type], fav ? 0x10000u : 1u);                                  // totals row (Pattern.cpp:147, 390-393)
    const int endcell = cell_at + static_cast<int>((w0 >> 27) & 1u) * back_step;
    const uint32_t score = dir >= 2 ? (w1 >> 16) : (w1 & 0xFFFFu);                  // int(1.2 * score) on diagonals (Pattern.cpp:151-152)
    // Group(favour, favour) is the owner's view, Group(favour, -favour) the opponent's (Pattern.h:159-161): '_' adds the score to both,
    // '^' to the opponent's only.  The two views of a cell are one 64-bit word of the block (white: own low / opp high, black: opp low /
    // own high): one add either way, the value decides.
    unsigned long long* pair = reinterpret_cast<unsigned long long*>(s_scores) + fav;
    const uint32_t lo_opp = fav ? score : 0u, hi_opp = fav ? 0u : score;
    const uint32_t tslot = 5u - static_cast<uint32_t>(type);                         // LiveThree 0, DeadThree 1, LiveTwo 2 feed compounds (no branches: one compare)
    const bool feeds = tslot < 3u;
    uint32_t* cnt = s_cnt + (feeds ? tslot : 0u) * kCells;
    const uint32_t one = 1u << (4 * (fav * 4 + dir));
    const int n_dep = (w0 >> 8) & 7;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d >= n_dep) break;
        const uint32_t f = (w0 >> (11 + 4 * d)) & 15u;
        const int c = endcell + static_cast<int>(f & 7u) * back_step;
        const bool both = (f & 8u) != 0u;                                           // '_'
        atomicAdd(&pair[2 * c], (static_cast<unsigned long long>(both ? score : hi_opp) << 32) | (both ? score : lo_opp));
        if (both && feeds) atomicAdd(&cnt[c], one);
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

// ... and its other scored blanks get +600 in the opponent's view (opp = the opponent group's word of cell 0 in the [cell][4] block)
__device__ __forceinline__ void add_counter_cells(uint32_t w0, int back, int q, int stride, uint32_t* opp) {
    const int n_dep = (w0 >> 8) & 7, endcell = q + back * stride;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t f = (w0 >> (11 + 4 * d)) & 15u;
        if (d < n_dep && static_cast<int>(f & 7u) != back) atomicAdd(&opp[4 * (endcell - static_cast<int>(f & 7u) * stride)], 600u);
    }
}

// ---- phase D: the density stencil of sixteen boards on the matrix cores ----
// Evaluator::Updater::updateBlock (Pattern.cpp:236-272) adds, for every stone, the 7x7 BlockWeights (Pattern.cpp:598-609) around
// it to its colour's weight plane and their non-zero mask to its count plane: as a function of the position,
//   weight[c][q] = sum over cells q' of W[q' - q] * stone_c[q'],  count[c][q] likewise with W != 0,
// a product of a constant banded 225 x 225 matrix with the stone planes.  Sixteen boards x two colours are the 32 columns
// of v_mfma_i32_32x32x32_i8; an M tile is 32 consecutive cells (7 tiles cover cells 0..223, cell 224 is done by hand), a K tile
// two board rows of 16 (k = 16 y + x: the stones of a row become the bytes of its tile as they lie).  A tile of cells only
// reaches rows y0-3 .. y1+3, so 31 (M, K) tile pairs per plane kind hold non-zeros.
[[maybe_unused]] constexpr int kDensTiles = 7;       // M tiles of 32 cells (cells 0 .. 223; cell 224 is done by hand)
__host__ __device__ constexpr int dens_kt_lo(int m) { return (((32 * m) / 15 - 3) < 0 ? 0 : (32 * m) / 15 - 3) / 2; }
__host__ __device__ constexpr int dens_kt_hi(int m) { return (((32 * m + 31) / 15 + 3) > 14 ? 14 : (32 * m + 31) / 15 + 3) / 2; }
// The weight operand of lane (cell c = (yo, xo), k half h) at K tile kt holds the taps from row y = 2 kt + h, columns 0..15, to c:
// byte j = w(y - yo, j - xo).  It depends on (dy, xo) only, so ALL weight operands are rows of one small table in LDS,
// [plane kind][dy + 6][xo] x 16 bytes (6 240 B), read with one ds_read_b128 per MFMA; going up one K tile is +2 in dy = +480 bytes,
// an immediate offset.  (A table in global memory, 62 KB read by every wavefront for every group, cost 380 MB of L2 traffic per
// launch and a global-load latency per MFMA.)
constexpr int kWtabDy = 13, kWtabEntryWords = 4, kWtabKindWords = kWtabDy * 15 * kWtabEntryWords, kWtabWords = 2 * kWtabKindWords;
// BlockWeights by |dy| and dx + 3 (the matrix is symmetric in both axes)
__host__ __device__ constexpr int block_weight(int dy, int dx) {
    constexpr int w[4][7] = {{1, 3, 4, 0, 4, 3, 1}, {0, 3, 5, 4, 5, 3, 0}, {0, 4, 3, 3, 3, 4, 0}, {2, 0, 0, 1, 0, 0, 2}};
    return w[dy < 0 ? -dy : dy][dx + 3];
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ---- phase D: one plane kind of one M tile of a group of sixteen boards leaves for HBM ----
// Here the BOARDS are on the accumulator rows (A = stones, B = weights): lane (n, h) holds cell 32 M + n of sixteen board-colour
// columns, so a store instruction writes 2 x 128 contiguous bytes (with the cells on the rows a lane would own 16 bytes of 64
// different cache lines: measured 2.4 TB/s for the density planes alone).  Spreading the fourteen passes over the group's board
// iterations spreads the 3.6 KB per board over the kernel's run time instead of one burst at its start.
// The stones are read again (64 B per board from L2) and become operand bytes through a 256-entry table in LDS (byte -> 8 bytes).
template <int KIND, int M>
__device__ __forceinline__ void density_tile_pass(const uint32_t (&own_rows)[8], const uint32_t (&other_rows)[8], int n_boards, int first_board, int lane,
                                                  const v4i* __restrict__ s_wtab, int32_t* __restrict__ out_density,
                                                  const uint2* __restrict__ s_lut, int plane_stride) {
    constexpr int lo = dens_kt_lo(M), hi = dens_kt_hi(M), nk = hi - lo + 1;
    constexpr int y0 = (32 * M) / 15, y1 = (32 * M + 31) / 15;
    const int n = lane & 31, h = lane >> 5;
    // occupied cells 32 M .. 32 M + 31 of this lane's board (rows y0 .. y1 of both planes)
    uint32_t occ = 0;
#pragma unroll
    for (int y = y0; y <= y1; ++y) {
        const uint32_t both = own_rows[y / 2] | other_rows[y / 2];
        const uint32_t r = ((y & 1) ? both >> 16 : both) & 0x7FFFu;
        const int at = 15 * y - 32 * M;
        occ |= at >= 0 ? r << at : r >> -at;
    }
    const int cell = 32 * M + n, yo = (cell * 0x8889) >> 19, xo = cell - 15 * yo;
    const v4i* w = s_wtab + KIND * (kWtabKindWords / kWtabEntryWords) + (2 * lo + h - yo + 6) * 15 + xo;
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < nk; ++k) {
        const uint32_t r = (own_rows[lo + k] >> (16 * h)) & 0x7FFFu;
        const uint2 b0 = s_lut[r & 255u], b1 = s_lut[r >> 8];
        const v4i stones = {static_cast<int>(b0.x), static_cast<int>(b0.y), static_cast<int>(b1.x), static_cast<int>(b1.y)};
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(stones, w[30 * k], acc, 0, 0, 0);
    }
    // accumulator i = board-colour column 8 (i / 4) + 4 h + (i % 4) at cell 32 M + n; columns 2 b, 2 b + 1 are board b
    const int n_live = n_boards - first_board;      // >= 16 except in the last group
    int32_t* out = out_density + static_cast<size_t>(first_board) * 4 * plane_stride + KIND * plane_stride + 32 * M + n + h * (2 * 4 * plane_stride);
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const int b = 4 * (i / 4) + (i % 4) / 2;    // + 2 h
        const uint32_t wd = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(8 * b + 16 * h, static_cast<int>(occ)));
        const int neg = __builtin_amdgcn_sbfe(static_cast<int>(wd), n, 1);      // occupied cells hold -v - 1 = ~v (Pattern.cpp:253-265)
        if (b + 2 * h < n_live && plane_stride > 0) {
            out[b * 4 * plane_stride + 1 * 2 * plane_stride] = acc[i] ^ neg;                 // column 2 b: black, the second colour block
            out[b * 4 * plane_stride + 0 * 2 * plane_stride] = acc[i + 1] ^ neg;             // column 2 b + 1: white, the first
        }
    }
}

// All fourteen passes of a group, back to back: the group's density planes are one contiguous block of 16 x 3 600 bytes, and written
// within a few microseconds by one wavefront they reach DRAM as whole cache lines, row after row.  (One pass per board iteration --
// each line written in two pieces a board's work apart, each board's block in fourteen -- ran into the DRAM controller's write
// credits: TCC_EA0_WRREQ 9.7 M per launch instead of 3.9 M for the scores alone, a fifth of them 32-byte pieces.)
// The sixteen wavefronts of a workgroup take turns (the caller runs this in board iteration `wavefront number`), so that at any time
// one wavefront per CU is storing planes while fifteen evaluate boards.
__device__ __forceinline__ void density_planes_out(const uint16_t* __restrict__ planes, int n_boards, int first_board, int lane,
                                                   const v4i* __restrict__ s_wtab, int32_t* __restrict__ out_density,
                                                   const uint2* __restrict__ s_lut, int plane_stride) {
    const int n = lane & 31, plane = n & 1, board = first_board + (n >> 1);
    uint32_t own[8], other[8];
    {
        // (columns without a board read the group's first board: what they compute is never stored)
        const uint4* p = reinterpret_cast<const uint4*>(planes + static_cast<size_t>(board < n_boards ? board : first_board) * 32);
        const uint4 a0 = p[plane * 2], a1 = p[plane * 2 + 1], b0 = p[2 - plane * 2], b1 = p[3 - plane * 2];
        own[0] = a0.x; own[1] = a0.y; own[2] = a0.z; own[3] = a0.w; own[4] = a1.x; own[5] = a1.y; own[6] = a1.z; own[7] = a1.w;
        other[0] = b0.x; other[1] = b0.y; other[2] = b0.z; other[3] = b0.w; other[4] = b1.x; other[5] = b1.y; other[6] = b1.z; other[7] = b1.w;
    }
    // cell 224 = (14, 14), which no tile covers: the taps dy, dx in -3 .. 0 that lie on the board, by hand (one lane per column)
    if ((lane >> 5) == 0 && board < n_boards && plane_stride > 0) {
        uint32_t c224 = 0, w224 = 0;
#pragma unroll
        for (int dy = -3; dy <= 0; ++dy) {
            const int y = 14 + dy;
            const uint32_t r = (y & 1) ? own[y >> 1] >> 16 : own[y >> 1] & 0xFFFFu;
#pragma unroll
            for (int dx = -3; dx <= 0; ++dx) {
                if (block_weight(dy, dx) == 0) continue;
                const uint32_t bit = (r >> (14 + dx)) & 1u;
                c224 += bit;
                w224 += bit * static_cast<uint32_t>(block_weight(dy, dx));
            }
        }
        const uint32_t neg = 0u - (((own[7] | other[7]) >> 14) & 1u);             // occupied cells hold -v - 1 = ~v (Pattern.cpp:253-265)
        int32_t* out = out_density + static_cast<size_t>(board) * 4 * plane_stride + (1 - plane) * 2 * plane_stride;
        out[224] = static_cast<int32_t>(c224 ^ neg);
        out[plane_stride + 224] = static_cast<int32_t>(w224 ^ neg);
    }
#define GMK_PASS(K, M) density_tile_pass<K, M>(own, other, n_boards, first_board, lane, s_wtab, out_density, s_lut, plane_stride); __builtin_amdgcn_sched_barrier(0);
    GMK_PASS(0, 0) GMK_PASS(1, 0) GMK_PASS(0, 1) GMK_PASS(1, 1) GMK_PASS(0, 2) GMK_PASS(1, 2) GMK_PASS(0, 3) GMK_PASS(1, 3)
    GMK_PASS(0, 4) GMK_PASS(1, 4) GMK_PASS(0, 5) GMK_PASS(1, 5) GMK_PASS(0, 6) GMK_PASS(1, 6)
#undef GMK_PASS
}

__global__ __launch_bounds__(kThreads)
void eval_positions_kernel(const uint16_t* __restrict__ planes, int n_boards, int n_groups,
                           int32_t* __restrict__ out_scores, int32_t* __restrict__ out_density,
                           uint32_t* __restrict__ out_totals, int32_t* __restrict__ out_status,
                           const uint32_t* __restrict__ g_trans, const uint32_t* __restrict__ g_records,
                           int trans_words, int record_words, const uint32_t* __restrict__ g_wtab,
                           int phase_mask_arg, unsigned long long* __restrict__ prof) {
    // Profiling aids, -DGMK_PROFILE build only (the production build runs every phase and reads no clock): phase_mask bit p runs
    // phase p, bit 6 phase D (bit 8: its planes 1 KB apart, bit 9: passes without stores, bit 10: no passes; bit 11: no compounds =
    // phases 3b and 4 skipped; bit 12: consecutive groups to different workgroups), any mask but 0x7F sets the error bit of every
    // board's status word; prof != nullptr: s_memtime cycles per phase, summed over the wavefronts.
#ifdef GMK_PROFILE
    const int phase_mask = phase_mask_arg;
    unsigned long long t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime();
#define GMK_STAMP(k) do { if (prof) { const unsigned long long t_now_ = __builtin_amdgcn_s_memtime(); t_acc[k] += t_now_ - t_prev; t_prev = t_now_; } } while (0)
#else
    constexpr int phase_mask = 0x7F;
    (void)phase_mask_arg; (void)prof;
#define GMK_STAMP(k) do { } while (0)
#endif
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // layout: [trans (LDS address 0)][records (16-byte aligned)][lane jobs 128][initial line words 96][boards: kBoardsPerBlock * kBoardWords]
    const uint4* s_rec = reinterpret_cast<const uint4*>(lds + trans_words);
    uint32_t* s_jobs = lds + trans_words + record_words;

    for (int i = threadIdx.x; i < trans_words; i += kThreads) lds[i] = g_trans[i];
    for (int i = threadIdx.x; i < record_words; i += kThreads) lds[trans_words + i] = g_records[i];
    if (threadIdx.x < 128) s_jobs[threadIdx.x] = c_lane_jobs[threadIdx.x];
    uint2* s_lut = reinterpret_cast<uint2*>(s_jobs + 128 + kLineWords);      // byte -> its eight bits as bytes (the stone operands of phase D)
    if (threadIdx.x < 256) s_lut[threadIdx.x] = make_uint2(((threadIdx.x & 15u) * 0x204081u) & 0x01010101u, ((threadIdx.x >> 4) * 0x204081u) & 0x01010101u);
    uint32_t* s_wtab_words = s_jobs + 128 + kLineWords + 512;
    for (int i = threadIdx.x; i < kWtabWords; i += kThreads) s_wtab_words[i] = g_wtab[i];
    const v4i* s_wtab = reinterpret_cast<const v4i*>(s_wtab_words);
    uint32_t* s_handout = s_wtab_words + kWtabWords;        // [0] boards handed out, [1] density bursts handed out
    if (threadIdx.x < 4) s_handout[threadIdx.x] = threadIdx.x == 0 ? 2u * kBoardsPerBlock : 0u;      // (every wavefront's first two boards are its own: see below)

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane0 = threadIdx.x & 63;
    uint32_t* s_scores = lds + trans_words + record_words + kStaticTableWords + wave * kBoardWords;      // int32 scores, accumulated with ds_add
    uint32_t* s_cnt = s_scores + kScoreWords;
    uint32_t* s_lines = s_scores + kZeroWords;
    uint32_t* s_queue = s_lines + kLineWords;
    uint32_t* s_misc = s_queue + kQueueCap;
#ifndef GMK_K1_TOTALS_HOT
    uint32_t* const my_totals = s_queue + kQueueCap - 8 * kTotalsCopies + (lane0 & (kTotalsCopies - 1)) * 8;  // this lane's copy of the per-type totals (phase 0 says why)
#else
    uint32_t* const my_totals = s_misc + 4;
#endif

    // (no barrier here: the tables are on their way while every wavefront runs phase 0 of its first board, which needs none of them; the one
    // barrier of the kernel stands in front of that board's phase 1)
    GMK_STAMP(0);

    // ---- work distribution: see the hand-out counters below ----
    const bool planes_out = out_density != nullptr && (phase_mask & 64) && !(phase_mask & 1024);
    int lane = lane0;
    // (from constant memory, not from their staged copies: those are not there yet)
    const uint32_t job_a = c_lane_jobs[lane * 2], job_b = c_lane_jobs[lane * 2 + 1];
    const uint32_t line_init_lo = c_line_init[lane], line_init_hi = c_line_init[min(64 + lane, kLineWords - 1)];
    // where a line's first cell sits in its word: bit 2 x0 for a diagonal, 2 y0 for an anti-diagonal (first cell = job bits 12..19), 0 for rows and columns
    auto first_bit = [](uint32_t job) -> uint32_t {
        const uint32_t dir = (job >> 5) & 3u, first = (job >> 12) & 255u, y0 = (first * 0x8889u) >> 19, x0 = first - 15u * y0;
        return dir == 2u ? 2u * x0 : dir == 3u ? 2u * y0 : 0u;
    };
    const uint32_t norm_a = first_bit(job_a), norm_b = first_bit(job_b);
    uint32_t* s_rows = s_misc + 16;                      // row y at [3 + y]
    // the cells this lane owns in the four passes over the board (cell = 64 pass + lane): 4 x row and column, a byte per pass
    uint32_t cell_row4 = 0, cell_col = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int c = min(64 * pass + lane, kCells - 1), y = (c * 0x8889) >> 19;
        cell_row4 |= static_cast<uint32_t>(4 * y) << (8 * pass);
        cell_col |= static_cast<uint32_t>(c - 15 * y) << (8 * pass);
    }
    // a board's 64 B are fetched while the board before it is evaluated.  No branch around the loads (every lane reads some valid
    // row, the result is masked where it is used), and the two halves stay apart until they are used: any arithmetic on a loaded
    // value makes the compiler wait for it on the spot, and the wait covers every store issued before.
    const uint16_t* row_ptr = planes + (lane & 15);
    uint32_t next_black = 0, next_white = 0, cur_black = 0, cur_white = 0;
    auto fetch_row = [&](int b) {
        const uint16_t* p = row_ptr + static_cast<size_t>(min(b, n_boards - 1)) * 32;
        next_black = p[0];
        next_white = p[16];
    };
    auto take_row = [&](int b) -> uint32_t { return lane < 16 && b < n_boards ? cur_black | (cur_white << 16) : 0u; };

    // The workgroup owns one contiguous run of groups (what a CU writes at any time lies within a few hundred kilobytes) and hands its
    // boards out one at a time from a counter in LDS: the sixteen wavefronts -- four per SIMD, and a SIMD's vector unit is what the kernel
    // is bound by -- finish within one board of each other instead of 16 boards x (the spread of a board's work, ~ +-25 %) apart.
    // The density bursts of its groups are handed out the same way, one per wavefront and sixteen boards, wavefront w in its w-th board.
    // (Round 2 measured a dynamic hand-out -- chunks of four, stealing round a ring of workgroups -- as no gain and concluded the wavefronts'
    // waits were the limit.  They are not: tools/valu_probe.hip shows a wave-instruction costs a SIMD 4 cycles whatever its type, K1's ~950
    // per board make the vector unit ~90 % busy while all sixteen wavefronts run, and what the static hand-out lost was whole SIMDs idling
    // behind the slowest one: 0.1544 -> 0.1402 ms.)
    const int wg_g0 = static_cast<int>(static_cast<long long>(n_groups) * blockIdx.x / gridDim.x);
    const int wg_groups = static_cast<int>(static_cast<long long>(n_groups) * (blockIdx.x + 1) / gridDim.x) - wg_g0;
    const int wg_first = wg_g0 * kGroupBoards, wg_boards = min(wg_groups * kGroupBoards, n_boards - wg_first);
    auto hand_out = [&](int which) -> int {
        uint32_t v = 0;
        if (lane0 == 0) v = atomicAdd(&s_handout[which], 1u);
        return __builtin_amdgcn_readfirstlane(static_cast<int>(v));
    };
    // A wavefront's first two boards are fixed (boards w and 16 + w of the workgroup's run; the counter starts behind them): the first dynamic hand-out
    // then comes after the kernel's one barrier.
    int idx = wave, boards_done = 0;
    if (idx < wg_boards) fetch_row(wg_first + idx);
    asm volatile("" : "+v"(next_black), "+v"(next_white));        // (once: wait for them here)
#pragma unroll 1
    for (;;) {
        const bool live = idx < wg_boards;
        const int board = wg_first + idx;
        cur_black = next_black; cur_white = next_white;
        int idx_next = idx;
        if (live) { idx_next = boards_done == 0 ? kBoardsPerBlock + wave : hand_out(0); fetch_row(wg_first + min(idx_next, wg_boards - 1)); }
        GMK_STAMP(11);

        if (live) {
            // ---- phase 0: clear accumulators, take the two bit-planes (64 B), turn them into line words ----
            {
                // (the counters; the score block is written below, with the area bonus in it)
                uint4* z = reinterpret_cast<uint4*>(s_cnt) + lane;
    #pragma unroll
                for (int i = 0; i < kCntWords / 4; i += 64)
                    if (i + 64 <= kCntWords / 4 || lane < kCntWords / 4 - i) z[i] = make_uint4(0u, 0u, 0u, 0u);
            }
            s_lines[lane] = line_init_lo;                       // (the all-blank line words wait in two registers, not in LDS: no read before the write)
            if (lane < kLineWords - 64) s_lines[64 + lane] = line_init_hi;
            if (lane < kMiscWords) s_misc[lane] = 0;
#ifndef GMK_K1_TOTALS_HOT
            // Per-type totals: one atomic per match on eight addresses is 64 lanes on eight LDS words -- up to ~25 of them on the same one, served
            // one after the other (phase 2 is 41 % of the LDS pipe's cycles, profiles/r04_k1_lds_phases.txt).  kTotalsCopies copies, a lane adds to copy
            // lane & (kTotalsCopies - 1); they live in the last words of the transition queue (a board that is not flagged queues < kQueueCap - 64 entries) and are
            // summed behind phase 2.
            if (lane < 8 * kTotalsCopies) s_queue[kQueueCap - 8 * kTotalsCopies + lane] = 0;
#endif
            const uint32_t my_row = take_row(board);
            wave_phase_fence();
            {
                // stones per colour: a row reduction in registers (an atomicAdd of fifteen lanes on one word is turned by the compiler
                // into a serial loop over those lanes)
                uint32_t cnt = static_cast<uint32_t>(__popc(my_row & 0x7FFFu)) | (static_cast<uint32_t>(__popc(my_row >> 16)) << 16);
                cnt += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(cnt), 0x118, 0xF, 0xF, false));       // row_shr:8
                cnt += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(cnt), 0x114, 0xF, 0xF, false));       // row_shr:4
                cnt += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(cnt), 0x112, 0xF, 0xF, false));       // row_shr:2
                cnt += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(cnt), 0x111, 0xF, 0xF, false));       // row_shr:1
                if (lane == 15) s_misc[0] = cnt;                // lane 15 holds the sum of lanes 0 .. 15 (my_row is zero in lane 15)
            }
            {
                // Where is a colour's density count positive (<=> its weight positive: the +160 of Pattern.cpp:268)?  Wherever a stone
                // of the colour lies under the 7x7 BlockWeights mask (Pattern.cpp:598-609) around the cell: the rows, dilated by the
                // mask's row patterns -- 1001001 three rows away, 0111110 one and two rows away, 1110111 in the row itself -- and OR-ed
                // over the seven rows, for both colours at once (black in the low, white in the high half word), restricted to empty cells.
                // (a shift must not carry bits from one half word into the other: they are cut off first)
                const uint32_t r = my_row;
                const uint32_t l1 = (r & 0x3FFF3FFFu) << 1, l2 = (r & 0x1FFF1FFFu) << 2, l3 = (r & 0x0FFF0FFFu) << 3;
                const uint32_t r1 = (r & 0x7FFE7FFEu) >> 1, r2 = (r & 0x7FFC7FFCu) >> 2, r3 = (r & 0x7FF87FF8u) >> 3;
                const int far = static_cast<int>(r | l3 | r3), near = static_cast<int>(r | l1 | l2 | r1 | r2);
                uint32_t g = l1 | l2 | l3 | r1 | r2 | r3;
                g |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, near, 0x111, 0xF, 0xF, true)) | static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, near, 0x101, 0xF, 0xF, true));     // rows y -+ 1
                g |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, near, 0x112, 0xF, 0xF, true)) | static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, near, 0x102, 0xF, 0xF, true));     // rows y -+ 2
                g |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, far, 0x113, 0xF, 0xF, true)) | static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, far, 0x103, 0xF, 0xF, true));       // rows y -+ 3
                const uint32_t empty = ~(r | (r >> 16)) & 0x7FFFu;
                if (lane < 15) s_rows[3 + lane] = r;            // (three zero rows on either side: the density gate of phase 3b reads rows y - 3 .. y + 3 unchecked)
                // The score block starts at the area bonus instead of zero: +160 in a colour's own view where its gate bit is set
                // (Pattern.cpp:268), one 16-byte write per cell [white own, white opp, black opp, black own]; the gate word of the
                // cell's row comes from that row's lane.
                const int gate = static_cast<int>(g & (empty | (empty << 16)));
    #pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const uint32_t gw = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>((cell_row4 >> (8 * pass)) & 0xFFu), gate)) >> ((cell_col >> (8 * pass)) & 0xFFu);
                    if (pass < 3 || lane < kCells - 192)
                        reinterpret_cast<uint4*>(s_scores)[64 * pass + lane] = make_uint4(((gw >> 16) & 1u) * 160u, 0u, 0u, (gw & 1u) * 160u);
                }
            }
            {
                // a stone turns its cell's blank (3) into black (0) or white (1) in the four lines through it: one XOR each.
                // Four lanes share a row (cells 0-3, 4-7, 8-11, 12-14), so the loop runs as long as the fullest quarter row.
                const int y = lane >> 2, part = lane & 3;
                const uint32_t row = __shfl(my_row, min(y, 14));
                uint32_t row_sym = 0;
                for (uint32_t m = lane < 60 ? (row | (row >> 16)) & (0xFu << (4 * part)) & 0x7FFFu : 0u; m; m &= m - 1u) {
                    const int x = __ffs(m) - 1;
                    const uint32_t code = ((row >> x) & 1u) ? 3u : 2u;
                    const uint32_t at_x = code << (2 * x), at_y = code << (2 * y);
                    row_sym |= at_x;
                    atomicXor(&s_lines[kColBase + x], at_y);
                    atomicXor(&s_lines[kDiagBase + x - y + 14], at_x);
                    atomicXor(&s_lines[kAntiBase + x + y], at_y);
                }
                if (row_sym) atomicXor(&s_lines[y], row_sym);
            }
            wave_phase_fence();
            GMK_STAMP(1);
        }
        if (boards_done == 0) __syncthreads();                  // the tables are staged (the automaton at LDS address 0, the records, jobs, the tables of phase D,
                                                                // the hand-out counters): from here on the wavefronts never wait for each other
        if (live) {
            // ---- phase 1: walk the DFA along this lane's lines; transitions that emit go to the queue ----
            // The lane's one or two lines become ONE stream of 2-bit DFA symbols:
            //   '?' cells '?' '?'  ['?' cells '?' '?']  '?' '?' ...
            // (after "??" the automaton sits in its '?' self-loop state, which the next line's leading '?' keeps: no reset
            // between the two lines).  The stream is kept shifted left by 2, so a step is: symbol * 4 = low word & 12, LDS
            // address = (previous word's next-row offset) | symbol * 4, one lookup.  Emitting transitions are queued raw
            // (record, lane, step); the slot is a ballot prefix (this wave is the only producer), decoding happens in phase 2.
            int n_queued = 0;                                       // wave-uniform
            if (phase_mask & 2) {
                const int len_a = static_cast<int>(job_a & 31u) - 3, len_b = static_cast<int>(job_b & 31u) - 3;
                uint64_t syms = line_symbols(s_lines[(job_a >> 20) & 127u] >> norm_a, len_a);      // (a diagonal's word shifted down to its first cell)
                syms |= line_symbols(s_lines[(job_b >> 20) & 127u] >> norm_b, len_b) << (2 * len_a + 6);
                syms |= 0xAAAAAAAAAAAAAAAAull << (2 * (len_a + len_b) + 12);
                // The stream is kept shifted left by 2: symbol s at bits 2 s + 2, so that the four bits from 2 s up are (symbol s - 1, symbol s) and
                // one v_bfe + one v_bfi make the lookup address: (table word & 0x3FF3) | (those four bits & 12) -- the row offsets are multiples of 16.
                const uint32_t cur = static_cast<uint32_t>(syms) << 2;        // symbols 0..14 at bits 2..31
                const uint32_t rest = static_cast<uint32_t>(syms >> 28);      // symbols 15.. at bits 2.. (bits 0, 1: symbol 14)
                uint32_t tw = 0;                                    // the previous step's table word (row 0 = root)
                // The emission of step s is queued BEHIND the lookup of step s + 1 (the next address needs the table word only): the
                // prefix count and the queue write then run in the shadow of that lookup's LDS round trip.  A queue entry is the table
                // word itself with the low 17 bits (next row, kinds) replaced by lane | step << 6: the record number stays where it is.
                // Where the queue stands is a byte address kept by the scalar unit; a lane adds its prefix count (the vector unit is
                // what this kernel is bound by: 4 cycles per wave-instruction on a SIMD whatever the lanes do).
                const uint32_t q_base = static_cast<uint32_t>(s_queue - lds) * 4u;     // (the kernel's LDS starts at address 0)
                // The scalar unit keeps the fill level at most 64 entries below the capacity: a step's (at most 64) entries then fit whatever the
                // lanes add, without a per-lane clamp (a vector instruction per step); a board that gets there is flagged below.
                const uint32_t q_full = q_base + 4u * (kQueueCap - 64);
                uint32_t q_at = q_base;                             // wave-uniform
                auto push = [&](uint32_t word, int step) {
                    const bool emits = word > 0x1FFFFu;
                    const unsigned long long emitters = __ballot(emits);
                    if (emitters) {
                        if (emits) {
                            const uint32_t ahead = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(emitters >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(emitters), 0u));
                            *lds_word_rw(q_at + 4u * ahead) = (word & ~0x1FFFFu) | (static_cast<uint32_t>(lane) | static_cast<uint32_t>(step) << 6);
                        }
                        q_at = min(q_at + 4u * static_cast<uint32_t>(__popcll(emitters)), q_full);
                    }
                };
#pragma unroll
                for (int step = 0; step < kScanSteps; ++step) {
                    const uint32_t four = step < 15 ? __builtin_amdgcn_ubfe(cur, 2 * step, 4) : __builtin_amdgcn_ubfe(rest, 2 * (step - 15), 4);
                    const uint32_t addr = dfa_address(tw, four);
                    const uint32_t before = tw;
                    tw = *lds_word(addr);                       // (the table is at LDS address 0: the address is used as it is)
                    if (step > 0) push(before, step - 1);
                }
                push(tw, kScanSteps - 1);
                n_queued = static_cast<int>((q_at - q_base) >> 2);
                if (q_at >= q_full) s_misc[2] = 1;                  // (flagged at kQueueCap - 64 entries: 2.5 x the most a board of the test sets queues)
            }
            wave_phase_fence();
            GMK_STAMP(2);

            // ---- phase 2: one lane per emitting transition: the score deposits of its 1-2 matches ----
            if (phase_mask & 4) {
                uint32_t qe_next = s_queue[min(lane, kQueueCap - 1)];      // (entry: record number << 17 | step << 6 | lane)
                for (int m = lane; m < n_queued; m += 64) {
                    const uint32_t qe = qe_next;
                    qe_next = s_queue[min(m + 64, kQueueCap - 1)];      // (the next round's entry is on its way while this one is worked on;
                                                                         //  fetching its record and line jobs ahead as well measured no gain)
                    // which line of which lane, and where on it (symbol 0 of a line's segment is its leading pad)
                    const int src_lane = qe & 63, src_step = (qe >> 6) & 31;
                    const uint4 rec_now = s_rec[qe >> 17];
                    const uint32_t ja_now = s_jobs[src_lane * 2], jb_now = s_jobs[src_lane * 2 + 1];
                    const int seg_a = ja_now & 31u;
                    const bool second = src_step >= seg_a;
                    const uint32_t job = second ? jb_now : ja_now;
                    const int pos = src_step - 1 - (second ? seg_a : 0);
                    const int dir = (job >> 5) & 3, stride = (job >> 7) & 31;
                    const int cell_at = static_cast<int>((job >> 12) & 255u) + pos * stride;
                    deposit_match(rec_now.x, rec_now.y, cell_at, dir, -stride, s_scores, s_cnt, s_misc, my_totals);
                    if (rec_now.z) deposit_match(rec_now.z, rec_now.w, cell_at, dir, -stride, s_scores, s_cnt, s_misc, my_totals);
                }
            }
            wave_phase_fence();
#ifndef GMK_K1_TOTALS_HOT
            // the copies of the per-type totals (see my_totals) become the totals row; their place is the queue's again from phase 3b on
            if (lane < 8) {
                const uint32_t* t = s_queue + kQueueCap - 8 * kTotalsCopies + lane;
                uint32_t sum = 0;
#pragma unroll
                for (int c = 0; c < kTotalsCopies; ++c) sum += t[8 * c];
                s_misc[4 + lane] = sum;
            }
#endif
            GMK_STAMP(3);

            // ---- phase 3: one lane per cell: compound candidates (the area bonus is in the block since phase 0) ----
            int n_cand = 0;                                         // wave-uniform
            if (phase_mask & 8) {
                // (the reads of all four passes first: four round trips in a row are four waits)
                uint32_t any_cnt[4];
    #pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int qc = min(64 * pass + lane, kCells - 1);               // the last pass has 33 cells
                    any_cnt[pass] = s_cnt[qc] | s_cnt[kCells + qc] | s_cnt[2 * kCells + qc];
                }
    #pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int q = 64 * pass + lane;
                    // compound candidates (Compound::Test, Pattern.cpp:424-433): cells whose LiveThree / DeadThree / LiveTwo '_' counters, each
                    // clipped to 2 (the reference's 2-bit shift flags), OR-ed over the types, sum to two or more over the directions (only
                    // empty cells have counters: a '_' piece is a blank).  Decided in phase 3b, with the density gate of Pattern.cpp:182.
                    const uint32_t any = any_cnt[pass];
                    const uint32_t upper = (any >> 1) | (any >> 2) | (any >> 3);
                    // one nibble per (colour, direction): 0, 1 or 2 = the clipped count; the nibbles of a colour summed by one multiplication
                    const uint32_t clipped = ((any | upper) & 0x11111111u) + (upper & 0x11111111u);
                    uint32_t cand = 0;
                    // (24-bit multiplies: bits 12..15 of the product depend on the low sixteen bits of the factor only, and a full 32-bit multiply --
                    // which the compiler picks once it has dropped the mask -- runs at a quarter of the rate)
                    if ((mul24(clipped, 0x1111u) & 0xF000u) >= 0x2000u) cand |= 1u;
                    if ((mul24(clipped >> 16, 0x1111u) & 0xF000u) >= 0x2000u) cand |= 2u;
                    if (pass == 3 && q >= kCells) cand = 0u;
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
            }
            wave_phase_fence();
            GMK_STAMP(4);

            // ---- phase 3b: one lane per candidate cell and colour: the compound decision (Pattern.cpp:440-486), critical-point deposits,
            //      counter-move rescans queued in the upper half of the queue ----
            if (n_cand > kQueueCap / 2) { s_misc[2] = 1; n_cand = kQueueCap / 2; }
            if (phase_mask & 2048) n_cand = 0;
            if (phase_mask & 8)
            for (int v = lane; v < 2 * n_cand; v += 64) {                      // one lane per (candidate cell, colour): 0 white, 1 black
                const uint32_t ce = s_queue[v >> 1];
                const int q = ce & 255, c = v & 1;
                if (!((ce >> (8 + c)) & 1u)) continue;
                // this colour's four direction nibbles of the three counter words (read together with the rows below: one round trip)
                const uint32_t f3 = (s_cnt[q] >> (16 * c)) & 0xFFFFu, fd = (s_cnt[kCells + q] >> (16 * c)) & 0xFFFFu, f2 = (s_cnt[2 * kCells + q] >> (16 * c)) & 0xFFFFu;
                // the density gate (Pattern.cpp:182): the colour's density COUNT at the cell must be two or more: its stones under the
                // non-zero cells of the 7x7 BlockWeights mask around q, counted from the rows (black low, white high half word)
                uint32_t dens = 0;
                {
                    const int qy = (q * 0x8889) >> 19, qx = q - 15 * qy, half = c ? 0 : 16;
    #pragma unroll
                    for (int dy = -3; dy <= 3; ++dy) {
                        const uint32_t pattern = dy == 0 ? 0x77u : (dy == 3 || dy == -3) ? 0x49u : 0x3Eu;      // 1110111, 1001001, 0111110
                        const uint32_t mask = ((pattern << qx) >> 3) & 0x7FFFu;
                        dens += __popc((s_rows[3 + qy + dy] >> half) & mask);
                    }
                }
                if (dens < 2u) continue;
                // The reference walks the directions in order through a state machine S0, L2, LD3, To33, To43, To44 (Pattern.cpp:440-486); in
                // each direction the component is the first of LiveThree, DeadThree, LiveTwo with a non-zero counter, taken once or twice (the
                // counters count like its 2-bit shift flags: 0, 1, 2 or more, Pattern.cpp:395-400).  Its outcome does not depend on the order:
                // with n components of which s are threes (weight 2, a LiveTwo 1), the state ends at 3 + min(2, s) for n >= 2 -- the first two
                // transitions add w1 + w2 + 1, every further one w - 1, capped at 5 -- and the "triple" flag is n >= 3.  So: nibble masks.
                const uint32_t up3 = (f3 >> 1) | (f3 >> 2) | (f3 >> 3), upd = (fd >> 1) | (fd >> 2) | (fd >> 3), up2 = (f2 >> 1) | (f2 >> 2) | (f2 >> 3);
                const uint32_t sel3 = (f3 | up3) & 0x1111u;                                          // directions whose component is a LiveThree
                const uint32_t seld = (fd | upd) & 0x1111u & ~sel3;                                  // ... a DeadThree
                const uint32_t sel2 = (f2 | up2) & 0x1111u & ~(sel3 | seld);                         // ... a LiveTwo
                const uint32_t strong = sel3 | seld, any_dir = strong | sel2;
                const uint32_t twice_strong = (sel3 & up3) | (seld & upd), twice = twice_strong | (sel2 & up2);     // taken twice (counter >= 2)
                const int n_comp = __popc(any_dir) + __popc(twice), threes = __popc(strong) + __popc(twice_strong);
                if (n_comp < 2) { s_misc[2] = 1; continue; }                                         // reference reads out of bounds here
                const int ctype = min(threes, 2);
                atomicAdd(&s_misc[12 + ctype], c ? 0x10000u : 1u);
                // updateCritical, both perspectives: the colour's pair of the cell
                const uint32_t crit = 600u * static_cast<uint32_t>(n_comp);
                atomicAdd(reinterpret_cast<unsigned long long*>(s_scores) + 2 * q + c, (static_cast<unsigned long long>(crit) << 32) | crit);
                if (n_comp >= 3 || sel3) continue;                                                   // a triple cross, or a live three among them
                // exactly two components, in direction order: queue their counter-move rescans
                const int d1 = (__ffs(any_dir) - 1) >> 2;
                const uint32_t others = any_dir & (any_dir - 1u);
                const int d2 = others ? (__ffs(others) - 1) >> 2 : d1;
                const uint32_t cd1 = static_cast<uint32_t>(d1) | (((seld >> (4 * d1)) & 1u) ? 4u : 8u), cd2 = static_cast<uint32_t>(d2) | (((seld >> (4 * d2)) & 1u) ? 4u : 8u);
                const uint32_t slot = atomicAdd(&s_misc[3], 2u);
                const uint32_t head = static_cast<uint32_t>(q) | (static_cast<uint32_t>(c) << 8);
                if (slot < kQueueCap / 2) s_queue[kQueueCap / 2 + slot] = head | (cd1 << 9);
                if (slot + 1 < kQueueCap / 2) s_queue[kQueueCap / 2 + slot + 1] = head | (cd2 << 9);
                else s_misc[2] = 1;
            }
            wave_phase_fence();
            GMK_STAMP(5);

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
                        const int first_at = dir == 2 ? 2 * (x - at) : dir == 3 ? 2 * (y - at) : 0;       // the line's first cell in its word (diagonals: bits 2 x / 2 y)
                        const int len = dir < 2 ? 15 : dir == 2 ? 15 - abs(diag - 14) : min(anti, 28 - anti) + 1;
                        // six '?' | cells | six '?', then the 13 symbols starting six before q
                        const uint64_t syms = (0xAAAull | (static_cast<uint64_t>(s_lines[line] >> first_at) << 12) | (0xAAAull << (2 * len + 12))) >> (2 * at);
                        const int start = k > 7 ? k - 7 : 0;
                        const uint32_t cur = static_cast<uint32_t>(syms >> (2 * start)) << 2;      // kept shifted left by 2 as in phase 1
                        uint32_t tw = 0;
    #pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (start + i <= k) tw = *lds_word(dfa_address(tw, __builtin_amdgcn_ubfe(cur, 2 * i, 4)));      // v_bfe + v_bfi, as in phase 1
                        if ((gmk::dev_trans_kinds(tw) >> tslot) & 1u) {                     // the record holds the wanted type
                            const uint4 rec = s_rec[gmk::dev_trans_record(tw)];
                            hit_back = counter_match(rec.x, k, want);
                            hit_w0 = rec.x;
                            if (hit_back < 0) { hit_back = counter_match(rec.z, k, want); hit_w0 = rec.z; }
                        }
                    }
                    const unsigned long long hits = __ballot(hit_back >= 0);
                    const uint32_t mine = static_cast<uint32_t>(hits >> (lane & ~7)) & 0xFFu;
                    if (hit_back >= 0 && (mine & ((1u << kk) - 1u)) == 0u) add_counter_cells(hit_w0, hit_back, q, stride, s_scores + (c ? 2 : 1));
                }
            }
            wave_phase_fence();
            GMK_STAMP(6);

        }

        asm volatile("" : "+v"(next_black), "+v"(next_white));    // the wait for the next board's planes (a use the compiler must honour)
        GMK_STAMP(7);

        // ---- phase D: density planes of the workgroup's groups: one burst in this wavefront's turn (its w-th board of every sixteen),
        //      and whatever is left once the boards have run out ----
        if (planes_out && (!live || (boards_done & 15) == wave)) {
            for (;;) {
                const int burst = hand_out(1);
                if (burst >= wg_groups) break;
                int lane_p = lane0, first_p = (wg_g0 + burst) * kGroupBoards;      // opaque copies: what the passes derive from them is computed here
                asm volatile("" : "+v"(lane_p), "+s"(first_p));
                density_planes_out(planes, n_boards, first_p, lane_p, s_wtab, out_density, s_lut, (phase_mask & 512) ? 0 : (phase_mask & 256) ? 256 : kCells);
                GMK_STAMP(8);
                if (live) break;
            }
        }
        if (!live) break;
        // ---- phase 5: results leave LDS ----
        if (live && (phase_mask & 32)) {
            int lane_b = lane;                                  // (a copy the compiler cannot see through: the store addresses are computed here,
            asm volatile("" : "+v"(lane_b));                    // not kept in registers across the density pass)
            if (out_scores) {
                // the block is [cell][group] in LDS and [group][cell] in memory: a lane reads the sixteen bytes of its cell and writes one
                // word into each group's plane, 256 contiguous bytes per store instruction
                int32_t* dst = out_scores + static_cast<size_t>(board) * kScoreWords + lane_b;
                const int4* src = reinterpret_cast<const int4*>(s_scores);
                // (all reads first: one LDS round trip instead of four)
                const int4 v0 = src[lane_b], v1 = src[lane_b + 64], v2 = src[lane_b + 128], v3 = src[min(lane_b + 192, kCells - 1)];
#ifdef GMK_K1_PLAIN_SCORES
#define GMK_STORE(p, v) (*(p) = (v))
#else
#define GMK_STORE(p, v) __builtin_nontemporal_store(v, p)       // (non-temporal: 0.1575 -> 0.1543 ms)
#endif
                // in ADDRESS order (the board's 3 600 bytes are one contiguous run: piece after piece, so that the two parts of a cache line that two
                // pieces share reach the L2 back to back)
                const bool tail = lane_b + 192 < kCells;
                GMK_STORE(dst, v0.x); GMK_STORE(dst + 64, v1.x); GMK_STORE(dst + 128, v2.x); if (tail) GMK_STORE(dst + 192, v3.x);
                GMK_STORE(dst + kCells, v0.y); GMK_STORE(dst + kCells + 64, v1.y); GMK_STORE(dst + kCells + 128, v2.y); if (tail) GMK_STORE(dst + kCells + 192, v3.y);
                GMK_STORE(dst + 2 * kCells, v0.z); GMK_STORE(dst + 2 * kCells + 64, v1.z); GMK_STORE(dst + 2 * kCells + 128, v2.z); if (tail) GMK_STORE(dst + 2 * kCells + 192, v3.z);
                GMK_STORE(dst + 3 * kCells, v0.w); GMK_STORE(dst + 3 * kCells + 64, v1.w); GMK_STORE(dst + 3 * kCells + 128, v2.w); if (tail) GMK_STORE(dst + 3 * kCells + 192, v3.w);
#undef GMK_STORE
            }
            if (out_totals && lane_b < 11) out_totals[static_cast<size_t>(board) * 11 + lane_b] = s_misc[4 + lane_b];
            if (out_status && lane_b == 0) {
                const uint32_t wbits = s_misc[1], stones = s_misc[0];
                const int stones_b = stones & 0xFFFFu, stones_w = stones >> 16;
                // the side that completed five is the only one that can own a Five (the game stops there)
                const int winner = (wbits & 1u) ? 1 : (wbits & 2u) ? -1 : 0;
                const bool over = winner != 0 || stones_b + stones_w == kCells;
                const int to_move = over ? 0 : (stones_b == stones_w ? 1 : -1);
                out_status[board] = (over ? 1 : 0) | ((s_misc[2] || phase_mask != 0x7F) ? 2 : 0) | ((winner & 0xFF) << 8) | ((to_move & 0xFF) << 16);
            }
        }
        wave_phase_fence();
        GMK_STAMP(9);
        idx = idx_next; ++boards_done;
    }
#ifdef GMK_PROFILE
    // [0] table staging, [1] phase 0, [2] scan, [3] deposits, [4] phase 3, [5] phase 3b, [6] rescans, [7] wait for the next planes,
    // [8] phase D, [9] phase 5, [11] loop head; [15] wavefronts
    if (prof && lane0 == 0) {
        for (int k = 0; k < 12; ++k) atomicAdd(&prof[k], t_acc[k]);
        atomicAdd(&prof[15], 1ull);
    }
#endif
#undef GMK_STAMP
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
    for (uint32_t& j : jobs) j = 3u | (15u << 20);                // "no line": length 0 (three pads) on a line word that stays zero
    auto pack = [](const LineJob& l) {
        const int line = l.dir == 0 ? l.y0 : l.dir == 1 ? kColBase + l.x0 : l.dir == 2 ? kDiagBase + l.x0 - l.y0 + 14 : kAntiBase + l.x0 + l.y0;
        const int stride = l.dir == 0 ? 1 : l.dir == 1 ? 15 : l.dir == 2 ? 16 : 14;
        return static_cast<uint32_t>((l.len + 3) | (l.dir << 5) | (stride << 7) | ((l.y0 * 15 + l.x0) << 12) | (line << 20));
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
    // (a diagonal's cells sit at bits 2 x, an anti-diagonal's at bits 2 y: the line starts at x = max(0, x - y) resp. y = max(0, x + y - 14))
    for (int d = 0; d <= 28; ++d) init[kDiagBase + d] = init[kAntiBase + d] = ((1u << (2 * (15 - std::abs(d - 14)))) - 1u) << (2 * std::max(0, d - 14));
    GMK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_line_init), init, sizeof init));
    GMK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_lane_jobs), jobs, sizeof jobs));
    if (steps != kScanSteps) { gmk::set_error("lane jobs: %d scan steps, the kernel is built for %d", steps, kScanSteps); return GMK_ERR_STATE; }
    return GMK_OK;
}

// The weight table of phase D (kWtab*): [plane kind: count, weight][dy + 6][xo] x 16 bytes, byte j = the tap from column j of a row
// dy below the cell's to a cell in column xo: BlockWeights (Pattern.cpp:598-609) at (dy, j - xo), its non-zero mask for the count planes.
int upload_density_weights(uint32_t** d_out) {
    std::vector<int8_t> t(static_cast<size_t>(kWtabWords) * 4, 0);
    for (int kind = 0; kind < 2; ++kind)
        for (int dy = -3; dy <= 3; ++dy)
            for (int xo = 0; xo < 15; ++xo)
                for (int j = 0; j < 15; ++j) {
                    if (std::abs(j - xo) > 3) continue;
                    const int w = block_weight(dy, j - xo);
                    t[((static_cast<size_t>(kind) * kWtabDy + dy + 6) * 15 + xo) * 16 + j] = static_cast<int8_t>(kind == 0 ? (w != 0) : w);
                }
    // every row offset a lane can ask for lies inside the table: dy = 2 kt + h - yo for the K tiles its M tile visits
    for (int cell = 0; cell < 224; ++cell)
        for (int kt = dens_kt_lo(cell / 32); kt <= dens_kt_hi(cell / 32); ++kt)
            for (int h = 0; h < 2; ++h) {
                const int dy = 2 * kt + h - cell / 15;
                if (dy < -6 || dy > 6) { gmk::set_error("density weight table: dy %d of cell %d out of range", dy, cell); return GMK_ERR_STATE; }
            }
    // ... and every tap of every cell lies in one of the K tiles its M tile visits
    for (int cell = 0; cell < 224; ++cell)
        for (int dy = -3; dy <= 3; ++dy) {
            const int y = cell / 15 + dy, m = cell / 32;
            if (y < 0 || y > 14) continue;
            if (y / 2 < dens_kt_lo(m) || y / 2 > dens_kt_hi(m)) { gmk::set_error("density weight table: row %d of cell %d is not covered", y, cell); return GMK_ERR_STATE; }
        }
    GMK_HIP_CHECK(hipMalloc(d_out, t.size()));
    GMK_HIP_CHECK(hipMemcpy(*d_out, t.data(), t.size(), hipMemcpyHostToDevice));
    return GMK_OK;
}

struct Launch { int grid, n_groups; size_t lds; };

Launch plan_launch(int n, const gmk::DeviceState& st) {
    Launch l;
    l.n_groups = (n + kGroupBoards - 1) / kGroupBoards;
    l.grid = std::max(1, std::min(l.n_groups, st.cu_count * kMaxBlocksPerCu));
    l.lds = static_cast<size_t>(kBoardsPerBlock * kBoardWords + st.n_states * 4 + st.n_records * 4 + kStaticTableWords) * 4;
    return l;
}

bool g_jobs_uploaded = false;
uint32_t* g_wtab = nullptr;
}  // namespace

extern "C" int gmk_eval_batch(const uint16_t* d_planes, int n, int32_t* d_scores, int32_t* d_density,
                              uint32_t* d_totals, int32_t* d_status, void* stream) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (n < 0 || (n > 0 && !d_planes)) { gmk::set_error("gmk_eval_batch: bad arguments"); return GMK_ERR_ARG; }
    if (n == 0) return GMK_OK;
    if (!g_jobs_uploaded) {
        int rc = upload_lane_jobs();
        if (rc != GMK_OK) return rc;
        rc = upload_density_weights(&g_wtab);
        if (rc != GMK_OK) return rc;
        GMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(eval_positions_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        g_jobs_uploaded = true;
    }
    const Launch l = plan_launch(n, st);
    if (l.lds > 160u * 1024u) { gmk::set_error("gmk_eval_batch: tables do not fit in LDS (%zu bytes)", l.lds); return GMK_ERR_CAPACITY; }
    static const int phase_mask = gmk::profile_env("GMK_EVAL_PHASE_MASK") ? std::atoi(gmk::profile_env("GMK_EVAL_PHASE_MASK")) : 0x7F;
    unsigned long long* d_prof = nullptr;
    static const bool stamps = gmk::profile_env("GMK_EVAL_PROFILE") != nullptr;
    if (stamps) {
        GMK_HIP_CHECK(hipMalloc(&d_prof, 16 * sizeof(unsigned long long)));
        GMK_HIP_CHECK(hipMemset(d_prof, 0, 16 * sizeof(unsigned long long)));
    }
    hipLaunchKernelGGL(eval_positions_kernel, dim3(l.grid), dim3(kThreads), l.lds, static_cast<hipStream_t>(stream),
                       d_planes, n, l.n_groups, d_scores, d_density, d_totals, d_status,
                       st.d_trans, st.d_records, st.n_states * 4, st.n_records * 4, g_wtab, phase_mask, d_prof);
    GMK_HIP_CHECK(hipGetLastError());
    if (stamps) {                                               // (profile build only) cycles per phase and board, mean over the wavefronts
        unsigned long long h[16];
        GMK_HIP_CHECK(hipDeviceSynchronize());
        GMK_HIP_CHECK(hipMemcpy(h, d_prof, sizeof h, hipMemcpyDeviceToHost));
        (void)hipFree(d_prof);
        const char* names[12] = {"staging", "phase0", "scan", "deposits", "phase3", "phase3b", "rescans", "wait-planes", "phaseD", "phase5", "-", "loop-head"};
        const double boards = static_cast<double>(n), waves = static_cast<double>(h[15] ? h[15] : 1);
        double total = 0;
        for (int k = 0; k < 12; ++k) total += static_cast<double>(h[k]);
        std::fprintf(stderr, "[GMK_EVAL_PROFILE] s_memtime cycles per board (wavefront time; %.0f wavefronts, %.0f cycles each):", waves, total / waves);
        for (int k = 0; k < 12; ++k) if (k != 10) std::fprintf(stderr, " %s %.0f", names[k], static_cast<double>(h[k]) / boards);
        std::fprintf(stderr, "\n");
    }
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
