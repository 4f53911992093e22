// store_probe.hip -- profiling aid (not product code): what the write path takes for K1's output pattern.
//
// K1 (eval_kernel.hip) writes 65 536 x (3 600 B scores + 3 600 B density) per launch; with next to no arithmetic left in it the
// kernel still takes 0.19 ms while torch.fill_ of the same bytes takes 0.07 ms.  This program writes the same bytes with the same
// work distribution (256 workgroups x 16 wavefronts, a wavefront owns 16 consecutive boards and writes them one after the other)
// and switches the suspects on and off one at a time:
//   bit 0   score blocks: 4 x 16 B-per-lane stores per board (900 words, contiguous, 16-byte aligned)
//   bit 1   ... non-temporal
//   bit 2   density planes as K1 writes them: one burst per group, dword stores, 2 x 128 B per instruction (boards on the MFMA rows)
//   bit 3   density planes as 16 B-per-lane stores over the group's contiguous 57.6 KB, one burst per group
//   bit 4   s_waitcnt vmcnt(0) at the end of every board iteration (what a per-board global load forces on the loop)
//   bit 5   boards interleaved over the chip (board = iteration * 4096 + wavefront) instead of 16 consecutive boards per wavefront
//   bit 6   density planes with the board's score block, 16 B per lane, every iteration (no burst)
//   bit 7   the burst in iteration 0 for every wavefront (instead of iteration == wavefront number)
//   bit 8   density 16 B-per-lane stores non-temporal as well
//   bit 9   board stride 3 584 B (28 x 128: every 1 KB piece starts on a cache line) instead of 3 600 B -- diagnostic only, the C-ABI fixes 3 600
//   bit 12  score blocks as 16 dword stores per board (one word per lane into each of the four group planes)
//   bit 11  600 idle cycles between the fourteen passes of a density burst
//   bit 10  no board structure at all: iteration i, piece k of all wavefronts are one contiguous sweep (what fill_ does, in K1's launch shape)
// The launch takes 158 KB of LDS per workgroup like K1, so that every CU gets exactly one workgroup.
// `spin` = cycles of s_sleep per board iteration in place of the evaluation (0: stores back to back).
// usage: store_probe [spin cycles ...]      prints one line per (mode, spin)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kCells = 225, kScoreWords = 900;

__device__ __forceinline__ void spin_cycles(int cycles) {
    if (cycles <= 0) return;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    while (static_cast<int64_t>(__builtin_amdgcn_s_memtime() - t0) < cycles) __builtin_amdgcn_s_sleep(4);
}

template <bool NT>
__device__ __forceinline__ void store_block(int32_t* dst, int words, int lane, int tag) {
    v4i* d = reinterpret_cast<v4i*>(dst);
    const v4i v = {tag, lane, tag ^ lane, 7};
    for (int i = lane; i < words / 4; i += 64) {
        if (NT) __builtin_nontemporal_store(v, &d[i]); else d[i] = v;
    }
}

template <int MODE>
__global__ __launch_bounds__(1024) void pattern_kernel(int32_t* __restrict__ scores, int32_t* __restrict__ density, int n_boards, int spin) {
    extern __shared__ uint32_t lds[];
    if (spin == -12345) lds[threadIdx.x] = MODE;          // (keeps the allocation alive)
    constexpr int mode = MODE;
    constexpr int stride = (mode & 512) ? 896 : kScoreWords;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n_groups = n_boards / 16;
    const int waves_total = gridDim.x * 16;
    for (int group = blockIdx.x * 16 + wave; group < n_groups; group += waves_total) {
        const int first_board = group * 16;
#pragma unroll 1
        for (int bi = 0; bi < 16; ++bi) {
            const int board = (mode & 32) ? bi * waves_total + group : first_board + bi;
            spin_cycles(spin);
            if (mode & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (mode & 1024) {
                // four 1 KB pieces per wavefront and iteration, each piece index one contiguous 4 MB sweep over all wavefronts
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v4i* d = reinterpret_cast<v4i*>(scores) + ((static_cast<size_t>(bi) * 4 + k) * waves_total + group) * 64 + lane;
                    const v4i v = {bi, lane, k, 7};
                    if (static_cast<size_t>((bi * 4 + k)) * waves_total * 1024 + 1024 * group + 1024 <= static_cast<size_t>(n_boards) * 3600) {
                        if (mode & 2) __builtin_nontemporal_store(v, d); else *d = v;
                    }
                }
            } else if (mode & 4096) {
                // K1 round 3: the score block is [cell][group] in LDS: a lane writes one word into each group's plane, 256 contiguous bytes per instruction
                int32_t* dst = scores + static_cast<size_t>(board) * kScoreWords + lane;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (p < 3 || lane < 33) { if (mode & 2) __builtin_nontemporal_store(board + g, dst + 64 * p + g * kCells); else dst[64 * p + g * kCells] = board + g; }
            } else if (mode & 1) {
                store_block<(mode & 2) != 0>(scores + static_cast<size_t>(board) * stride, stride, lane, board);
            }
            if (mode & 64) store_block<(mode & 256) != 0>(density + static_cast<size_t>(board) * kScoreWords, kScoreWords, lane, board);
            const bool my_turn = (mode & 128) ? bi == 0 : bi == wave;
            if (my_turn && (mode & 4)) {
                // K1's phase D: 14 passes (7 tiles of 32 cells x 2 plane kinds), 8 x 2 dword stores each; lane (n, h): cell 32 M + n of boards b + 2 h
                const int n = lane & 31, h = lane >> 5, ps = kCells;
#pragma unroll 1
                for (int pass = 0; pass < 14; ++pass) {
                    const int m = pass >> 1, kind = pass & 1;
                    int off = kind * ps + 32 * m + n + h * (2 * 4 * ps);
                    asm volatile("" : "+v"(off));               // (the sixteen addresses of a pass are computed in the pass, as in K1)
                    int32_t* out = density + static_cast<size_t>(first_board) * 4 * ps + off;
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {
                        const int b = 4 * (i / 4) + (i % 4) / 2;
                        out[b * 4 * ps + 1 * 2 * ps] = m + i;
                        out[b * 4 * ps + 0 * 2 * ps] = m - i;
                    }
                    if (mode & 2048) spin_cycles(600);          // (K1 issues 3-5 MFMAs and their LDS reads between the passes' stores)
                }
                if (lane < 32) {
                    int32_t* out = density + static_cast<size_t>(first_board + (lane >> 1)) * 4 * ps + (1 - (lane & 1)) * 2 * ps;
                    out[224] = lane;
                    out[ps + 224] = -lane;
                }
            }
            if (my_turn && (mode & 8)) store_block<(mode & 256) != 0>(density + static_cast<size_t>(first_board) * kScoreWords, 16 * kScoreWords, lane, group);
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(v4i* __restrict__ dst, size_t n16) {
    const v4i v = {1, 2, 3, 4};
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, &dst[i]); else dst[i] = v;
    }
}

// the launch-shape question: every wavefront writes `pieces` 1 KB pieces, piece p of all wavefronts one contiguous sweep
__global__ void shape_kernel(v4i* __restrict__ dst, int pieces, int nt) {
    extern __shared__ uint32_t lds[];
    if (pieces == -12345) lds[threadIdx.x] = nt;
    const int waves_per_wg = blockDim.x >> 6, wave_global = blockIdx.x * waves_per_wg + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t waves_total = static_cast<size_t>(gridDim.x) * waves_per_wg;
    const v4i v = {1, 2, 3, static_cast<int>(lane)};
    for (int p = 0; p < pieces; ++p) {
        v4i* d = dst + (p * waves_total + wave_global) * 64 + lane;
        if (nt) __builtin_nontemporal_store(v, d); else *d = v;
    }
}

int main(int argc, char** argv) {
    const int n_boards = 65536, reps = 30;
    std::vector<int> spins;
    for (int i = 1; i < argc; ++i) if (atoi(argv[i]) >= 0) spins.push_back(atoi(argv[i]));
    if (spins.empty()) spins = {0, 12000, 20000};
    int32_t *scores, *density;
    const size_t bytes = static_cast<size_t>(n_boards) * kScoreWords * 4;
    CHECK(hipMalloc(&scores, bytes));
    CHECK(hipMalloc(&density, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time_it = [&](auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps;
    };
    {
        // reference: one contiguous sweep over both buffers (they are separate allocations: two launches' worth in one kernel each)
        for (int nt = 0; nt < 2; ++nt) {
            const float ms = time_it([&] {
                if (nt) { fill_kernel<true><<<2048, 256>>>(reinterpret_cast<v4i*>(scores), bytes / 16); fill_kernel<true><<<2048, 256>>>(reinterpret_cast<v4i*>(density), bytes / 16); }
                else { fill_kernel<false><<<2048, 256>>>(reinterpret_cast<v4i*>(scores), bytes / 16); fill_kernel<false><<<2048, 256>>>(reinterpret_cast<v4i*>(density), bytes / 16); }
            });
            printf("fill%s  2 x %.0f MB  %.4f ms  %.2f TB/s\n", nt ? "_nt" : "   ", bytes / 1e6, ms, 2 * bytes / ms / 1e9);
        }
    }
    {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(shape_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        struct Shape { int grid, block, lds_kb; };
        const Shape shapes[] = {{256, 1024, 158}, {256, 1024, 0}, {512, 1024, 0}, {512, 512, 79}, {1024, 256, 39}, {2048, 256, 0}, {2048, 256, 19}, {4096, 256, 0}, {8192, 256, 0}, {16384, 256, 0}, {65536, 64, 0}, {230400, 64, 0}};
        for (const Shape& sh : shapes)
            for (int nt = 0; nt < 2; ++nt) {
                const size_t waves_total = static_cast<size_t>(sh.grid) * (sh.block / 64);
                const int pieces = static_cast<int>(bytes / 1024 / waves_total);
                const float ms = time_it([&] { shape_kernel<<<sh.grid, sh.block, sh.lds_kb * 1024>>>(reinterpret_cast<v4i*>(scores), pieces, nt); });
                printf("shape grid %6d block %4d lds %3d KB  pieces/wave %3d  %s  %.4f ms  %.2f TB/s\n", sh.grid, sh.block, sh.lds_kb, pieces, nt ? "nt   " : "plain", ms, pieces * waves_total * 1024.0 / ms / 1e9);
            }
    }
    struct Mode { int bits; const char* what; void (*launch)(int32_t*, int32_t*, int, int); };
#define M(bits, what) {bits, what, [](int32_t* sc, int32_t* de, int n, int spin) { \
        static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(pattern_kernel<bits>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true); (void)once; \
        pattern_kernel<bits><<<256, 1024, 158 * 1024>>>(sc, de, n, spin); }}
    const Mode modes[] = {
        M(0, "no stores at all (launch + spin only)"),
        M(1024, "sweep: 4 x 1 KB per wavefront and iteration, contiguous over the chip, plain"),
        M(1024 | 2, "sweep nt"),
        M(1 | 2 | 512, "scores nt, board stride 3584 B (pieces on cache lines)"),
        M(1 | 512, "scores plain, board stride 3584 B"),
        M(1 | 2, "scores nt"),
        M(1, "scores plain"),
        M(1 | 2 | 16, "scores nt, vmcnt(0) per board"),
        M(1 | 2 | 32, "scores nt, boards interleaved over the chip"),
        M(4, "density K1 pattern (dword, 2x128B), burst by turn"),
        M(8, "density 16B/lane, burst by turn"),
        M(8 | 256, "density 16B/lane nt, burst by turn"),
        M(1 | 2 | 4, "K1: scores nt + density dword burst"),
        M(1 | 2 | 4 | 16, "K1 + vmcnt(0) per board"),
        M(1 | 2 | 8, "scores nt + density 16B/lane burst"),
        M(1 | 2 | 8 | 256, "scores nt + density 16B/lane nt burst"),
        M(1 | 2 | 8 | 16, "scores nt + density 16B/lane burst + vmcnt(0)"),
        M(1 | 2 | 64, "scores nt + density 16B/lane with every board"),
        M(1 | 2 | 64 | 256, "scores nt + density 16B/lane nt with every board"),
        M(1 | 64, "scores + density 16B/lane with every board, plain"),
        M(1 | 2 | 4 | 128, "K1 pattern, every wavefront's burst in iteration 0"),
        M(1 | 2 | 4 | 2048, "K1 pattern, 600 cycles between the passes of a burst"),
        M(4 | 2048, "density dword burst alone, 600 cycles between passes"),
        M(1 | 2 | 4096, "scores as 16 dword stores nt (256 B per instruction)"),
        M(1 | 4096, "scores as 16 dword stores plain"),
        M(1 | 2 | 4 | 4096, "round-3 K1: scores 16 dword nt + density dword burst"),
        M(1 | 2 | 8 | 4096, "scores 16 dword nt + density 16B/lane burst"),
    };
    for (int spin : spins)
        for (const Mode& m : modes) {
            const float ms = time_it([&] { m.launch(scores, density, n_boards, spin); });
            const double out_bytes = ((m.bits & (1 | 1024)) ? bytes : 0) + ((m.bits & (4 | 8 | 64)) ? bytes : 0);
            
            printf("spin %6d  mode %3d  %.4f ms  %.2f TB/s  %s\n", spin, m.bits, ms, out_bytes / ms / 1e9, m.what);
        }
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    return 0;
}
