// What does one v_mfma_f32_32x32x2_f32 cost when it is issued the way K9's layer 3 issues it?  256 workgroups x 4 wavefronts (one per SIMD, the
// chip under full matrix load), each running REPS rounds of 7 independent accumulators; variants add, per MFMA, what the real loop has beside it:
//   0  MFMAs only                     1  + one ds_read_b32 behind each MFMA (value used 7 MFMAs later)
//   2  + one global_load_dword per 7 (used 7 MFMAs later: NOT how K9 does it, it loads two steps ahead; shows what a late weight costs)
//   8  as 1 + one global_load_dword per round that nobody waits for, and s_waitcnt vmcnt(15)    9  as 1 + one v_fma per round    10  both
//   6, 7  as 1, the seven reads in one burst behind the round's first / third MFMA (one s_waitcnt per round)
//   4  one ds_read_b128 per four MFMAs instead (the same bytes)    5  ds_read_b32 behind each MFMA whose value nobody waits for (issue cost alone)
// Prints shader clocks (s_memtime) per MFMA for wavefront 0 of workgroup 0 and the wall time per MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_issue_probe tools/mfma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int REPS = 2048;
template <int V>
__global__ __launch_bounds__(512) void probe(const float* w, float* out, unsigned long long* clk) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 36000; i += blockDim.x) lds[i] = 0.001f * (i & 127);
    __syncthreads();
    f32x16 acc[7] = {};
    float b[2][7];
    uint32_t base[7];
    for (int t = 0; t < 7; ++t) base[t] = ((lane >> 5) * 289 + 18 + t * 32 + (lane & 31)) * 4;
    const char* in = reinterpret_cast<const char*>(lds);
    for (int t = 0; t < 7; ++t) b[0][t] = *reinterpret_cast<const float*>(in + base[t]);
    float a = w[lane], a_next = a, sinkv = 0.0f, sinkf = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS; r += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cur = h;
            if (V >= 1) {
#pragma unroll
                for (int t = 0; t < 7; ++t) b[cur ^ 1][t] = *reinterpret_cast<const float*>(in + ((r + h) & 31) * 2 * 289 * 4 + base[t]);
            }
            if (V == 2) a_next = w[((r + h) & 255) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 7; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[V >= 1 ? cur : 0][t], acc[t], 0, 0, 0);
            if (V == 6 || V == 7) {                          // the reads of a round in one burst behind its first (6) / third (7) MFMA: one s_waitcnt per round
                __builtin_amdgcn_sched_group_barrier(0x008, V == 6 ? 1 : 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, V == 6 ? 6 : 4, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (V >= 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            if (V == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (V == 2) a = a_next;
            if (V == 8 || V == 10) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(15)" : "=v"(sinkv) : "v"(w + ((r + h) & 255) * 64 + lane));
            if (V == 9 || V == 10) sinkf = __builtin_fmaf(a, b[cur][3], sinkf);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)");
    float s = sinkv + sinkf;
    for (int t = 0; t < 7; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[(blockIdx.x * 256 + threadIdx.x) & 65535] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}
template <int V>
__global__ __launch_bounds__(256) void probe2(const float* w, float* out, unsigned long long* clk) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 36000; i += blockDim.x) lds[i] = 0.001f * (i & 127);
    __syncthreads();
    f32x16 acc[8] = {};
    const float a = w[lane];
    const uint32_t base = lane * 16;
    float4 b[2][2];
    b[0][0] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lds) + base);
    b[0][1] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lds) + base + 1024);
    float sink = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS; r += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (V == 4) {
                b[h ^ 1][0] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lds) + ((r + h) & 31) * 2048 + base);
                b[h ^ 1][1] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lds) + ((r + h) & 31) * 2048 + base + 1024);
                const float* bb = reinterpret_cast<const float*>(&b[h][0]);
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb[t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[0][0].x, acc[t], 0, 0, 0);
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(sink) : "v"(base), "n"(0));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = sink;
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[(blockIdx.x * 256 + threadIdx.x) & 65535] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}
// 11: as 7 (the seven reads of a round in one burst behind its third MFMA), but for the round AFTER the next one (three operand sets): does the cost of
// variant 7 come from waiting for reads that are 4 MFMAs old?
__global__ __launch_bounds__(256) void probe3(const float* w, float* out, unsigned long long* clk) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 36000; i += blockDim.x) lds[i] = 0.001f * (i & 127);
    __syncthreads();
    f32x16 acc[7] = {};
    float b[3][7];
    uint32_t base[7];
    for (int t = 0; t < 7; ++t) base[t] = ((lane >> 5) * 289 + 18 + t * 32 + (lane & 31)) * 4;
    const char* in = reinterpret_cast<const char*>(lds);
    for (int t = 0; t < 7; ++t) { b[0][t] = *reinterpret_cast<const float*>(in + base[t]); b[1][t] = *reinterpret_cast<const float*>(in + 2312 + base[t]); }
    const float a = w[lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS - 2; r += 3) {
#pragma unroll
        for (int h = 0; h < 3; ++h) {
#pragma unroll
            for (int t = 0; t < 7; ++t) b[(h + 2) % 3][t] = *reinterpret_cast<const float*>(in + ((r + h) & 31) * 2 * 289 * 4 + base[t]);
#pragma unroll
            for (int t = 0; t < 7; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[h][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int t = 0; t < 7; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[(blockIdx.x * 256 + threadIdx.x) & 65535] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}
// 12: two rounds' worth per scheduling block: 14 reads in one burst behind the third MFMA, 14 MFMAs (7 accumulators, two k each): is variant 7's
// cost per round or per burst?
__global__ __launch_bounds__(256) void probe4(const float* w, float* out, unsigned long long* clk) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 36000; i += blockDim.x) lds[i] = 0.001f * (i & 127);
    __syncthreads();
    f32x16 acc[7] = {};
    float b[2][14];
    uint32_t base[7];
    for (int t = 0; t < 7; ++t) base[t] = ((lane >> 5) * 289 + 18 + t * 32 + (lane & 31)) * 4;
    const char* in = reinterpret_cast<const char*>(lds);
    for (int t = 0; t < 14; ++t) b[0][t] = *reinterpret_cast<const float*>(in + (t / 7) * 2312 + base[t % 7]);
    const float a = w[lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS; r += 4) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int t = 0; t < 14; ++t) b[h ^ 1][t] = *reinterpret_cast<const float*>(in + ((r + 2 * h + t / 7) & 15) * 2 * 289 * 4 + base[t % 7]);
#pragma unroll
            for (int t = 0; t < 14; ++t) acc[t % 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[h][t], acc[t % 7], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int t = 0; t < 7; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[(blockIdx.x * 256 + threadIdx.x) & 65535] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}
// 17: as 13 with a global_load_dwordx4
// 13 / 14: as 7 (NT MFMAs + NT reads per round, NT = 7 / 3 / 3) + one global_load_dword per round with NO address arithmetic (scalar base + lane
// offset + immediate), nobody waits for it: what the weight fetch itself costs; 16: NT = 3 without the load
template <int NT, int LOAD>
__global__ __launch_bounds__(512) void probe5(const float* w, float* out, unsigned long long* clk) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 36000; i += blockDim.x) lds[i] = 0.001f * (i & 127);
    __syncthreads();
    f32x16 acc[NT] = {};
    float b[2][NT];
    uint32_t base[NT];
    for (int t = 0; t < NT; ++t) base[t] = ((lane >> 5) * 289 + 18 + t * 32 + (lane & 31)) * 4;
    const char* in = reinterpret_cast<const char*>(lds);
    for (int t = 0; t < NT; ++t) b[0][t] = *reinterpret_cast<const float*>(in + base[t]);
    const float a = w[lane];
    const uint32_t lane_off = lane * 4, lane_off4 = lane * 16;
    float sink = 0.0f;
    float4 sink4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS; r += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int t = 0; t < NT; ++t) b[h ^ 1][t] = *reinterpret_cast<const float*>(in + ((r + h) & 31) * 2 * 289 * 4 + base[t]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[h][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NT >= 4 ? 3 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NT - (NT >= 4 ? 3 : 1), 0);
            __builtin_amdgcn_sched_barrier(0);
            if (LOAD == 1) asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(sink) : "v"(lane_off), "s"(w), "n"(256 * 3));
            if (LOAD == 4) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(sink4) : "v"(lane_off4), "s"(w), "n"(1024));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = sink + sink4.x + sink4.w;
    for (int t = 0; t < NT; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[(blockIdx.x * 256 + threadIdx.x) & 65535] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}
// 18: eight MFMAs per round fed by FOUR ds_read2st64_b32 (two dwords 2 560 bytes apart per instruction) in one burst behind the third MFMA:
// does an LDS read cost per instruction or per dword?
__global__ __launch_bounds__(256) void probe6(const float* w, float* out, unsigned long long* clk) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 36000; i += blockDim.x) lds[i] = 0.001f * (i & 127);
    __syncthreads();
    f32x16 acc[8] = {};
    float2 b[2][4];
    uint32_t base[4];
    for (int t = 0; t < 4; ++t) base[t] = ((lane >> 5) * 320 + 18 + t * 32 + (lane & 31)) * 4;
    auto read2 = [](uint32_t addr) { float2 v; asm volatile("ds_read2st64_b32 %0, %1 offset0:0 offset1:10" : "=v"(v) : "v"(addr)); return v; };
    for (int t = 0; t < 4; ++t) b[0][t] = read2(base[t]);
    asm volatile("s_waitcnt lgkmcnt(0)");
    const float a = w[lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS; r += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[h][0].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[h][0].y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[h][1].x, acc[2], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) b[h ^ 1][t] = read2(base[t] + ((r + h) & 7) * 5120);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[h][1].y, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[h][2].x, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[h][2].y, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[h][3].x, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[h][3].y, acc[7], 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[(blockIdx.x * 256 + threadIdx.x) & 65535] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}
template <class K>
void run(K kernel, int variant, double per_rep, const float* w, float* out, unsigned long long* clk, int threads = 256) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kernel<<<256, threads, 150 * 1024>>>(w, out, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) kernel<<<256, threads, 150 * 1024>>>(w, out, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double n = per_rep * REPS;
    if (threads > 256) std::printf("(two wavefronts per SIMD: halve the ticks for the SIMD's cost per MFMA) ");
    std::printf("variant %d: %.2f s_memtime ticks per MFMA, %.2f ns per MFMA wall (whole kernel) -> the clock ran at %.2f GHz; kernel %.3f ms\n", variant, c / n, ms * 1e6 / n, (c / n) / (ms * 1e6 / n), ms);
}
int main() {
    float *w, *out; unsigned long long* clk;
    hipMalloc(&w, 256 * 64 * 4); hipMemset(w, 0, 256 * 64 * 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 8);
    run(probe<0>, 0, 7, w, out, clk); run(probe<1>, 1, 7, w, out, clk); run(probe<2>, 2, 7, w, out, clk);
    run(probe<8>, 8, 7, w, out, clk); run(probe<9>, 9, 7, w, out, clk); run(probe<10>, 10, 7, w, out, clk);
    run(probe<6>, 6, 7, w, out, clk); run(probe<7>, 7, 7, w, out, clk);
    run(probe3, 11, 7.0 * (REPS / 3 * 3) / REPS, w, out, clk);
    run(probe4, 12, 7, w, out, clk);
    run(probe5<7, 1>, 13, 7, w, out, clk); run(probe5<3, 1>, 14, 3, w, out, clk); run(probe5<3, 0>, 16, 3, w, out, clk); run(probe5<7, 4>, 17, 7, w, out, clk);
    run(probe6, 18, 8, w, out, clk);
    run(probe5<7, 1>, 13, 7, w, out, clk, 512); run(probe5<3, 1>, 14, 3, w, out, clk, 512); run(probe5<7, 4>, 17, 7, w, out, clk, 512); run(probe<9>, 9, 7, w, out, clk, 512);
    run(probe2<4>, 4, 8, w, out, clk); run(probe2<5>, 5, 8, w, out, clk);
    return 0;
}
