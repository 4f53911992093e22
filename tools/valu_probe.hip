// valu_probe.hip -- profiling aid (not product code): wave64 issue rate of the integer / bit-field VALU instructions K1 is made of,
// at 1, 2 and 4 wavefronts per SIMD (K1 runs 4).  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

template <int KIND>
__global__ void valu_kernel(uint32_t* out, int iters, unsigned long long* cycles) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 ^ 9, a5 = a0 + 11, a6 = a0 | 13, a7 = a0 + 1;
    const uint32_t k = out[0];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {            // v_and_or / v_lshl_add / v_xor / v_add
                a0 = (a0 & k) | a1; a1 = (a1 << 2) + a2; a2 ^= a3; a3 += a4; a4 = (a4 & k) | a5; a5 = (a5 << 2) + a6; a6 ^= a7; a7 += a0;
            } else if (KIND == 1) {     // v_bfe_u32, v_cndmask, v_mul_u32_u24, v_min
                a0 = __builtin_amdgcn_ubfe(a0, a1 & 31, 5) + a0; a1 = a1 > a2 ? a3 : a1; a2 = (a2 & 0xFFFFFF) * (a3 & 0xFFFFFF); a3 = min(a3, a4) + 1;
                a4 = __builtin_amdgcn_ubfe(a4, a5 & 31, 5) + a4; a5 = a5 > a6 ? a7 : a5; a6 = (a6 & 0xFFFFFF) * (a7 & 0xFFFFFF); a7 = min(a7, a0) + 1;
            } else if (KIND == 2) {     // DPP moves and mbcnt
                a0 += __builtin_amdgcn_update_dpp(0, a1, 0x111, 0xF, 0xF, true); a1 += __builtin_amdgcn_update_dpp(0, a2, 0x112, 0xF, 0xF, true);
                a2 += __builtin_amdgcn_mbcnt_lo(a3, a2); a3 += __builtin_amdgcn_update_dpp(0, a4, 0xB1, 0xF, 0xF, true);
                a4 += __builtin_amdgcn_update_dpp(0, a5, 0x111, 0xF, 0xF, true); a5 += __builtin_amdgcn_update_dpp(0, a6, 0x112, 0xF, 0xF, true);
                a6 += __builtin_amdgcn_mbcnt_lo(a7, a6); a7 += __builtin_amdgcn_update_dpp(0, a0, 0xB1, 0xF, 0xF, true);
            } else {                    // f32 fma for reference
                float f0 = __uint_as_float(a0), f1 = __uint_as_float(a1), f2 = __uint_as_float(a2), f3 = __uint_as_float(a3);
                f0 = fmaf(f0, f1, f2); f1 = fmaf(f1, f2, f3); f2 = fmaf(f2, f3, f0); f3 = fmaf(f3, f0, f1);
                a0 = __float_as_uint(f0); a1 = __float_as_uint(f1); a2 = __float_as_uint(f2); a3 = __float_as_uint(f3);
                float g0 = __uint_as_float(a4), g1 = __uint_as_float(a5), g2 = __uint_as_float(a6), g3 = __uint_as_float(a7);
                g0 = fmaf(g0, g1, g2); g1 = fmaf(g1, g2, g3); g2 = fmaf(g2, g3, g0); g3 = fmaf(g3, g0, g1);
                a4 = __float_as_uint(g0); a5 = __float_as_uint(g1); a6 = __float_as_uint(g2); a7 = __float_as_uint(g3);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x + 1] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    uint32_t* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, (256 * 1024 + 1) * 4)); CHECK(hipMemset(out, 0xFF, 4));
    CHECK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    const char* names[4] = {"and_or/lshl_add/xor/add", "bfe/cndmask/mul24/min", "dpp/mbcnt", "fma_f32"};
    for (int kind = 0; kind < 4; ++kind)
        for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
            const int block = 256 * waves_per_simd;
            auto launch = [&] {
                if (kind == 0) valu_kernel<0><<<256, block>>>(out, iters, cyc);
                if (kind == 1) valu_kernel<1><<<256, block>>>(out, iters, cyc);
                if (kind == 2) valu_kernel<2><<<256, block>>>(out, iters, cyc);
                if (kind == 3) valu_kernel<3><<<256, block>>>(out, iters, cyc);
            };
            launch(); CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h; CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            // instructions per wave: see the ISA; reported per statement group instead: 64 statements per iteration
            printf("%-26s %d waves/SIMD: %.3f ms, wave 0: %llu cycles for %d iterations = %.1f cycles per iteration per wave, %.2f per iteration per SIMD-slot\n",
                   names[kind], waves_per_simd, ms, h, iters, (double)h / iters, (double)h / iters / waves_per_simd);
        }
    return 0;
}
