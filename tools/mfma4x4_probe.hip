// Fragment layout of v_mfma_f32_4x4x1_16B_f32, found by experiment: every lane feeds A = 100 + lane, B = 1000 * (lane + 1) into one instruction;
// the product printed per (register, lane) tells which A and B lanes meet where.  Build: hipcc --offload-arch=gfx950 -o /tmp/p tools/mfma4x4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    f32x4 c = {};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(static_cast<float>(100 + lane), 1000.0f * (lane + 1), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[r * 64 + lane] = c[r];
}
int main() {
    float* d; float h[256];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    probe<<<1, 64>>>(d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int bad = 0;
    for (int r = 0; r < 4; ++r)
        for (int lane = 0; lane < 64; ++lane) {
            // expectation: D[block = lane / 4][row = r][column = lane % 4] = A[block][row] * B[block][column], A on lane 4 * block + row, B on lane 4 * block + column
            const float want = static_cast<float>(100 + 4 * (lane / 4) + r) * (1000.0f * (lane + 1));
            if (h[r * 64 + lane] != want) { if (bad < 8) std::printf("reg %d lane %d: %.0f, expected %.0f\n", r, lane, h[r * 64 + lane], want); ++bad; }
        }
    std::printf("4x4x1_16B layout %s (%d mismatches)\n", bad ? "DIFFERS" : "as expected: D[reg i][lane 4b + j] = A[lane 4b + i] * B[lane 4b + j]", bad);
    return bad != 0;
}
