// pvnet_kernel.hip -- K9: the policy-value network on the f32 matrix cores: its convolutional trunk as ONE fused kernel, its dense layers as a second.
//
// The network the network-guided search (K7) calls at every leaf is the reference's PolicyValueNetwork
// (network/model_tf.py:28-66): conv3x3 6->32->64->128 (+ReLU), a 1x1 policy head (128->4, ReLU) and a 1x1 value head
// (128->2, ReLU) on 15x15 boards, then small dense layers.  The convolutions are 99 % of its arithmetic (43 MFLOP per
// position).  pvnet_trunk_kernel runs them for a batch of positions in float32 on v_mfma_f32_32x32x2_f32 (exact f32, same rate as
// the vector FMA peak, MI355X_MICROARCH.md): one workgroup (4 wavefronts, one per SIMD) takes one position at a time and
// keeps every activation in LDS; nothing but the 6 input planes is read from and nothing but the 1 350 head activations is
// written to HBM per position.  pvnet_dense_kernel (further down) turns those into the value and the 225 probabilities.
//
// Formulation: every layer is Out^T[cout][pixel] = sum_k W^T[cout][k] * Act[k][pixel], k = (channel pair, tap):
//   A operand (lane l: A[i = l & 31][k = l >> 5]) = weights, packed on the host in exactly that lane order, read from L2;
//   B operand (lane l: B[k = l >> 5][j = l & 31]) = activations of 32 pixels for two adjacent input channels at one tap,
//     one ds_read_b32 per MFMA: activations live in LDS as act[channel][17 x 17] with a zero border, so a tap is a
//     constant address offset and "same" padding costs nothing;
//   C/D (column = pixel on the lane, rows = 16 output channels in registers per lane half) start from the bias -> ReLU, one LDS
//     write per register, conflict-free (lanes = consecutive pixels).
// 225 pixels are seven tiles of 32 and the corner (14, 14), which rides on v_mfma_f32_4x4x1_16B_f32 beside them (conv_tiles()).
// The 1x1 heads use the layer-3 accumulators as B operands directly (the idiom of cdna_hip_programming.md "An accumulator tile as
// the next MFMA's operand": no transpose, no LDS round trip), on the 4x4x1 form as well: six output rows fill it, a 32 x 32 tile not.
// Wave w owns output channels [32 w, 32 w + 32) of layer 3 for the 7 pixel tiles + the corner (112 + 4 accumulator registers);
// layer 2 gives each wave one of 2 channel tiles, 3 pixel tiles and half the k range of a fourth; layer 1 the 8 pixel tiles, two each.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "capi_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPix = 225;
constexpr int kPad = 289;                        // 17 x 17: the image with a zero border, per channel
constexpr int kTiles = 8;                        // pixel tiles of 32 (225 -> 256)
constexpr int kHeadRows = 6;                     // 4 policy + 2 value channels
// LDS (floats): input planes | layer-1 activations | layer-2 activations | the four waves' partial head outputs [4][8][256] | the three layers' biases
constexpr int oIn = 0, oAct1 = oIn + 6 * kPad, oAct2 = oAct1 + 32 * kPad, oPart = oAct2 + 64 * kPad, oBias = oPart + 4 * 8 * 256, kLdsFloats = oBias + 32 + 64 + 128;
constexpr int kInPerThread = (6 * kPix + 255) / 256;

struct PvParams {
    const float* states;                         // [n][6][225]
    int n;
    const float* w1;                             // packed A operands, see pack_layer(): [1 tile][27 k-pairs][64 lanes]
    const float* w2;                             // [2 tiles][144][64]
    const float* w3;                             // [4 tiles][288][64]
    const float* whc;                            // heads for the corner pixel: [4 waves][6 rows][32 channels]
    const float* wh;                             // heads: [4 waves][16 registers policy, 16 registers value][64 lanes], A operands of the 4x4x1 form
    const float* b1; const float* b2; const float* b3; const float* bh;      // biases [32], [64], [128], [6]
    float* pflat;                                // [n][900]  relu(policy conv), flattened (pixel, channel) like tf.layers.flatten of NHWC
    float* vflat;                                // [n][450]  relu(value conv), flattened (pixel, channel)
    unsigned long long* prof;                    // GMK_PVNET_PROFILE: shader clocks of workgroup 0 per stage (diagnostic runs only), else null
};

__device__ __forceinline__ int padded_index(int p) { return (p / 15 + 1) * 17 + (p % 15) + 1; }
// row of the 32 x 32 C/D tile that register r of this lane holds (cdna_hip_programming.md, fragment layout)
__device__ __forceinline__ int cd_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// The order in which a layer's k-pairs are walked (and packed): chunks of 2 * CP input channels; inside a chunk the nine taps; inside a tap
// ("step") the chunk's CP channel pairs.
__host__ __device__ inline void step_chunk_tap(int step, int& chunk, int& tap) { chunk = step / 9; tap = step - 9 * chunk; }

// One convolution layer for NP pixel tiles and one tile of 32 output channels: acc[t] += W^T * Act over all k-pairs (in pack_layer()'s order).
// `in` points at the layer's input activations in LDS, base[t] is this lane's byte offset for tile t:
// ((lane >> 5) * kPad + padded_index(pixel) - 18) * 4, so that tap (ky, kx) of channel c adds (c * kPad + ky * 17 + kx) * 4.
//
// What shapes this loop (tools/mfma_issue_probe.hip, profiles/r04_k9_mfma_issue_probe.txt; one wavefront per SIMD): back-to-back
// v_mfma_f32_32x32x2_f32 issue every 64.1 clocks and NOTHING but another MFMA overlaps them for free: an MFMA whose B operand an LDS read wrote
// costs ~2 clocks more, a global load ~20 (28 for 16 bytes per lane), EVERY VECTOR-ALU INSTRUCTION 20-30 (the f32 matrix instruction runs at the
// vector FMA rate: it has no shadow for vector work).  So the loop contains no vector instruction -- the nine taps and the CP channel pairs of a
// chunk are unrolled, which makes every LDS address `a register that changes once per chunk + an immediate` and every weight address `a scalar
// base + the lane + an immediate`; the only vector instructions left per chunk are the NB address increments -- and as few loads as it can.
//   * Operands are fetched AHEAD of the MFMAs that use them, and the scheduling barriers keep the fetches where they are written (left alone, the
//     compiler sinks every load to just before its use, and each MFMA then waits for LDS or L2 behind it): a step's CP weights two steps ahead, one
//     16-byte global load per four k-pairs, into one of three rotating register sets (three steps per rotation: no register moves); the NB
//     activation reads of a k-pair AHEAD k-pairs ahead (one everywhere: two or three for the layers whose k-pairs are two to four MFMAs measured
//     the same), in one burst behind the third MFMA of the current k-pair.  The fetches past the layer's last k-pair read two steps of padding
//     behind the weights and the LDS behind the activations (both there, both unused).
//   * CORNER: 225 pixels are seven tiles and ONE pixel, the corner (14, 14).  A 32 x 32 MFMA for it would compute 31 columns nobody reads (64
//     clocks) and a v_fma costs 20-30 here, so it rides on v_mfma_f32_4x4x1_16B_f32 (8 clocks): A = the A operand the lane holds anyway
//     (W[channel l & 31][k = 2 kp + (l >> 5)]), B = the corner's activation for that k (one more LDS read per k-pair, base[NB - 1]: the same address
//     on all lanes of a half, so all four columns of a block are the corner).  Block l >> 2 multiplies the A values of its four lanes with that
//     activation: register i of lane l accumulates channel 4 ((l >> 2) & 7) + i over the k of the lane's half.  The corner's taps with ky = 2 or
//     kx = 2 read the zero border: those five of nine taps leave it out.
//   * SHARED: one more pixel tile, acc[NP] / base[NP], takes part in chunk `shared_chunk` only: two waves share that tile's k range and one of them
//     adds the other's partial sums afterwards -- how layer 2's fourteen tile jobs become 3.5 per wave.
template <int CIN, int NP, bool CORNER = false, bool SHARED = false, int AHEAD = 1>
__device__ __forceinline__ void conv_tiles(const char* in, const float* __restrict__ w, int lane, const uint32_t (&base)[NP + (SHARED ? 1 : 0) + (CORNER ? 1 : 0)],
                                           f32x16 (&acc)[NP + (SHARED ? 1 : 0)], f32x4& corner, const float (&first)[CIN >= 16 ? 8 : 9 * (CIN / 2)], int shared_chunk = -1) {
    constexpr int CP = CIN >= 16 ? 8 : CIN / 2;
    constexpr int CHUNKS = CIN / 2 / CP;
    constexpr int NB = NP + (SHARED ? 1 : 0) + (CORNER ? 1 : 0); // B operands fetched per k-pair
    constexpr int kChunkBytes = 2 * CP * kPad * 4;               // LDS distance of two chunks
    uint32_t addr[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) addr[t] = base[t];
    // weights: [step][CP / 4 groups][64 lanes][4 k-pairs]: a lane fetches the A operands of four consecutive k-pairs with one 16-byte load (a
    // global load costs the MFMA stream ~20 clocks whatever its width, tools/mfma_issue_probe.hip)
    const float4* wl_ptr = reinterpret_cast<const float4*>(w) + lane;
    constexpr bool kAllHeld = CIN < 16;                         // layer 1: the caller holds all 27 A operands in registers, nothing is fetched
    static_assert(CHUNKS == 1 || (9 * CP) % (AHEAD + 1) == 0, "the operand sets rotate through a whole chunk");
    static_assert(AHEAD * NB <= 15, "LDS reads in flight (lgkmcnt counts to 15)");
    float wA[CP], wB[CP], wC[CP], b[AHEAD + 1][NB];
#pragma unroll
    for (int i = 0; i < CP; ++i) wA[i] = first[i];              // step 0's are the same for every position: the kernel keeps them, nothing to wait for
    if constexpr (!kAllHeld) {
#pragma unroll
        for (int g = 0; g < CP / 4; ++g) { const float4 v = wl_ptr[(CP / 4 + g) * 64]; wB[4 * g] = v.x; wB[4 * g + 1] = v.y; wB[4 * g + 2] = v.z; wB[4 * g + 3] = v.w; }
    }
    // byte offset of k-pair kp of a chunk from the chunk's base (past the chunk's end: into the next one)
    auto kp_offset = [](int kp) { const int c = kp / (9 * CP), k = kp - c * 9 * CP, tap = k / CP; return c * kChunkBytes + ((tap / 3) * 17 + (tap % 3)) * 4 + 2 * (k - tap * CP) * kPad * 4; };
#pragma unroll
    for (int d = 0; d < AHEAD; ++d)
#pragma unroll
        for (int t = 0; t < NB; ++t) b[d][t] = *reinterpret_cast<const float*>(in + addr[t] + kp_offset(d));
    auto do_step = [&](auto tap_c, const float (&wc)[CP], float (&wl)[CP], auto with_shared) {
        constexpr int tap = decltype(tap_c)::value;
        constexpr bool with_corner = CORNER && tap / 3 < 2 && tap % 3 < 2;
        constexpr int NT = NP + (decltype(with_shared)::value ? 1 : 0);    // MFMAs per k-pair
        constexpr int lead = NT >= 4 ? 3 : 1;                              // (2 .. 5 measure the same)
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
            const int kp = tap * CP + cp, cur = kp % (AHEAD + 1), ahead = (kp + AHEAD) % (AHEAD + 1);
#pragma unroll
            for (int t = 0; t < NB; ++t) b[ahead][t] = *reinterpret_cast<const float*>(in + addr[t] + kp_offset(kp + AHEAD));
            constexpr bool fetch = !kAllHeld;
            if constexpr (fetch) {
                if (cp % 4 == 0) { const float4 v = wl_ptr[((tap + 2) * (CP / 4) + cp / 4) * 64]; wl[cp] = v.x; wl[cp + 1] = v.y; wl[cp + 2] = v.z; wl[cp + 3] = v.w; }
            }
            const float a = kAllHeld ? first[kp] : wc[cp];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[cur][t], acc[t], 0, 0, 0);
            if (with_corner) corner = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b[cur][NB - 1], corner, 0, 0, 0);
            // issue order inside this block: `lead` MFMAs, the LDS reads and (every fourth k-pair) the weight load in their shadow, the other MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, lead, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
            if (fetch && cp % 4 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NT - lead + (with_corner ? 1 : 0), 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto chunk_steps = [&](auto with_shared) {
        do_step(std::integral_constant<int, 0>{}, wA, wC, with_shared); do_step(std::integral_constant<int, 1>{}, wB, wA, with_shared); do_step(std::integral_constant<int, 2>{}, wC, wB, with_shared);
        do_step(std::integral_constant<int, 3>{}, wA, wC, with_shared); do_step(std::integral_constant<int, 4>{}, wB, wA, with_shared); do_step(std::integral_constant<int, 5>{}, wC, wB, with_shared);
        do_step(std::integral_constant<int, 6>{}, wA, wC, with_shared); do_step(std::integral_constant<int, 7>{}, wB, wA, with_shared); do_step(std::integral_constant<int, 8>{}, wC, wB, with_shared);
    };
#pragma unroll 1
    for (int chunk = 0; chunk < CHUNKS; ++chunk) {
        if (SHARED && chunk == shared_chunk) chunk_steps(std::integral_constant<bool, SHARED>{});
        else chunk_steps(std::integral_constant<bool, false>{});
#pragma unroll
        for (int t = 0; t < NB; ++t) addr[t] += kChunkBytes;
        wl_ptr += 9 * (CP / 4) * 64;
    }
}

// lanes 0..31 get a[l] + a[l + 32], lanes 32..63 get b[l - 32] + b[l]: one half exchange adds the lane halves of two registers
__device__ __forceinline__ float add_halves(float a, float b) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

// ReLU on a C/D tile of 32 channels x 32 pixels (the bias is what the accumulators started from) and its store into the next layer's activations
// (valid pixels only)
__device__ __forceinline__ void store_tile(float* out /* [channel][kPad] */, const f32x16& acc, int cout0, int tile, int lane) {
    const int p = tile * 32 + (lane & 31);
    const int q = padded_index(min(p, kPix - 1));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = cout0 + cd_row(r, lane);
        const float v = fmaxf(acc[r], 0.0f);
        if (p < kPix) out[c * kPad + q] = v;
    }
}

__global__ __launch_bounds__(256)
void pvnet_trunk_kernel(PvParams prm) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < kLdsFloats; i += 256) lds[i] = i >= oBias ? prm.b1[i - oBias] : 0.0f;       // the zero borders are written once
    // this lane's read offset for pixel tile t (a pixel past the image reads pixel 224's neighbourhood; its column is never stored)
    auto tile_base = [lane](int t) { return static_cast<uint32_t>(((lane >> 5) * kPad + padded_index(min(t * 32 + (lane & 31), kPix - 1)) - 18) * 4); };
    uint32_t base_all[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) base_all[t] = tile_base(t);
    __syncthreads();

    // the input planes of a position travel through registers: fetched while the previous position computes, written into LDS
    // (float32 [6][225] -> [6][17 x 17]) once layer 1 has read the previous ones
    float in_next[kInPerThread];
    int in_at[kInPerThread];                                                  // where this thread's j-th input value goes in LDS (-1: none)
#pragma unroll
    for (int j = 0; j < kInPerThread; ++j) { const int i = threadIdx.x + 256 * j; in_at[j] = i < 6 * kPix ? oIn + (i / kPix) * kPad + padded_index(i % kPix) : -1; }
    auto fetch_input = [&](int img) {
        const float* src = prm.states + static_cast<size_t>(min(img, prm.n - 1)) * 6 * kPix;
#pragma unroll
        for (int j = 0; j < kInPerThread; ++j) { const int i = threadIdx.x + 256 * j; in_next[j] = i < 6 * kPix ? src[i] : 0.0f; }
    };
    auto store_input = [&]() {
#pragma unroll
        for (int j = 0; j < kInPerThread; ++j) if (in_at[j] >= 0) lds[in_at[j]] = in_next[j];
    };
    // the output stage's indices are the same for every position too: output i of this thread = head row j of pixel p, bias bh[j]
    constexpr int kOutPerThread = (kHeadRows * kPix + 255) / 256;
    int out_from[kOutPerThread];                                              // j * 256 + p in a wave's partials, -1: none
    float out_bias[kOutPerThread];
#pragma unroll
    for (int it = 0; it < kOutPerThread; ++it) {
        const int i = threadIdx.x + 256 * it;
        const bool policy = i < 4 * kPix;
        const int k = policy ? i : i - 4 * kPix;
        const int p = policy ? k >> 2 : k >> 1, j = policy ? (k & 3) : 4 + (k & 1);
        out_from[it] = i < kHeadRows * kPix ? j * 256 + p : -1;
        out_bias[it] = prm.bh[min(j, kHeadRows - 1)];
    }
    fetch_input(blockIdx.x);
    store_input();
    __syncthreads();

    f32x4 no_corner = {};                                                     // layers 1 and 2 compute the corner with its tile
    unsigned long long t_mark = 0, t_stage[12] = {};
    auto stamp = [&](int stage) {
        if (prm.prof) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_stage[stage] += t - t_mark; t_mark = t; }
    };
    // What is the same for every position and would otherwise be waited for at the start of a layer (an L2 round trip each): the biases the
    // accumulators start from (in LDS: prm.b1 | b2 | b3 are one array) and the A operands of each layer's first step (in registers).
    const int ct2 = wave & 1;                                                 // layer 2: this wave's channel tile
    const float* w2_mine = prm.w2 + static_cast<size_t>(ct2) * 144 * 64;
    const float* w3_mine = prm.w3 + static_cast<size_t>(wave) * 288 * 64;
    float w1_all[27], w2_first[8], w3_first[8];                               // layer 1: all 27 k-pairs (packed as seven groups of four)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float4 v2 = reinterpret_cast<const float4*>(w2_mine)[g * 64 + lane], v3 = reinterpret_cast<const float4*>(w3_mine)[g * 64 + lane];
        w2_first[4 * g] = v2.x; w2_first[4 * g + 1] = v2.y; w2_first[4 * g + 2] = v2.z; w2_first[4 * g + 3] = v2.w;
        w3_first[4 * g] = v3.x; w3_first[4 * g + 1] = v3.y; w3_first[4 * g + 2] = v3.z; w3_first[4 * g + 3] = v3.w;
    }
#pragma unroll
    for (int g = 0; g < 7; ++g) {
        const float4 v = reinterpret_cast<const float4*>(prm.w1)[g * 64 + lane];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) if (4 * g + i < 27) w1_all[4 * g + i] = e[i];
    }
    if (prm.prof) t_mark = __builtin_amdgcn_s_memtime();
    for (int img = blockIdx.x; img < prm.n; img += gridDim.x) {
        fetch_input(img + gridDim.x);

        // ---- layer 1: 6 -> 32, wave w takes pixel tiles 2w, 2w+1 ----
        {
            const uint32_t base[2] = {tile_base(2 * wave), tile_base(2 * wave + 1)};
            f32x16 acc[2];                                                    // the accumulators start from the bias of the channel they hold
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = lds[oBias + cd_row(r, lane)];
            conv_tiles<6, 2>(reinterpret_cast<const char*>(lds + oIn), prm.w1, lane, base, acc, no_corner, w1_all);
            store_tile(lds + oAct1, acc[0], 0, 2 * wave, lane);
            store_tile(lds + oAct1, acc[1], 0, 2 * wave + 1, lane);
        }
        __syncthreads();
        store_input();                                                        // layer 1 is done with the current planes
        stamp(0);

        // ---- layer 2: 32 -> 64.  Two channel tiles x (seven pixel tiles + the corner) over four waves: wave w takes channel tile w & 1, three
        //      pixel tiles of its own (0..2 or 4..6), HALF the k range of pixel tile 3 (its partner w ^ 2 takes the other half; the upper wave
        //      hands its partial sums over through LDS) and the corner pixel beside the matrix cores (the upper wave's is the one that is kept) ----
        {
            const int ct = ct2, upper = wave >> 1;
            f32x16 acc[4];
            f32x4 corner2 = {};
            const uint32_t base[5] = {tile_base(4 * upper), tile_base(4 * upper + 1), tile_base(4 * upper + 2), tile_base(3), tile_base(7)};
#pragma unroll
            for (int r = 0; r < 16; ++r) {                                    // start from the bias; of the shared tile's two partial sums only one does
                const float bias = lds[oBias + 32 + 32 * ct + cd_row(r, lane)];
                acc[0][r] = acc[1][r] = acc[2][r] = bias;
                acc[3][r] = upper ? 0.0f : bias;
            }
            const int corner_channel = 32 * ct + 4 * ((lane >> 2) & 7);       // + i for register i of the corner's accumulator
            const float4 bias2c = *reinterpret_cast<const float4*>(prm.b2 + corner_channel);
            conv_tiles<32, 3, true, true>(reinterpret_cast<const char*>(lds + oAct1), w2_mine, lane, base, acc, corner2, w2_first, upper);
            stamp(6);
#pragma unroll
            for (int t = 0; t < 3; ++t) store_tile(lds + oAct2, acc[t], 32 * ct, 4 * upper + t, lane);
            float* handover = lds + oPart + ct * 16 * 64;                     // the head partials' place is free until this position's heads
            if (upper) {
                const float bias[4] = {bias2c.x, bias2c.y, bias2c.z, bias2c.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = fmaxf(add_halves(corner2[i], corner2[i]) + bias[i], 0.0f);
                    if ((lane & 35) == 0) lds[oAct2 + (corner_channel + i) * kPad + padded_index(kPix - 1)] = v;     // lanes 0, 4, .. 28
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) handover[r * 64 + lane] = acc[3][r];
            }
            stamp(7);
            __syncthreads();
            stamp(8);
            if (!upper) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[3][r] += handover[r * 64 + lane];
                store_tile(lds + oAct2, acc[3], 32 * ct, 3, lane);
            }
        }
        stamp(1);
        __syncthreads();
        stamp(2);

        // ---- layer 3: 64 -> 128, wave w takes channel tile w, the seven full pixel tiles on the matrix cores and the corner pixel beside them;
        //      its output never leaves the registers ----
        f32x16 acc[kTiles - 1];
        f32x4 corner = {};
        float whp[16], whv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int t = 0; t < kTiles - 1; ++t) acc[t][r] = lds[oBias + 96 + 32 * wave + cd_row(r, lane)];
            whp[r] = prm.wh[(wave * 32 + r) * 64 + lane];
            whv[r] = prm.wh[(wave * 32 + 16 + r) * 64 + lane];
        }
        const int corner_channel = 4 * ((lane >> 2) & 7);                     // of this wave's 32, + i for register i of the corner's accumulator
        const float4 bias3c = *reinterpret_cast<const float4*>(prm.b3 + 32 * wave + corner_channel);
        float4 whc[kHeadRows];
#pragma unroll
        for (int j = 0; j < kHeadRows; ++j) whc[j] = *reinterpret_cast<const float4*>(prm.whc + (wave * kHeadRows + j) * 32 + corner_channel);
        stamp(9);
        conv_tiles<64, kTiles - 1, true>(reinterpret_cast<const char*>(lds + oAct2), w3_mine, lane, base_all, acc, corner, w3_first);
        stamp(3);

        // ---- heads: Out6^T[j][pixel] = sum_c W6^T[j][c] * relu(Out3^T[c][pixel]).  Six output rows would leave a 32 x 32 tile four fifths
        //      empty, so the heads run on v_mfma_f32_4x4x1_16B_f32 (16 independent 4 x 4 outer products per instruction, 8 cycles; layout checked
        //      by tools/mfma4x4_probe.hip): block l / 4 is the four pixels of lanes 4 (l / 4) .. + 3, B = this lane's activation (accumulator
        //      register r: channel cd_row(r, .) of this wave's 32), A = the head weights of that channel (policy rows 0..3 in one instruction,
        //      value rows 0..1 in a second), D register i = head row i for this lane's pixel, summed over the 16 registers = the 16 channels of
        //      this lane half.  All ReLUs first, then the MFMAs back to back (vector work between matrix instructions is what costs). ----
#pragma unroll
        for (int t = 0; t < kTiles - 1; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = fmaxf(acc[t][r], 0.0f);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 hp[kTiles - 1] = {}, hv[kTiles - 1] = {};
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int t = 0; t < kTiles - 1; ++t) {
                hp[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(whp[r], acc[t][r], hp[t], 0, 0, 0);
                hv[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(whv[r], acc[t][r], hv[t], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        // the two lane halves hold the two halves of the wave's channels for the same pixel: one half exchange adds two rows at once (lanes 0..31
        // end with row a, lanes 32..63 with row b of the pair)
        float* part = lds + oPart;                                            // [wave][8 rows][256 pixels]
#pragma unroll
        for (int t = 0; t < kTiles - 1; ++t) {
            float* dst = part + (wave * 8 + (lane >> 5)) * 256 + t * 32 + (lane & 31);
            dst[0 * 256] = add_halves(hp[t][0], hp[t][1]);
            dst[2 * 256] = add_halves(hp[t][2], hp[t][3]);
            dst[4 * 256] = add_halves(hv[t][0], hv[t][1]);
        }
        stamp(10);
        // the corner pixel: the halves' k sums added, bias, ReLU -- four channels per lane, the same four on the four lanes of a block and on both
        // halves -- then six dot products over the wave's 32 channels = over the eight blocks of a half
        {
            const float bias[4] = {bias3c.x, bias3c.y, bias3c.z, bias3c.w};
            float x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = fmaxf(add_halves(corner[i], corner[i]) + bias[i], 0.0f);
#pragma unroll
            for (int j = 0; j < kHeadRows; ++j) {
                float v = ((x[0] * whc[j].x + x[1] * whc[j].y) + x[2] * whc[j].z) + x[3] * whc[j].w;
                // the eight blocks of a half = lanes l, l + 4, .. l + 28: two row rotations (DPP) add the four of a row of 16, one exchange the two rows
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));      // row_ror:8
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));      // row_ror:4
                v += __shfl_xor(v, 16, 64);
                if (lane == 0) part[(wave * 8 + j) * 256 + kPix - 1] = v;
            }
        }
        stamp(11);
        __syncthreads();
        stamp(4);
        // ---- sum the four waves' partial head outputs, bias, ReLU, flatten (pixel, channel) ----
        float* pf = prm.pflat + static_cast<size_t>(img) * 4 * kPix;
        float* vf = prm.vflat + static_cast<size_t>(img) * 2 * kPix;
        // policy outputs first (index = pixel * 4 + channel), then value outputs (pixel * 2 + channel): coalesced stores
#pragma unroll
        for (int it = 0; it < kOutPerThread; ++it) {
            const int i = threadIdx.x + 256 * it;
            if (out_from[it] < 0) continue;
            const float* src = part + out_from[it];
            const float v = fmaxf((((src[0 * 8 * 256] + src[1 * 8 * 256]) + src[2 * 8 * 256]) + src[3 * 8 * 256]) + out_bias[it], 0.0f);
            if (i < 4 * kPix) pf[i] = v; else vf[i - 4 * kPix] = v;
        }
        // no barrier here: the partials are next written behind the two barriers of the next position's layers 1 and 2
        stamp(5);
    }
    if (prm.prof && blockIdx.x == 0 && threadIdx.x == 0)
        for (int k = 0; k < 12; ++k) prm.prof[k] = t_stage[k];
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The three dense layers behind the trunk (network/model_tf.py:53-66): policy = softmax(dense 900 -> 225), value = tanh(dense 64 -> 1 of
// relu(dense 450 -> 64)), 0.46 MFLOP per position -- 1 % of the network, one small kernel.  One workgroup takes 16 positions: their 1 350 head
// activations are staged in LDS, every layer is Out[position][output] = sum_k Act[position][k] W^T[k][output] on v_mfma_f32_16x16x4_f32 (A =
// activations, one conflict-free ds_read_b32 per step: row stride 900 and 452 floats = 4 banks; B = weights, packed on the host in lane order
// per (wave, step of 4 k, output tile) and streamed from L2).  Wave w owns the policy's output tiles w, w + 4, w + 8, w + 12 (225 outputs = 15 tiles of
// 16; the 16th is zero weights) and tile w of the 64 hidden units; the hidden layer's 450 k ride along the first 120 steps of the policy's 228.
// Steps come in bodies of twelve with all addresses `register + immediate` (vector instructions between f32 MFMAs cost 20-30 clocks each, a
// load ~20 whatever its width: see conv_tiles()), a lane's B operands of four steps are one 16-byte buffer load, and a body fetches the NEXT
// body's weights before its own MFMAs start (two register sets).  Then bias, the logits and hidden
// units through LDS, and per position one wavefront's softmax / dot product + tanh.
constexpr int kDensePos = 16;                      // positions per workgroup
constexpr int kDenseBody = 12;                     // steps (of 4 k) per body = three groups of four (one 16-byte weight load per lane and group)
constexpr int kPolicySteps = 225;                  // -> 228 = 19 bodies; the hidden layer's 113 steps -> 120 ride along bodies 0 .. 9
constexpr int kDenseBodies = 19, kHiddenBodies = 10;
constexpr int kVfStride = 452;
constexpr int oDensePf = 0, oDenseVf = kDensePos * 900, kDenseLdsFloats = oDenseVf + kDensePos * kVfStride + 32;
// after the MFMAs the logits [16][256] take the place of the policy activations, the hidden units [16][64] that of the value activations

struct DenseParams {
    const float* pflat; const float* vflat; int n;
    const float* w;                                // [4 waves][19 + 1 bodies][4 policy tiles (wave + 4 q) | 1 hidden tile (wave)][3 groups][64 lanes][4 steps]:
                                                   // what a wave fetches for a body is one contiguous 15 KB run of 16-byte loads
    const float* bp;                               // [256]: the policy's biases, 0 behind the 225th
    const float* bhid; const float* wo; float bo;  // [64], [64]
    float* value; float* probs;                    // [n], [n][225]
};

__global__ __launch_bounds__(256)
void pvnet_dense_kernel(DenseParams prm) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pos0 = blockIdx.x * kDensePos, npos = min(kDensePos, prm.n - pos0);
    // ---- stage the 16 positions' activations (rows of positions past the batch are zeros): all loads first (16 and 8 bytes per lane), then the
    //      LDS writes; the policy rows are contiguous in both places, the value rows go from 450 to 452 floats apart ----
    {
        constexpr int kPfVec = kDensePos * 900 / 4;                                       // 3 600 float4, and as many float2 of the value rows
        static_assert(kPfVec == kDensePos * 450 / 2, "one index walks both");
        constexpr int kPerThread = (kPfVec + 255) / 256;                                  // 15 of each per thread
        const float4* src = reinterpret_cast<const float4*>(prm.pflat + static_cast<size_t>(pos0) * 900);
        const float2* srcv = reinterpret_cast<const float2*>(prm.vflat + static_cast<size_t>(pos0) * 450);
        float4 p4[kPerThread];
        float2 v2[kPerThread];
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const int i = threadIdx.x + 256 * j;
            p4[j] = i < npos * 225 ? src[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);      // 225 float4 per row
            v2[j] = i < npos * 225 ? srcv[i] : make_float2(0.0f, 0.0f);                  // 225 float2 per row
        }
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < kPfVec) {
                reinterpret_cast<float4*>(lds + oDensePf)[i] = p4[j];
                const int r = i / 225, c = i - r * 225;
                *reinterpret_cast<float2*>(lds + oDenseVf + r * kVfStride + 2 * c) = v2[j];
            }
        }
        if (threadIdx.x < kDensePos + 16) {                                             // the two pad columns of every row and the 32 floats behind the last
            if (threadIdx.x < kDensePos) *reinterpret_cast<float2*>(lds + oDenseVf + threadIdx.x * kVfStride + 450) = make_float2(0.0f, 0.0f);
            else *reinterpret_cast<float2*>(lds + oDenseVf + kDensePos * kVfStride + 2 * (threadIdx.x - kDensePos)) = make_float2(0.0f, 0.0f);
        }
    }
    __syncthreads();
    f32x4 acc[4] = {}, acch = {};
    const char* a_pf = reinterpret_cast<const char*>(lds + oDensePf + (lane & 15) * 900 + (lane >> 4));        // + 16 bytes per step
    const char* a_vf = reinterpret_cast<const char*>(lds + oDenseVf + (lane & 15) * kVfStride + (lane >> 4));
    // Weight fetches are buffer loads: descriptor of this wave's run + (the lane's byte offset, one register) + a scalar offset that moves once per
    // body + an immediate -- no vector arithmetic inside the MFMA stream (flat loads made the compiler carry 64-bit pointers in vector registers).
    constexpr int kBodyBytes = kDenseBody * 5 * 64 * 4, kWaveBytes = (kDenseBodies + 1) * kBodyBytes;
    const auto w_mine = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(prm.w) + static_cast<size_t>(__builtin_amdgcn_readfirstlane(wave)) * (kWaveBytes / 4), 0, kWaveBytes, 0x00020000);
    const int lane_bytes = lane * 16;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v bw[2][kDenseBody / 4][4], bh[2][kDenseBody / 4];                   // [register set][group of four steps][tile]: a lane's B operands of four steps
    auto fetch = [&](f32x4v (&w4)[kDenseBody / 4][4], f32x4v (&w1)[kDenseBody / 4], int body, auto with_hidden) {
#pragma unroll
        for (int g = 0; g < kDenseBody / 4; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) w4[g][q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(w_mine, lane_bytes + (q * 3 + g) * 1024, body * kBodyBytes, 0));
            if (decltype(with_hidden)::value) w1[g] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(w_mine, lane_bytes + (4 * 3 + g) * 1024, body * kBodyBytes, 0));
        }
    };
    auto body_mfmas = [&](const f32x4v (&w4)[kDenseBody / 4][4], const f32x4v (&w1)[kDenseBody / 4], int body, auto with_hidden) {
        const char* ap = a_pf + body * kDenseBody * 16;
        const char* av = a_vf + body * kDenseBody * 16;
#pragma unroll
        for (int s = 0; s < kDenseBody; ++s) {
            const float a = *reinterpret_cast<const float*>(ap + s * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w4[s / 4][q][s % 4], acc[q], 0, 0, 0);
            if (decltype(with_hidden)::value) acch = __builtin_amdgcn_mfma_f32_16x16x4f32(*reinterpret_cast<const float*>(av + s * 16), w1[s / 4][s % 4], acch, 0, 0, 0);
        }
    };
    constexpr std::integral_constant<bool, true> yes{};
    constexpr std::integral_constant<bool, false> no{};
    // The 19 bodies are unrolled (straight-line code): as a loop the two register sets are loop-carried values that the compiler copies at the
    // loop header behind an s_waitcnt vmcnt(0) -- which exposes the fetch it was meant to hide.  The hidden layer's weights are fetched for
    // bodies 0 .. 9 only; body 18 fetches the padding.
    fetch(bw[0], bh[0], 0, yes);
#pragma unroll
    for (int body = 0; body < kDenseBodies; ++body) {
        if (body + 1 < kHiddenBodies) fetch(bw[(body + 1) & 1], bh[(body + 1) & 1], body + 1, yes); else fetch(bw[(body + 1) & 1], bh[(body + 1) & 1], body + 1, no);
        __builtin_amdgcn_sched_barrier(0);
        if (body < kHiddenBodies) body_mfmas(bw[body & 1], bh[body & 1], body, yes); else body_mfmas(bw[body & 1], bh[body & 1], body, no);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                                          // everybody is done with the activations: their place is reused
    // ---- C/D: column (output) = lane & 15, row (position) = 4 (lane >> 4) + register ----
    float* logits = lds + oDensePf;                                           // [16][256]
    float* hidden = lds + oDenseVf;                                           // [16][64]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = (wave + 4 * q) * 16 + (lane & 15);
        const float bias = prm.bp[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) logits[(4 * (lane >> 4) + r) * 256 + col] = acc[q][r] + bias;
    }
    {
        const int col = wave * 16 + (lane & 15);
        const float bias = prm.bhid[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) hidden[(4 * (lane >> 4) + r) * 64 + col] = fmaxf(acch[r] + bias, 0.0f);
    }
    __syncthreads();
    // ---- per position (four per wave, side by side so that their reductions overlap): softmax over the 225 logits, tanh of the hidden units' dot
    //      product ----
    const float wo = prm.wo[lane];
    float x[4][4], m[4], sum[4], v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * wave + i;
        m[i] = -INFINITY;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int c = lane + 64 * e; x[i][e] = c < kPix ? logits[row * 256 + c] : -INFINITY; m[i] = fmaxf(m[i], x[i][e]); }
        v[i] = hidden[row * 64 + lane] * wo;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) { m[i] = fmaxf(m[i], __shfl_xor(m[i], s, 64)); v[i] += __shfl_xor(v[i], s, 64); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sum[i] = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[i][e] = expf(x[i][e] - m[i]); sum[i] += x[i][e]; }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1)
#pragma unroll
        for (int i = 0; i < 4; ++i) sum[i] += __shfl_xor(sum[i], s, 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * wave + i;
        if (row >= npos) break;
        float* out = prm.probs + static_cast<size_t>(pos0 + row) * kPix;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int c = lane + 64 * e; if (c < kPix) out[c] = x[i][e] / sum[i]; }
        if (lane == 0) prm.value[pos0 + row] = tanhf(v[i] + prm.bo);
    }
}

// A operands of one layer in lane order, four consecutive k-pairs of a lane side by side: [cout tile][k-pair / 4][64 lanes][4]; k-pair order as
// conv_tiles() walks it (layer 1's 27 k-pairs: seven groups, the last one padded)
void pack_layer(const float* w /* [cout][cin][3][3] */, int cin, int cout, std::vector<float>& out) {
    const int CP = cin >= 16 ? 8 : cin / 2, chunks = cin / 2 / CP, steps = chunks * 9, kps = steps * CP, groups = (kps + 3) / 4;
    out.assign((static_cast<size_t>(cout / 32) * groups + 2 * CP / 4 + 1) * 256, 0.0f);      // + the two steps conv_tiles() fetches past the last tile's end
    for (int tile = 0; tile < cout / 32; ++tile)
        for (int step = 0; step < steps; ++step) {
            int chunk, tap;
            step_chunk_tap(step, chunk, tap);
            for (int cp = 0; cp < CP; ++cp) {
                const int kp = step * CP + cp;
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = tile * 32 + (lane & 31), ci = chunk * 2 * CP + 2 * cp + (lane >> 5);
                    out[((static_cast<size_t>(tile) * groups + kp / 4) * 64 + lane) * 4 + (kp & 3)] = w[(static_cast<size_t>(co) * cin + ci) * 9 + tap];
                }
            }
        }
}

}  // namespace

struct gmk_pvnet {
    float *d_w1 = nullptr, *d_w2 = nullptr, *d_w3 = nullptr, *d_wh = nullptr, *d_b = nullptr;
    bool attr_set = false;
    // the dense layers (gmk_pvnet_set_dense): packed weights, biases (policy [256] | hidden [64] | output weights [64]), the output bias, and the
    // head activations between the two kernels of gmk_pvnet_evaluate ([capacity][900 + 450], grown on demand)
    float *d_wp = nullptr, *d_dense = nullptr, *d_flat = nullptr;
    float b_out = 0.0f;
    bool has_dense = false, dense_attr_set = false;
    int flat_capacity = 0;
};

extern "C" int gmk_pvnet_destroy(gmk_pvnet* net) {
    if (!net) return GMK_OK;
    (void)hipFree(net->d_w1); (void)hipFree(net->d_w2); (void)hipFree(net->d_w3); (void)hipFree(net->d_wh); (void)hipFree(net->d_b);
    (void)hipFree(net->d_wp); (void)hipFree(net->d_dense); (void)gmk::device_free(net->d_flat);
    delete net;
    return GMK_OK;
}

extern "C" int gmk_pvnet_create(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                                const float* wp, const float* bp, const float* wv, const float* bv, gmk_pvnet** out) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !wp || !bp || !wv || !bv || !out) { gmk::set_error("gmk_pvnet_create: bad arguments"); return GMK_ERR_ARG; }
    std::vector<float> p1, p2, p3, ph(4 * 32 * 64 + 4 * kHeadRows * 32, 0.0f), bias(32 + 64 + 128 + 8, 0.0f);
    pack_layer(w1, 6, 32, p1);
    pack_layer(w2, 32, 64, p2);
    pack_layer(w3, 64, 128, p3);
    for (int wave = 0; wave < 4; ++wave)                         // 4x4x1 A operands: lane l carries row l % 4 of its block, for the channel that accumulator
        for (int r = 0; r < 16; ++r)                             // register r of wave `wave` holds on that lane half
            for (int lane = 0; lane < 64; ++lane) {
                const int c = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 3;
                ph[(wave * 32 + r) * 64 + lane] = wp[j * 128 + c];
                ph[(wave * 32 + 16 + r) * 64 + lane] = j < 2 ? wv[j * 128 + c] : 0.0f;
            }
    for (int wave = 0; wave < 4; ++wave)                         // ... and per channel for the corner pixel
        for (int j = 0; j < kHeadRows; ++j)
            for (int c = 0; c < 32; ++c) ph[4 * 32 * 64 + (wave * kHeadRows + j) * 32 + c] = j < 4 ? wp[j * 128 + 32 * wave + c] : wv[(j - 4) * 128 + 32 * wave + c];
    std::memcpy(&bias[0], b1, 32 * 4); std::memcpy(&bias[32], b2, 64 * 4); std::memcpy(&bias[96], b3, 128 * 4);
    std::memcpy(&bias[224], bp, 4 * 4); std::memcpy(&bias[228], bv, 2 * 4);
    gmk_pvnet* net = new gmk_pvnet;
    const bool ok = hipMalloc(&net->d_w1, p1.size() * 4) == hipSuccess && hipMalloc(&net->d_w2, p2.size() * 4) == hipSuccess &&
                    hipMalloc(&net->d_w3, p3.size() * 4) == hipSuccess && hipMalloc(&net->d_wh, ph.size() * 4) == hipSuccess &&
                    hipMalloc(&net->d_b, bias.size() * 4) == hipSuccess &&
                    hipMemcpy(net->d_w1, p1.data(), p1.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                    hipMemcpy(net->d_w2, p2.data(), p2.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                    hipMemcpy(net->d_w3, p3.data(), p3.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                    hipMemcpy(net->d_wh, ph.data(), ph.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                    hipMemcpy(net->d_b, bias.data(), bias.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { gmk_pvnet_destroy(net); gmk::set_error("gmk_pvnet_create: device allocation or copy failed"); return GMK_ERR_HIP; }
    *out = net;
    return GMK_OK;
}

namespace {

int launch_trunk(gmk_pvnet* net, const float* d_states, int n, float* d_pflat, float* d_vflat, hipStream_t stream) {
    gmk::DeviceState& st = gmk::device_state();
    const size_t lds = static_cast<size_t>(kLdsFloats) * 4;
    if (!net->attr_set) {
        GMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pvnet_trunk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        net->attr_set = true;
    }
    PvParams prm;
    prm.states = d_states; prm.n = n;
    prm.w1 = net->d_w1; prm.w2 = net->d_w2; prm.w3 = net->d_w3; prm.wh = net->d_wh; prm.whc = net->d_wh + 4 * 32 * 64;
    prm.b1 = net->d_b; prm.b2 = net->d_b + 32; prm.b3 = net->d_b + 96; prm.bh = net->d_b + 224;
    prm.pflat = d_pflat; prm.vflat = d_vflat;
    static const bool profile = gmk::profile_env("GMK_PVNET_PROFILE") != nullptr;
    prm.prof = nullptr;
    if (profile) GMK_HIP_CHECK(hipMalloc(&prm.prof, 12 * sizeof(unsigned long long)));
    const int grid = std::min(n, st.cu_count > 0 ? st.cu_count : 256);        // one workgroup per CU, each takes every grid-th position
    hipLaunchKernelGGL(pvnet_trunk_kernel, dim3(grid), dim3(256), lds, stream, prm);
    GMK_HIP_CHECK(hipGetLastError());
    if (profile) {
        unsigned long long t[12];
        GMK_HIP_CHECK(hipDeviceSynchronize());
        GMK_HIP_CHECK(hipMemcpy(t, prm.prof, sizeof t, hipMemcpyDeviceToHost));
        (void)hipFree(prm.prof);
        const double per = 1.0 / ((n + grid - 1) / grid);
        std::fprintf(stderr, "[GMK_PVNET_PROFILE] shader clocks per position (workgroup 0): layer 1 %.0f, layer 2 %.0f (MFMA loop %.0f, stores %.0f, barrier %.0f, shared tile %.0f), "
                             "wait %.0f, layer 3 %.0f (operands %.0f, MFMA loop %.0f), heads %.0f (MFMAs + stores %.0f, corner %.0f, barrier %.0f), output %.0f\n",
                     t[0] * per, (t[6] + t[7] + t[8] + t[1]) * per, t[6] * per, t[7] * per, t[8] * per, t[1] * per, t[2] * per, (t[9] + t[3]) * per, t[9] * per, t[3] * per,
                     (t[10] + t[11] + t[4]) * per, t[10] * per, t[11] * per, t[4] * per, t[5] * per);
    }
    return GMK_OK;
}

}  // namespace

extern "C" int gmk_pvnet_forward(gmk_pvnet* net, const float* d_states, int n, float* d_pflat, float* d_vflat, void* stream) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (!net || n < 0 || (n > 0 && (!d_states || !d_pflat || !d_vflat))) { gmk::set_error("gmk_pvnet_forward: bad arguments"); return GMK_ERR_ARG; }
    if (n == 0) return GMK_OK;
    return launch_trunk(net, d_states, n, d_pflat, d_vflat, static_cast<hipStream_t>(stream));
}

extern "C" int gmk_pvnet_set_dense(gmk_pvnet* net, const float* w_policy, const float* b_policy, const float* w_hidden, const float* b_hidden,
                                   const float* w_out, float b_out) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (!net || !w_policy || !b_policy || !w_hidden || !b_hidden || !w_out) { gmk::set_error("gmk_pvnet_set_dense: bad arguments"); return GMK_ERR_ARG; }
    // B operands of v_mfma_f32_16x16x4_f32 in lane order: lane l carries W[output = 16 tile + (l & 15)][k = 4 step + (l >> 4)]; per wave and body
    // the nine steps' four policy tiles and hidden tile side by side (zeros: the 16th policy tile, k >= 450 of the hidden layer, the pad body)
    std::vector<float> wp(static_cast<size_t>(4) * (kDenseBodies + 1) * kDenseBody * 5 * 64, 0.0f), dense(256 + 64 + 64, 0.0f);
    for (int wave = 0; wave < 4; ++wave)
        for (int step = 0; step < kPolicySteps; ++step)
            for (int q = 0; q < 5; ++q)
                for (int lane = 0; lane < 64; ++lane) {
                    const int k = 4 * step + (lane >> 4), body = step / kDenseBody, g = (step % kDenseBody) / 4;
                    float v = 0.0f;
                    if (q < 4) { const int o = 16 * (wave + 4 * q) + (lane & 15); if (o < kPix) v = w_policy[static_cast<size_t>(o) * 900 + k]; }
                    else if (k < 450) v = w_hidden[static_cast<size_t>(16 * wave + (lane & 15)) * 450 + k];
                    wp[((((static_cast<size_t>(wave) * (kDenseBodies + 1) + body) * 5 + q) * 3 + g) * 64 + lane) * 4 + (step & 3)] = v;
                }
    std::memcpy(&dense[0], b_policy, kPix * 4); std::memcpy(&dense[256], b_hidden, 64 * 4); std::memcpy(&dense[320], w_out, 64 * 4);
    if (!net->d_wp) {
        const bool ok = hipMalloc(&net->d_wp, wp.size() * 4) == hipSuccess && hipMalloc(&net->d_dense, dense.size() * 4) == hipSuccess;
        if (!ok) { gmk::set_error("gmk_pvnet_set_dense: device allocation failed"); return GMK_ERR_HIP; }
    }
    GMK_HIP_CHECK(hipMemcpy(net->d_wp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
    GMK_HIP_CHECK(hipMemcpy(net->d_dense, dense.data(), dense.size() * 4, hipMemcpyHostToDevice));
    net->b_out = b_out;
    net->has_dense = true;
    return GMK_OK;
}

extern "C" int gmk_pvnet_evaluate(gmk_pvnet* net, const float* d_states, int n, float* d_value, float* d_probs, void* stream) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (!net || n < 0 || (n > 0 && (!d_states || !d_value || !d_probs))) { gmk::set_error("gmk_pvnet_evaluate: bad arguments"); return GMK_ERR_ARG; }
    if (!net->has_dense) { gmk::set_error("gmk_pvnet_evaluate: the dense layers have not been set (gmk_pvnet_set_dense)"); return GMK_ERR_STATE; }
    if (n == 0) return GMK_OK;
    if (n > net->flat_capacity) {                                // the head activations between the two kernels; grows, never shrinks
        if (net->d_flat) { GMK_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream))); (void)gmk::device_free(net->d_flat); net->d_flat = nullptr; net->flat_capacity = 0; }
        const int capacity = std::max(n, 1024);
        GMK_HIP_CHECK(gmk::device_malloc(&net->d_flat, static_cast<size_t>(capacity) * 1350 * sizeof(float)));
        net->flat_capacity = capacity;
    }
    float* d_pflat = net->d_flat;
    float* d_vflat = net->d_flat + static_cast<size_t>(net->flat_capacity) * 900;
    const int rc = launch_trunk(net, d_states, n, d_pflat, d_vflat, static_cast<hipStream_t>(stream));
    if (rc != GMK_OK) return rc;
    if (!net->dense_attr_set) {
        GMK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pvnet_dense_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        net->dense_attr_set = true;
    }
    DenseParams prm;
    prm.pflat = d_pflat; prm.vflat = d_vflat; prm.n = n;
    prm.w = net->d_wp; prm.bp = net->d_dense; prm.bhid = net->d_dense + 256; prm.wo = net->d_dense + 320; prm.bo = net->b_out;
    prm.value = d_value; prm.probs = d_probs;
    hipLaunchKernelGGL(pvnet_dense_kernel, dim3((n + kDensePos - 1) / kDensePos), dim3(256), static_cast<size_t>(kDenseLdsFloats) * 4, static_cast<hipStream_t>(stream), prm);
    GMK_HIP_CHECK(hipGetLastError());
    return GMK_OK;
}
