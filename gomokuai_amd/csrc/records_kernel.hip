// records_kernel.hip -- K4 + K5: game records -> training tuples on the device.
//
// A self-play record (moves, per-move root visit counts, winner) is what gmk_mcts_advance leaves in HBM.  The
// reference turns every move of a game into a sample while it plays (agents/utils.py:29-41, 55-59):
//     ( Board.encoded_states()  uint8[6][15][15]   (core/py_ext/src/game_ext.hpp:87-104)
//     , Player.calc_score(player to move, winner)  float
//     , MCTS::evalState's pi        float32[225]   (core/lib/src/MCTS.cpp:104-117, Statistical.hpp:37-42) )
// and then augments it eight-fold (network/data_helper.py:36-55: rot90^i and fliplr of it, i = 0..3).
// Here one workgroup produces one sample (x8 when augmenting) straight from the record:
//   K4  the six feature planes are rebuilt by replaying the first t moves of the record,
//   K5  pi = softmax(log(v / |v|_2 + [v != 0] + eps) / tau) in double, tau = 1 below 15 stones else 0.01.
#include "capi_common.h"

namespace {

constexpr int kCells = 225;

// index permutation of network/data_helper.py:36-55: out[j] = in[perm(j)] for np.rot90(a, k) then optional np.fliplr
__device__ __forceinline__ int augment_source(int j, int k, bool flip) {
    int r = j / 15, c = j % 15;
    if (flip) c = 14 - c;                     // fliplr(b)[r][c] = b[r][14 - c]
    for (int i = 0; i < k; ++i) {             // rot90(a)[r][c] = a[c][14 - r]
        const int nr = c, nc = 14 - r;
        r = nr; c = nc;
    }
    return r * 15 + c;
}

__global__ __launch_bounds__(256)
void samples_from_records_kernel(const uint8_t* __restrict__ moves, const int32_t* __restrict__ lens, const uint16_t* __restrict__ visits,
                                 const int8_t* __restrict__ winner, const int32_t* __restrict__ sample_game, const int32_t* __restrict__ sample_move,
                                 int n_samples, int augment, uint8_t* __restrict__ out_states, float* __restrict__ out_values, float* __restrict__ out_pi) {
    __shared__ int8_t s_cell[kCells];
    __shared__ float s_pi[kCells];
    __shared__ float s_red[256];
    __shared__ double s_redd[256];
    const int s = blockIdx.x;
    if (s >= n_samples) return;
    const int g = sample_game[s], t = sample_move[s], tid = threadIdx.x;
    const uint8_t* mv = moves + static_cast<size_t>(g) * 225;

    // ---- K4: position before move t (black moves first), player to move, last two moves ----
    if (tid < kCells) s_cell[tid] = 0;
    __syncthreads();
    if (tid < t) s_cell[mv[tid]] = (tid & 1) ? -1 : 1;
    __syncthreads();
    const int cur = (t & 1) ? -1 : 1;

    // ---- K5: pi from the visit counts of move t ----
    const uint16_t* vrow = visits + (static_cast<size_t>(g) * 225 + static_cast<size_t>(t)) * 225;
    float v = tid < kCells ? static_cast<float>(vrow[tid]) : 0.0f;
    s_red[tid] = v * v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) s_red[tid] += s_red[tid + w]; __syncthreads(); }
    const float sq = s_red[0];
    if (sq > 0.0f) v = v / sqrtf(sq);                                 // VectorXf::normalized()
    v = v ? v + 1.0f : v;                                             // MCTS.cpp:112
    const float temperature = t < 15 ? 1.0f : 0.01f;                  // MCTS.cpp:114 (stones on the board = t)
    const float eps = 1.1920929e-07f;
    const double e = tid < kCells ? exp(static_cast<double>(logf(v + eps) / temperature)) : 0.0;   // Statistical.hpp:38-39
    __syncthreads();
    s_redd[tid] = e;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) s_redd[tid] += s_redd[tid + w]; __syncthreads(); }
    if (tid < kCells) { const float p = static_cast<float>(e / s_redd[0]); s_pi[tid] = p > eps ? p : 0.0f; }
    __syncthreads();

    const int copies = augment ? 8 : 1;
    for (int a = 0; a < copies; ++a) {
        const size_t o = static_cast<size_t>(s) * copies + a;
        const int k = a >> 1;
        const bool flip = (a & 1) != 0;
        if (tid < kCells) {
            const int src = augment ? augment_source(tid, k, flip) : tid;
            const int8_t c = s_cell[src];
            uint8_t* st = out_states + o * 6 * kCells;
            st[0 * kCells + tid] = c == cur;                                          // stones of the player to move
            st[1 * kCells + tid] = c == -cur;                                         // opponent's stones
            st[2 * kCells + tid] = c == 0;                                            // empties
            st[3 * kCells + tid] = (t >= 1 && mv[t - 1] == src);                      // last move
            st[4 * kCells + tid] = (t >= 2 && mv[t - 2] == src);                      // the move before it
            st[5 * kCells + tid] = cur == 1;                                          // all ones iff black is to move
            out_pi[o * kCells + tid] = s_pi[src];
        }
        if (tid == 0) out_values[o] = static_cast<float>(cur) * static_cast<float>(winner[g]);   // CalcScore (Game.h:34-36)
    }
}

}  // namespace

extern "C" int gmk_samples_from_records(const uint8_t* d_moves, const int32_t* d_lens, const uint16_t* d_visits, const int8_t* d_winner,
                                        const int32_t* d_sample_game, const int32_t* d_sample_move, int n_samples, int augment,
                                        uint8_t* d_states, float* d_values, float* d_pi, void* stream) {
    gmk::DeviceState& st = gmk::device_state();
    if (!st.ready) { gmk::set_error("gmk_init has not succeeded (no CPU fallback)"); return GMK_ERR_STATE; }
    if (n_samples < 0 || (n_samples > 0 && (!d_moves || !d_lens || !d_visits || !d_winner || !d_sample_game || !d_sample_move || !d_states || !d_values || !d_pi))) {
        gmk::set_error("gmk_samples_from_records: bad arguments");
        return GMK_ERR_ARG;
    }
    if (n_samples == 0) return GMK_OK;
    hipLaunchKernelGGL(samples_from_records_kernel, dim3(n_samples), dim3(256), 0, static_cast<hipStream_t>(stream),
                       d_moves, d_lens, d_visits, d_winner, d_sample_game, d_sample_move, n_samples, augment, d_states, d_values, d_pi);
    GMK_HIP_CHECK(hipGetLastError());
    return GMK_OK;
}
