// board_device.h -- bit-board helpers shared by the search kernels (mcts_kernel.hip, az_kernel.hip).
// A position is 16 row words: black stones in bits 0..14, white stones in bits 16..30 of word y.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace gmk {

__device__ __forceinline__ bool run_of_five(uint32_t m) { return (m & (m >> 1) & (m >> 2) & (m >> 3) & (m >> 4)) != 0; }

// Five or more in a row through (x, y) for the colour in bits [shift, shift+15) of the row words
// (Board::checkGameEnd, core/lib/src/Game.cpp:88-136).  rows[y * Stride].
template <int Stride>
__device__ __forceinline__ bool five_through(const uint32_t* rows, int x, int y, int shift) {
    const uint32_t own = (rows[y * Stride] >> shift) & 0x7FFFu;
    if (run_of_five(own)) return true;
    uint32_t v = 16u, d1 = 16u, d2 = 16u;                       // bit 4 = the stone itself
#pragma unroll
    for (int i = 1; i <= 4; ++i) {
        if (y + i < 15) {
            const uint32_t o = ((rows[(y + i) * Stride] >> shift) & 0x7FFFu) << 4;
            v |= ((o >> (x + 4)) & 1u) << (4 + i);
            d1 |= ((o >> (x + i + 4)) & 1u) << (4 + i);
            d2 |= ((o >> (x - i + 4)) & 1u) << (4 + i);
        }
        if (y - i >= 0) {
            const uint32_t o = ((rows[(y - i) * Stride] >> shift) & 0x7FFFu) << 4;
            v |= ((o >> (x + 4)) & 1u) << (4 - i);
            d1 |= ((o >> (x - i + 4)) & 1u) << (4 - i);
            d2 |= ((o >> (x + i + 4)) & 1u) << (4 - i);
        }
    }
    return run_of_five(v) || run_of_five(d1) || run_of_five(d2);
}

}  // namespace gmk
