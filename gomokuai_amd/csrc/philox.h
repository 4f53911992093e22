// philox.h -- Philox4x32-10 counter-based RNG (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as
// easy as 1, 2, 3", SC'11), shared by host workload generation and the device kernels.  Replaces the
// reference's random_device-seeded std::mt19937 (core/lib/src/Game.cpp:11-12), which has no seed API.
#pragma once
#include <cstdint>

#if defined(__HIP__)
#include <hip/hip_runtime.h>
#define GMK_HD __host__ __device__ __forceinline__
#else
#define GMK_HD inline
#endif

namespace gmk {

struct Philox4 { uint32_t v[4]; };

GMK_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
        const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
        const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
        c1 = static_cast<uint32_t>(p1);
        c3 = static_cast<uint32_t>(p0);
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

}  // namespace gmk
