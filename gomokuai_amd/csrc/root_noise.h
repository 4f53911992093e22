// root_noise.h -- Default::AddNoise (core/lib/include/algorithms/MonteCarlo.hpp:97-108) on the host, shared by the tree searches
// (K3 gmk_mcts_add_root_noise, K6 / K8 gmk_trad_add_root_noise, K7 gmk_az_add_root_noise):
//   P <- (1 - epsilon) * P + epsilon * normalized(gamma(alpha, 1) per entry with P != 0)        (Statistical.hpp:29-34)
// The draws use the toolchain's own std::gamma_distribution<float> over std::mt19937, the distribution code the reference
// runs; only the engine's seed differs (the reference: random_device; here: Philox of (seed; game id, stones on the root
// board, 'nois')), so searches stay reproducible.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <random>
#include <thread>
#include <vector>

#include "philox.h"

namespace gmk {

inline uint32_t root_noise_engine_seed(uint64_t seed, uint32_t game_id, uint32_t stones) {
    return philox4x32_10(game_id, stones, 0x6E6F6973u /* 'nois' */, 0u, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32)).v[0];
}

// p[0 .. n): the priors in the order the reference's vector holds them (ascending cell); entries that are 0 take no draw
inline void mix_root_noise(float* p, int n, float alpha, float epsilon, uint32_t engine_seed) {
    std::mt19937 engine(engine_seed);
    std::gamma_distribution<float> gamma(alpha, 1.0f);
    float noise[225], sq = 0.0f;
    for (int i = 0; i < n; ++i) {
        p[i] *= 1 - epsilon;                                       // prior_probs *= 1 - epsilon
        noise[i] = p[i] ? gamma(engine) : 0.0f;
        sq += noise[i] * noise[i];
    }
    const float norm = sq > 0.0f ? std::sqrt(sq) : 1.0f;           // VectorXf::normalized(): a zero vector stays zero
    for (int i = 0; i < n; ++i) p[i] += epsilon * (sq > 0.0f ? noise[i] / norm : noise[i]);
}

// fn(g) for g in [0, n) on the host's cores: a game's draws depend on nothing but its own seed, and seeding a Mersenne twister
// plus ~220 gamma draws per game is ~20 us, which at thousands of games per search would rival the search itself
template <class Fn>
inline void for_each_game(size_t n, Fn fn) {
    const size_t workers = std::min<size_t>({n / 64 + 1, std::max(1u, std::thread::hardware_concurrency()), 16});
    if (workers <= 1) { for (size_t g = 0; g < n; ++g) fn(g); return; }
    std::vector<std::thread> pool;
    for (size_t w = 0; w < workers; ++w)
        pool.emplace_back([=]() { for (size_t g = n * w / workers; g < n * (w + 1) / workers; ++g) fn(g); });
    for (std::thread& t : pool) t.join();
}

}  // namespace gmk
