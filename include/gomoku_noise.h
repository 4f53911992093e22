/* gomoku_noise.h -- the counter-based Dirichlet sampler of Default::AddNoise ("sampler 1").
 *
 * The reference mixes root noise into the priors before EVERY search (core/lib/src/MCTS.cpp:179-183,
 * core/lib/include/algorithms/MonteCarlo.hpp:97-108):
 *     P <- (1 - epsilon) P + epsilon * normalized(gamma(alpha, 1) per entry with P != 0)      (Statistical.hpp:29-34)
 * and draws from std::gamma_distribution<float> over a random_device-seeded std::mt19937 -- a stream nobody can reproduce
 * (no seed API) and one that is sequential, i.e. it can only be drawn on the host, one game after the other.
 *
 * This header defines the stream the device-resident self-play loops draw from instead: every draw is a pure function of
 *     (seed; global game id, stones on the root board, cell)
 * through Philox4x32-10, so a wavefront draws a root's 225 values side by side, inside the kernel that searches, and the
 * CPU oracle (oracle/go_mcts.c, oracle/go_trad.c) draws the same values one by one.  The file is written ONCE and compiled
 * by gcc (C99, the oracle), by g++ and by hipcc (host and gfx950 device code): every operation in it is an IEEE-754 binary64
 * +, -, *, /, sqrt or an integer operation -- no libm / OCML transcendental, whose last bits differ between the two sides --
 * so the draws are bit-identical wherever they are computed.  Compile with -ffp-contract=off (a fused multiply-add rounds once).
 *
 * Algorithm: Marsaglia & Tsang, "A simple method for generating gamma variables", ACM TOMS 26(3), 2000, with the alpha < 1
 * boost gamma(alpha) = gamma(alpha + 1) * U^(1 / alpha); the normal variate by Marsaglia's polar method.  One ATTEMPT consumes
 * exactly one Philox block (words 0, 1: the polar pair; word 2: the acceptance test; word 3: the boost), a rejected attempt
 * moves on to the next block: counter = (game id, stones, 'dirc', cell | attempt << 8), key = seed.
 * The result is returned as float (the reference's distribution is gamma_distribution<float>); values below 1e-18 are returned
 * as 0 so that neither they nor their squares are subnormal floats anywhere downstream (with alpha = 0.05 a third of all draws
 * are that small; after the normalisation they are far below one ulp of the prior they are added to).
 */
#ifndef GOMOKU_NOISE_H_
#define GOMOKU_NOISE_H_

#include <stdint.h>
#if !defined(__HIP_DEVICE_COMPILE__)
#include <math.h>
#endif

#if defined(__HIPCC__)
#define GMK_NOISE_FN __host__ __device__ static inline
#else
#define GMK_NOISE_FN static inline
#endif

#define GMK_NOISE_TAG 0x64697263u /* 'dirc' */
#define GMK_NOISE_MAX_ATTEMPTS 64u

GMK_NOISE_FN void gmk_noise_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

GMK_NOISE_FN double gmk_noise_from_bits(uint64_t b) { double d; __builtin_memcpy(&d, &b, 8); return d; }
GMK_NOISE_FN uint64_t gmk_noise_to_bits(double d) { uint64_t b; __builtin_memcpy(&b, &d, 8); return b; }

/* a 32-bit word as a uniform variate in (0, 1): (w + 1/2) / 2^32, exact in binary64 */
GMK_NOISE_FN double gmk_noise_uniform(uint32_t w) { return ((double)w + 0.5) * 2.3283064365386962890625e-10; }

/* natural logarithm of a positive NORMAL double: x = m 2^e with m in (sqrt(1/2), sqrt(2)], log m = 2 atanh((m - 1) / (m + 1)) as its
 * odd series up to the 23rd power (|s| <= 0.1716: the first term left out is < 2e-19 relative), Horner, fixed order */
GMK_NOISE_FN double gmk_noise_log(double x) {
    const uint64_t b = gmk_noise_to_bits(x);
    int e = (int)(b >> 52) - 1023;
    double m = gmk_noise_from_bits((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), s2 = s * s;
    double p = 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0;
    p = p * s2 + 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0;
    p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2 + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * (s * p);
}

/* exp(x) for x <= 0: x = k ln 2 + r with |r| <= ln 2 / 2 (ln 2 in two parts), exp r as its Taylor polynomial of degree 13
 * (the first term left out is < 4e-18), times 2^k through the exponent field; below -700 the result is 0 (far below the 1e-18
 * every caller flushes at) */
GMK_NOISE_FN double gmk_noise_exp(double x) {
    if (x < -700.0) return 0.0;
    const int k = (int)(x * 1.4426950408889634 - 0.5);
    const double r = (x - (double)k * 0.693147180369123816490) - (double)k * 1.90821492927058770002e-10;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    return p * gmk_noise_from_bits((uint64_t)(k + 1023) << 52);
}

/* one gamma(alpha, 1) variate: the draw of Stats::DirichletNoise for the entry `cell` of the root that game `game_id` reaches with
 * `stones` stones on the board */
GMK_NOISE_FN float gmk_noise_gamma(float alpha, uint32_t game_id, uint32_t stones, uint32_t cell, uint32_t seed_lo, uint32_t seed_hi) {
    const double a = (double)alpha, a1 = alpha < 1.0f ? a + 1.0 : a;
    const double d = a1 - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (uint32_t attempt = 0; attempt < GMK_NOISE_MAX_ATTEMPTS; ++attempt) {
        uint32_t w[4];
        gmk_noise_philox(game_id, stones, GMK_NOISE_TAG, cell | attempt << 8, seed_lo, seed_hi, w);
        const double v1 = 2.0 * gmk_noise_uniform(w[0]) - 1.0, v2 = 2.0 * gmk_noise_uniform(w[1]) - 1.0;
        const double s = v1 * v1 + v2 * v2;
        if (!(s < 1.0) || s < 1e-300) continue;                       /* outside the unit disc (polar method) */
        const double x = v1 * sqrt(-2.0 * gmk_noise_log(s) / s);     /* a standard normal variate */
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        const double u = gmk_noise_uniform(w[2]), x2 = x * x;
        if (!(u < 1.0 - 0.0331 * (x2 * x2)) && !(gmk_noise_log(u) < 0.5 * x2 + d * (1.0 - v + gmk_noise_log(v)))) continue;
        double g = d * v;
        if (alpha < 1.0f) g = g * gmk_noise_exp(gmk_noise_log(gmk_noise_uniform(w[3])) / a);
        return g < 1e-18 ? 0.0f : (float)g;
    }
    return 0.0f;                                                      /* (sixty-four rejections in a row: probability < 1e-38) */
}

/* The order in which the 225 squares of a root's draws are added up for VectorXf::normalized() (Eigen's own order is not a
 * contract): the one a wavefront computes without a shuffle through memory, and the one oracle/go_trad.c already fixes for the
 * Heuristic's norms.  Lane l of 64 adds the entries l, l + 64, l + 128, l + 192 in that order; then a binary tree inside every row
 * of 16 lanes (lane i takes lane i + 8, then + 4, + 2, + 1; a lane beyond the row contributes 0); then (row 0 + row 1) + (row 2 + row 3). */
GMK_NOISE_FN float gmk_noise_sum225(const float* v) {
    float p[64];
    for (int l = 0; l < 64; ++l) {
        p[l] = v[l];
        for (int j = 1; j < 4; ++j) if (l + 64 * j < 225) p[l] += v[l + 64 * j];
    }
    for (int off = 8; off >= 1; off >>= 1)
        for (int l = 0; l < 64; ++l) p[l] = p[l] + (((l & 15) + off < 16) ? p[l + off] : 0.0f);
    return (p[0] + p[16]) + (p[32] + p[48]);
}

/* Default::AddNoise on a root's priors by cell (p[i] == 0: no child, no draw): the serial statement of what the kernels do with one
 * cell per lane and round */
GMK_NOISE_FN void gmk_noise_mix225(float* p, float alpha, float epsilon, uint32_t game_id, uint32_t stones, uint32_t seed_lo, uint32_t seed_hi) {
    float noise[225], sq[225];
    for (int i = 0; i < 225; ++i) {
        p[i] *= 1 - epsilon;                                          /* prior_probs *= 1 - epsilon */
        noise[i] = p[i] != 0.0f ? gmk_noise_gamma(alpha, game_id, stones, (uint32_t)i, seed_lo, seed_hi) : 0.0f;
        sq[i] = noise[i] * noise[i];
    }
    const float z = gmk_noise_sum225(sq);
    if (z > 0.0f) {                                                   /* normalized(): a zero vector stays zero */
        const float nrm = sqrtf(z);
        for (int i = 0; i < 225; ++i) noise[i] = noise[i] / nrm;
    }
    for (int i = 0; i < 225; ++i) p[i] += epsilon * noise[i];
}

#endif /* GOMOKU_NOISE_H_ */
