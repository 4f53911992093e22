"""The counter-based Dirichlet sampler of Default::AddNoise (include/gomoku_noise.h), on the CPU: the header as gcc compiled it into the oracle
against an independent restatement in Python floats (IEEE binary64, the same operations in the same order), against libm, and against the
distribution it claims to draw from.  The GPU side of the same header is held to the oracle by tests/test_selfplay_gpu.py."""
import math
import struct

import numpy as np
import pytest

M32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def from_bits(b):
    return struct.unpack("<d", struct.pack("<Q", b))[0]


def py_log(x):
    b = bits(x)
    e = (b >> 52) - 1023
    m = from_bits((b & 0x000FFFFFFFFFFFFF) | 0x3FF0000000000000)
    if m > 1.4142135623730951:
        m, e = m * 0.5, e + 1
    s = (m - 1.0) / (m + 1.0)
    s2 = s * s
    p = 1.0 / 23.0
    for d in (21, 19, 17, 15, 13, 11, 9, 7, 5, 3):
        p = p * s2 + 1.0 / d
    p = p * s2 + 1.0
    return float(e) * 0.6931471805599453 + 2.0 * (s * p)


def py_exp(x):
    if x < -700.0:
        return 0.0
    k = int(x * 1.4426950408889634 - 0.5)
    r = (x - float(k) * 0.693147180369123816490) - float(k) * 1.90821492927058770002e-10
    p = 1.0 / 6227020800.0
    for d in (479001600.0, 39916800.0, 3628800.0, 362880.0, 40320.0, 5040.0, 720.0, 120.0, 24.0, 6.0):
        p = p * r + 1.0 / d
    p = p * r + 0.5
    p = p * r + 1.0
    p = p * r + 1.0
    return p * from_bits((k + 1023) << 52)


def uniform(w):
    return (float(w) + 0.5) * 2.3283064365386962890625e-10


def py_gamma(alpha, game, stones, cell, seed):
    alpha = float(np.float32(alpha))
    a1 = alpha + 1.0 if alpha < 1.0 else alpha
    d = a1 - 1.0 / 3.0
    c = 1.0 / math.sqrt(9.0 * d)
    for attempt in range(64):
        w = philox4x32_10((game, stones, 0x64697263, cell | attempt << 8), (seed & M32, seed >> 32))
        v1, v2 = 2.0 * uniform(w[0]) - 1.0, 2.0 * uniform(w[1]) - 1.0
        s = v1 * v1 + v2 * v2
        if not (s < 1.0) or s < 1e-300:
            continue
        x = v1 * math.sqrt(-2.0 * py_log(s) / s)
        v = 1.0 + c * x
        if v <= 0.0:
            continue
        v = v * v * v
        u, x2 = uniform(w[2]), x * x
        if not (u < 1.0 - 0.0331 * (x2 * x2)) and not (py_log(u) < 0.5 * x2 + d * (1.0 - v + py_log(v))):
            continue
        g = d * v
        if alpha < 1.0:
            g = g * py_exp(py_log(uniform(w[3])) / alpha)
        return np.float32(0.0) if g < 1e-18 else np.float32(g)
    return np.float32(0.0)


def test_philox_known_answers():
    """Random123's known-answer vectors for philox4x32-10 (the generator the draws are keyed through)."""
    assert philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert philox4x32_10((M32, M32, M32, M32), (M32, M32)) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)


def test_log_and_exp_follow_the_restatement_and_libm(oracle):
    L = oracle.lib()
    rng = np.random.RandomState(3)
    xs = np.concatenate([10.0 ** rng.uniform(-300, 2, 2000), rng.uniform(0.5, 2.0, 2000), [1.0, 2.0, 0.5, 1.4142135623730951, 1.4142135623730954, 2.2250738585072014e-308]])
    for x in xs:
        got = L.go_noise_log(float(x))
        assert got == py_log(float(x))                                    # the same operations in the same order: bit for bit
        assert abs(got - math.log(x)) <= 4e-16 * max(1.0, abs(math.log(x)))
    for x in np.concatenate([-(10.0 ** rng.uniform(-12, 2.84, 3000)), [0.0, -700.0, -700.0001, -1e-300]]):
        got = L.go_noise_exp(float(x))
        assert got == py_exp(float(x))
        assert abs(got - math.exp(x)) <= 4e-16 * math.exp(x) or (x < -700.0 and got == 0.0)      # below -700 the header returns 0 (stated there)


def test_gamma_draws_follow_the_restatement(oracle):
    """Known answers: the header's draws (C, through the oracle library) == the Python restatement, for the reference's alpha and others, over many
    counters; and a frozen handful, so that neither side can drift."""
    L = oracle.lib()
    rng = np.random.RandomState(11)
    for alpha in (0.05, 0.3, 1.0, 2.5):
        for _ in range(400):
            game, stones, cell, seed = int(rng.randint(0, 2 ** 31)), int(rng.randint(0, 225)), int(rng.randint(0, 225)), int(rng.randint(0, 2 ** 62))
            got = np.float32(L.go_noise_gamma(alpha, game, stones, cell, seed))
            assert got == py_gamma(alpha, game, stones, cell, seed), (alpha, game, stones, cell, seed)
    frozen = [np.float32(L.go_noise_gamma(0.05, 7, 12, cell, 99)).view(np.uint32) for cell in range(8)]
    assert [int(x) for x in frozen] == [int(py_gamma(0.05, 7, 12, cell, 99).view(np.uint32)) for cell in range(8)]
    assert [hex(int(x)) for x in frozen] == FROZEN_ALPHA_005


# the eight draws (game 7, 12 stones, cells 0..7, seed 99) as float32 bit patterns
FROZEN_ALPHA_005 = ['0x3e0a6466', '0x2ffb4b4e', '0x3b63b32a', '0x2e82057e', '0x39393641', '0x36a4e416', '0x2bfa8b5a', '0x35b324af']


@pytest.mark.parametrize("alpha", [0.05, 0.3, 2.5])
def test_gamma_moments(oracle, alpha):
    """gamma(alpha, 1): mean alpha, variance alpha, third central moment 2 alpha (200 000 draws; bounds = five standard errors)."""
    L = oracle.lib()
    n = 200000
    g = np.array([L.go_noise_gamma(alpha, 5, 40, c % 225, 1000 + c // 225) for c in range(n)], dtype=np.float64)
    assert abs(g.mean() - alpha) < 5 * math.sqrt(alpha / n)
    var_se = math.sqrt((6 * alpha + 2 * alpha * alpha) / n)              # Var[(x - mu)^2] = mu4 - mu2^2 = 3a^2 + 6a - a^2
    assert abs(g.var() - alpha) < 5 * var_se
    assert abs(((g - alpha) ** 3).mean() - 2 * alpha) < 0.15 * max(1.0, 2 * alpha)
    if alpha < 1:       # P(X <= x) ~ x^alpha / Gamma(alpha + 1) for small x: the boost's tail
        x = 1e-6
        expect = x ** alpha / math.gamma(alpha + 1)
        assert abs((g <= x).mean() - expect) < 5 * math.sqrt(expect * (1 - expect) / n) + 1e-3


def test_mix_is_add_noise(oracle):
    """gmk_noise_mix225 = Default::AddNoise (MonteCarlo.hpp:97-108): P <- (1 - eps) P + eps * normalized(draws on the entries with P != 0), the
    L2 norm in the stated order; entries without a child stay 0; the draws do not depend on which other entries have children."""
    L = oracle.lib()
    rng = np.random.RandomState(5)
    p = np.zeros(225, dtype=np.float32)
    kids = np.sort(rng.choice(225, 140, replace=False))
    p[kids] = np.float32(1.0) / np.float32(140)
    q = p.copy()
    L.go_noise_mix225(q.ctypes.data, 0.05, 0.25, 31, 17, 4242)
    assert (q[p == 0] == 0).all() and (q[kids] > 0).all()
    draws = np.array([L.go_noise_gamma(0.05, 31, 17, int(c), 4242) if p[c] else 0.0 for c in range(225)], dtype=np.float32)
    sq = draws * draws
    part = np.zeros(64, dtype=np.float32)
    for l in range(64):                                                   # the order of gmk_noise_sum225
        part[l] = sq[l]
        for j in (1, 2, 3):
            if l + 64 * j < 225:
                part[l] = np.float32(part[l] + sq[l + 64 * j])
    for off in (8, 4, 2, 1):
        nxt = part.copy()
        for l in range(64):
            nxt[l] = np.float32(part[l] + (part[l + off] if (l & 15) + off < 16 else np.float32(0)))
        part = nxt
    z = np.float32(np.float32(part[0] + part[16]) + np.float32(part[32] + part[48]))
    nrm = np.sqrt(z, dtype=np.float32)
    expect = (p * np.float32(1 - np.float32(0.25))).astype(np.float32) + np.float32(0.25) * (draws / nrm).astype(np.float32)
    assert (q == expect.astype(np.float32)).all()
    assert abs(float(np.sqrt((((q - p * np.float32(0.75)) / np.float32(0.25)).astype(np.float64) ** 2).sum())) - 1.0) < 1e-5      # a unit vector was mixed in
