/*
 * go_stdsort.cpp -- TEST INFRASTRUCTURE (see gomoku_oracle.h).
 * AhoCorasickBuilder::sortPatterns (core/lib/src/utils/ACAutomata.cpp:78-80) orders pattern
 * indices with std::sort on a numeric key.  Four pairs of patterns have EQUAL keys, and the
 * order std::sort leaves equal elements in is implementation-defined; it changes the trie the
 * builder produces.  The oracle therefore calls the toolchain's own std::sort (libstdc++, what a
 * g++ build of the reference uses) with the reference's comparator instead of restating a sort.
 */
#include <algorithm>
#include <numeric>

extern "C" void go__std_sort_indices(const int *codes, int *indices, int n) {
    std::iota(indices, indices + n, 0);
    std::sort(indices, indices + n, [codes](int lhs, int rhs) { return codes[lhs] < codes[rhs]; });
}

/*
 * Default::AddNoise draws from std::gamma_distribution<float>(alpha, 1) over std::mt19937
 * (core/lib/include/algorithms/Statistical.hpp:24-34); the distribution's algorithm is implementation-defined,
 * so the oracle calls the toolchain's own template, as a g++ build of the reference does.
 */
#include <random>
extern "C" void go__gamma_draws(unsigned seed, float alpha, int n, float *out) {
    std::mt19937 engine(seed);
    std::gamma_distribution<float> gamma(alpha, 1.0f);
    for (int i = 0; i < n; ++i) out[i] = gamma(engine);
}
