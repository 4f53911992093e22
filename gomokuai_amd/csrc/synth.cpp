// synth.cpp -- host-side synthetic workloads for the evaluator / MCTS benchmarks.
#include <cstring>

#include "../../include/gomoku_hip.h"
#include "philox.h"

namespace {

struct HostBoard {
    uint16_t plane[2][16];
    uint8_t occupied[225];
    int n_moves;
    void clear() { std::memset(this, 0, sizeof *this); }
    bool stone(int colour, int x, int y) const { return x >= 0 && x < 15 && y >= 0 && y < 15 && ((plane[colour][y] >> x) & 1); }
    // five or more in a row through (x,y) for `colour` (core/lib/src/Game.cpp:88-136)
    bool wins(int colour, int x, int y) const {
        static const int dirs[4][2] = {{1, 0}, {0, 1}, {1, -1}, {1, 1}};
        for (auto& d : dirs) {
            int run = 1;
            for (int s = -1; s <= 1; s += 2)
                for (int i = 1; i <= 5 && stone(colour, x + s * i * d[0], y + s * i * d[1]); ++i) ++run;
            if (run >= 5) return true;
        }
        return false;
    }
    // returns true when the move ended the game
    bool play(int id) {
        const int colour = n_moves & 1, x = id % 15, y = id / 15;      // black (plane 0) moves first
        plane[colour][y] = static_cast<uint16_t>(plane[colour][y] | (1u << x));
        occupied[id] = 1;
        ++n_moves;
        return wins(colour, x, y) || n_moves == 225;
    }
    // uniform draw then linear probe to the next empty cell with wrap (core/lib/src/Game.cpp:68-72)
    int probe(uint32_t r) const {
        int id = static_cast<int>(r % 225u);
        while (occupied[id]) id = (id + 1) % 225;
        return id;
    }
};

}  // namespace

extern "C" int gmk_synth_boards(uint64_t seed, uint32_t first_board, int n, int kind,
                                uint8_t* h_moves, int stride, int32_t* h_lens, uint16_t* h_planes) {
    if (n < 0 || stride < 64 || !h_moves || !h_lens || kind < 0 || kind > 1) return GMK_ERR_ARG;
    const uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
    HostBoard b;
    for (int i = 0; i < n; ++i) {
        b.clear();
        const uint32_t id = first_board + static_cast<uint32_t>(i);
        const uint32_t plies = 8u + gmk::philox4x32_10(id, 0, static_cast<uint32_t>(kind), 0, k0, k1).v[0] % 53u;
        uint8_t* mv = h_moves + static_cast<size_t>(i) * stride;
        std::memset(mv, 0, static_cast<size_t>(stride));
        for (uint32_t p = 0; p < plies; ++p) {
            const gmk::Philox4 r = gmk::philox4x32_10(id, p + 1, static_cast<uint32_t>(kind), 0, k0, k1);
            int cell = -1;
            if (kind == 1 && p > 0 && r.v[1] % 10u < 9u) {
                const int anchor = mv[r.v[2] % p];
                const int x = anchor % 15 + static_cast<int>(r.v[3] % 5u) - 2;
                const int y = anchor / 15 + static_cast<int>((r.v[3] / 5u) % 5u) - 2;
                if (x >= 0 && x < 15 && y >= 0 && y < 15 && !b.occupied[y * 15 + x]) cell = y * 15 + x;
            }
            if (cell < 0) cell = b.probe(r.v[0]);
            mv[b.n_moves] = static_cast<uint8_t>(cell);
            if (b.play(cell)) break;
        }
        h_lens[i] = b.n_moves;
        if (h_planes) std::memcpy(h_planes + static_cast<size_t>(i) * 32, b.plane, sizeof b.plane);
    }
    return GMK_OK;
}

extern "C" int gmk_moves_to_planes(const uint8_t* h_moves, int stride, const int32_t* h_lens, int n, uint16_t* h_planes) {
    if (n < 0 || !h_moves || !h_lens || !h_planes) return GMK_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        uint16_t* pl = h_planes + static_cast<size_t>(i) * 32;
        std::memset(pl, 0, 64);
        for (int m = 0; m < h_lens[i]; ++m) {
            const int id = h_moves[static_cast<size_t>(i) * stride + m];
            if (id >= 225) return GMK_ERR_ARG;
            pl[(m & 1) * 16 + id / 15] = static_cast<uint16_t>(pl[(m & 1) * 16 + id / 15] | (1u << (id % 15)));
        }
    }
    return GMK_OK;
}
