"""K1 parity on the GPU: gmk_eval_batch (HIP, through the C-ABI) vs the CPU oracle's in-order replay
of the same move lists.  Integer outputs: bit-exact."""
import ctypes as C

import numpy as np
import pytest

from gomokuai_amd import lib as G

pytestmark = pytest.mark.gpu

NAMES = ("scores", "density", "totals", "status")


def _compare(ref, got):
    for name, a, b in zip(NAMES, ref, got):
        bad = np.nonzero((a.reshape(len(a), -1) != b.reshape(len(b), -1)).any(axis=1))[0]
        assert len(bad) == 0, "%s differs on %d boards, first %d" % (name, len(bad), bad[0])


@pytest.mark.parametrize("kind,n", [(0, 4096), (1, 4096), (0, 5), (1, 1), (0, 257)])
def test_eval_matches_oracle_replay(oracle, kind, n):
    moves, lens, planes = G.synth_boards(n, kind, first_board=1000 * kind)
    ref = oracle.replay_batch(moves, lens)
    assert not (ref[3] & 2).any()
    _compare(ref, G.eval_batch_host(planes))


def _planes_from(moves_list):
    n = len(moves_list)
    moves = np.zeros((n, 256), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    for i, m in enumerate(moves_list):
        moves[i, :len(m)] = m
        lens[i] = len(m)
    return moves, lens, G.moves_to_planes(moves, lens)


def test_edge_positions(oracle):
    pos = lambda x, y: y * 15 + x
    empty = []
    one_corner = [pos(0, 0)]
    black_five = [pos(3, 3), pos(3, 4), pos(4, 4), pos(3, 5), pos(5, 5), pos(3, 6), pos(6, 6), pos(3, 7), pos(7, 7)]   # board_integrationtest.cpp:70
    white_five = [pos(3, 3), pos(3, 4), pos(4, 4), pos(3, 5), pos(5, 5), pos(3, 6), pos(6, 6), pos(3, 7), pos(8, 8), pos(3, 8)]   # :83
    # black completes SIX in a row (overline wins, Game.cpp:88-136): stones at x=0..2,4..5 then x=3
    six = [pos(0, 0), pos(0, 14), pos(1, 0), pos(1, 14), pos(2, 0), pos(2, 14), pos(4, 0), pos(4, 14), pos(5, 0), pos(7, 14), pos(3, 0)]
    kifu = [pos(7, 7), pos(8, 7), pos(7, 6), pos(7, 8), pos(6, 9)]                                                     # boardmap_unittest.cpp:36
    edge_hug = [pos(0, 0), pos(14, 14), pos(1, 0), pos(13, 14), pos(2, 0), pos(12, 14), pos(0, 1), pos(14, 13), pos(0, 2), pos(14, 12)]
    cases = [empty, one_corner, black_five, white_five, six, kifu, edge_hug]
    moves, lens, planes = _planes_from(cases)
    ref = oracle.replay_batch(moves, lens)
    got = G.eval_batch_host(planes)
    _compare(ref, got)
    st = got[3]
    assert st[0] == (1 << 16)                                  # empty board: not over, black to move
    assert st[2] & 1 and ((st[2] >> 8) & 0xFF) == 1            # black won
    assert st[3] & 1 and ((st[3] >> 8) & 0xFF) == 0xFF         # white won
    assert st[4] & 1 and ((st[4] >> 8) & 0xFF) == 1            # overline counts


def test_full_board_tie(oracle):
    """board_integrationtest.cpp:98-123: the row-interleaved fill ends in a tie on move 225."""
    order = []
    for j in range(15):
        y = 2 * j if j <= 7 else 2 * (j - 7) - 1
        order += [y * 15 + i for i in range(15)]
    moves, lens, planes = _planes_from([order, order[:224], order[:120]])
    ref = oracle.replay_batch(moves, lens)
    got = G.eval_batch_host(planes)
    _compare(ref, got)
    assert got[3][0] & 1 and ((got[3][0] >> 8) & 0xFF) == 0 and not got[0][0].any()


def test_long_games(oracle):
    """Boards far denser than the benchmark distribution (up to 200 plies without a winner)."""
    rng = np.random.RandomState(11)
    games = []
    L = oracle.lib()
    while len(games) < 64:
        b = oracle.new_board()
        seq = []
        target = rng.randint(80, 200)
        while b.cur_player != 0 and len(seq) < target:
            mv = L.go_board_random_move(C.byref(b), int(rng.randint(0, 225)))
            L.go_board_apply(C.byref(b), mv, 1)
            seq.append(mv)
        games.append(seq)
    moves, lens, planes = _planes_from(games)
    ref = oracle.replay_batch(moves, lens)
    keep = np.nonzero((ref[3] & 2) == 0)[0]
    got = G.eval_batch_host(planes)
    _compare([r[keep] for r in ref], [g[keep] for g in got])


def test_properties_at_full_size(oracle):
    """BASELINE config 2 size (65 536 boards): size-independent properties on all boards + oracle parity
    on a strided sample.  Reference invariant (Pattern.cpp:314-333): occupied cell => score 0, empty => >= 0."""
    n = 65536
    moves, lens, planes = G.synth_boards(n, 1, first_board=500000)
    scores, density, totals, status = G.eval_batch_host(planes)
    occ = np.zeros((n, 225), dtype=bool)
    for c in range(2):
        for y in range(15):
            row = planes[:, c, y].astype(np.int64)
            for x in range(15):
                occ[:, y * 15 + x] |= ((row >> x) & 1).astype(bool)
    assert (scores >= 0).all()
    assert not scores[np.broadcast_to(occ[:, None, :], scores.shape)].any()
    assert ((density < 0) == np.broadcast_to(occ[:, None, None, :], density.shape)).all()
    assert not (status & 2).any()
    # evaluation is a function of the position: the same planes in another batch slot give the same answer
    perm = np.random.RandomState(5).permutation(n)[:4096]
    again = G.eval_batch_host(planes[perm])
    assert (again[0] == scores[perm]).all() and (again[1] == density[perm]).all() and (again[2] == totals[perm]).all()
    sample = np.arange(0, n, 64)
    ref = oracle.replay_batch(moves[sample], lens[sample])
    _compare(ref, (scores[sample], density[sample], totals[sample], status[sample]))


def test_dense_boards_match_oracle(oracle):
    """100 .. 225 stones without a five (prefixes of shuffled tie games): the sizes the synthetic 8..60-ply boards never reach --
    full queues, saturated counters, the full-board tie."""
    rng = np.random.RandomState(11)
    cls = lambda c: ((c % 15) // 2 + c // 15) % 2             # two colour classes that never line up five
    blacks, whites = [c for c in range(225) if cls(c) == 0], [c for c in range(225) if cls(c) == 1]
    n = 384
    moves = np.zeros((n, 225), np.uint8)
    lens = np.zeros(n, np.int32)
    for g in range(n):
        b, w = list(rng.permutation(blacks)), list(rng.permutation(whites))
        seq = []
        while b or w:
            if b:
                seq.append(b.pop())
            if w:
                seq.append(w.pop())
        moves[g] = seq
        lens[g] = 225 if g < 8 else rng.randint(100, 226)
    got = G.eval_batch_host(G.moves_to_planes(moves, lens))
    ref = oracle.replay_batch(moves, lens)
    for name, a, b in zip(("scores", "density", "totals", "status"), ref, got):
        np.testing.assert_array_equal(a, b, name)
    assert (got[3][:8] & 1).all() and not (got[3] & 2).any()   # the full boards are over (ties), nothing overflowed
