"""K6 parity: the device pattern-guided search (gmk_trad_*) against the oracle's TraditionalPolicy restatement
(oracle/go_trad.c; Traditional.h:17-69, Heuristic.hpp, MonteCarlo.hpp:149-184).  The search has no random numbers;
both sides use the same float summation order, so everything is compared exactly: visit counts, the BITS of values
and priors, the chosen move, tree size, the number of evaluator updates the cached apply / revert logic performed,
and the evaluator state the search leaves behind (flag words included)."""
import numpy as np
import pytest

from gomokuai_amd import lib as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gmk():
    G.init()
    return G


def _positions(G, n, max_len, first=0):
    moves, lens, _ = G.synth_boards(n, 1, first_board=first)      # clustered boards: compounds and threats
    out = []
    for g in range(n):
        k = int(min(lens[g], max_len, (g * 7) % (max_len + 1)))
        out.append([int(m) for m in moves[g, :k]])
    return out


def _compare(stats, g, orc):
    v, q, p, best = orc.root_children()
    assert stats["status"][g] == 0
    np.testing.assert_array_equal(stats["visits"][g], v)
    np.testing.assert_array_equal(stats["priors"][g].view(np.uint32), p.view(np.uint32))
    np.testing.assert_array_equal(stats["values"][g].view(np.uint32), q.view(np.uint32))
    assert stats["best"][g] == best
    assert stats["root_visits"][g] == orc.root_visits
    assert np.float32(stats["root_value"][g]).view(np.uint32) == np.float32(orc.root_value).view(np.uint32)
    assert stats["n_nodes"][g] == orc.n_nodes
    assert stats["evaluator_updates"][g] == orc.evaluator_updates


def _compare_evaluator(G, O, dev, g, orc):
    import ctypes as C
    L = O.lib()
    ev = L.go_trad_evaluator(orc.h)
    scores = np.zeros((4, 225), np.int32); density = np.zeros((2, 2, 225), np.int32)
    pd = np.zeros((226, 8), np.uint32); cd = np.zeros((226, 3), np.uint32)
    L.go_eval_get_scores(C.c_void_p(ev), scores.ctypes.data); L.go_eval_get_density(C.c_void_p(ev), density.ctypes.data)
    L.go_eval_get_pattern_dist(C.c_void_p(ev), pd.ctypes.data); L.go_eval_get_compound_dist(C.c_void_p(ev), cd.ctypes.data)
    np.testing.assert_array_equal(dev["scores"][g], scores)
    np.testing.assert_array_equal(dev["density"][g], density)
    np.testing.assert_array_equal(dev["pattern_dist"][g], pd)
    np.testing.assert_array_equal(dev["compound_dist"][g], cd)


def test_search_matches_oracle(gmk, oracle):
    G, O = gmk, oracle
    n, playouts = 24, 400
    pos = _positions(G, n, 40)
    t = G.TraditionalMCTS(n, node_capacity=1 << 17)
    t.set_positions(pos)
    t.run(playouts)
    stats = t.root_stats()
    dev = t.read_evaluators()
    for g in range(n):
        orc = O.TraditionalMCTS(5.0)
        orc.search(pos[g], playouts)
        _compare(stats, g, orc)
        _compare_evaluator(G, O, dev, g, orc)
    t.close()


def test_evaluators_persist_across_searches(gmk, oracle):
    """The policy's evaluator lives across searches and is synchronised (Pattern.cpp:356-368), not rebuilt: its
    history-dependent flag words carry over, on both sides."""
    G, O = gmk, oracle
    n = 8
    base = _positions(G, n, 30, first=100)
    moves, lens, _ = G.synth_boards(n, 1, first_board=100)
    t = G.TraditionalMCTS(n, node_capacity=1 << 17)
    orcs = [O.TraditionalMCTS(5.0) for _ in range(n)]
    rounds = [base,
              [[int(m) for m in moves[g, :min(int(lens[g]), len(base[g]) + 2)]] for g in range(n)],      # two plies further
              _positions(G, n, 20, first=300)]                                                             # somewhere else entirely
    for r, pos in enumerate(rounds):
        t.set_positions(pos)
        t.run(150 + 50 * r)
        stats = t.root_stats()
        dev = t.read_evaluators()
        for g in range(n):
            orcs[g].search(pos[g], 150 + 50 * r)
            _compare(stats, g, orcs[g])
            _compare_evaluator(G, O, dev, g, orcs[g])
    t.close()


def test_split_runs_equal_one_run(gmk):
    G = gmk
    n = 6
    pos = _positions(G, n, 24, first=40)
    a = G.TraditionalMCTS(n, node_capacity=1 << 16); a.set_positions(pos); a.run(300)
    b = G.TraditionalMCTS(n, node_capacity=1 << 16); b.set_positions(pos); b.run(100); b.run(200)
    sa, sb = a.root_stats(), b.root_stats()
    for k in ("visits", "best", "root_visits", "n_nodes", "evaluator_updates"):
        np.testing.assert_array_equal(sa[k], sb[k])
    np.testing.assert_array_equal(sa["values"].view(np.uint32), sb["values"].view(np.uint32))
    a.close(); b.close()


def test_decisive_positions(gmk):
    """An open four must be completed, a four of the opponent must be blocked (DecisiveFilter, Heuristic.hpp:94-161)."""
    G = gmk
    c = lambda y, x: y * 15 + x
    own_four = [c(7, 7), c(0, 0), c(7, 8), c(0, 2), c(7, 9), c(0, 4), c(7, 10), c(0, 6)]           # black to move, black has four
    rival_four = [c(0, 0), c(7, 7), c(0, 2), c(7, 8), c(0, 4), c(7, 9), c(14, 14), c(7, 10)]       # black to move, white has four
    t = G.TraditionalMCTS(2, node_capacity=1 << 12)
    t.set_positions([own_four, rival_four])
    t.run(200)
    s = t.root_stats()
    assert s["best"][0] in (c(7, 6), c(7, 11)) and s["best"][1] in (c(7, 6), c(7, 11))
    assert set(np.nonzero(s["visits"][0])[0]) <= {c(7, 6), c(7, 11)}
    assert set(np.nonzero(s["visits"][1])[0]) <= {c(7, 6), c(7, 11)}
    assert s["root_value"][0] == -1.0                      # the side that just moved (white) has lost
    t.close()


def test_node_capacity_is_reported(gmk):
    G = gmk
    t = G.TraditionalMCTS(1, node_capacity=256)
    t.set_positions([[112, 113, 127]])
    t.run(500)
    s = t.root_stats()
    assert s["status"][0] & 1 and s["n_nodes"][0] <= 256
    t.close()
