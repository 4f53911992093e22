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


def test_step_keeps_the_subtree_and_noise(gmk, oracle):
    """gmk_trad_step + gmk_trad_add_root_noise (MCTS::stepForward / syncWithBoard, Default::AddNoise) for a batch of games:
    searching, stepping to the best child, stepping to an arbitrary reply, searching again equals the oracle's persistent
    MCTS object driven the same way.  Games that are over stop moving (a step on their childless root changes nothing)."""
    import ctypes as C
    G, O = gmk, oracle
    n = 6
    pos = _positions(G, n, 16, first=900)
    pos = [p if len(p) >= 2 else [112, 113] for p in pos]
    t = G.TraditionalMCTS(n, node_capacity=1 << 18)
    orcs = [O.TraditionalMCTS(5.0) for _ in range(n)]
    boards = []
    for g, o in enumerate(orcs):
        o.set_noise(0.05, 0.25, 4242, 50 + g)
        b = O.new_board()
        for mv in pos[g]:
            O.lib().go_board_apply(C.byref(b), mv, 1)
        boards.append(b)
    t.set_positions(pos)
    lists = [list(p) for p in pos]
    compared = kept_visits = 0
    for rnd in range(3):
        t.add_root_noise(0.05, 0.25, seed=4242, first_game_id=50)
        t.run(250)
        st = t.root_stats()
        live = [g for g in range(n) if boards[g].cur_player != 0]
        for g in live:
            orcs[g].run(lists[g], 250)
            v, q, p, best = orcs[g].root_children()
            where = "round %d game %d" % (rnd, g)
            np.testing.assert_array_equal(st["priors"][g].view(np.uint32), p.view(np.uint32), where)
            np.testing.assert_array_equal(st["visits"][g], v, where)
            np.testing.assert_array_equal(st["values"][g].view(np.uint32), q.view(np.uint32), where)
            assert st["best"][g] == best and st["root_visits"][g] == orcs[g].root_visits, where
            assert np.float32(st["root_value"][g]).view(np.uint32) == np.float32(orcs[g].root_value).view(np.uint32), where
            kept_visits = max(kept_visits, int(st["root_visits"][g]) - 250)
            compared += 1
        # everybody still playing steps to the most visited child, then the opponent answers somewhere: the first free
        # cell (rarely a child: a new node) in odd games, a free cell in the middle in even ones
        first = np.full(n, -1, np.int16)
        for g in live:
            first[g] = st["best"][g]
            assert orcs[g].step_forward() == st["best"][g]
            lists[g].append(int(first[g]))
            O.lib().go_board_apply(C.byref(boards[g]), int(first[g]), 1)
        t.step(first)
        replies = np.full(n, -1, np.int16)
        for g in live:
            if boards[g].cur_player == 0:
                continue
            free = [c for c in range(225) if c not in lists[g]]
            kid_visits = orcs[g].root_children()[0]              # the new root's children, when the kept subtree has some
            replies[g] = free[0] if g % 2 else (int(np.argmax(kid_visits)) if kid_visits.any() else free[len(free) // 2])
            lists[g].append(int(replies[g]))
            O.lib().go_board_apply(C.byref(boards[g]), int(replies[g]), 1)
        t.step(replies)
        assert (t.root_stats()["status"] == 0).all()
    assert compared >= 10 and kept_visits > 0            # some searches started from a kept subtree
    t.close()


def _dense_positions(n, lo, hi, seed):
    """prefixes (lo .. hi stones) of shuffled games between two colour classes that never line up five"""
    rng = np.random.RandomState(seed)
    cls = lambda c: ((c % 15) // 2 + c // 15) % 2
    blacks, whites = [c for c in range(225) if cls(c) == 0], [c for c in range(225) if cls(c) == 1]
    out = []
    for g in range(n):
        b, w = list(rng.permutation(blacks)), list(rng.permutation(whites))
        seq = []
        while b or w:
            if b:
                seq.append(int(b.pop()))
            if w:
                seq.append(int(w.pop()))
        out.append(seq[:int(rng.randint(lo, hi + 1))])
    return out


def test_nearly_full_boards(gmk, oracle):
    """Roots with 1 .. 12 empty cells: the search runs into full boards (Evaluator::checkGameEnd's tie, Pattern.cpp:344-354) and
    exhausts its tree; visits, values, chosen move and evaluator updates still equal the oracle's."""
    G, O = gmk, oracle
    pos = _dense_positions(10, 213, 224, 3)
    t = G.TraditionalMCTS(len(pos), node_capacity=1 << 16)
    t.set_positions(pos)
    t.run(300)
    stats = t.root_stats()
    for g in range(len(pos)):
        orc = O.TraditionalMCTS(5.0)
        orc.search(pos[g], 300)
        _compare(stats, g, orc)
    t.close()
