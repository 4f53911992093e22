"""K8 parity: the device PoolRAVE search (gmk_trad_run_poolrave on the K6 tree) against the oracle's restatement of
PoolRAVEPolicy (oracle/go_rave.c; PoolRAVE.h:7-52, MonteCarlo.hpp:113-184).  Both sides draw the rollouts from the same
Philox counters and evaluate PUCB / the HandSelect weighting in double and the running means in float, so everything is
compared exactly: visit counts, the BITS of values, priors and AMAF values, AMAF visit counts, the chosen move and the
tree size."""
import numpy as np
import pytest

from gomokuai_amd import lib as G

pytestmark = pytest.mark.gpu

SEED = 0x1234ABCD5678EF01


@pytest.fixture(scope="module")
def gmk():
    G.init()
    return G


def _positions(G, n, max_len, first=0):
    moves, lens, _ = G.synth_boards(n, 0, first_board=first)      # random openings
    out = []
    for g in range(n):
        k = int(min(lens[g], max_len, (g * 5) % (max_len + 1)))
        out.append([int(m) for m in moves[g, :k]])
    return out


def _compare(stats, g, orc, where=""):
    v, q, p, av, aq, best = orc.root_children()
    assert stats["status"][g] == 0, where
    np.testing.assert_array_equal(stats["visits"][g], v, where)
    np.testing.assert_array_equal(stats["priors"][g].view(np.uint32), p.view(np.uint32), where)
    np.testing.assert_array_equal(stats["values"][g].view(np.uint32), q.view(np.uint32), where)
    np.testing.assert_array_equal(stats["amaf_visits"][g], av, where)
    np.testing.assert_array_equal(stats["amaf_values"][g].view(np.uint32), aq.view(np.uint32), where)
    assert stats["best"][g] == best, where
    assert stats["root_visits"][g] == orc.root_visits, where
    assert np.float32(stats["root_value"][g]).view(np.uint32) == np.float32(orc.root_value).view(np.uint32), where


def test_search_matches_oracle(gmk, oracle):
    G, O = gmk, oracle
    n, playouts = 20, 300
    pos = _positions(G, n, 40)
    t = G.PoolRAVEMCTS(n, node_capacity=1 << 17, c_puct=2.0, seed=SEED, first_game_id=7)
    t.set_positions(pos)
    t.run(playouts)
    stats = t.root_stats()
    for g in range(n):
        orc = O.PoolRAVEMCTS(2.0, 0.0, seed=SEED, game_id=7 + g)
        orc.run(pos[g], playouts)
        _compare(stats, g, orc, "game %d" % g)
        assert stats["n_nodes"][g] == orc.size
    assert stats["amaf_visits"].max() > 50 and stats["visits"].max() > 10
    t.close()


def test_other_constants_and_late_positions(gmk, oracle):
    """c_puct = 1e-4 (the class default, PoolRAVE.h:13) and positions late in the game, where rollouts are short and
    terminal leaves appear inside the tree."""
    G, O = gmk, oracle
    n, playouts = 8, 400
    moves, lens, _ = G.synth_boards(n, 0, first_board=500)
    pos = [[int(m) for m in moves[g, :max(0, int(lens[g]) - 1 - g % 3)]] for g in range(n)]
    t = G.PoolRAVEMCTS(n, node_capacity=1 << 17, c_puct=1e-4, seed=SEED)
    t.set_positions(pos)
    t.run(playouts)
    stats = t.root_stats()
    for g in range(n):
        orc = O.PoolRAVEMCTS(1e-4, 0.1, seed=SEED, game_id=g)
        orc.run(pos[g], playouts)
        _compare(stats, g, orc, "game %d" % g)
    t.close()


def test_split_runs_equal_one_run_and_seeds_matter(gmk):
    G = gmk
    n = 6
    pos = _positions(G, n, 24, first=40)
    a = G.PoolRAVEMCTS(n, node_capacity=1 << 17, seed=SEED); a.set_positions(pos); a.run(300)
    b = G.PoolRAVEMCTS(n, node_capacity=1 << 17, seed=SEED); b.set_positions(pos); b.run(100); b.run(200)
    c = G.PoolRAVEMCTS(n, node_capacity=1 << 17, seed=SEED + 1); c.set_positions(pos); c.run(300)
    sa, sb, sc = a.root_stats(), b.root_stats(), c.root_stats()
    for k in ("visits", "best", "root_visits", "n_nodes", "amaf_visits"):
        np.testing.assert_array_equal(sa[k], sb[k])
    np.testing.assert_array_equal(sa["values"].view(np.uint32), sb["values"].view(np.uint32))
    np.testing.assert_array_equal(sa["amaf_values"].view(np.uint32), sb["amaf_values"].view(np.uint32))
    assert (sa["visits"] != sc["visits"]).any()
    a.close(); b.close(); c.close()


def test_node_capacity_is_reported(gmk):
    G = gmk
    t = G.PoolRAVEMCTS(1, node_capacity=1024)
    t.set_positions([[112, 113, 127]])
    t.run(50)
    s = t.root_stats()
    assert s["status"][0] & 1 and s["n_nodes"][0] <= 1024
    t.close()


def test_step_keeps_the_subtree_and_noise(gmk, oracle):
    """The agent loop for a batch of games: AddNoise, search, step to the most visited child, an arbitrary reply, search again
    from the kept subtree -- equal to the oracle's persistent MCTS object driven the same way."""
    import ctypes as C
    G, O = gmk, oracle
    n, playouts = 6, 200
    pos = _positions(G, n, 16, first=900)
    pos = [p if len(p) >= 2 else [112, 113] for p in pos]
    t = G.PoolRAVEMCTS(n, node_capacity=1 << 17, c_puct=2.0, seed=SEED, first_game_id=50)
    orcs = [O.PoolRAVEMCTS(2.0, 0.0, seed=SEED, game_id=50 + g) for g in range(n)]
    boards = []
    for g, o in enumerate(orcs):
        o.set_noise(0.05, 0.25)
        b = O.new_board()
        for mv in pos[g]:
            O.lib().go_board_apply(C.byref(b), mv, 1)
        boards.append(b)
    t.set_positions(pos)
    lists = [list(p) for p in pos]
    compared = kept_visits = 0
    for rnd in range(3):
        t.add_root_noise(0.05, 0.25)
        t.run(playouts)
        st = t.root_stats()
        live = [g for g in range(n) if boards[g].cur_player != 0]
        for g in live:
            orcs[g].run(lists[g], playouts)
            _compare(st, g, orcs[g], "round %d game %d" % (rnd, g))
            kept_visits = max(kept_visits, int(st["root_visits"][g]) - playouts)
            compared += 1
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
            kid_visits = orcs[g].root_children()[0]
            replies[g] = free[0] if g % 2 else (int(np.argmax(kid_visits)) if kid_visits.any() else free[len(free) // 2])
            lists[g].append(int(replies[g]))
            O.lib().go_board_apply(C.byref(boards[g]), int(replies[g]), 1)
        t.step(replies)
        assert (t.root_stats()["status"] == 0).all()
    assert compared >= 10 and kept_visits > 0
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
    """Roots with 1 .. 12 empty cells: rollouts of a few plies, ties at full boards, trees that run out of leaves."""
    G, O = gmk, oracle
    pos = _dense_positions(10, 213, 224, 4)
    t = G.PoolRAVEMCTS(len(pos), node_capacity=1 << 16, c_puct=2.0, seed=SEED, first_game_id=90)
    t.set_positions(pos)
    t.run(300)
    stats = t.root_stats()
    for g in range(len(pos)):
        orc = O.PoolRAVEMCTS(2.0, 0.0, seed=SEED, game_id=90 + g)
        orc.run(pos[g], 300)
        _compare(stats, g, orc, "game %d" % g)
    t.close()
