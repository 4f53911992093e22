"""The PoolRAVE restatement (oracle/go_rave.c) on the CPU: the invariants the reference's algorithm implies
(PoolRAVE.h:7-52, MonteCarlo.hpp:113-184, MCTS.cpp:158-198).  The reference holds no test for this policy
(core/test has none), so these are properties, not golden values."""
import numpy as np


def test_counts_and_determinism(oracle):
    O = oracle
    moves = [112, 113, 97, 98]
    a = O.PoolRAVEMCTS(2.0, 0.0, seed=11, game_id=3)
    a.run(moves, 500)
    v, q, p, av, aq, best = a.root_children()
    assert a.root_visits == 500
    assert v.sum() == 499                                   # the first playout expands the root, every later one visits a child
    assert a.size > 1 + 221 + 218 * 400                     # nearly every playout expands a leaf with ~220 children
    assert best == int(np.argmax(v)) or v[best] == v.max()
    assert (p[np.array(moves)] == 0).all() and np.count_nonzero(p) == 221
    assert np.allclose(p[p > 0], np.float32(1.0) / np.float32(221))
    assert (av >= v).sum() > 200                            # AMAF sees every playout whose later moves include the cell
    assert (np.abs(q) <= 1).all() and (np.abs(aq) <= 1).all()
    b = O.PoolRAVEMCTS(2.0, 0.5, seed=11, game_id=3)        # c_bias is dead code in the reference (MonteCarlo.hpp:130-139)
    b.run(moves, 500)
    for x, y in zip(a.root_children(), b.root_children()):
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
    c = O.PoolRAVEMCTS(2.0, 0.0, seed=12, game_id=3)
    c.run(moves, 500)
    assert (c.root_children()[0] != v).any()


def test_kept_subtree_and_terminal_root(oracle):
    O = oracle
    moves = [112, 113]
    t = O.PoolRAVEMCTS(2.0, 0.0, seed=5)
    t.run(moves, 300)
    v = t.root_children()[0]
    mv = t.step_forward()
    assert v[mv] == v.max()
    t.run(moves + [mv], 100)
    assert t.root_visits == 100 + v[mv]                     # the kept child brings its visits along
    c = lambda y, x: y * 15 + x
    won = [c(7, 3), c(0, 0), c(7, 4), c(0, 2), c(7, 5), c(0, 4), c(7, 6), c(0, 6), c(7, 7)]
    w = O.PoolRAVEMCTS(2.0, 0.0, seed=5)
    w.run(won, 50)
    assert w.root_visits == 50 and w.root_value == 1.0 and w.root_children()[5] == -1 and w.size == 1
