"""The oracle's restatement of the pattern-guided search (oracle/go_trad.c: Traditional.h:17-69, Heuristic.hpp:16-45,
94-200, MonteCarlo.hpp:149-184).  The reference has no tests for this part and cannot be built here, so
the float part is "parity unpinned"; these tests hold the behaviour its code prescribes."""
import numpy as np

c = lambda y, x: y * 15 + x


def test_empty_board_prior_is_the_centre(oracle):
    probs, value = oracle.trad_heuristic([])
    assert list(np.nonzero(probs)[0]) == [c(7, 7)] and probs[c(7, 7)] == 1.0        # Heuristic.hpp:23-26
    t = oracle.TraditionalMCTS(5.0)
    t.search([], 50)
    v, q, p, best = t.root_children()
    assert best == c(7, 7) and v[c(7, 7)] == 49 and t.root_visits == 50


def test_decisive_filter_keeps_only_the_decisive_cells(oracle):
    own_four = [c(7, 7), c(0, 0), c(7, 8), c(0, 2), c(7, 9), c(0, 4), c(7, 10), c(0, 6)]
    probs, value = oracle.trad_heuristic(own_four)
    assert set(np.nonzero(probs)[0]) == {c(7, 6), c(7, 11)}
    assert abs(float(np.sqrt((probs.astype(np.float64) ** 2).sum())) - 1.0) < 1e-6   # L2-normalised (Eigen normalize)
    assert value > 0.99
    rival_four = [c(0, 0), c(7, 7), c(0, 2), c(7, 8), c(0, 4), c(7, 9), c(14, 14), c(7, 10)]
    probs, value = oracle.trad_heuristic(rival_four)
    assert set(np.nonzero(probs)[0]) == {c(7, 6), c(7, 11)}
    # a live three of the side to move outranks everything but fours: the cells that make it a four
    own_three = [c(7, 7), c(0, 0), c(7, 8), c(0, 2), c(7, 9), c(0, 5)]
    probs, _ = oracle.trad_heuristic(own_three)
    assert set(np.nonzero(probs)[0]) <= {c(7, 5), c(7, 6), c(7, 10), c(7, 11)} and probs[c(7, 6)] > 0 and probs[c(7, 10)] > 0


def test_search_plays_the_win_and_counts_add_up(oracle):
    own_four = [c(7, 7), c(0, 0), c(7, 8), c(0, 2), c(7, 9), c(0, 4), c(7, 10), c(0, 6)]
    t = oracle.TraditionalMCTS(5.0)
    t.search(own_four, 300)
    v, q, p, best = t.root_children()
    assert best in (c(7, 6), c(7, 11)) and int(v.sum()) == 299 and t.root_visits == 300
    assert t.root_value == -1.0                       # every line ends in black's five: the root's player (white) has lost
    assert t.n_nodes == 3                             # the winning children are terminal, never expanded


def test_search_is_deterministic_and_evaluator_persists(oracle):
    pos = [c(7, 7), c(7, 8), c(8, 8), c(6, 6), c(8, 7), c(8, 6)]
    a, b = oracle.TraditionalMCTS(5.0), oracle.TraditionalMCTS(5.0)
    a.search(pos, 500); b.search(pos, 500)
    va, qa, pa, ba = a.root_children(); vb, qb, pb, bb = b.root_children()
    assert (va == vb).all() and (qa.view(np.uint32) == qb.view(np.uint32)).all() and ba == bb
    assert int(va.sum()) == 499
    # a second search from a later position reuses the evaluator (sync, not rebuild): far fewer updates than a replay per playout
    before = a.evaluator_updates
    a.search(pos + [c(9, 6), c(5, 9)], 200)
    assert a.evaluator_updates - before < 200 * 8
