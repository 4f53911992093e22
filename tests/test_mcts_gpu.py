"""K3 parity on the GPU: gmk_mcts_* (persistent playout kernel) vs the CPU oracle's restatement of
MCTS + RandomPolicy under the same Philox stream.  Visit counts, tree size: exact.  Root Q (f32): bitwise."""
import ctypes as C

import numpy as np
import pytest

from gomokuai_amd import lib as G

pytestmark = pytest.mark.gpu

SEED = G.DEFAULT_SEED


def _oracle_search(O, moves, length, playouts, game_id, c_puct=5.0, c_rollouts=5, seed=SEED):
    b = O.new_board()
    L = O.lib()
    for i in range(length):
        L.go_board_apply(C.byref(b), int(moves[i]), 1)
    m = O.MCTS(playouts, c_puct, c_rollouts, seed, game_id)
    q, pi, visits = m.eval_state(b)
    assert b.nrec == length                     # mcts_unittest.cpp:26-35: the board is left untouched
    return visits, np.float32(q), m.root_visits, m.size, m.alg_bytes


def _openings(n, plies, first=0):
    """n random openings of exactly `plies` stones (the synthetic generator truncated)."""
    moves, lens, _ = G.synth_boards(n, 0, first_board=first)
    lens = np.minimum(lens, plies).astype(np.int32)
    planes = G.moves_to_planes(moves, lens)
    last = np.array([moves[i, lens[i] - 1] if lens[i] > 0 else -1 for i in range(n)], dtype=np.int16)
    return moves, lens, planes, last


@pytest.mark.parametrize("n_games,plies,playouts", [(24, 4, 200), (13, 0, 120), (5, 9, 300)])
def test_visits_match_oracle(oracle, n_games, plies, playouts):
    moves, lens, planes, last = _openings(n_games, plies, first=77)
    t = G.BatchedMCTS(n_games, playouts_capacity=playouts)
    t.set_roots(planes, last, first_game_id=1000)
    t.run(playouts)
    visits, q, rv, nodes, status = t.root_stats()
    total_bytes = 0
    for g in range(n_games):
        ov, oq, orv, osize, obytes = _oracle_search(oracle, moves[g], int(lens[g]), playouts, 1000 + g)
        assert (visits[g] == ov).all(), "game %d visits differ" % g
        assert rv[g] == orv == playouts
        assert nodes[g] == osize
        assert q[g].tobytes() == oq.tobytes(), "game %d root Q %r vs %r" % (g, q[g], oq)
        total_bytes += obytes
    assert not status.any()
    assert t.alg_bytes() == total_bytes
    t.close()


def test_two_handles_with_different_rollout_counts(oracle):
    """The rollout-sum -> value table belongs to the handle: a supervisor with c_rollouts = 5 and a candidate with 10 live side
    by side (play_match_games, two CorePyExt MCTS(RandomPolicy(c, r)) agents) and both must search like the oracle, in either
    order of creation and use."""
    moves, lens, planes, last = _openings(6, 4, first=31)
    a = G.BatchedMCTS(6, playouts_capacity=150, c_rollouts=5)
    b = G.BatchedMCTS(6, playouts_capacity=150, c_rollouts=10)          # created second: a one-table-per-process build would now serve a with b's table
    c = G.BatchedMCTS(6, playouts_capacity=150, c_rollouts=3)
    d = G.BatchedMCTS(6, playouts_capacity=150, c_rollouts=1)           # RandomPolicy(c, 1): the divisor 1 has no reciprocal constant
    for t, r in ((a, 5), (b, 10), (c, 3), (d, 1), (a, 5)):
        t.set_roots(planes, last, first_game_id=40)
        t.run(150)
        visits, q, _, nodes, status = t.root_stats()
        for g in range(6):
            ov, oq, _, osize, _ = _oracle_search(oracle, moves[g], int(lens[g]), 150, 40 + g, c_rollouts=r)
            assert (visits[g] == ov).all() and nodes[g] == osize and q[g].tobytes() == oq.tobytes(), "c_rollouts %d, game %d" % (r, g)
        assert not status.any()
    for t in (a, b, c, d):
        t.close()


def _no_five_positions(n, lo, hi, seed):
    """n move sequences of lo .. hi-1 stones in which neither colour ever lines up five (two colour classes that cannot)."""
    rng = np.random.RandomState(seed)
    cls = lambda c: ((c % 15) // 2 + c // 15) % 2
    blacks, whites = [c for c in range(225) if cls(c) == 0], [c for c in range(225) if cls(c) == 1]
    moves = np.zeros((n, 225), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    for g in range(n):
        b, w = list(rng.permutation(blacks)), list(rng.permutation(whites))
        seq = []
        while b or w:
            if b:
                seq.append(b.pop())
            if w:
                seq.append(w.pop())
        moves[g] = seq
        lens[g] = rng.randint(lo, hi)
    return moves, lens


@pytest.mark.parametrize("c_rollouts,form", [(5, "quads"), (16, "quads, all 64 lanes"), (20, "pairs"), (32, "pairs, all 64 lanes"), (40, "one lane")])
def test_every_lane_form_of_the_rollouts(oracle, c_rollouts, form):
    """A rollout runs on four lanes (one line direction each), on two, or on one, whichever the wavefront's games x rollouts leave room for
    (mcts_kernel.hip / rollout_device.h); with one game per wavefront the rollout count picks the form.  All of them must play the
    oracle's rollouts: visit counts, tree size and the bits of the root value, from openings and from late positions (boards that fill up
    inside a rollout block)."""
    for late, playouts in ((False, 120), (True, 60)):
        if late:
            moves, lens = _no_five_positions(5, 150, 215, seed=c_rollouts)
            planes = G.moves_to_planes(moves, lens)
            last = np.array([moves[g, lens[g] - 1] for g in range(5)], dtype=np.int16)
        else:
            moves, lens, planes, last = _openings(5, 4, first=11)
        t = G.BatchedMCTS(5, playouts_capacity=playouts, c_rollouts=c_rollouts)
        assert t.launch_info()["grid"] == 5                       # one game per wavefront
        t.set_roots(planes, last, first_game_id=300)
        t.run(playouts)
        visits, q, rv, nodes, status = t.root_stats()
        for g in range(5):
            ov, oq, _, osize, _ = _oracle_search(oracle, moves[g], int(lens[g]), playouts, 300 + g, c_rollouts=c_rollouts)
            assert (visits[g] == ov).all() and nodes[g] == osize and q[g].tobytes() == oq.tobytes(), "%s, %d stones, game %d" % (form, int(lens[g]), g)
        t.close()


def test_one_game_one_rollout(oracle):
    """The smallest search handle: one game, c_rollouts = 1 (one rollout lane in the workgroup: both fast-division constants are for 1)."""
    moves, lens, planes, last = _openings(1, 4, first=77)
    t = G.BatchedMCTS(1, playouts_capacity=200, c_rollouts=1)
    t.set_roots(planes, last, first_game_id=77)
    t.run(200)
    visits, q, _, nodes, status = t.root_stats()
    ov, oq, _, osize, _ = _oracle_search(oracle, moves[0], int(lens[0]), 200, 77, c_rollouts=1)
    assert (visits[0] == ov).all() and nodes[0] == osize and q[0].tobytes() == oq.tobytes() and not status.any()
    t.close()


def test_results_independent_of_batching(oracle):
    """Game g's result depends on its global id only: 30 games in one handle == the same games run as 7 + 23."""
    moves, lens, planes, last = _openings(30, 4, first=5)
    a = G.BatchedMCTS(30, playouts_capacity=100)
    a.set_roots(planes, last, first_game_id=0)
    a.run(100)
    va = a.root_stats()[0]
    b1 = G.BatchedMCTS(7, playouts_capacity=100)
    b1.set_roots(planes[:7], last[:7], first_game_id=0)
    b1.run(100)
    b2 = G.BatchedMCTS(23, playouts_capacity=100)
    b2.set_roots(planes[7:], last[7:], first_game_id=7)
    b2.run(100)
    assert (va[:7] == b1.root_stats()[0]).all() and (va[7:] == b2.root_stats()[0]).all()


def test_terminal_and_near_terminal_roots(oracle):
    pos = lambda x, y: y * 15 + x
    won = [pos(3, 3), pos(3, 4), pos(4, 4), pos(3, 5), pos(5, 5), pos(3, 6), pos(6, 6), pos(3, 7), pos(7, 7)]      # black has five
    four = won[:-1]                                                                                               # black to move, one move wins
    games = [won, four, four + [pos(0, 0)]]
    n = len(games)
    moves = np.zeros((n, 64), dtype=np.uint8)
    lens = np.array([len(g) for g in games], dtype=np.int32)
    for i, g in enumerate(games):
        moves[i, :len(g)] = g
    planes = G.moves_to_planes(moves, lens)
    last = np.array([g[-1] for g in games], dtype=np.int16)
    t = G.BatchedMCTS(n, playouts_capacity=150)
    t.set_roots(planes, last, first_game_id=9)
    t.run(150)
    visits, q, rv, nodes, status = t.root_stats()
    # finished game: every playout ends at the root with value +1 for the player who made the last move
    assert rv[0] == 150 and nodes[0] == 1 and q[0] == np.float32(1.0) and not visits[0].any()
    for g in (1, 2):
        ov, oq, orv, osize, _ = _oracle_search(oracle, moves[g], int(lens[g]), 150, 9 + g)
        assert (visits[g] == ov).all() and nodes[g] == osize and q[g].tobytes() == oq.tobytes()


def test_arena_capacity_flag():
    moves, lens, planes, last = _openings(3, 4)
    t = G.BatchedMCTS(3, node_capacity=1000)
    t.set_roots(planes, last)
    t.run(50)
    _, _, rv, nodes, status = t.root_stats()
    assert (status & 2).all() and (nodes <= 1000).all() and (rv == 50).all()


def test_pi_from_visits(oracle):
    """gmk_visits_to_pi (host) vs the oracle's restatement of MCTS::evalState's pi; float32 transcendental
    order is unpinned in the reference (Eigen), so the comparison carries a tolerance of 1e-6 absolute."""
    rng = np.random.RandomState(0)
    for stones in (0, 14, 15, 60):
        v = np.zeros(225, dtype=np.uint32)
        idx = rng.choice(225, size=120, replace=False)
        v[idx] = rng.randint(0, 60, size=120)
        a = G.visits_to_pi(v, stones)
        b = oracle.visits_to_pi(v, stones)
        assert np.abs(a - b).max() <= 1e-6 and abs(float(a.sum()) - 1.0) < 1e-4


def test_nearly_full_boards(oracle):
    """Roots with 1 .. 12 empty cells (shuffled tie games): rollouts of a few plies, full-board ties, trees that run out of leaves."""
    rng = np.random.RandomState(8)
    cls = lambda c: ((c % 15) // 2 + c // 15) % 2             # two colour classes that never line up five
    blacks, whites = [c for c in range(225) if cls(c) == 0], [c for c in range(225) if cls(c) == 1]
    n = 10
    moves = np.zeros((n, 225), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    for g in range(n):
        b, w = list(rng.permutation(blacks)), list(rng.permutation(whites))
        seq = []
        while b or w:
            if b:
                seq.append(b.pop())
            if w:
                seq.append(w.pop())
        moves[g] = seq
        lens[g] = rng.randint(213, 225)
    planes = G.moves_to_planes(moves, lens)
    last = np.array([moves[g, lens[g] - 1] for g in range(n)], dtype=np.int16)
    t = G.BatchedMCTS(n, playouts_capacity=300)
    t.set_roots(planes, last, first_game_id=70)
    t.run(300)
    visits, q, rv, nodes, status = t.root_stats()
    for g in range(n):
        ov, oq, orv, osize, _ = _oracle_search(oracle, moves[g], int(lens[g]), 300, 70 + g)
        assert (visits[g] == ov).all() and nodes[g] == osize and q[g].tobytes() == oq.tobytes() and rv[g] == orv, "game %d" % g
    t.close()


def test_reserved_arenas_serve_the_lock_step_calls_too():
    """gmk_mcts_reserve: a handle whose two arenas are the halves of one block (what the persistent loop with kept subtrees uses) runs the
    call-by-call API -- set_roots, run, advance with reuse_subtree, root_stats -- exactly as a handle that allocates as it goes."""
    import torch
    n, playouts = 6, 40
    moves, lens, _ = G.synth_boards(n, 0, first_board=77)
    lens = np.minimum(lens, 3).astype(np.int32)
    planes = G.moves_to_planes(moves, lens)
    last = np.array([moves[i, lens[i] - 1] if lens[i] > 0 else -1 for i in range(n)], dtype=np.int16)
    dev = torch.device("cuda", 0)
    out = []
    for reserve in (None, False, True):
        t = G.BatchedMCTS(n, node_capacity=3 * playouts * 225 + 1)
        if reserve is not None:
            t.reserve(two_arenas=reserve)
        t.set_roots(planes, last, first_game_id=5)
        d_moves = torch.zeros((n, 225), dtype=torch.uint8, device=dev); d_lens = torch.zeros(n, dtype=torch.int32, device=dev)
        d_winner = torch.zeros(n, dtype=torch.int8, device=dev); d_unfinished = torch.zeros(1, dtype=torch.int32, device=dev)
        d_visits = torch.zeros((n, 225, 225), dtype=torch.int16, device=dev)
        for _ in range(4):
            t.run(playouts)
            t.advance(d_moves.data_ptr(), d_visits.data_ptr(), d_lens.data_ptr(), d_winner.data_ptr(), d_unfinished.data_ptr(), reuse_subtree=True)
        t.run(playouts)
        visits, q, rv, nodes, status = t.root_stats()
        out.append((d_moves.cpu().numpy().copy(), d_visits.cpu().numpy().copy(), visits.copy(), nodes.copy()))
        t.close()
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert (a == b).all()
