"""BASELINE.json's configurations at FULL size on the GPU, inside the suite the driver runs.

  configs[1]  65 536 boards through K1: every board of both synthetic distributions against the oracle (scores, density,
              totals, status: bit-exact), i.e. what bench.py times, checked in full;
  configs[2]  4 096 games x 800 playouts through K3 (one launch, 1 024 workgroups): properties on every game, oracle parity
              (visit counts, root Q bits, tree size) on a strided sample of 32 games;
  plus the larger differential runs that used to live under tools/ (K3 at 96 x 800, K6 at 14 x 6 000 playouts, K8 at 64 x 3 000
  with a kept-subtree step, K1 on 8 192 boards of 100 .. 225 stones)."""
import ctypes as C

import numpy as np
import pytest

from gomokuai_amd import lib as G

pytestmark = pytest.mark.gpu


def _mismatching_boards(ref, got):
    n = len(ref[0])
    return {name: int((a.reshape(n, -1) != b.reshape(n, -1)).any(1).sum()) for name, a, b in zip(("scores", "density", "totals", "status"), ref, got)}


@pytest.mark.parametrize("kind", [0, 1])
def test_config2_all_65536_boards_match_the_oracle(oracle, kind):
    """BASELINE configs[1]: the bench's own board set (kind 0, first_board 0 = rank 0's shard) and the clustered set."""
    n = 65536
    moves, lens, planes = G.synth_boards(n, kind, first_board=0)
    got = G.eval_batch_host(planes)
    ref = oracle.replay_batch(moves, lens)
    assert not (ref[3] & 2).any() and not (got[3] & 2).any()
    assert _mismatching_boards(ref, got) == {"scores": 0, "density": 0, "totals": 0, "status": 0}
    if kind == 1:
        assert ((got[2][:, 8:] != 0).any(1)).mean() > 0.4          # the clustered set is the one with compounds on about half its boards


def test_config2_ragged_batch_sizes(oracle):
    """Batch sizes around the kernel's group of sixteen boards and its grid: the last group partial, fewer groups than CUs, one board."""
    for n in (1, 15, 16, 17, 255, 4097, 12289):
        moves, lens, planes = G.synth_boards(n, n % 2, first_board=90000 + n)
        assert _mismatching_boards(oracle.replay_batch(moves, lens), G.eval_batch_host(planes)) == {"scores": 0, "density": 0, "totals": 0, "status": 0}, n


def test_config3_4096_games_800_playouts(oracle):
    """BASELINE configs[2] as bench.py runs it: 4 096 games from 4-ply random openings, RandomPolicy(c_puct 5, c_rollouts 5), fresh
    roots, 800 playouts in one launch."""
    n, P = 4096, 800
    moves, lens, _ = G.synth_boards(n, 0, first_board=0)
    lens = np.minimum(lens, 4).astype(np.int32)
    planes = G.moves_to_planes(moves, lens)
    last = np.array([moves[i, lens[i] - 1] for i in range(n)], dtype=np.int16)
    tree = G.BatchedMCTS(n, playouts_capacity=P)
    tree.set_roots(planes, last, first_game_id=0)
    tree.run(P)
    visits, q, rv, nodes, status = tree.root_stats()
    # every game: all playouts ran, none hit the arena's end, the children's visits add up (the first playout expands the root)
    assert (rv == P).all() and not status.any()
    assert (visits.sum(1) == P - 1).all()
    assert (nodes > P).all() and (nodes <= P * 225 + 1).all()
    assert (np.abs(q) <= 1.0).all()
    # a strided sample against the oracle: visit counts, root Q bits, tree size
    alg = 0
    for g in range(0, n, n // 32):
        b = oracle.new_board()
        for i in range(int(lens[g])):
            oracle.lib().go_board_apply(C.byref(b), int(moves[g, i]), 1)
        om = oracle.MCTS(P, 5.0, 5, G.DEFAULT_SEED, g)
        om.run_playouts(b)
        ov, _, _ = om.root_children()
        assert (ov == visits[g]).all(), "game %d" % g
        assert np.float32(q[g]).tobytes() == np.float32(om.root_value).tobytes() and nodes[g] == om.size
        alg += om.alg_bytes
    assert tree.alg_bytes() > alg                            # (the kernel's byte count covers all 4 096 games)
    tree.close()


@pytest.mark.parametrize("n,per_wave", [(5000, 3), (7000, 4)])
def test_k3_three_and_four_games_per_wavefront(oracle, n, per_wave):
    """Batches that put three (rollouts on quads, select on quarter-waves) and four (rollouts on lane pairs) games into a wavefront --
    gmk_mcts_create picks that from the number of games -- against the oracle on a strided sample, 60 playouts each; the last wavefront
    is partly filled in both."""
    P = 60
    moves, lens, _ = G.synth_boards(n, 0, first_board=400000)
    lens = np.minimum(lens, 6).astype(np.int32)
    planes = G.moves_to_planes(moves, lens)
    last = np.array([moves[i, lens[i] - 1] for i in range(n)], dtype=np.int16)
    tree = G.BatchedMCTS(n, playouts_capacity=P)
    assert tree.launch_info()["grid"] == -(-n // per_wave)
    tree.set_roots(planes, last, first_game_id=123)
    tree.run(P)
    visits, q, rv, nodes, status = tree.root_stats()
    assert (rv == P).all() and not status.any() and (visits.sum(1) == P - 1).all()
    for g in list(range(0, n, n // 24)) + [n - 2, n - 1]:
        b = oracle.new_board()
        for i in range(int(lens[g])):
            oracle.lib().go_board_apply(C.byref(b), int(moves[g, i]), 1)
        om = oracle.MCTS(P, 5.0, 5, G.DEFAULT_SEED, 123 + g)
        om.run_playouts(b)
        ov, _, _ = om.root_children()
        assert (ov == visits[g]).all() and np.float32(q[g]).tobytes() == np.float32(om.root_value).tobytes() and nodes[g] == om.size, "game %d" % g
    tree.close()


def test_config4_32768_games_through_the_self_play_loop(oracle):
    """BASELINE configs[3] at its full game count, in the suite (only bench.py ran this size): 32 768 games through
    selfplay.play_games (device-resident continuous batching, two search handles), at a small playout budget so that it takes seconds.
    Size-independent properties on EVERY game: the record replays move by move on the oracle's board (Board::applyMove with its victory
    check) -- every move legal, the game over exactly at its recorded length, the recorded winner -- and every searched ply's root visit
    counts add up to playouts - 1 (the first playout of a fresh tree expands the root, every other one passes through a child)."""
    import torch
    from gomokuai_amd import selfplay
    n, P = 32768, 24
    rec = selfplay.play_games(n, P, first_game_id=0)
    assert not rec.overflow and len(rec) == n
    lens = rec.lens.cpu().numpy()
    legal, end_ply, winner = oracle.replay_games(rec.moves.cpu().numpy(), lens)
    assert legal.all() and (end_ply == lens).all() and (winner == rec.winner.cpu().numpy()).all()
    assert lens.min() >= 9 and lens.max() <= 225
    sums = rec.visits.view(n, 225, 225).to(torch.int32).sum(2)                      # [game][ply]
    searched = torch.arange(225, device=sums.device)[None, :] < rec.lens[:, None]
    assert bool((sums[searched] == P - 1).all()) and bool((sums[~searched] == 0).all())
    # the move played is the first most-visited child of its search (MCTS::stepForward: std::max_element)
    best = rec.visits.view(n, 225, 225).to(torch.int32).argmax(2)
    assert bool((best[searched] == rec.moves.to(torch.int64)[searched]).all())
    # and the games do not depend on how they were batched: a strided sample played alone gives the same records
    for g in (0, 12345, 32767):
        one = selfplay.play_games(1, P, first_game_id=g, slots=None).cpu()
        assert int(one.lens[0]) == int(lens[g]) and (one.moves[0, :lens[g]] == rec.moves[g, :lens[g]].cpu()).all()


def test_k3_96_games_800_playouts(oracle):
    n, P = 96, 800
    moves, lens, _ = G.synth_boards(n, 0, first_board=31337)
    lens = np.minimum(lens, 6).astype(np.int32)
    planes = G.moves_to_planes(moves, lens)
    last = np.array([moves[i, lens[i] - 1] for i in range(n)], dtype=np.int16)
    tree = G.BatchedMCTS(n, playouts_capacity=P)
    tree.set_roots(planes, last, first_game_id=1000)
    tree.run(P)
    visits, q, rv, nodes, status = tree.root_stats()
    for g in range(n):
        b = oracle.new_board()
        for i in range(int(lens[g])):
            oracle.lib().go_board_apply(C.byref(b), int(moves[g, i]), 1)
        om = oracle.MCTS(P, 5.0, 5, G.DEFAULT_SEED, 1000 + g)
        om.run_playouts(b)
        ov, _, _ = om.root_children()
        assert (ov == visits[g]).all() and np.float32(q[g]).tobytes() == np.float32(om.root_value).tobytes() and nodes[g] == om.size, "game %d" % g
    tree.close()


def test_k6_14_games_6000_playouts(oracle):
    n, P = 14, 6000
    moves, lens, _ = G.synth_boards(n, 1, first_board=555)
    pos = [[int(m) for m in moves[g, :min(int(lens[g]), 3 + 2 * g)]] for g in range(n)]
    t = G.TraditionalMCTS(n, node_capacity=1 << 20)
    t.set_positions(pos)
    t.run(P)
    st = t.root_stats()
    for g in range(n):
        o = oracle.TraditionalMCTS(5.0)
        o.search(pos[g], P)
        v, qq, p, best = o.root_children()
        assert (v == st["visits"][g]).all() and (qq.view(np.uint32) == st["values"][g].view(np.uint32)).all(), "game %d" % g
        assert best == st["best"][g] and o.n_nodes == st["n_nodes"][g] and o.evaluator_updates == st["evaluator_updates"][g], "game %d" % g
    assert not (st["status"] & ~np.int32(0)).any()
    t.close()


def test_k8_64_games_3000_playouts_and_a_kept_subtree(oracle):
    n, P = 64, 3000
    moves, lens, _ = G.synth_boards(n, 0, first_board=777)
    pos = [[int(m) for m in moves[g, :min(int(lens[g]), g % 30)]] for g in range(n)]
    t = G.PoolRAVEMCTS(n, node_capacity=(P + 1200) * 225, c_puct=2.0, first_game_id=300)
    t.set_positions(pos)
    t.run(P)
    orcs = [oracle.PoolRAVEMCTS(2.0, 0.0, seed=G.DEFAULT_SEED, game_id=300 + g) for g in range(n)]

    def check(st):
        for g in range(n):
            v, qq, p, av, aq, best = orcs[g].root_children()
            assert (v == st["visits"][g]).all() and (qq.view(np.uint32) == st["values"][g].view(np.uint32)).all() and best == st["best"][g], "game %d" % g
            assert (av == st["amaf_visits"][g]).all() and (aq.view(np.uint32) == st["amaf_values"][g].view(np.uint32)).all(), "game %d" % g

    for g in range(n):
        orcs[g].run(pos[g], P)
    st = t.root_stats()
    check(st)
    t.step()
    for g in range(n):
        if st["best"][g] >= 0:
            pos[g] = pos[g] + [orcs[g].step_forward()]
    t.run(1000)
    for g in range(n):
        orcs[g].run(pos[g], 1000)
    check(t.root_stats())
    t.close()


def test_k1_8192_dense_boards(oracle):
    """Boards of 100 .. 225 stones without a five (prefixes of shuffled tie games), far denser than the synthetic 8 .. 60-ply sets:
    queue capacities, saturated counters, many compounds, full boards."""
    rng = np.random.RandomState(5)
    cls = lambda c: ((c % 15) // 2 + c // 15) % 2             # two colour classes that never line up five (pairs of columns, shifted per row)
    blacks = [c for c in range(225) if cls(c) == 0]
    whites = [c for c in range(225) if cls(c) == 1]
    if len(blacks) < len(whites):
        blacks, whites = whites, blacks
    n = 8192
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
        lens[g] = rng.randint(100, 226)
    got = G.eval_batch_host(G.moves_to_planes(moves, lens))
    ref = oracle.replay_batch(moves, lens)
    assert not (got[3] & 2).any() and not (ref[3] & 2).any()
    assert _mismatching_boards(ref, got) == {"scores": 0, "density": 0, "totals": 0, "status": 0}


# ---- extended sweeps: more of the same differential runs on other boards, games and seeds.  The FIRST case of every sweep is part of the default
#      suite (one more set of 65 536 boards, one more K3 / K6 / K8 batch); the others (minutes of oracle time) run with GMK_EXTENDED=1, by hand on
#      the GPU box after a kernel change ----
import os

extended = pytest.mark.skipif(not os.environ.get("GMK_EXTENDED"), reason="extended sweep: set GMK_EXTENDED=1")


def _first_by_default(cases):
    return [cases[0]] + [pytest.param(*c, marks=extended) for c in cases[1:]]


@pytest.mark.parametrize("kind,first", _first_by_default([(0, 65536), (1, 65536), (0, 1 << 20), (1, 1 << 20), (0, 7777777), (1, 7777777)]))
def test_extended_k1_other_boards(oracle, kind, first):
    n = 65536
    moves, lens, planes = G.synth_boards(n, kind, first_board=first)
    assert _mismatching_boards(oracle.replay_batch(moves, lens), G.eval_batch_host(planes)) == {"scores": 0, "density": 0, "totals": 0, "status": 0}


@pytest.mark.parametrize("seed,first,plies,playouts", _first_by_default([(1, 100000, 0, 800), (2, 5000000, 9, 500), (0xDEADBEEF, 31, 30, 300), (77, 1 << 30, 60, 200)]))
def test_extended_k3_other_games_and_seeds(oracle, seed, first, plies, playouts):
    n = 128
    moves, lens, _ = G.synth_boards(n, 0, first_board=first)
    lens = np.minimum(lens, plies).astype(np.int32)
    planes = G.moves_to_planes(moves, lens)
    last = np.array([moves[i, lens[i] - 1] if lens[i] > 0 else -1 for i in range(n)], dtype=np.int16)
    tree = G.BatchedMCTS(n, playouts_capacity=playouts, seed=seed)
    tree.set_roots(planes, last, first_game_id=first)
    tree.run(playouts)
    visits, q, rv, nodes, status = tree.root_stats()
    for g in range(0, n, 4):
        b = oracle.new_board()
        for i in range(int(lens[g])):
            oracle.lib().go_board_apply(C.byref(b), int(moves[g, i]), 1)
        om = oracle.MCTS(playouts, 5.0, 5, seed, first + g)
        om.run_playouts(b)
        ov, _, _ = om.root_children()
        assert (ov == visits[g]).all(), "game %d" % g
        assert np.float32(q[g]).tobytes() == np.float32(om.root_value).tobytes() and nodes[g] == om.size
    tree.close()


@pytest.mark.parametrize("first,kind,playouts", _first_by_default([(9000, 1, 2500), (123456, 0, 2500), (424242, 1, 4000)]))
def test_extended_k6_other_positions(oracle, first, kind, playouts):
    n = 28
    moves, lens, _ = G.synth_boards(n, kind, first_board=first)
    pos = [[int(m) for m in moves[g, :min(int(lens[g]), 2 + g)]] for g in range(n)]
    t = G.TraditionalMCTS(n, node_capacity=1 << 20)
    t.set_positions(pos)
    t.run(playouts)
    st = t.root_stats()
    for g in range(n):
        o = oracle.TraditionalMCTS(5.0)
        o.search(pos[g], playouts)
        v, qq, p, best = o.root_children()
        assert (v == st["visits"][g]).all() and (qq.view(np.uint32) == st["values"][g].view(np.uint32)).all(), "game %d" % g
        assert best == st["best"][g] and o.n_nodes == st["n_nodes"][g] and o.evaluator_updates == st["evaluator_updates"][g], "game %d" % g
    t.close()


@pytest.mark.parametrize("first,gid", _first_by_default([(1000, 0), (888888, 70000)]))
def test_extended_k8_other_games(oracle, first, gid):
    n, P = 64, 2000
    moves, lens, _ = G.synth_boards(n, 0, first_board=first)
    pos = [[int(m) for m in moves[g, :min(int(lens[g]), (g * 7) % 40)]] for g in range(n)]
    t = G.PoolRAVEMCTS(n, node_capacity=(P + 200) * 225, c_puct=2.0, first_game_id=gid)
    t.set_positions(pos)
    t.run(P)
    st = t.root_stats()
    for g in range(n):
        o = oracle.PoolRAVEMCTS(2.0, 0.0, seed=G.DEFAULT_SEED, game_id=gid + g)
        o.run(pos[g], P)
        v, qq, p, av, aq, best = o.root_children()
        assert (v == st["visits"][g]).all() and (qq.view(np.uint32) == st["values"][g].view(np.uint32)).all() and best == st["best"][g], "game %d" % g
        assert (av == st["amaf_visits"][g]).all() and (aq.view(np.uint32) == st["amaf_values"][g].view(np.uint32)).all(), "game %d" % g
    t.close()
