"""The library's device-block pool hands blocks back UNCLEARED (the driver's clear costs 1.5 s per 24 GB): searches on reused blocks that were filled
with 0xA5 first (gmk_pool_poison, a switch of the production library) must give what searches on fresh blocks give -- i.e. no kernel reads a node,
a header field or a record row that nothing wrote since the block changed hands."""
import ctypes as C

import numpy as np
import pytest

from gomokuai_amd import lib as G
from gomokuai_amd import selfplay

pytestmark = pytest.mark.gpu


@pytest.fixture
def poisoned_pool():
    G.init(0)
    G.release_pool()
    G.pool_poison(True)
    yield
    G.pool_poison(False)
    G.release_pool()


def _k3_search(n, playouts, first):
    moves, lens, _ = G.synth_boards(n, 0, first_board=first)
    lens = np.minimum(lens, 6).astype(np.int32)
    planes = G.moves_to_planes(moves, lens)
    last = np.array([moves[i, lens[i] - 1] if lens[i] > 0 else -1 for i in range(n)], dtype=np.int16)
    tree = G.BatchedMCTS(n, playouts_capacity=playouts)
    tree.set_roots(planes, last, first_game_id=first)
    tree.run(playouts)
    out = tree.root_stats()
    tree.close()                                    # its arenas (>= 16 MB each) go to the pool
    return moves, lens, out


def test_k3_searches_on_poisoned_blocks(oracle, poisoned_pool):
    n, playouts = 96, 200                           # 96 x 45 001 nodes: 34.6 MB of statistics, 17.3 MB of links and of parents -- all pooled
    _k3_search(n, playouts, 5000)                   # fills the pool
    moves, lens, (visits, q, rv, nodes, status) = _k3_search(n, playouts, 777)      # ... and this one searches on the poisoned blocks
    assert not status.any() and (rv == playouts).all()
    for g in range(0, n, 5):
        b = oracle.new_board()
        for i in range(int(lens[g])):
            oracle.lib().go_board_apply(C.byref(b), int(moves[g, i]), 1)
        om = oracle.MCTS(playouts, 5.0, 5, G.DEFAULT_SEED, 777 + g)
        om.run_playouts(b)
        assert (om.root_children()[0] == visits[g]).all() and nodes[g] == om.size and np.float32(q[g]).tobytes() == np.float32(om.root_value).tobytes(), "game %d" % g


def test_k6_searches_on_poisoned_blocks(oracle, poisoned_pool):
    n, playouts, cap = 40, 300, 1 << 19             # 40 x 524 288 nodes: every one of the five arrays is >= 16 MB
    def search(first):
        moves, lens, _ = G.synth_boards(n, 1, first_board=first)
        pos = [[int(m) for m in moves[g, :min(int(lens[g]), 4 + g % 9)]] for g in range(n)]
        t = G.TraditionalMCTS(n, node_capacity=cap)
        t.set_positions(pos)
        t.run(playouts)
        st = t.root_stats()
        t.close()
        return pos, st
    search(31)
    pos, st = search(4242)
    assert not (st["status"] & ~G.TraditionalMCTS.STATUS_ARENA_FULL).any()
    for g in range(0, n, 3):
        o = oracle.TraditionalMCTS(5.0)
        o.search(pos[g], playouts)
        v, qq, p, best = o.root_children()
        assert (v == st["visits"][g]).all() and (qq.view(np.uint32) == st["values"][g].view(np.uint32)).all() and best == st["best"][g] and o.n_nodes == st["n_nodes"][g], "game %d" % g


def test_self_play_with_kept_subtrees_on_poisoned_blocks(poisoned_pool):
    """The persistent loops with the reference agent's semantics (two arenas per slot, compacted subtrees, root priors written by the kernel) on
    poisoned blocks play the games they play on fresh ones (K3 and K6)."""
    kw3 = dict(seed=11, first_game_id=40, reuse_subtree=True, root_noise=(0.05, 0.25), slots=32, node_capacity=80000)       # 41 / 20 / 20 MB blocks: pooled
    G.pool_poison(False)
    fresh3 = selfplay.play_games(60, 40, **kw3).cpu()
    kw6 = dict(seed=3, first_game_id=9, opening_plies=2, reuse_subtree=True, root_noise=(0.05, 0.25), slots=8, node_capacity=1 << 18)
    fresh6 = selfplay.play_supervisor_games(20, 60, **kw6).cpu()
    G.release_pool()
    G.pool_poison(True)
    for _ in range(2):                               # the first run fills the pool, the second one plays on poisoned blocks
        again3 = selfplay.play_games(60, 40, **kw3).cpu()
        again6 = selfplay.play_supervisor_games(20, 60, **kw6).cpu()
    assert not again3.overflow and (fresh3.lens == again3.lens).all() and (fresh3.moves == again3.moves).all() and (fresh3.visits == again3.visits).all()
    assert not again6.overflow and (fresh6.lens == again6.lens).all() and (fresh6.moves == again6.moves).all() and (fresh6.visits == again6.visits).all()
