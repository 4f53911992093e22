"""Error behaviour of the C-ABI on a live device: bad arguments and out-of-order calls come back as status codes with a
message (the Python layer raises RuntimeError), capacities are reported per game, nothing crashes."""
import ctypes as C

import numpy as np
import pytest

from gomokuai_amd import lib as G

pytestmark = pytest.mark.gpu


def test_out_of_order_and_bad_arguments():
    G.init()
    L = G.load()
    h = C.c_void_p()
    assert L.gmk_mcts_create(1, 1 << 24, 5.0, 5, 0, C.byref(h)) == -3            # GMK_ERR_ARG: child index would not fit
    assert b"2^24" in L.gmk_last_error()
    assert L.gmk_trad_create(1, 100, C.byref(h)) == -3
    assert L.gmk_az_create(0, 1000, 5.0, C.byref(h)) == -3
    t = G.TraditionalMCTS(2, node_capacity=1024)
    with pytest.raises(RuntimeError, match="set_positions"):
        t.run(10)
    with pytest.raises(RuntimeError, match="set_positions"):
        t.step()
    with pytest.raises(RuntimeError, match="set_positions"):
        t.root_stats()                                           # the arenas hold nothing to read yet
    t.close()
    m = G.BatchedMCTS(2, playouts_capacity=10)
    for call in (lambda: m.run(5), lambda: m.root_stats(), lambda: m.add_root_noise()):
        with pytest.raises(RuntimeError, match="set_roots"):
            call()
    m.close()
    r = G.PoolRAVEMCTS(2, node_capacity=1024)
    with pytest.raises(RuntimeError, match="set_positions"):
        r.run(10)
    with pytest.raises(RuntimeError, match="set_positions"):
        r.root_stats()
    r.set_positions([[112], [112, 113]])
    with pytest.raises(RuntimeError, match="run_poolrave"):
        r.root_stats()                                           # positioned, but no AMAF statistics before the first PoolRAVE search
    r.run(2)                                                     # 225 + 224 + 223 nodes do not fit in 1024... the second expansion does not
    assert (r.root_stats()["n_nodes"] <= 1024).all()
    with pytest.raises(RuntimeError, match="run_poolrave"):
        G.TraditionalMCTS.run(r, 5)                              # one policy per handle
    r.close()
    a = G.AlphaZeroMCTS(2, node_capacity=1024)
    with pytest.raises(RuntimeError, match="set_roots"):
        a.select()
    with pytest.raises(RuntimeError, match="set_roots"):
        a.root_stats()
    with pytest.raises(RuntimeError, match="invalid position"):
        bad = np.zeros((2, 2, 16), np.uint16); bad[0, 0, 3] = 1; bad[0, 1, 3] = 1     # a cell that is black and white
        a.set_roots(bad, np.full((2, 2), -1, np.int16))
    a.close()
    assert G.eval_batch_host(np.zeros((0, 2, 16), np.uint16))[0].shape == (0, 4, 225)   # empty batch


def test_illegal_steps_are_reported_not_played():
    import torch
    G.init()
    t = G.TraditionalMCTS(2, node_capacity=1 << 14)
    t.set_positions([[112, 113], [112, 113, 127]])
    t.run(50)
    t.step(np.array([112, -1], np.int16))                        # game 0: the cell is taken; game 1: best child
    st = t.root_stats()
    assert st["status"][0] & 8 and st["status"][1] == 0
    t.close()
    a = G.AlphaZeroMCTS(1, node_capacity=1 << 12)
    planes = G.moves_to_planes(np.array([[112, 113] + [0] * 62], np.uint8), np.array([2], np.int32))
    a.set_roots(planes, np.array([[113, 112]], np.int16))
    a.search(lambda s: (torch.zeros(1, device="cuda"), torch.full((1, 225), 1 / 225.0, device="cuda")), 5)
    a.step(np.array([113], np.int16))
    assert a.root_stats()["status"][0] & 4
    a.close()


def test_persistent_self_play_ends_when_nothing_can_be_played():
    """The one-launch self-play loop ends when every slot's game has; configurations in which a game can never move must not spin: an arena
    too small for a single expansion stops its slot and is reported as an overflow, and zero playouts per move are refused up front."""
    from gomokuai_amd import selfplay
    rec = selfplay.play_games(6, 10, node_capacity=100, slots=4)         # 100 nodes: the root's 225 children do not fit
    assert rec.overflow and int(rec.lens.max()) == 0
    with pytest.raises(G.GmkError):
        selfplay.play_games(6, 0, slots=4)
