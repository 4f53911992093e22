"""tests/golden/regression.json: digests of the oracle's outputs on fixed seeded inputs (made by tests/golden/make_golden.py).
CPU: the oracle still produces them.  GPU: the device kernels produce them too, without the oracle in the loop."""
import importlib.util
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "regression.json")))


def _maker():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_oracle_reproduces_the_golden_digests(oracle):
    assert _maker().cases() == GOLDEN


@pytest.mark.gpu
def test_device_reproduces_the_golden_digests():
    from gomokuai_amd import lib as G
    mk = _maker()
    G.init()
    for kind in (0, 1):
        case = GOLDEN["eval_kind%d" % kind]
        _, _, planes = G.synth_boards(case["boards"], kind, first_board=case["first_board"])
        assert mk.digest(*G.eval_batch_host(planes)) == case["sha256"]
    case = GOLDEN["mcts_random"]
    n = case["games"]
    moves, lens, _ = G.synth_boards(n, 0, first_board=case["first_board"])
    lens = np.minimum(lens, case["plies"]).astype(np.int32)
    tree = G.BatchedMCTS(n, playouts_capacity=case["playouts"])
    tree.set_roots(G.moves_to_planes(moves, lens), np.array([moves[g, lens[g] - 1] for g in range(n)], np.int16), first_game_id=case["first_game_id"])
    tree.run(case["playouts"])
    assert mk.digest(tree.root_stats()[0]) == case["sha256"]
    tree.close()
    case = GOLDEN["mcts_traditional"]
    n = case["games"]
    moves, lens, _ = G.synth_boards(n, 1, first_board=case["first_board"])
    t = G.TraditionalMCTS(n, node_capacity=1 << 17)
    t.set_positions([[int(x) for x in moves[g, :min(int(lens[g]), 4 + 2 * g)]] for g in range(n)])
    t.run(case["playouts"])
    st = t.root_stats()
    stats = []
    for g in range(n):
        stats += [st["visits"][g], st["values"][g].view(np.uint32), st["priors"][g].view(np.uint32), np.array([st["best"][g], st["n_nodes"][g]], np.int64)]
    assert mk.digest(*stats) == case["sha256"]
    t.close()
    case = GOLDEN["mcts_poolrave"]
    n = case["games"]
    moves, lens, _ = G.synth_boards(n, 0, first_board=case["first_board"])
    t = G.PoolRAVEMCTS(n, node_capacity=1 << 18, c_puct=2.0, first_game_id=case["first_game_id"])
    t.set_positions([[int(x) for x in moves[g, :min(int(lens[g]), 3 + g)]] for g in range(n)])
    t.run(case["playouts"][0])
    t.step()
    t.run(case["playouts"][1])
    st = t.root_stats()
    stats = []
    for g in range(n):
        stats += [st["visits"][g], st["values"][g].view(np.uint32), st["priors"][g].view(np.uint32), st["amaf_visits"][g], st["amaf_values"][g].view(np.uint32),
                  np.array([st["best"][g], st["root_visits"][g]], np.int64)]
    assert mk.digest(*stats) == case["sha256"]
    t.close()
