#!/usr/bin/env python3
"""Writes tests/golden/regression.json: digests of what the CPU oracle computes on fixed, seeded inputs.

The reference itself cannot be built or imported in this image (Eigen is absent), so these are NOT
reference outputs: the reference's own test vectors live as literals in tests/test_oracle_golden.py.  This file freezes
the oracle (and through tests/test_golden.py the device kernels) against drift from one round to the next.
Usage: python tests/golden/make_golden.py   (CPU only)"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gomokuai_amd import lib as G          # host-side generator of the synthetic boards (no GPU needed)  # noqa: E402
from oracle import oracle as O             # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def cases():
    out = {}
    for kind in (0, 1):
        moves, lens, _ = G.synth_boards(256, kind, first_board=424242)
        scores, density, totals, status = O.replay_batch(moves, lens)
        out["eval_kind%d" % kind] = {"boards": 256, "first_board": 424242, "sha256": digest(scores, density, totals, status)}
    # K3: seeded random-rollout searches
    moves, lens, _ = G.synth_boards(6, 0, first_board=777)
    visits = []
    for g in range(6):
        b = O.new_board()
        for i in range(min(int(lens[g]), 5)):
            O.lib().go_board_apply(C.byref(b), int(moves[g, i]), 1)
        m = O.MCTS(300, 5.0, 5, G.DEFAULT_SEED, 40 + g)
        m.run_playouts(b)
        visits.append(m.root_children()[0])
    out["mcts_random"] = {"games": 6, "playouts": 300, "plies": 5, "first_board": 777, "first_game_id": 40, "sha256": digest(np.stack(visits))}
    # K6: pattern-guided searches
    moves, lens, _ = G.synth_boards(6, 1, first_board=999)
    stats = []
    for g in range(6):
        t = O.TraditionalMCTS(5.0)
        t.search([int(x) for x in moves[g, :min(int(lens[g]), 4 + 2 * g)]], 500)
        v, q, p, best = t.root_children()
        stats += [v, q.view(np.uint32), p.view(np.uint32), np.array([best, t.n_nodes], np.int64)]
    out["mcts_traditional"] = {"games": 6, "playouts": 500, "first_board": 999, "sha256": digest(*stats)}
    # K8: PoolRAVE searches, a step to the most visited child in between
    moves, lens, _ = G.synth_boards(5, 0, first_board=1234)
    stats = []
    for g in range(5):
        pos = [int(x) for x in moves[g, :min(int(lens[g]), 3 + g)]]
        t = O.PoolRAVEMCTS(2.0, 0.0, seed=G.DEFAULT_SEED, game_id=60 + g)
        t.run(pos, 300)
        pos.append(t.step_forward())
        t.run(pos, 200)
        v, q, p, av, aq, best = t.root_children()
        stats += [v, q.view(np.uint32), p.view(np.uint32), av, aq.view(np.uint32), np.array([best, t.root_visits], np.int64)]
    out["mcts_poolrave"] = {"games": 5, "playouts": [300, 200], "first_board": 1234, "first_game_id": 60, "sha256": digest(*stats)}
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "regression.json")
    json.dump(cases(), open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
