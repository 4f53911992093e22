#!/usr/bin/env python3
"""Writes tests/golden/reference_tuples.npz: training tuples produced by the REFERENCE's own game loop.

Container-only (the reference tree does not exist on the GPU box): imports /root/reference/agents/agent.py and
agents/utils.py -- numpy-only files -- with this repo's CorePyExt standing in for the extension module the reference never
shipped, and plays `dual_play(agents, verbose=True)` (agents/utils.py:29-63) with the reference's random `Agent`
(agents/agent.py:6-27; no search, so no GPU is needed).  What is stored is data only: per game the move list and the winner,
and per move the reference loop's (encoded states, score, probabilities) tuple.  tests/test_selfplay_gpu.py feeds the move lists
to the device kernels K4 + K5 and compares their states / values with these; tests/test_selfplay.py does the same with the host
path.  The random agent's probabilities are uniform 1/225 (agents/agent.py:24): pi is pinned by shape and value only.

Usage: python tests/golden/make_reference_tuples.py   (CPU only; Board.random_move is seeded through core.set_seed)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REFERENCE = "/root/reference"


def main():
    from gomokuai_amd import core as our_core
    sys.path.insert(0, our_core.module_path)                     # CorePyExt
    sys.path.insert(0, REFERENCE)
    for k in [k for k in sys.modules if k == "core" or k.startswith("core.")]:
        sys.modules.pop(k)
    ref_core = importlib.import_module("core")                   # /root/reference/core/__init__.py re-exporting CorePyExt
    agent_mod = importlib.import_module("agents.agent")
    utils = importlib.import_module("agents.utils")
    assert ref_core.Board is our_core.Board
    if hasattr(our_core, "set_seed"):
        our_core.set_seed(20260410)
    agent = agent_mod.Agent()
    games, rows = 6, []
    moves = np.zeros((games, 225), dtype=np.uint8)
    lens = np.zeros(games, dtype=np.int32)
    winner = np.zeros(games, dtype=np.int8)
    states, values, probs, owner = [], [], [], []
    for g in range(games):
        board = ref_core.Board()
        data = utils.dual_play({ref_core.Player.black: agent, ref_core.Player.white: agent}, board=board, verbose=True)
        record = [p.id for p in board.move_record]
        assert len(record) == len(data) and board.status["is_end"]
        lens[g] = len(record)
        moves[g, :len(record)] = record
        winner[g] = int(float(board.status["winner"]))
        for s, v, p in data:
            states.append(np.asarray(s, dtype=np.uint8))
            values.append(float(v))
            probs.append(np.asarray(p, dtype=np.float64).reshape(-1))
            owner.append(g)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tuples.npz")
    np.savez_compressed(out, moves=moves, lens=lens, winner=winner, states=np.stack(states), values=np.array(values, dtype=np.float64),
                        probs=np.stack(probs).astype(np.float32), game=np.array(owner, dtype=np.int32))
    print("wrote %s: %d games, %d tuples, %d bytes" % (out, games, len(values), os.path.getsize(out)))


if __name__ == "__main__":
    main()
