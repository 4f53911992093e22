#!/usr/bin/env python3
"""Where a ply of the batched supervisor self-play goes (1 792 games, 1 000 playouts, kept subtrees, root noise): the search itself
vs. the host-side steps around it.  Measured: noise 1.9, search 64.9, root statistics 0.5, host boards 0.8, re-rooting 5.5 ms per ply."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gomokuai_amd import lib as G
from gomokuai_amd.selfplay import _HostGames
torch.cuda.set_device(0); G.init(0)
n = 2048
games = _HostGames(n)
m, l, _ = G.synth_boards(n, 0)
games.open_with(m, l, 2)
tree = G.TraditionalMCTS(n, node_capacity=3 * 1000 * 226 + 1)
T = {k: 0.0 for k in ("setpos", "noise", "run", "stats", "host", "step")}
def tick(key, t0):
    torch.cuda.synchronize(); T[key] += time.perf_counter() - t0
tree.set_positions(games.moves, games.lens)
for ply in range(12):
    t0 = time.perf_counter(); tree.add_root_noise(0.05, 0.25); tick("noise", t0)
    t0 = time.perf_counter(); tree.run(1000); tick("run", t0)
    t0 = time.perf_counter(); st = tree.root_stats(); tick("stats", t0)
    t0 = time.perf_counter()
    played = np.where(games.over, -1, st["best"]).astype(np.int16)
    at = games.lens.copy(); moved = games.apply(played); tick("host", t0)
    t0 = time.perf_counter(); tree.step(played); tick("step", t0)
print({k: round(v / 12 * 1e3, 1) for k, v in T.items()}, "ms per ply")
