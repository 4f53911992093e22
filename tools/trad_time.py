#!/usr/bin/env python3
"""K6 timing aid: n games x playouts of the pattern-guided search, whole-launch time; and the oracle on one core."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gomokuai_amd import lib as G
torch.cuda.set_device(0); G.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
playouts = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 18
moves, lens, _ = G.synth_boards(n, 1)
pos = [[int(m) for m in moves[g, :min(int(lens[g]), 12)]] for g in range(n)]
t = G.TraditionalMCTS(n, node_capacity=cap)
t.set_positions(pos); t.run(10); torch.cuda.synchronize()
t.set_positions(pos)
t0 = time.perf_counter(); t.run(playouts); torch.cuda.synchronize(); dt = time.perf_counter() - t0
s = t.root_stats()
print("gpu: %d games x %d playouts in %.3f s = %.2f M playouts/s; nodes/game mean %.0f max %d; status!=0: %d; updates/playout %.2f" %
      (n, playouts, dt, n * playouts / dt / 1e6, s["n_nodes"].mean(), s["n_nodes"].max(), int((s["status"] != 0).sum()), s["evaluator_updates"].mean() / playouts))
from oracle import oracle as O
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 5 and k < n:
    o = O.TraditionalMCTS(5.0); o.search(pos[k], playouts); k += 1
dt = time.perf_counter() - t0
print("cpu oracle, 1 core: %d searches in %.2f s = %.1f k playouts/s" % (k, dt, k * playouts / dt / 1e3))
