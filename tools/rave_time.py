#!/usr/bin/env python3
"""K8 timing aid: n games x playouts of the PoolRAVE search, whole-launch time; and the oracle on one core."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gomokuai_amd import lib as G
torch.cuda.set_device(0); G.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
playouts = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cap = int(sys.argv[3], 0) if len(sys.argv) > 3 and int(sys.argv[3], 0) > 0 else playouts * 222 + 512
moves, lens, _ = G.synth_boards(n, 0)
pos = [[int(m) for m in moves[g, :4]] for g in range(n)]
t = G.PoolRAVEMCTS(n, node_capacity=cap)
t.set_positions(pos); t.run(10); torch.cuda.synchronize()
for rep in range(2):
    t.set_positions(pos)
    t0 = time.perf_counter(); t.run(playouts); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = t.root_stats()
    print("gpu: %d games x %d playouts in %.3f s = %.2f M playouts/s; nodes/game mean %.0f max %d; status!=0: %d" %
          (n, playouts, dt, n * playouts / dt / 1e6, s["n_nodes"].mean(), s["n_nodes"].max(), int((s["status"] != 0).sum())))
if len(sys.argv) > 4:
    sys.exit(0)
from oracle import oracle as O
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 5 and k < n:
    o = O.PoolRAVEMCTS(2.0, 0.0, game_id=k); o.run(pos[k], playouts); k += 1
dt = time.perf_counter() - t0
print("cpu oracle, 1 core: %d searches in %.2f s = %.1f k playouts/s" % (k, dt, k * playouts / dt / 1e3))
