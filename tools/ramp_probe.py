#!/usr/bin/env python3
"""Profiling aid: K1's time per launch as a function of how long the GPU has been busy (the driver's bench times 20 launches after 5 warm-up
launches: 3 ms; tools/eval_time.py times 100-200).  Prints the mean of consecutive blocks of launches from a cold start, with and without
a hipGraph, and after an idle gap."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gomokuai_amd import lib as G
torch.cuda.set_device(0); G.init(0)
n = 65536
_, _, planes = G.synth_boards(n, 0)
dev = torch.device("cuda", 0)
d_planes = torch.from_numpy(planes.view(np.int16).reshape(n, 32)).to(dev)
d_scores = torch.empty((n, 900), dtype=torch.int32, device=dev)
d_density = torch.empty((n, 900), dtype=torch.int32, device=dev)
d_totals = torch.empty((n, 11), dtype=torch.int32, device=dev)
d_status = torch.empty((n,), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
f = lambda: G.eval_batch(d_planes.data_ptr(), n, d_scores.data_ptr(), d_density.data_ptr(), d_totals.data_ptr(), d_status.data_ptr(), stream)
torch.cuda.synchronize(); time.sleep(0.5)
def series(blocks, per):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    evs[0].record()
    for b in range(blocks):
        for _ in range(per): f()
        evs[b + 1].record()
    torch.cuda.synchronize()
    return [evs[b].elapsed_time(evs[b + 1]) / per for b in range(blocks)]
print("cold, blocks of 5 launches:", " ".join("%.4f" % x for x in series(24, 5)))
time.sleep(0.5)
print("after 0.5 s idle, blocks of 5:", " ".join("%.4f" % x for x in series(12, 5)))
print("no idle, blocks of 20:", " ".join("%.4f" % x for x in series(10, 20)))
torch.cuda.synchronize()
print("after a synchronize, blocks of 20:", " ".join("%.4f" % x for x in series(5, 20)))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): f()
g.replay(); torch.cuda.synchronize()
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("graph of 20, replay %d: %.4f ms per launch" % (rep, e0.elapsed_time(e1) / 20))

# ---- what puts the chip into the fast state, and how long an idle gap it survives ----
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
_, _, mplanes, mlast = B.mcts_openings(G, np, 4096, 0)
tree = G.BatchedMCTS(4096, playouts_capacity=800)
def k3():
    tree.set_roots(mplanes, mlast, first_game_id=0); tree.run(800, stream)
big = torch.empty(472 * 1024 * 1024 // 4, dtype=torch.int32, device=dev)
torch.cuda.synchronize(); time.sleep(0.5)
for _ in range(3): k3()
print("idle, then 3 K3 searches, then K1 blocks of 5:", " ".join("%.4f" % x for x in series(10, 5)))
torch.cuda.synchronize(); time.sleep(0.5)
for _ in range(300): big.fill_(3)
print("idle, then 300 fills of 472 MB, then K1 blocks of 5:", " ".join("%.4f" % x for x in series(10, 5)))
for gap in (0.001, 0.005, 0.02, 0.1, 0.3):
    series(40, 5); torch.cuda.synchronize(); time.sleep(gap)
    print("steady K1, %.0f ms idle, then blocks of 5:" % (gap * 1e3), " ".join("%.4f" % x for x in series(8, 5)))
