"""K3 at BASELINE configs[0]: ONE game, 1 000 playouts per search (the reference's own CPU-runnable case) -- the latency of a search."""
import time, sys
import numpy as np
sys.path.insert(0, '.')
from gomokuai_amd import lib as G
import torch
P = 1000
moves, lens, _ = G.synth_boards(1, 0)
lens = np.minimum(lens, 4).astype(np.int32)
planes = G.moves_to_planes(moves, lens)
last = np.array([moves[0, lens[0] - 1]], dtype=np.int16)
t = G.BatchedMCTS(1, playouts_capacity=P, c_rollouts=5)
print(t.launch_info())
best = 1e9
for rep in range(5):
    t.set_roots(planes, last, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); t.run(P); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    best = min(best, dt)
v, q, rv, nodes, st = t.root_stats()
print("one game x %d playouts: %.2f ms per search = %.0f k playouts/s; visits digest %d" % (P, best * 1e3, P / best / 1e3, int((v.astype(np.int64) * np.arange(225)).sum())))
