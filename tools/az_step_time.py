#!/usr/bin/env python3
"""K7 + K9 timing aid: one lock-step playout of 4 096 games (select kernel, the network's two kernels, expand kernel), eager and replayed from a
hipGraph; under `rocprofv3 --kernel-trace --stats` the per-kernel share."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gomokuai_amd import lib as G
from gomokuai_amd.network import FusedPolicyValueNetwork, PolicyValueNetwork
torch.cuda.set_device(0); G.init(0)
n, P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 32
moves, lens, _ = G.synth_boards(n, 0)
lens = np.minimum(lens, 4).astype(np.int32)
planes = G.moves_to_planes(moves, lens)
last = np.stack([moves[np.arange(n), lens - 1], moves[np.arange(n), lens - 2]], 1).astype(np.int16)
fused = FusedPolicyValueNetwork(PolicyValueNetwork(seed=1).cuda().eval())
for graph in (False, True, False, True):
    tree = G.AlphaZeroMCTS(n, node_capacity=(P + 8) * 225 + 1)
    tree.set_roots(planes, last)
    with torch.no_grad():
        tree.search(fused, 4, graph=graph)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); tree.search(fused, P, graph=graph); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / P
    print("n=%d %s: %.4f ms per step = %.3f M playouts/s" % (n, "hipGraph" if graph else "eager   ", ms, n / ms / 1e3))
    tree.close()
