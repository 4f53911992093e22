#!/usr/bin/env python3
"""K7 timing aid: lock-step network-guided search, n games, with and without hipGraph replay of a playout step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gomokuai_amd import lib as G
from gomokuai_amd.network import PolicyValueNetwork
torch.cuda.set_device(0); G.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
playouts = int(sys.argv[2]) if len(sys.argv) > 2 else 200
moves, lens, _ = G.synth_boards(n, 0)
lens = np.minimum(lens, 4).astype(np.int32)
planes = G.moves_to_planes(moves, lens)
last = np.stack([moves[np.arange(n), lens - 1], moves[np.arange(n), lens - 2]], 1).astype(np.int16)
net = PolicyValueNetwork(seed=1).cuda().eval()
for dtype in (torch.float32, torch.bfloat16):
    tree = G.AlphaZeroMCTS(n, node_capacity=playouts * 225 + 1)
    tree.set_roots(planes, last)
    def network(states):
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
            v, p = net(states)
        return v.float(), p.float()
    tree.search(network, 3); torch.cuda.synchronize()
    t0 = time.perf_counter(); tree.search(network, playouts); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    # the network alone
    s = tree.states.clone(); torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(20): network(s)
    torch.cuda.synchronize(); dn = (time.perf_counter() - t1) / 20
    print("%s: %d games x %d playouts in %.3f s = %.2f M playouts/s (%.2f ms per step, network alone %.2f ms)" %
          (str(dtype).split(".")[1], n, playouts, dt, n * playouts / dt / 1e6, dt / playouts * 1e3, dn * 1e3))
    tree.close()
