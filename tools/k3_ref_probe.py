#!/usr/bin/env python3
"""Where the time of a kept-subtree self-play batch goes: handle creation, the persistent launch, teardown.  tools/k3_ref_probe.py [games] [slots] [playouts] [cap_mult]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gomokuai_amd import lib as G

games = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
playouts = int(sys.argv[3]) if len(sys.argv) > 3 else 800
mult = float(sys.argv[4]) if len(sys.argv) > 4 else 3
G.init(0)
dev = torch.device("cuda", 0)
for name, reuse, noise in (("new roots", False, None), ("kept", True, None), ("kept + noise", True, (0.05, 0.25)), ("kept + noise again", True, (0.05, 0.25))):
    cap = int(playouts * 225 * (mult if reuse else 1)) + 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tree = G.BatchedMCTS(slots, node_capacity=cap)
    tree.set_option(G.OPT_NOISE_SAMPLER, 1)
    d_moves = torch.zeros((games, 225), dtype=torch.uint8, device=dev); d_lens = torch.zeros(games, dtype=torch.int32, device=dev)
    d_winner = torch.zeros(games, dtype=torch.int8, device=dev); d_visits = torch.zeros((games, 225, 225), dtype=torch.int16, device=dev)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    tree.selfplay_run(games, 0, playouts, d_moves.data_ptr(), d_visits.data_ptr(), d_lens.data_ptr(), d_winner.data_ptr(), None, None, reuse, noise, None)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st = tree.root_stats()
    tree.close()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    moves = int(d_lens.sum())
    print(json.dumps({"mode": name, "games": games, "slots": slots, "cap": cap, "create_s": round(t1 - t0, 3), "run_s": round(t2 - t1, 3), "close_s": round(t3 - t2, 3), "moves": moves,
                      "playouts_per_s_run": moves * playouts / (t2 - t1), "overflow": bool((st[4] & 2).any()), "max_nodes": int(st[3].max())}), flush=True)
