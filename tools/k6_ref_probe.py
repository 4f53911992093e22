#!/usr/bin/env python3
"""K6 self-play modes side by side: seconds, searched moves, playouts/s and evaluator updates per playout.  tools/k6_ref_probe.py [games] [slots] [playouts]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gomokuai_amd import lib as G

games = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
playouts = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
G.init(0)
dev = torch.device("cuda", 0)
m, l, _ = G.synth_boards(games, 0)
open_moves, open_lens = m, np.minimum(l, 2).astype(np.int32)
modes = (("new roots, persistent", False, None, True), ("kept, persistent", True, None, True), ("kept + noise, persistent", True, (0.05, 0.25), True),
         ("kept + noise, lock step", True, (0.05, 0.25), False), ("new roots, lock step", False, None, False))
if len(sys.argv) > 4:
    modes = [modes[int(i)] for i in sys.argv[4].split(",")]
for name, reuse, noise, persistent in modes:
    cap = (3 if reuse else 1) * (1 << 18)
    tree = G.TraditionalMCTS(slots, node_capacity=cap, c_puct=5.0)
    tree.set_option(G.OPT_NOISE_SAMPLER, 1)
    d_moves = torch.zeros((games, 225), dtype=torch.uint8, device=dev); d_lens = torch.zeros(games, dtype=torch.int32, device=dev)
    d_winner = torch.zeros(games, dtype=torch.int8, device=dev); d_visits = torch.zeros((games, 225, 225), dtype=torch.int16, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps, overflow = tree.selfplay_run(games, 0, playouts, d_moves.data_ptr(), d_visits.data_ptr(), d_lens.data_ptr(), d_winner.data_ptr(), open_moves, open_lens, reuse, noise,
                                        G.DEFAULT_SEED, None, 0, persistent)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    L = G.load()
    upd = np.zeros(slots, np.uint64); nn = np.zeros(slots, np.int32)
    tree.positioned = True
    import ctypes as C
    # evaluator updates are summed per slot in the headers; read them without repositioning
    L.gmk_trad_set_positions(tree.h, np.zeros((slots, 225), np.uint8).ctypes.data, np.zeros(slots, np.int32).ctypes.data)
    L.gmk_trad_root_stats(tree.h, None, None, None, None, None, None, nn.ctypes.data, None, upd.ctypes.data)
    tree.close()
    lens = d_lens.cpu().numpy()
    searched = int(lens.sum() - np.minimum(lens, 2).sum())
    print(json.dumps({"mode": name, "seconds": round(dt, 3), "searched_moves": searched, "mean_len": float(lens.mean()), "max_len": int(lens.max()), "percentiles": {str(q): int(np.percentile(lens, q)) for q in (50, 90, 99, 99.9)}, "games_of_225": int((lens == 225).sum()), "games_over_100": int((lens > 100).sum()), "playouts_per_s": searched * playouts / dt,
                      "games_per_s": games / dt, "evaluator_updates_per_playout": float(upd.sum()) / (searched * playouts), "overflow": overflow, "steps": steps}), flush=True)
