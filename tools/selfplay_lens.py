#!/usr/bin/env python3
"""Game-length distribution of the K3 self-play modes (what the tail of a finite batch is made of): tools/selfplay_lens.py [games] [playouts]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gomokuai_amd import selfplay

games = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
playouts = int(sys.argv[2]) if len(sys.argv) > 2 else 800
for name, kw in (("new roots", {}), ("kept subtrees", {"reuse_subtree": True}), ("kept subtrees + noise", {"reuse_subtree": True, "root_noise": (0.05, 0.25)})):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec = selfplay.play_games(games, playouts, record_visits=False, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lens = rec.lens.cpu().numpy()
    pct = {str(p): int(np.percentile(lens, p)) for p in (50, 90, 99, 99.9, 100)}
    print(json.dumps({"mode": name, "games": games, "playouts": playouts, "seconds": round(dt, 3), "moves": int(lens.sum()), "mean": float(lens.mean()), "percentiles": pct,
                      "games_over_100": int((lens > 100).sum()), "games_over_150": int((lens > 150).sum()), "games_of_225": int((lens == 225).sum()),
                      "playouts_per_s": float(lens.sum()) * playouts / dt, "draws": int((rec.winner.cpu().numpy() == 0).sum())}), flush=True)
