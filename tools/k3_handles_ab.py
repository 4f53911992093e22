#!/usr/bin/env python3
"""A/B in one process: the new-root self-play pipeline with two handles x 8 192 slots (the default plan) against one handle of 8 192 / 16 384 slots.
Arenas are allocated before each timed run (prepare_only), runs alternate.  tools/k3_handles_ab.py [games] [playouts]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gomokuai_amd import selfplay

games = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
playouts = int(sys.argv[2]) if len(sys.argv) > 2 else 800
configs = [("two handles x 8192 slots (default)", dict(slots="auto", handles="auto")), ("one handle, 8192 slots", dict(slots=8192, handles=1)),
           ("one handle, 16384 slots", dict(slots=16384, handles=1)), ("two handles x 4096 slots", dict(slots=8192, handles=2))]
for rep in range(2):
    for name, kw in configs:
        selfplay.play_games(games, playouts, prepare_only=True, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rec = selfplay.play_games(games, playouts, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        moves = int(rec.lens.sum())
        print(json.dumps({"rep": rep, "config": name, "seconds": round(dt, 3), "games_per_s": games / dt, "playouts_per_s": moves * playouts / dt}), flush=True)
        del rec
