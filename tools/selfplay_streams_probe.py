#!/usr/bin/env python3
"""Probe: whole-game RandomPolicy self-play with the games split over k handles that run side by side on k streams (one host thread
each), so that one handle's slowest wavefronts of a search overlap the other handles' work.  usage: selfplay_streams_probe.py GAMES K SLOTS_PER_HANDLE"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gomokuai_amd import lib as G

games, k, slots = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
P, N = 800, 225
G.init(0)
dev = torch.device("cuda", 0)
d_moves = torch.zeros((games, N), dtype=torch.uint8, device=dev)
d_lens = torch.zeros(games, dtype=torch.int32, device=dev)
d_winner = torch.zeros(games, dtype=torch.int8, device=dev)
d_visits = torch.zeros((games, N, N), dtype=torch.int16, device=dev)
parts = [((games * i) // k, (games * (i + 1)) // k) for i in range(k)]
trees = [G.BatchedMCTS(min(slots, hi - lo), node_capacity=P * N + 1) for lo, hi in parts]
streams = [torch.cuda.Stream(dev) for _ in range(k)]
torch.cuda.synchronize()
t0 = time.perf_counter()
def work(i):
    lo, hi = parts[i]
    trees[i].selfplay_run(hi - lo, lo, P, d_moves[lo:].data_ptr(), d_visits[lo:].data_ptr(), d_lens[lo:].data_ptr(), d_winner[lo:].data_ptr(), None, None, False, None,
                          streams[i].cuda_stream)
th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
moves = int(d_lens.sum())
print("games %d handles %d slots/handle %d: %.3f s, %.1f M playouts/s, moves %d, checksum %d" % (games, k, slots, dt, moves * P / dt / 1e6, moves, int(d_moves.to(torch.int64).sum())))
