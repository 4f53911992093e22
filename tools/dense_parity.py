#!/usr/bin/env python3
"""K1 on DENSE boards (100..225 stones, no five in a row: prefixes of shuffled tie games), which the synthetic generator's 8..60-ply
boards never reach: queue capacities, saturated counters, many compounds.  Compares every output with the oracle's in-order replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomokuai_amd import lib as G
from oracle import oracle as O
G.init(); O.lib()
rng = np.random.RandomState(5)
cls = lambda c: ((c % 15) // 2 + c // 15) % 2                 # two colour classes that never line up five (pairs of columns, shifted per row)
blacks = [c for c in range(225) if cls(c) == 0]
whites = [c for c in range(225) if cls(c) == 1]
if len(blacks) < len(whites):
    blacks, whites = whites, blacks
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
moves = np.zeros((n, 225), np.uint8)
lens = np.zeros(n, np.int32)
for g in range(n):
    b, w = list(rng.permutation(blacks)), list(rng.permutation(whites))
    seq = []
    while b or w:
        if b: seq.append(b.pop())
        if w: seq.append(w.pop())
    moves[g] = seq
    lens[g] = rng.randint(100, 226)
t0 = time.time()
planes = G.moves_to_planes(moves, lens)
got = G.eval_batch_host(planes)
ref = O.replay_batch(moves, lens)
bad = sum(int((a != b).reshape(n, -1).any(1).sum()) for a, b in zip(ref, got))
print("K1 dense: %d boards of %d..%d stones, mismatching arrays: %d; device error flags %d, oracle error flags %d, finished %d  (%.1f s)" %
      (n, lens.min(), lens.max(), bad, int((got[3] & 2).astype(bool).sum()), int((ref[3] & 2).astype(bool).sum()), int((ref[3] & 1).sum()), time.time() - t0))
sys.exit(1 if bad else 0)
