#!/usr/bin/env python3
"""K1 against the oracle on many more boards than the suite holds: `rounds` x 65 536 boards of both synthetic kinds from different
first-board offsets (the oracle replays them on all host cores).  usage: stress_parity.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from gomokuai_amd import lib as G
from oracle import oracle as O
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n, cores = 65536, min(len(os.sched_getaffinity(0)), 16)
O.lib()
_m, _l, _ = G.synth_boards(2, 0)
O.replay_batch(_m, _l)            # the oracle builds its tables on first use, and not under a lock: once, before the threads start
bad = 0
t0 = time.time()
for r in range(rounds):
    for kind in (0, 1):
        first = 1000003 * (r + 1) + 17 * kind
        moves, lens, planes = G.synth_boards(n, kind, first_board=first)
        got = G.eval_batch_host(planes)
        cuts = [n * i // cores for i in range(cores + 1)]
        with ThreadPoolExecutor(cores) as pool:
            parts = list(pool.map(lambda i: O.replay_batch(moves[cuts[i]:cuts[i + 1]], lens[cuts[i]:cuts[i + 1]]), range(cores)))
        ref = [np.concatenate([p[k] for p in parts]) for k in range(4)]
        wrong = [int((a.reshape(n, -1) != b.reshape(n, -1)).any(1).sum()) for a, b in zip(ref, got)]
        compounds = int((ref[2][:, 8:11] != 0).any(1).sum())
        print("round %d kind %d first_board %d: mismatching boards (scores, density, totals, status) %s; %d boards with compounds, %.0f s" % (r, kind, first, wrong, compounds, time.time() - t0), flush=True)
        bad += sum(wrong)
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)
