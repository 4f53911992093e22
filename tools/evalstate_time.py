#!/usr/bin/env python3
"""K2 timing aid: time per evaluator update (apply) per game-wavefront."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gomokuai_amd import lib as G
torch.cuda.set_device(0); G.init(0)
n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 2304, 40
moves, lens, _ = G.synth_boards(n, 1)
scr = np.full((n, k), -1, np.int16)
for g in range(n):
    m = min(int(lens[g]), k); scr[g, :m] = moves[g, :m]
d = torch.from_numpy(scr).cuda()
e = G.EvaluatorStates(n)
L = G.load()
back = torch.full((n, k), -2, dtype=torch.int16).cuda()      # take everything back again
reps = 40                                                     # long enough for the clocks to settle
L.gmk_evalstate_update(e.h, d.data_ptr(), k, 0); L.gmk_evalstate_update(e.h, back.data_ptr(), k, 0); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    L.gmk_evalstate_update(e.h, d.data_ptr(), k, 0); L.gmk_evalstate_update(e.h, back.data_ptr(), k, 0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
ups = 2 * (scr >= 0).sum(1).mean()
print("K2-UPDATES-PER-LAUNCH %d" % int((scr >= 0).sum()))                     # (apply or revert: what tools/pmc_k2_phases.sh normalises its counters by)
print("K2: %d games x %.1f updates (apply, then revert) in %.3f ms -> %.2f us per update per game (%d games per launch), %.1f M updates/s" % (n, ups, dt * 1e3, dt * 1e6 / ups, n, n * ups / dt / 1e6))
