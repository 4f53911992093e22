#!/usr/bin/env python3
"""Turns rocprofv3 --pmc passes (one directory per pass, as tools/profile.sh leaves them under gpurun_out/) into the
text summary committed under profiles/.  Usage: summarize_pmc.py <dir with pmc_*/ subdirs> <boards per launch> <playouts per launch>"""
import collections
import csv
import glob
import sys

root, n_boards, n_playouts = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
vals = collections.defaultdict(dict)
for path in sorted(glob.glob(root + "/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = "eval_positions_kernel" if "eval_positions" in r["Kernel_Name"] else "mcts_playouts_kernel" if "mcts_playouts" in r["Kernel_Name"] else None
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        for c, v in agg[k].items():
            vals[k][c] = sum(v) / len(v)
print("# rocprofv3 --pmc, one pass per counter group (separate runs), command: python3 bench.py --steps 5 --warmup 2 --mcts-reps 1 --no-cpu-baseline")
print("# FETCH_SIZE / WRITE_SIZE are in KB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads half the bytes of wide coalesced reads (x2 below);")
print("# narrower accesses are uncalibrated (the 8-byte node reads of the MCTS kernel are given uncorrected and corrected).")
for k, unit, per in (("eval_positions_kernel", "board", n_boards), ("mcts_playouts_kernel", "playout", n_playouts)):
    v = vals.get(k)
    if not v:
        continue
    print("\n## %s  (mean per launch; one launch = %d %ss)" % (k, per, unit))
    for c in sorted(v):
        print("%-24s %.6g" % (c, v[c]))
    print("per %s: VALU %.0f, SALU %.0f, LDS %.0f wave-instructions" % (unit, v["SQ_INSTS_VALU"] / per, v["SQ_INSTS_SALU"] / per, v["SQ_INSTS_LDS"] / per))
    print("HBM traffic per launch: read %.1f MB uncorrected / %.1f MB with the x2 correction, write %.1f MB" %
          (v["FETCH_SIZE"] * 1024 / 1e6, 2 * v["FETCH_SIZE"] * 1024 / 1e6, v["WRITE_SIZE"] * 1024 / 1e6))
    print("wave time (quad-cycles): active %.0f%%, waiting %.0f%%, issue-stalled %.0f%%" %
          (100 * v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 100 * v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
    print("LDS bank-conflict cycles / LDS active cycles: %.0f%%" % (100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]))
