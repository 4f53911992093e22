#!/usr/bin/env python3
"""Turns rocprofv3 --pmc passes (one directory per pass, as tools/profile.sh leaves them under gpurun_out/) into the
text summary committed under profiles/.
Usage: summarize_pmc.py <dir with pmc_*/ subdirs> <boards per K1 launch> <playouts per K3 launch> [<playouts per K6 launch> [<playouts per K8 launch> [<updates per pair of K2 launches>]]]"""
import collections
import csv
import glob
import sys

root, n_boards, n_playouts = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n_trad = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n_rave = int(sys.argv[5]) if len(sys.argv) > 5 else 0
n_evs = int(sys.argv[6]) if len(sys.argv) > 6 else 0
vals = collections.defaultdict(dict)
for path in sorted(glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        k = "eval_positions_kernel" if "eval_positions" in name else "mcts_playouts_kernel" if "mcts_playouts" in name else "trad_playouts_kernel" if "trad_playouts" in name else "rave_playouts_kernel" if "rave_playouts" in name else "evalstate_update_kernel" if "evalstate_update" in name else None
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        for c, v in agg[k].items():
            if k == "trad_playouts_kernel":                    # bench.py launches a short warm-up search first: keep the two full searches
                v = sorted(v)[-2:]
            if k == "rave_playouts_kernel":                    # likewise: the three full searches
                v = sorted(v)[-3:]
            if k == "evalstate_update_kernel":                 # per PAIR of launches (apply, then revert), as bench.py counts its updates
                vals[k][c] = 2 * sum(v) / len(v)
                continue
            vals[k][c] = sum(v) / len(v)
print("# rocprofv3 --pmc, one pass per counter group (separate runs), command: python3 bench.py --steps 5 --warmup 2 --mcts-reps 1 --no-cpu-baseline --az-games 0")
print("# FETCH_SIZE / WRITE_SIZE are in KB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads half the bytes of wide coalesced reads (x2 below);")
print("# narrower accesses are uncalibrated (the 8-byte node reads of the MCTS kernel are given uncorrected and corrected).")
for k, unit, per in (("eval_positions_kernel", "board", n_boards), ("mcts_playouts_kernel", "playout", n_playouts), ("trad_playouts_kernel", "playout", n_trad), ("rave_playouts_kernel", "playout", n_rave),
                     ("evalstate_update_kernel", "update", n_evs)):
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
    print("VALU share of a wave's time: %.1f%% (x resident waves per SIMD = VALU busy); branches per %s: %.1f" %
          (100 * v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], unit, v.get("SQ_INSTS_BRANCH", 0) / per))
    if "SQ_THREAD_CYCLES_VALU" in v:
        # SQ_THREAD_CYCLES_VALU counts active lanes x cycles, SQ_ACTIVE_INST_VALU wave cycles (both in quad-cycles): their quotient / 64 = the share of VALU lanes doing work
        print("VALU lane utilisation: %.1f%% (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)): the share of the 64 lanes that were active in the average vector instruction"
              % (100 * v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"])))
    if v.get("SQ_INSTS_VALU_MFMA_I8"):
        print("matrix cores: %.1f v_mfma_i32_32x32x32_i8 per %s, busy %.0f cycles per %s" % (v["SQ_INSTS_VALU_MFMA_I8"] / per, unit, v["SQ_VALU_MFMA_BUSY_CYCLES"] / per, unit))
