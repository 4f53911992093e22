#!/bin/bash
# K2 per-phase figures (profiling aid): time per update and LDS / VALU counters of evalstate_update_kernel with subsets of update_move's
# phases enabled (GMK_EVS_PHASE_MASK: 1 match, 2 compounds -, 4 patterns -, 8 7x7 block, 16 patterns +, 32 compounds +; the stone itself is
# always placed, so the boards evolve as in the full run).  The states are wrong unless the mask is 63.
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
out=gpurun_out/pmc_k2p; rm -rf $out; mkdir -p $out
export GMK_HIP_LIB=prof          # the phase masks live in the profiling flavour of the library only
for m in ${MASKS:-1 21 29 31 63}; do
  t=$(GMK_EVS_PHASE_MASK=$m timeout -k 10 120 python3 tools/evalstate_time.py | tail -1 | sed 's/.*-> //')
  GMK_EVS_PHASE_MASK=$m timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/m_$m -o p --output-format csv -- python3 tools/evalstate_time.py > $out/run_$m.log 2>&1 || { echo "mask $m failed"; tail -5 $out/run_$m.log; exit 1; }
  python3 - $m "$t" <<'PY'
import csv, glob, collections, sys
m = sys.argv[1]
v = collections.defaultdict(list)
for p in glob.glob("gpurun_out/pmc_k2p/m_%s/**/*counter_collection.csv" % m, recursive=True):
    for r in csv.DictReader(open(p)):
        if "evalstate_update" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
upd = None                      # updates per launch (apply or revert), as tools/evalstate_time.py printed it for THIS run (its game count has changed before)
for line in open("gpurun_out/pmc_k2p/run_%s.log" % m):
    if line.startswith("K2-UPDATES-PER-LAUNCH"):
        upd = float(line.split()[1])
assert upd, "tools/evalstate_time.py did not print its update count"
print("mask %3s  %s | per update: " % (m, sys.argv[2]) + " ".join("%s %.1f" % (k[3:], sum(x) / len(x) / upd) for k, x in sorted(v.items())))
PY
  rm -rf $out/m_$m
done
