#!/bin/bash
# VALU / SALU / LDS wave-instructions per board of K1 with phases 0..p enabled (profiling aid).
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
out=gpurun_out/pmc_k1p; rm -rf $out; mkdir -p $out
for m in ${MASKS:-1 3 7 15 31 63}; do
  GMK_EVAL_PHASE_MASK=$m timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $out/m_$m -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --mcts-games 0 --no-cpu-baseline > $out/run_$m.log 2>&1 || { echo "mask $m failed"; tail -5 $out/run_$m.log; exit 1; }
  python3 - $m <<'PY'
import csv, glob, collections, sys
m = sys.argv[1]
v = collections.defaultdict(list)
for p in glob.glob("gpurun_out/pmc_k1p/m_%s/**/*counter_collection.csv" % m, recursive=True):
    for r in csv.DictReader(open(p)):
        if "eval_positions" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("mask", m, " ".join("%s %.1f" % (k[3:], sum(x) / len(x) / 65536) for k, x in sorted(v.items())))
PY
done
