#!/bin/bash
# K1-only PMC passes (one rocprofv3 run per counter group; never combined with trace domains).  Output: gpurun_out/pmc_k1/
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
out=gpurun_out/pmc_k1; rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp -d $out/pmc_$i -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --mcts-games 0 --no-cpu-baseline > $out/run_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/run_$i.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
v = collections.defaultdict(list)
for p in glob.glob("gpurun_out/pmc_k1/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "eval_positions" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(x) / len(x) for k, x in v.items()}
n = 65536
for k in sorted(m): print("%-24s %.6g  per board %.1f" % (k, m[k], m[k] / n))
PY
