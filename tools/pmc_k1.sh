#!/bin/bash
# K1-only PMC passes (one rocprofv3 run per counter group; never combined with trace domains).  Output: gpurun_out/pmc_k1/
# usage: tools/pmc_k1.sh [eval_time.py variant, default all]
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
variant=${1:-all}
out=gpurun_out/pmc_k1_$variant; rm -rf $out; mkdir -p $out
export GMK_EVAL_REPS=10
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp -d $out/pmc_$i -o p --output-format csv -- python3 tools/eval_time.py $variant > $out/run_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/run_$i.log; }
done
python3 - $out <<'PY'
import csv, glob, collections, sys
v = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "eval_positions" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(x) / len(x) for k, x in v.items()}
n = 65536
for k in sorted(m): print("%-28s %.6g  per board %.1f" % (k, m[k], m[k] / n))
PY
find $out -name "*counter_collection.csv" -size +2M -delete
find $out -name "*agent_info.csv" -delete
