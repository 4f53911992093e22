#!/bin/bash
# K1 write-path counters of the PRODUCTION library (profiling aid), one rocprofv3 run per group.  usage: tools/pmc_k1_prod.sh [out dir]
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
out=${1:-gpurun_out/pmc_k1_prod}; rm -rf $out; mkdir -p $out
export GMK_EVAL_REPS=10
i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_WRITE_sum TCC_WRITEBACK_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp -d $out/pmc_$i -o p --output-format csv -- python3 tools/eval_time.py all > $out/run_$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -3 $out/run_$i.log; }
done
python3 - $out <<'PY'
import csv, glob, collections, sys
v = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "eval_positions" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(v): print("%-44s %.6g   per board %.2f" % (k, sum(v[k]) / len(v[k]), sum(v[k]) / len(v[k]) / 65536))
PY
rm -rf $out/pmc_*
