#!/bin/bash
# K1 memory-path counters (profiling aid), one rocprofv3 run per group.  usage: tools/pmc_k1_mem.sh [variant]
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
variant=${1:-all}
out=gpurun_out/pmc_k1_mem_$variant; rm -rf $out; mkdir -p $out
export GMK_EVAL_REPS=10 GMK_HIP_LIB=prof          # the phase masks live in the profiling flavour of the library only
i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum" "TCC_WRITE_sum TCC_WRITEBACK_sum TCC_WRITE_SECTORS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp -d $out/pmc_$i -o p --output-format csv -- python3 tools/eval_time.py $variant > $out/run_$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -3 $out/run_$i.log; }
done
python3 - $out <<'PY'
import csv, glob, collections, sys
v = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "eval_positions" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(v): print("%-44s %.6g" % (k, sum(v[k]) / len(v[k])))
PY
rm -rf $out/pmc_*
