#!/bin/bash
# Collects the rocprofv3 evidence for profiles/: kernel-trace stats of the default bench and PMC passes (one rocprofv3 run
# per counter group, never combined with trace domains).  Usage on the GPU box: tools/profile.sh <tag>   (e.g. r01c)
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
tag=${1:-r01x}
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
BENCH="bench.py --steps 50 --warmup 5 --no-cpu-baseline --mcts-reps 1 --az-games 0 --mcts-saturated-games 0 --selfplay-games 0 --sup-games 0 --trad-saturated-games 0"   # K7 is the network (MIOpen kernels): not profiled here
echo "[profile] kernel trace"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $BENCH > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_BRANCH SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8"; do
  i=$((i+1))
  echo "[profile] pmc pass $i: $grp"
  timeout -k 10 400 rocprofv3 --pmc $grp -d $out/pmc_$i -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --mcts-reps 1 --no-cpu-baseline --az-games 0 --mcts-saturated-games 0 --selfplay-games 0 --sup-games 0 --trad-saturated-games 0 --settle-ms 0 > $out/pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pmc_$i.log; exit 1; }
done
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
# K9 (the fused network trunk) on its own: kernel stats and the matrix-core counters
echo "[profile] K9 kernel trace + pmc"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace9 -o t --output-format csv -- python3 tools/pvnet_time.py 4096 > $out/pvnet_trace.log 2>&1 || { echo "K9 trace failed"; tail -5 $out/pvnet_trace.log; exit 1; }
find $out/trace9 -name "*kernel_stats.csv" -exec cp {} $out/pvnet_kernel_stats.csv \;
rm -rf $out/trace9
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $out/pmc_k9 -o p --output-format csv -- python3 tools/pvnet_time.py 4096 > $out/pmc_k9.log 2>&1 || { echo "K9 pmc failed"; tail -5 $out/pmc_k9.log; exit 1; }
python3 - $out <<'PY' > $out/pvnet_pmc_summary.txt
import collections, csv, glob, sys
vals = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/pmc_k9/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "pvnet_trunk" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -- python3 tools/pvnet_time.py 4096")
print("## pvnet_trunk_kernel (mean per launch; one launch = 4096 positions, 256 workgroups x 4 waves)")
for k in sorted(vals):
    print("%-28s %.6g" % (k, sum(vals[k]) / len(vals[k])))
PY

rm -rf $out/trace                                     # the per-dispatch trace is large (gpurun returns at most 64 MiB); the stats are what profiles/ keeps
for d in $out/pmc_*/; do find $d -name "*agent_info.csv" -delete; done
python3 tools/summarize_pmc.py $out 65536 3276800 2048000 1638400 152808 > $out/pmc_summary.txt
find $out -name "*counter_collection.csv" -size +8M -delete
timeout -k 10 400 python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err
tail -c 600 $out/bench_n1.json
