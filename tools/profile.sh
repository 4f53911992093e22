#!/bin/bash
# Collects the rocprofv3 evidence for profiles/: kernel-trace stats of the default bench and PMC passes (one rocprofv3 run
# per counter group, never combined with trace domains).  Usage on the GPU box: tools/profile.sh <tag>   (e.g. r01c)
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
tag=${1:-r01x}
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
BENCH="bench.py --steps 50 --warmup 5 --no-cpu-baseline --mcts-reps 1 --az-games 0"   # K7 is the network (MIOpen kernels): not profiled here
echo "[profile] kernel trace"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $BENCH > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_BRANCH SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "[profile] pmc pass $i: $grp"
  timeout -k 10 400 rocprofv3 --pmc $grp -d $out/pmc_$i -o p --output-format csv -- python3 bench.py --steps 5 --warmup 2 --mcts-reps 1 --no-cpu-baseline --az-games 0 > $out/pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pmc_$i.log; exit 1; }
done
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/trace                                     # the per-dispatch trace is large (gpurun returns at most 64 MiB); the stats are what profiles/ keeps
for d in $out/pmc_*/; do find $d -name "*agent_info.csv" -delete; done
python3 tools/summarize_pmc.py $out 65536 3276800 1792000 1638400 > $out/pmc_summary.txt
find $out -name "*counter_collection.csv" -size +8M -delete
timeout -k 10 400 python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err
tail -c 600 $out/bench_n1.json
