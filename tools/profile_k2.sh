#!/bin/bash
# K2 alone (evalstate_update_kernel): kernel stats and the HBM byte counters of bench.py's `incremental` leg.  Usage on the GPU box: tools/profile_k2.sh <tag>
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
tag=${1:-r02x}
out=gpurun_out/prof_k2_$tag; rm -rf $out; mkdir -p $out
B="bench.py --steps 5 --warmup 1 --no-cpu-baseline --selfplay-games 0 --az-games 0 --trad-games 0 --rave-games 0 --mcts-games 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $B > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c -d $out/pmc_$c -o p --output-format csv -- python3 $B > $out/pmc_$c.log 2>&1 || { echo "$c failed"; tail -5 $out/pmc_$c.log; exit 1; }
done
python3 - $out <<'PY' | tee $out/summary.txt
import collections, csv, glob, sys
out = sys.argv[1]
for r in csv.DictReader(open(out + "/kernel_stats.csv")):
    if "evalstate_update" in r["Name"]:
        print("evalstate_update_kernel: %s launches, %.1f us average (rocprofv3 --kernel-trace --stats)" % (r["Calls"], float(r["AverageNs"]) / 1e3))
v = collections.defaultdict(list)
for p in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "evalstate_update" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(v):
    print("%s per launch: %.1f KB (mean of %d launches)" % (k, sum(v[k]) / len(v[k]), len(v[k])))
PY
