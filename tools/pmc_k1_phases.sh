#!/bin/bash
# K1 per-phase figures (profiling aid): kernel time and VALU / SALU / LDS wave-instructions per board with subsets of the phases
# enabled (GMK_EVAL_PHASE_MASK: 1 phase 0, 2 scan, 4 deposits, 8 phase 3, 16 rescans, 32 score stores, 64 phase D;
# 512 density passes without their stores, 1024 no density passes, 2048 no compounds = phases 3b and 4 skipped).  Results are wrong unless the mask is 127.
# COUNTERS="..." picks another counter group (e.g. the LDS ones: SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS).
cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
out=gpurun_out/pmc_k1p; rm -rf $out; mkdir -p $out
export GMK_EVAL_REPS=20 GMK_HIP_LIB=prof          # the phase masks live in the profiling flavour of the library only
for m in ${MASKS:-1 3 7 15 31 63 127 1151 639}; do
  t=$(GMK_EVAL_PHASE_MASK=$m GMK_EVAL_REPS=100 timeout -k 10 120 python3 tools/eval_time.py all | tail -1)
  GMK_EVAL_PHASE_MASK=$m timeout -k 10 240 rocprofv3 --pmc ${COUNTERS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU} -d $out/m_$m -o p --output-format csv -- python3 tools/eval_time.py all > $out/run_$m.log 2>&1 || { echo "mask $m failed"; tail -5 $out/run_$m.log; exit 1; }
  python3 - $m "$t" <<'PY'
import csv, glob, collections, sys
m = sys.argv[1]
v = collections.defaultdict(list)
for p in glob.glob("gpurun_out/pmc_k1p/m_%s/**/*counter_collection.csv" % m, recursive=True):
    for r in csv.DictReader(open(p)):
        if "eval_positions" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("mask %5s  %s  " % (m, sys.argv[2]) + " ".join("%s %.1f" % (k[3:], sum(x) / len(x) / 65536) for k, x in sorted(v.items())))
PY
  rm -rf $out/m_$m
done
