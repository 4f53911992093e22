#!/bin/bash
# K1 phase ablation (profiling aid): kernel time with subsets of phases enabled.  Results are wrong unless mask = 63.
for m in 63 1 33 3 7 15 31 55 47 59 61; do
  echo -n "mask $m: "
  GMK_EVAL_PHASE_MASK=$m timeout -k 10 120 python bench.py --mcts-games 0 --no-cpu-baseline --steps 100 | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline']['kernel_ms'])" || exit 1
done
