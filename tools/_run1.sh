mkdir -p gpurun_out/r03c
export GMK_HIP_LIB=prof
for m in 127 639 1151; do
  echo "mask $m" >> gpurun_out/r03c/masks.txt
  GMK_EVAL_PHASE_MASK=$m timeout -k 10 100 python3 tools/eval_time.py all >> gpurun_out/r03c/masks.txt 2>&1
  GMK_EVAL_PHASE_MASK=$m GMK_EVAL_PROFILE=1 GMK_EVAL_REPS=1 timeout -k 10 100 python3 tools/eval_time.py all 2>&1 | grep PROFILE | tail -1 >> gpurun_out/r03c/masks.txt
done
cat gpurun_out/r03c/masks.txt
