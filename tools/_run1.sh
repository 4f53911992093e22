mkdir -p gpurun_out/r03c
export GMK_HIP_LIB=prof
rm -f gpurun_out/r03c/masks2.txt
for m in 127 639 1151 95 63; do
  echo "mask $m" >> gpurun_out/r03c/masks2.txt
  GMK_EVAL_PHASE_MASK=$m timeout -k 10 100 python3 tools/eval_time.py all no-density neither >> gpurun_out/r03c/masks2.txt 2>&1
done
cat gpurun_out/r03c/masks2.txt
