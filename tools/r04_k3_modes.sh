set -e
for mode in "" "--slots 8192 --handles 1" "--reuse" "--reuse --noise"; do
  echo "== 32768 games 800 playouts $mode" >> gpurun_out/r04b_modes.txt
  python tools/selfplay_bench.py --games 32768 --playouts 800 $mode >> gpurun_out/r04b_modes.txt 2>&1
done
