cd "${GRAFT_REPO_ROOT:-.}" && export TMPDIR=/tmp
out=gpurun_out/prof_k9; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace9 -o t --output-format csv -- python3 tools/pvnet_time.py 4096 > $out/pvnet_trace.log 2>&1 || exit 1
find $out/trace9 -name "*kernel_stats.csv" -exec cp {} $out/pvnet_kernel_stats.csv \;
rm -rf $out/trace9
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $out/pmc_k9 -o p --output-format csv -- python3 tools/pvnet_time.py 4096 > $out/pmc_k9.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c -d $out/pmc_$c -o p --output-format csv -- python3 tools/pvnet_time.py 4096 > $out/pmc_$c.log 2>&1 || exit 1
done
