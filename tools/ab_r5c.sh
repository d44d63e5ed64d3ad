mkdir -p gpurun_out/r5c
for w in cfgL cfgL-chrom hap-16M; do
  for rep in 1 2; do
    FLATGFA_LIB=pollen_amd/lib_head/libflatgfa.so python3 tools/ab_kernels.py $w 2>/dev/null | tail -1
    python3 tools/ab_kernels.py $w 2>/dev/null | tail -1
    FLATGFA_NO_CLAIM=0 python3 tools/ab_kernels.py $w 2>/dev/null | tail -1
  done
done > gpurun_out/r5c/ab.log 2>&1
cat gpurun_out/r5c/ab.log
