mkdir -p gpurun_out/r5g
timeout 1800 python3 -m pytest tests -m gpu -q -x > gpurun_out/r5g/pytest.log 2>&1; tail -5 gpurun_out/r5g/pytest.log
{
for w in cfgL cfgL-chrom chrom-10k cfgL-32k hap-16M chr-like x16-16Mseg-chrom; do
  for rep in 1 2; do
    FLATGFA_PAIR_RECORDS=0 python3 tools/ab_kernels.py $w 2>/dev/null | tail -1
    python3 tools/ab_kernels.py $w 2>/dev/null | tail -1
  done
done
} > gpurun_out/r5g/ab.log 2>&1
cat gpurun_out/r5g/ab.log
