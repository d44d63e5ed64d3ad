mkdir -p gpurun_out/r5e
timeout 1500 python3 -m pytest tests -m gpu -q > gpurun_out/r5e/pytest.log 2>&1; tail -5 gpurun_out/r5e/pytest.log
{
for w in cfgL cfgL-chrom; do
  for rep in 1 2; do
    FLATGFA_LIB=pollen_amd/lib_head/libflatgfa.so python3 tools/ab_kernels.py $w 2>/dev/null | tail -1
    python3 tools/ab_kernels.py $w 2>/dev/null | tail -1
  done
done
python3 tools/ab_kernels.py hap-16M 2>/dev/null | tail -1
for k in 1 2 3; do
python3 bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --in-flight $k 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench in-flight $k', d['ms_per_step'], r['kernels_avg_ms'], r.get('cold',{}).get('kernels_avg_ms'), r['whole_call'].get('timed_region'))"
done
python3 bench.py --steps 200 --warmup 3 --no-extras --no-cpu-baseline --in-flight 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench 200 steps in-flight 2', d['ms_per_step'], r['kernels_avg_ms'], r['whole_call'].get('timed_region'))"
} > gpurun_out/r5e/ab.log 2>&1
cat gpurun_out/r5e/ab.log
