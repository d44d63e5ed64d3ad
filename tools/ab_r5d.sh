mkdir -p gpurun_out/r5d
{
python3 tools/ab_kernels.py cfgL 2>/dev/null | tail -1
AB_EXTRA_PLANS=1 python3 tools/ab_kernels.py cfgL 2>/dev/null | tail -1
AB_EXTRA_PLANS=2 python3 tools/ab_kernels.py cfgL 2>/dev/null | tail -1
AB_SIDE_STREAM=1 python3 tools/ab_kernels.py cfgL 2>/dev/null | tail -1
for k in 1 2; do
python3 bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --in-flight $k 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench in-flight $k', d['ms_per_step'], r['kernels_avg_ms'], r.get('cold',{}).get('kernels_avg_ms'), r['whole_call'].get('timed_region'))"
python3 bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-verify --in-flight $k 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench no-verify in-flight $k', d['ms_per_step'], r['kernels_avg_ms'], r.get('cold',{}).get('kernels_avg_ms'), r['whole_call'].get('timed_region'))"
done
} > gpurun_out/r5d/ab.log 2>&1
cat gpurun_out/r5d/ab.log
