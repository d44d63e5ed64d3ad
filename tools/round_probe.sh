#!/bin/bash
# (FLATGFA_DEBUG_SKIP / FLATGFA_ACC_SKIP / FLATGFA_ACC_PAIR / FLATGFA_ACC_SMALL exist in measurement builds only:
#  tools/variants.sh measure "-DFGFA_MEASURE" here, then FLATGFA_LIB=pollen_amd/lib_measure/libflatgfa.so on the GPU box)
# Quick look at a kernel change (run via gpurun): instruction mix, k_scan ablations, all workloads.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/prof_insts.sh cfgL > gpurun_out/probe_insts.log 2>&1
for m in 32 8 1; do
  echo "== FLATGFA_DEBUG_SKIP=$m" >> gpurun_out/probe_ablate.log
  FLATGFA_DEBUG_SKIP=$m timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-verify --no-extras 2>&1 | tail -4 | cut -c1-1500 >> gpurun_out/probe_ablate.log
done
for w in cfgL cfgL-uniform cfgL-short cfgL-fewlong cfgL-medium cfgL-4Mseg cfgS; do
  timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --workload $w 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d['bit_exact_vs_oracle'], d['roofline']['kernels_avg_ms'])" >> gpurun_out/probe_workloads.log 2>&1
done
cat gpurun_out/probe_workloads.log
