#!/bin/bash
# Same-box A/B of two versions of depth_fast.hip: tools/ab/<name>.hip against the tree's.
# usage (GPU box): bash tools/ab/ab.sh <old.hip> "<workloads>" [rounds]
cd $GRAFT_REPO_ROOT
OLD=$1; WLS=${2:-cfgL}; ROUNDS=${3:-2}
cp pollen_amd/csrc/depth_fast.hip /tmp/new.hip; cp pollen_amd/csrc/depth_fast.hpp /tmp/new.hpp; cp pollen_amd/csrc/depth_device.hip /tmp/new_dd.hip
one() {
  cp $2 pollen_amd/csrc/depth_fast.hip; cp ${2%.hip}.hpp pollen_amd/csrc/depth_fast.hpp; [ -f ${2%.hip}_dd.hip ] && cp ${2%.hip}_dd.hip pollen_amd/csrc/depth_device.hip; touch pollen_amd/csrc/*.hip pollen_amd/csrc/*.cpp
  make -C pollen_amd/csrc > /tmp/build.log 2>&1 || { echo "$1: build failed"; tail -3 /tmp/build.log; return; }
  for w in $WLS; do timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', '$w', d['ms_per_step'], d['bit_exact_vs_oracle'], d['roofline']['kernels_avg_ms'])"; done
}
for r in $(seq $ROUNDS); do one old tools/ab/$OLD; one new /tmp/new.hip; done
cp /tmp/new.hip pollen_amd/csrc/depth_fast.hip; cp /tmp/new.hpp pollen_amd/csrc/depth_fast.hpp; cp /tmp/new_dd.hip pollen_amd/csrc/depth_device.hip
