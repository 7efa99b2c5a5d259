#!/bin/bash
# round 3, run k: weight-ring depth of the 16-point SDF body (kRing16 = 4 / 5 / 6): small-batch latency + step time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k; mkdir -p $O
for lib in default build/exp/libhashmod_ring5.so build/exp/libhashmod_ring6.so; do
  if [ "$lib" != default ]; then export HM_LIB_PATH=$GRAFT_REPO_ROOT/$lib; else unset HM_LIB_PATH; fi
  timeout -k 10 120 python scripts/sdf_bench_small.py 2>/dev/null | grep -E "tile=16|n=  1024 tile= 8" | tee -a $O/small.log
  timeout -k 10 200 python bench.py --legs fixed --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib step', d['ms_per_step'])" | tee -a $O/small.log
done
