#!/bin/bash
# round 3, run e: z-ordered gather with padded rows / carried slab ids: encode tests, timing at C2 / C4, kernel trace,
# PMC passes (FETCH_SIZE, WRITE_SIZE) at C2 and C4; world-1 exchange after the workgroup-aggregated tracked scatter
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_encode_gpu.py tests/test_distributed_gpu.py -m gpu -q -x > $O/pytest_enc.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_enc.log | cut -c1-300
for cfg in C2 C4; do
  timeout -k 10 120 python bench.py --only gather --cfg $cfg > $O/gather_$cfg.log 2>&1; tail -1 $O/gather_$cfg.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['achieved'], d['frac'], d['avg_launch_ms'], d['min_launch_ms'])"
  timeout -k 10 120 python bench.py --only gather_bwd --cfg $cfg > $O/gather_bwd_$cfg.log 2>&1; tail -1 $O/gather_bwd_$cfg.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg bwd', d['achieved'], d['frac'], d['avg_launch_ms'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gather_C2 -- python bench.py --only gather --cfg C2 > $O/prof_gather_C2.log 2>&1; echo "trace C2 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gather_C4 -- python bench.py --only gather --cfg C4 > $O/prof_gather_C4.log 2>&1; echo "trace C4 rc=$?"
pmc() { # name counters -- cmd...
  local name=$1; local ctr=$2; shift 2
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_$name -- "$@" > $O/pmc_$name.log 2>&1
  echo "== $name [$ctr]"; python scripts/pmc_summary.py $O/pmc_$name encode_fwd zsort | cut -c1-1200
}
for cfg in C2 C4; do
  pmc gather_${cfg}_fetch FETCH_SIZE python bench.py --only gather --cfg $cfg
  pmc gather_${cfg}_write WRITE_SIZE python bench.py --only gather --cfg $cfg
done
HM_DIST_FORCE=1 timeout -k 10 200 python bench.py --legs fixed --no-extras --steps 20 --warmup 5 > $O/rccl1.log 2>&1; echo "rccl1 rc=$?"
grep -o '"ms_per_step": [0-9.]*' $O/rccl1.log | head -1
timeout -k 10 200 python bench.py --legs fixed --no-extras --steps 20 --warmup 5 > $O/plain.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $O/plain.log | head -1
