#!/bin/bash
# round-2 measurement batch B: MEMCPY-free graph test, gather cache-policy sweep, fixed-leg kernel trace, gloo rehearsal
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_graph_step_gpu.py tests/test_idr_step_gpu.py -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -3 $O/pytest.log
for cfg in C2 C4; do for fl in 0 1 2 3 4 5; do
  HM_ENCODE_FLAGS=$fl timeout -k 10 120 python bench.py --only gather --cfg $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$cfg flags=$fl', d['achieved'], d['avg_launch_ms'], d['min_launch_ms'])" | tee -a $O/gather_sweep.log
done; done
for g in 256 768 1024; do HM_ENCODE_GRID=$g timeout -k 10 120 python bench.py --only gather --cfg C4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('C4 grid=$g', d['achieved'], d['avg_launch_ms'])" | tee -a $O/gather_sweep.log; done
HM_ENCODE_SWEEP=0 timeout -k 10 120 python bench.py --only gather --cfg C4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('C4 tile-kernel', d['achieved'], d['avg_launch_ms'])" | tee -a $O/gather_sweep.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fixed -- python bench.py --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof_fixed.log 2>&1; echo "prof rc=$?"
tail -1 $O/prof_fixed.log | cut -c1-300
for n in 2 4; do
  HM_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 8 --warmup 3 --no-extras > $O/gloo_n$n.log 2>&1; echo "gloo n=$n rc=$?"
  tail -1 $O/gloo_n$n.log | cut -c1-1500
done
