#!/bin/bash
# round 3, run as: line search with all candidates in one round (21 rounds instead of 41) - tests, train-leg A/B, configs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3as; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_raytrace_gpu.py tests/test_graph_step_gpu.py tests/test_idr_step_gpu.py tests/test_nffb_gpu.py tests/test_bf16_gpu.py tests/test_split_gpu.py tests/test_distributed_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-250
for p in 1 0; do HM_TRACE_PERSISTENT=$p timeout -k 10 200 python bench.py --cfg C2 --legs train --no-extras --steps 600 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train 600 persistent=$p', d['ms_per_step'], d['value'], d['config']['sdf_evals_per_step']['mean'], d['config']['sdf_evals_per_step']['unfinished_max'])"; done
for cfg in C2 C4 C3 C5; do
  timeout -k 10 200 python bench.py --cfg $cfg --legs fixed --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'], d['config']['sdf_evals_per_step']['mean'])"
done
