#!/bin/bash
# round 3, run y: shared weight-gradient accumulators of the two MLP nodes + one dense table gradient per step (LocalTableGrad)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3y; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_graph_step_gpu.py tests/test_idr_step_gpu.py tests/test_distributed_gpu.py tests/test_nffb_gpu.py tests/test_sdf_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-250
for cfg in C2 C4 C3 C5; do
  timeout -k 10 200 python bench.py --cfg $cfg --legs fixed --no-extras --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])"
done
HM_LOCAL_TABLE_GRAD=0 timeout -k 10 200 python bench.py --cfg C4 --legs fixed --no-extras --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 without LocalTableGrad', d['ms_per_step'], d['value'])"
