#!/bin/bash
# round 3, run w: the static step's second SDF evaluation reuses the first one's MLP forward (mlp_grad._SdfMlpRows)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3w; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_graph_step_gpu.py tests/test_idr_step_gpu.py tests/test_sdf_gpu.py tests/test_distributed_gpu.py tests/test_nffb_gpu.py -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-250; grep "reuse of the ray rows" $O/pytest.log
for cfg in C2 C4 C3 C5; do
  timeout -k 10 200 python bench.py --cfg $cfg --legs fixed --no-extras --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])"
done
