#!/bin/bash
# round 3, run z: the sphere-tracing march as ONE persistent launch (a workgroup carries eight rays through all rounds)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_raytrace_gpu.py tests/test_graph_step_gpu.py tests/test_idr_step_gpu.py tests/test_nffb_gpu.py tests/test_bf16_gpu.py tests/test_split_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-250
for leg in fixed train; do
for p in 1 0; do
  HM_TRACE_PERSISTENT=$p timeout -k 10 200 python bench.py --cfg C2 --legs $leg --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 $leg persistent=$p', d['ms_per_step'], d['value'], d['config']['sdf_evals_per_step']['mean'])"
done
done
HM_TRACE_PERSISTENT=1 timeout -k 10 200 python bench.py --cfg C4 --legs fixed --no-extras --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 fixed', d['ms_per_step'], d['value'])"
