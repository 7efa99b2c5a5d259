#!/bin/bash
# round 3, run x: 64-point SDF kernel with 32-point half tiles for the remainder round (tail balance of the coarse scan)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3x; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_sdf_gpu.py tests/test_raytrace_gpu.py tests/test_graph_step_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-250
timeout -k 10 200 python bench.py --only mlp | cut -c1-400
for cfg in C2 C4 C3; do
  timeout -k 10 200 python bench.py --cfg $cfg --legs fixed --no-extras --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])"
done
