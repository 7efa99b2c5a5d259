#!/bin/bash
# round 3, run ak: pipelined GEMM with a K tail (K = 445, 257 of the backward sweeps) and dword-aligned k-contiguous operands
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ak; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gemm_ep_gpu.py tests/test_sdf_gpu.py tests/test_idr_step_gpu.py tests/test_graph_step_gpu.py tests/test_nffb_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-250
timeout -k 10 200 python bench.py --only gemm > $O/gemm.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3ak/gemm.json'))
print(d['achieved'], d['frac'])
for s in d['shapes']: print(s)
PY
for cfg in C2 C4 C3 C5; do
  timeout -k 10 200 python bench.py --cfg $cfg --legs fixed --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])"
done
