#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2r; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_encode_gpu.py tests/test_idr_step_gpu.py -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep "z-ordered\|mismatches\|passed\|failed\|Error\|assert" $O/pytest.log | cut -c1-260 | head -40
for cfg in C2 C4; do timeout -k 10 200 python bench.py --only gather_bwd --cfg $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg bwd', d['achieved'], d['avg_launch_ms'], 'all corners', d['all_corners']['achieved'], d['all_corners']['avg_launch_ms'])"; done
