#!/bin/bash
# round 3, run g: split kernel with cross-layer weight prefetch (bf16 kind)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_split_gpu.py -q -x > $O/pytest_split.log 2>&1; echo "split rc=$?"; tail -1 $O/pytest_split.log
for k in f16x2 bf16x2; do timeout -k 10 120 python bench.py --only mlp_split --split $k 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$k', d['avg_launch_ms'], d['fp32_equivalent_TFLOP/s'])"; done
