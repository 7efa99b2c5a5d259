#!/bin/bash
# round 3, run ai: buffer-load weight stream in the split-operand and bf16 SDF kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ai; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_split_gpu.py tests/test_bf16_gpu.py tests/test_idr_step_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-250
timeout -k 10 200 python bench.py --only mlp_split --split f16x2 | cut -c1-300
timeout -k 10 200 python bench.py --only mlp_split --split bf16x2 | cut -c1-300
timeout -k 10 200 python bench.py --only mlp_bf16 | cut -c1-300
timeout -k 10 200 python bench.py --cfg C5 --legs fixed --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5', d['ms_per_step'], d['value'])"
timeout -k 10 200 python bench.py --cfg C2 --split f16x2 --legs fixed --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2 f16x2', d['ms_per_step'], d['value'])"
