#!/bin/bash
# round 3, run i: packed-fp32 Softplus epilogue in every fused SDF kernel: tests + kernel timings + step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_sdf_gpu.py tests/test_raytrace_gpu.py tests/test_split_gpu.py tests/test_bf16_gpu.py tests/test_nffb_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-200
timeout -k 10 120 python bench.py --only mlp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32', d['avg_launch_ms'], d['achieved'], d['frac'])"
timeout -k 10 120 python bench.py --only mlp_bf16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16', d['avg_launch_ms'], d['achieved'])"
for k in f16x2 bf16x2; do timeout -k 10 120 python bench.py --only mlp_split --split $k 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$k', d['avg_launch_ms'], d['fp32_equivalent_TFLOP/s'])"; done
timeout -k 10 200 python bench.py --legs fixed --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['value'])"
