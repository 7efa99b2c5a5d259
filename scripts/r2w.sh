#!/bin/bash
# round 2, run w: SDF kernels with the embedding-input path as a template value - parity tests, kernel rooflines, legs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2w; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_sdf_gpu.py tests/test_nffb_gpu.py tests/test_raytrace_gpu.py tests/test_bf16_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -2 $O/pytest.log
for i in 1 2; do timeout -k 10 200 python bench.py --only mlp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mlp', d['achieved'], d['avg_launch_ms'])"; done
timeout -k 10 300 python bench.py --no-extras --steps 40 --warmup 5 2>/dev/null | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print('fixed', d['ms_per_step'], d['value'], 'train', d['train_leg']['ms_per_step'], 'lazy', d['lazy_sampler_leg']['ms_per_step'], d['lazy_sampler_leg']['value'])"
for c in C3 C5; do timeout -k 10 300 python bench.py --cfg $c --legs fixed --no-extras --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], d['value'])"; done
