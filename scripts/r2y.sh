#!/bin/bash
# round 2, run y: native-unit Softplus epilogues - parity tests, GEMM shapes, step legs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2y; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gemm_ep_gpu.py tests/test_sdf_gpu.py tests/test_idr_step_gpu.py tests/test_graph_step_gpu.py tests/test_nffb_gpu.py tests/test_loss_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -2 $O/pytest.log
timeout -k 10 200 python bench.py --only gemm 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('gemm', d['achieved'], [ (p['M'],p['K'],p['us']) for p in d['shapes']])"
timeout -k 10 300 python bench.py --no-extras --steps 40 --warmup 5 2>/dev/null | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print('fixed', d['ms_per_step'], d['value'], 'train', d['train_leg']['ms_per_step'], 'lazy', d['lazy_sampler_leg']['ms_per_step'], d['lazy_sampler_leg']['value'])"
