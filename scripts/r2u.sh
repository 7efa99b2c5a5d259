#!/bin/bash
# round 2, run u: lazy sampler - tracer / step / graph tests, then the fixed-weights leg with and without it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2u; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_raytrace_gpu.py tests/test_idr_step_gpu.py tests/test_graph_step_gpu.py tests/test_bf16_gpu.py tests/test_nffb_gpu.py -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep "lazy sampler\|passed\|failed\|Error" $O/pytest.log | cut -c1-400 | tail -20
timeout -k 10 300 python bench.py --no-extras --steps 40 --warmup 5 2>/dev/null | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print('fixed', d['ms_per_step'], d['value'], d['config']['sdf_evals_per_step']['per_ray_mean'], 'train', d.get('train_leg',{}).get('ms_per_step'), 'lazy', d['lazy_sampler_leg']['ms_per_step'], d['lazy_sampler_leg']['value'], d['lazy_sampler_leg']['sdf_evals_per_step']['per_ray_mean'])"
