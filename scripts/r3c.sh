#!/bin/bash
# round 3, run c: full GPU suite + full bench line (with the split legs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_gpu.log | cut -c1-300
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench.log > $O/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c/bench_line.json'))
print('headline', d['value'], d['ms_per_step'])
for k in ('train_leg','lazy_sampler_leg','split_f16x2_leg','config4_leg','config3_leg','config5_leg','config5_leg_plain_bf16'):
    if k in d: print(k, d[k]['value'], d[k]['ms_per_step'])
for k in d:
    if k.startswith('roofline'): print(k, d[k].get('achieved'), d[k].get('frac'), d[k].get('avg_launch_ms'))
print('cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
PY
