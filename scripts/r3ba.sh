#!/bin/bash
# round 3, run ba: full GPU suite + bench line with the scan + secant launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ba; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_gpu.log | cut -c1-200; grep "^FAILED" $O/pytest_gpu.log
timeout -k 10 400 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log > $O/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3ba/bench_line.json'))
print('headline', d['value'], d['ms_per_step'])
for k in ('train_leg','lazy_sampler_leg','split_f16x2_leg','config4_leg','config3_leg','config5_leg','config5_leg_plain_bf16'):
    if k in d: print(k, d[k]['value'], d[k]['ms_per_step'])
PY
