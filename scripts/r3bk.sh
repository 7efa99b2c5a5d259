#!/bin/bash
# round 3, run bk: final state after the DPP corner sums - smoke, full GPU suite, default bench line, kernel traces of the gather (C2 / C4) and of the C2 step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bk; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_gpu.log | cut -c1-200; grep "^FAILED" $O/pytest_gpu.log
timeout -k 10 400 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log > $O/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3bk/bench_line.json'))
print('headline', d['value'], d['ms_per_step'])
for k in ('train_leg','lazy_sampler_leg','split_f16x2_leg','config4_leg','config3_leg','config5_leg','config5_leg_plain_bf16'):
    if k in d: print(k, d[k]['value'], d[k]['ms_per_step'])
for k in d:
    if k.startswith('roofline'): print(k, d[k].get('achieved'), d[k].get('frac'), d[k].get('avg_launch_ms'))
PY
for cfg in C2 C4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gather_$cfg -- python bench.py --only gather --cfg $cfg > $O/prof_gather_$cfg.log 2>&1; echo "prof gather $cfg rc=$?"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_C2 -- python bench.py --cfg C2 --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof_C2.log 2>&1; echo "prof C2 rc=$?"
