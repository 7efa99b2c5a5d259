#!/bin/bash
# round 3, run bc: record run with the scan + secant launch and the filler tiles - smoke, full GPU suite, full bench line, kernel traces of the C2 / C3 / C5 steps, the world-1
# RCCL step (600 run-ahead steps + kernel trace), the self-spawned 2-rank rehearsal (gloo: both ranks on the one GPU)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bc; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_gpu.log | cut -c1-200; grep "^FAILED" $O/pytest_gpu.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log > $O/bench_line.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3bc/bench_line.json'))
print('headline', d['value'], d['ms_per_step'])
for k in ('train_leg','lazy_sampler_leg','split_f16x2_leg','config4_leg','config3_leg','config5_leg','config5_leg_plain_bf16'):
    if k in d: print(k, d[k]['value'], d[k]['ms_per_step'])
for k in d:
    if k.startswith('roofline'): print(k, d[k].get('achieved'), d[k].get('frac'), d[k].get('avg_launch_ms'))
print('cpu', d['cpu_baseline']['value'] if d.get('cpu_baseline') else None)
PY
for cfg in C2 C3 C5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -- python bench.py --cfg $cfg --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof_$cfg.log 2>&1; echo "prof $cfg rc=$?"
done
export HM_DIST_FORCE=1
timeout -k 10 300 python bench.py --legs fixed --no-extras --steps 600 --warmup 5 > $O/rccl1_600.log 2>&1; echo "rccl1_600 rc=$?"
grep -o '"ms_per_step": [0-9.]*' $O/rccl1_600.log | head -1; grep -o '"sdf_evals_per_step": {[^}]*}' $O/rccl1_600.log | head -1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rccl1 -- python bench.py --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof_rccl1.log 2>&1; echo "prof rccl1 rc=$?"
unset HM_DIST_FORCE
echo "gloo2 next"; timeout -k 10 300 python bench.py --gpus 2 --legs fixed --no-extras --steps 10 --warmup 3 > $O/gloo2.log 2>&1; echo "gloo2 rc=$?"; tail -1 $O/gloo2.log | cut -c1-330
