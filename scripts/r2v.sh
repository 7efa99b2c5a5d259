#!/bin/bash
# round 2, run v: step time + kernel-trace launch counts after the fused embedding-row input gradient
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2v; mkdir -p $O
timeout -k 10 300 python bench.py --no-extras --steps 40 --warmup 5 2>/dev/null | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print('fixed', d['ms_per_step'], d['value'], 'train', d['train_leg']['ms_per_step'], 'lazy', d['lazy_sampler_leg']['ms_per_step'], d['lazy_sampler_leg']['value'])"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o step --output-format csv -- python bench.py --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/step_kernel_stats.csv
rm -rf $O/prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r2v/step_kernel_stats.csv')))
steps=13
tr=('sdf_fwd','trace_','secant','sampler','closest','ray_samples','tail_prepare')
tot=0; gem=0; gemms=0
for r in rows:
    n=int(r['Calls'])/steps
    if any(t in r['Name'] for t in tr): continue
    tot+=n
    if 'gemm_f32' in r['Name']: gem+=n; gemms+=float(r['TotalDurationNs'])/1e6/steps
print('non-tracer launches/step', round(tot,1), 'gemm launches', round(gem,1), 'gemm ms', round(gemms,3))
PY
