#!/bin/bash
# round 2, run aa: repeatability of the full bench line (three runs back to back)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2aa; mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 500 python bench.py > $O/bench$i.log 2>&1; tail -1 $O/bench$i.log > $O/bench_line$i.json
  python -c "
import json; d=json.load(open('$O/bench_line$i.json'))
print($i, d['value'], d['ms_per_step'], 'lazy', d['lazy_sampler_leg']['ms_per_step'], 'train', d['train_leg']['ms_per_step'], 'mlp', d['roofline_mlp']['achieved'], d['roofline_mlp']['avg_launch_ms'], d['roofline_mlp']['min_launch_ms'], 'gather', d['roofline']['achieved'], 'bf16', d['roofline_mlp_bf16']['achieved'], 'gemm', d['roofline_gemm']['achieved'], 'c3', d['config3_leg']['ms_per_step'], 'c5', d['config5_leg']['ms_per_step'], 'cpu', d['cpu_baseline']['value'])"
done
