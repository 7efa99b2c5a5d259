#!/bin/bash
# round 3, run am: the default bench twice more (run-to-run spread of the side legs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3am; mkdir -p $O
for i in 1 2; do
timeout -k 10 400 python bench.py > $O/bench$i.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench$i.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'])
for k in ('config3_leg','config5_leg','config5_leg_plain_bf16','config4_leg','split_f16x2_leg','lazy_sampler_leg','train_leg'): print(k, d[k]['ms_per_step'])
print(d['roofline_mlp']['achieved'], d['roofline_gemm']['achieved'], d['roofline']['achieved'])
"
done
