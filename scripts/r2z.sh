#!/bin/bash
# round 2, run z: long run-ahead replays (no per-step sync) of all three legs with the final code:
# 600 steps each of fixed / train / lazy; tracer counters must stay sane (unfinished 0, non-finite 0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2z; mkdir -p $O
timeout -k 10 500 python bench.py --no-extras --steps 600 --warmup 5 > $O/bench600.log 2>&1; echo "rc=$?"
tail -1 $O/bench600.log | python -c "
import sys,json; d=json.loads(sys.stdin.read())
for k,l in (('fixed',d),('train',d['train_leg']),('lazy',d['lazy_sampler_leg'])):
    st=l['config']['sdf_evals_per_step'] if k=='fixed' else l['sdf_evals_per_step']
    print(k, l['ms_per_step'], l['value'], 'evals', st['mean'], st['min'], st['max'], 'unfinished', st['unfinished_max'], 'nonfinite', st['nonfinite_sdf_max'], 'loss', l['final_loss'])"
grep -c "Memory access fault" $O/bench600.log
