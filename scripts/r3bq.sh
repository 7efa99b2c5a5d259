#!/bin/bash
# round 3, run bq: the whole march inside the persistent kernel (HM_TRACE_TAIL_FIRST=1) against the default hybrid, final code
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bq; mkdir -p $O
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export HM_TRACE_TAIL_FIRST=1; else unset HM_TRACE_TAIL_FIRST; fi
  timeout -k 10 300 python bench.py --cfg C2 --legs both --no-extras --steps 200 --warmup 10 > $O/b_$v.log 2>&1 && tail -1 $O/b_$v.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("tail_first='$v'", "fixed", d["ms_per_step"], "train", d["train_leg"]["ms_per_step"], "lazy", d["lazy_sampler_leg"]["ms_per_step"])'
done
