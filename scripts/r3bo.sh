#!/bin/bash
# round 3, run bo: 1000 run-ahead steps of the fixed / train / lazy legs (C2) and 300 of C4 with the round's final code
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bo; mkdir -p $O
timeout -k 10 500 python bench.py --cfg C2 --legs both --no-extras --steps 1000 --warmup 10 > $O/c2.log 2>&1; echo "c2 rc=$?"
tail -1 $O/c2.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("fixed", d["ms_per_step"], d["config"]["sdf_evals_per_step"]); print("train", d["train_leg"]["ms_per_step"], d["train_leg"]["sdf_evals_per_step"]); print("lazy", d["lazy_sampler_leg"]["ms_per_step"], d["lazy_sampler_leg"]["sdf_evals_per_step"]); print("final losses", d["final_loss"], d["train_leg"]["final_loss"], d["lazy_sampler_leg"]["final_loss"])'
timeout -k 10 300 python bench.py --cfg C4 --legs fixed --no-extras --steps 300 --warmup 10 > $O/c4.log 2>&1; echo "c4 rc=$?"
tail -1 $O/c4.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("C4 fixed", d["ms_per_step"], d["config"]["sdf_evals_per_step"])'
