#!/bin/bash
# round 3, run bp: self-spawned 4-rank rehearsal (gloo: four ranks on the one GPU) of the final code, all default legs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bp; mkdir -p $O
timeout -k 10 600 python bench.py --gpus 4 --steps 5 --warmup 3 > $O/gloo4.log 2>&1; echo "gloo4 rc=$?"; tail -1 $O/gloo4.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["n_gpus"], d["value"], d["ms_per_step"], d["config"]["backend"], d["config"]["ranks_seen"], d["config"]["exchange"], {k: d[k]["ms_per_step"] for k in ("train_leg","lazy_sampler_leg","config4_leg") if k in d}, "roofline" in d)'
