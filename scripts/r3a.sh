#!/bin/bash
# round 3, run a: new tests (C5-bf16, z-ordered gather vs oracle, static exchange), bench line, world-1 RCCL rehearsal,
# self-spawned 2-rank gloo rehearsal through bench.py's launcher
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "zordered or bf16 or distributed or static_exchange or idr_training_steps or nffb" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest_new.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; echo "bench rc=$?"
tail -1 $O/bench.log | cut -c1-1500
HM_DIST_FORCE=1 timeout -k 10 200 python bench.py --legs fixed --no-extras --steps 20 --warmup 5 > $O/rccl1.log 2>&1; echo "rccl1 rc=$?"
tail -1 $O/rccl1.log | cut -c1-600
timeout -k 10 300 python bench.py --gpus 2 --legs fixed --no-extras --steps 10 --warmup 3 > $O/gloo2.log 2>&1; echo "gloo2 rc=$?"
tail -1 $O/gloo2.log | cut -c1-600
