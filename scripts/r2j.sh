#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2j; mkdir -p $O
timeout -k 10 120 python bench.py --only mlp_bf16 2>/dev/null | tail -1 | tee $O/mlp_bf16.log
timeout -k 10 120 python bench.py --only mlp 2>/dev/null | tail -1 | cut -c1-200
for cfg in C5 C2; do
timeout -k 10 300 python bench.py --cfg $cfg --bf16 1 --steps 10 --warmup 3 --no-extras > $O/bench_${cfg}_bf16.log 2>&1; echo "$cfg bf16 rc=$?"; tail -1 $O/bench_${cfg}_bf16.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['sdf_evals_per_step'], d['train_leg']['ms_per_step'])"
done
