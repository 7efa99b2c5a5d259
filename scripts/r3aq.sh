#!/bin/bash
# round 3, run aq: 600 run-ahead steps of the train / lazy / fixed legs with the final code (persistent march tail and
# secant kernels under moving weights), and the same with the tail kernel carrying whole marches (HM_TRACE_TAIL_FIRST=1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3aq; mkdir -p $O
for leg in train lazy fixed; do
  timeout -k 10 300 python bench.py --cfg C2 --legs $leg --no-extras --steps 600 --warmup 5 2>$O/$leg.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$leg 600', d['ms_per_step'], d['value'], d['final_loss'], d['config']['sdf_evals_per_step'])"
done
HM_TRACE_TAIL_FIRST=1 timeout -k 10 300 python bench.py --cfg C2 --legs train --no-extras --steps 600 --warmup 5 2>$O/tail1.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train 600, whole marches in the tail kernel', d['ms_per_step'], d['value'], d['final_loss'], d['config']['sdf_evals_per_step'])"
timeout -k 10 300 python bench.py --cfg C4 --legs train --no-extras --steps 300 --warmup 5 2>$O/c4.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 train 300', d['ms_per_step'], d['value'], d['final_loss'], d['config']['sdf_evals_per_step'])"
