#!/bin/bash
# round 3, run d: (1) 600 run-ahead steps of the world-1 RCCL step with StaticGradExchange (no per-step sync) - tracer
# counters must stay sane; (2) rocprofv3 kernel trace of that step; (3) the same for the plain single-GPU step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
export HM_DIST_FORCE=1
timeout -k 10 300 python bench.py --legs fixed --no-extras --steps 600 --warmup 5 > $O/rccl1_600.log 2>&1; echo "rccl1_600 rc=$?"
tail -1 $O/rccl1_600.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); st=d['config']['sdf_evals_per_step']
print('rccl1 600 steps:', d['ms_per_step'], 'ms', d['config']['exchange'], d['config']['backend'], 'evals', st['mean'], st['min'], st['max'], 'unfinished', st['unfinished_max'], 'nonfinite', st['nonfinite_sdf_max'], 'loss', d['final_loss'])"
grep -c "Memory access fault" $O/rccl1_600.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rccl1 -- python bench.py --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof_rccl1.log 2>&1; echo "prof rccl1 rc=$?"
unset HM_DIST_FORCE
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -- python bench.py --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof_step.log 2>&1; echo "prof step rc=$?"
find $O -name "*kernel_stats.csv" | head
