#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
for cfg in C3 C5; do
timeout -k 10 300 python bench.py --cfg $cfg --steps 10 --warmup 3 --no-extras > $O/bench_$cfg.log 2>&1; echo "$cfg rc=$?"; tail -2 $O/bench_$cfg.log | cut -c1-1800
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_C3 -- python bench.py --cfg C3 --legs fixed --no-extras --steps 6 --warmup 3 > $O/prof_C3.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r2h/prof_C3/**/*kernel_stats.csv',recursive=True):
    rows=list(csv.DictReader(open(f))); tot=sum(float(r['TotalDurationNs']) for r in rows); print('total ms',tot/1e6)
    for r in rows[:16]: print(r['Name'][:80], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2), round(float(r['AverageNs'])/1e3,1))
PY
