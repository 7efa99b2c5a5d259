#!/bin/bash
# round 3, run az: kernel trace of the C2 step with the scan + secant launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3az; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_C2 -- python bench.py --cfg C2 --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof_C2.log 2>&1; echo "prof C2 rc=$?"
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3az/prof_C2/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
