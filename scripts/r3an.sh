#!/bin/bash
# round 3, run an: 16-byte loads in the sampler / closest-approach reductions, persistent secant - tests, step, kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3an; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_raytrace_gpu.py tests/test_graph_step_gpu.py tests/test_idr_step_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-250
for cfg in C2 C4; do
  timeout -k 10 200 python bench.py --cfg $cfg --legs fixed --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_C2 -- python bench.py --cfg C2 --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof_C2.log 2>&1; echo "prof rc=$?"
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3an/prof_C2/*/*_kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
print('launches/iter', sum(int(r['Calls']) for r in rows)/13)
for r in rows:
    if any(k in r['Name'] for k in ('reduce','secant','trace_','ray_samples','tail_prepare')): print(r['Name'][:60], int(r['Calls'])/13, round(float(r['AverageNs'])/1e3,1))
PY
