#!/bin/bash
# round 3, run l: grouped weight-gradient GEMM: grad-path tests (pinned tolerances) + step time + GEMM share
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_sdf_gpu.py tests/test_idr_step_gpu.py tests/test_gemm_ep_gpu.py tests/test_graph_step_gpu.py tests/test_nffb_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-250
timeout -k 10 200 python bench.py --legs fixed --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['value'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -- python bench.py --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof_step.log 2>&1; echo "prof rc=$?"
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3l/prof_step/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
g=[(r['Name'][20:90],int(r['Calls'])/13,float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/13/1e6) for r in rows if 'gemm' in r['Name']]
for x in g: print(x)
print('gemm total ms/step', sum(x[3] for x in g), 'launches', sum(x[1] for x in g), 'all launches/step', sum(int(r['Calls']) for r in rows)/13)
PY
