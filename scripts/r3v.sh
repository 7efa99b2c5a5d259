#!/bin/bash
# round 3, run v: level-batched grad path of the filter-bank embedders: tests, config-3 / config-5 steps, launch counts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3v; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_nffb_gpu.py tests/test_idr_step_gpu.py tests/test_bf16_gpu.py tests/test_split_gpu.py -m gpu -q -x -k "nffb or filter_bank or embedder or stylemod" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-250
for cfg in C3 C5; do
  timeout -k 10 200 python bench.py --cfg $cfg --legs fixed --no-extras --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -- python bench.py --cfg $cfg --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof_$cfg.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob('gpurun_out/r3v/prof_$cfg/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print('$cfg launches/step', sum(int(r['Calls']) for r in rows)/13, 'kernel ms/step', sum(float(r['TotalDurationNs']) for r in rows)/13/1e6)
PY
done
