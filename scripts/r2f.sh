#!/bin/bash
# round-2 batch F: z-ordered gather (tests, A/B, PMC), deterministic ClipAdam + 2-rank gloo GPU test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_encode_gpu.py tests/test_optim_gpu.py tests/test_distributed_gpu.py -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -5 $O/pytest.log | cut -c1-400
grep "rank-0\|largest\|max param" $O/pytest.log | cut -c1-600
g() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['achieved'], d['avg_launch_ms'], d['min_launch_ms'])" | tee -a $O/gather_sweep.log; }
for cfg in C2 C4; do
  HM_ENCODE_ZORDER=0 timeout -k 10 120 python bench.py --only gather --cfg $cfg 2>/dev/null | g "$cfg sweep"
  for zg in 256 512 1024; do HM_ENCODE_ZGRID=$zg timeout -k 10 120 python bench.py --only gather --cfg $cfg 2>/dev/null | g "$cfg zorder grid=$zg"; done
done
pmc() { local name=$1; local ctr=$2; shift 2
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_$name -- "$@" > $O/pmc_$name.log 2>&1
  echo "== $name [$ctr]"; python scripts/pmc_summary.py $O/pmc_$name encode_fwd zsort | cut -c1-1500; }
for cfg in C2 C4; do
  pmc gather_${cfg}_fetch FETCH_SIZE python bench.py --only gather --cfg $cfg
  pmc gather_${cfg}_write WRITE_SIZE python bench.py --only gather --cfg $cfg
  pmc gather_${cfg}_rdreq "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum" python bench.py --only gather --cfg $cfg
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gather_C4 -- python bench.py --only gather --cfg C4 > $O/prof_gather_C4.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r2f/prof_gather_C4/**/*kernel_stats.csv',recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
