#!/bin/bash
# round-2 batch C: full GPU suite with the p32 kernel + merged sampler launch, SDF kernel A/B, gather flag combos, bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -rf -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -4 $O/pytest.log
for p in 0 1; do HM_SDF_P32=$p timeout -k 10 120 python bench.py --only mlp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mlp p32=$p', d['achieved'], d['avg_launch_ms'])" | tee -a $O/mlp_ab.log; done
for cfg in C2 C4; do for fl in 2 6 8 10 14 13; do
  HM_ENCODE_FLAGS=$fl timeout -k 10 120 python bench.py --only gather --cfg $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$cfg flags=$fl', d['achieved'], d['avg_launch_ms'], d['min_launch_ms'])" | tee -a $O/gather_sweep.log
done; done
for g in 256 384; do for fl in 2 13 14; do HM_ENCODE_FLAGS=$fl HM_ENCODE_GRID=$g timeout -k 10 120 python bench.py --only gather --cfg C4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('C4 grid=$g flags=$fl', d['achieved'], d['avg_launch_ms'])" | tee -a $O/gather_sweep.log; done; done
timeout -k 10 280 python bench.py --no-extras > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-1200
