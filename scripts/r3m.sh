#!/bin/bash
# round 3, run m: matrix-core filter-bank embedder for big batches: tests + config-3 / config-5 steps, A/B against the
# 8-lanes-per-point kernel (HM_NFFB_MFMA=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_nffb_gpu.py tests/test_idr_step_gpu.py -m gpu -q -x -k "nffb or filter_bank or embedder" -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-200; grep "matrix-core kernel vs" $O/pytest.log | cut -c1-200
for m in 0 1; do for cfg in C3 C5; do
  HM_NFFB_MFMA=$m timeout -k 10 200 python bench.py --cfg $cfg --legs fixed --no-extras --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mfma=$m $cfg', d['ms_per_step'], d['value'])"
done; done
