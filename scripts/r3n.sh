#!/bin/bash
# round 3, run n: matrix-core filter-bank embedder for the tracer's small rounds too (HM_NFFB_MFMA=2) vs the 32-lane VALU kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 1 2; do for cfg in C3 C5; do
  HM_NFFB_MFMA=$m timeout -k 10 200 python bench.py --cfg $cfg --legs fixed --no-extras --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mfma=$m $cfg', d['ms_per_step'], d['value'])"
done; done
