#!/bin/bash
# round 3, run ag: buffer-load weight stream in the 16- / 8- / 4-point bodies too
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ag; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_sdf_gpu.py tests/test_raytrace_gpu.py tests/test_nffb_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-250
python scripts/small_tile_probe.py 2>/dev/null | grep "n="
for cfg in C2 C3 C5; do
  timeout -k 10 200 python bench.py --cfg $cfg --legs fixed --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])"
done
