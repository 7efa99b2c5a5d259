#!/bin/bash
# round 3, run ae: 64-point SDF kernel with the B fragments of the next octet requested before the current octet's MFMAs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ae; mkdir -p $O
HM_LIB_PATH=scripts/libhashmod_probe.so python scripts/sdf_phase_probe.py 2>/dev/null | tail -12
timeout -k 10 900 python -m pytest tests/test_sdf_gpu.py tests/test_raytrace_gpu.py tests/test_nffb_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-250
timeout -k 10 200 python bench.py --only mlp | cut -c1-330
for cfg in C2 C3; do
  timeout -k 10 200 python bench.py --cfg $cfg --legs fixed --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])"
done
