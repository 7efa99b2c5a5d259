#!/bin/bash
# round 3, run bb: edge cases of the scan + secant launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bb; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_raytrace_gpu.py -k "one_launch" -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log | cut -c1-200; grep "^FAILED\|^E  " $O/pytest.log | head -20
