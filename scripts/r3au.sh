#!/bin/bash
# round 3, run au: fused camera-rays kernel - its test (the A/B of the first pass: C2 10.11 -> 10.00 ms)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3au; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_raytrace_gpu.py -k camera -m gpu -q -s > $O/pytest2.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest2.log | cut -c1-200; grep "^FAILED" $O/pytest2.log; grep "camera_rays:" $O/pytest2.log
