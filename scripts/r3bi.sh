#!/bin/bash
# round 3, run bi: trilinear instantiations of the device tracer against the generic tracer; the re-bounded bf16 loss-curve test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bi; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_raytrace_gpu.py -k "trilinear" tests/test_bf16_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log | cut -c1-200; grep "^FAILED\|^E  " $O/pytest.log | head -20
