#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_nffb_gpu.py tests/test_sdf_gpu.py tests/test_raytrace_gpu.py -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -5 $O/pytest.log | cut -c1-300; grep "fused " $O/pytest.log | head
