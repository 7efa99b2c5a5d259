#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep "bf16\|loss curves\|passed\|failed\|Error" $O/pytest.log | cut -c1-400 | head -20
