#!/bin/bash
# round 3, run ad: the pieces of the record run r3ac that its silence kill cut off - kernel trace of the world-1 RCCL step,
# the self-spawned 2-rank rehearsal (gloo: both ranks on the one GPU), the re-toleranced table-backward test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ad; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_encode_gpu.py tests/test_distributed_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log | cut -c1-200
export HM_DIST_FORCE=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rccl1 -- python bench.py --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof_rccl1.log 2>&1; echo "prof rccl1 rc=$?"
unset HM_DIST_FORCE
timeout -k 10 300 python bench.py --gpus 2 --legs fixed --no-extras --steps 10 --warmup 3 > $O/gloo2.log 2>&1; echo "gloo2 rc=$?"; tail -1 $O/gloo2.log | cut -c1-330
