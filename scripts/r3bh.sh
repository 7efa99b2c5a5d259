#!/bin/bash
# round 3, run bh: the GPU suite three more times (flake check of the final code)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bh; mkdir -p $O
for i in 1 2 3; do
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:randomly > $O/pytest_$i.log 2>&1; echo "run $i rc=$?"; tail -1 $O/pytest_$i.log | cut -c1-200; grep "^FAILED" $O/pytest_$i.log
done
