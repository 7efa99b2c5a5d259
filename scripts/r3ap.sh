#!/bin/bash
# round 3, run ap: the full GPU suite twice more (stability of the pinned tolerances / chaotic loss-curve bounds with the final code)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ap; mkdir -p $O
for i in 1 2; do
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu$i.log 2>&1; echo "pytest $i rc=$?"; tail -1 $O/pytest_gpu$i.log | cut -c1-200; grep "^FAILED" $O/pytest_gpu$i.log
done
