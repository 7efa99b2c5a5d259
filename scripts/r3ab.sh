#!/bin/bash
# round 3, run ab: kernel trace of the C2 step after the half-tile schedule / shared accumulators / march tail / 96-row tiles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ab; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_C2 -- python bench.py --cfg C2 --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof_C2.log 2>&1; echo "prof rc=$?"
ls $O/prof_C2/*/ | head
