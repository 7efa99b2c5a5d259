#!/bin/bash
# round 3, run bm: time of the z-ordered gather with only some of its levels (probe builds, scripts/gather_level_probe.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bm; mkdir -p $O
timeout -k 10 900 python scripts/gather_level_probe.py > $O/levels.log 2>&1; echo "rc=$?"; cat $O/levels.log | grep levels
