#!/bin/bash
# round 3, run bd: points per march round over 300 training steps (how many line-search tries a straggler takes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bd; mkdir -p $O
HM_TRACE_PERSISTENT=0 timeout -k 10 600 python scripts/round_counts.py 300 > $O/rounds.log 2>&1; echo "rc=$?"; grep "^step" $O/rounds.log | cut -c1-900
