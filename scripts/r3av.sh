#!/bin/bash
# round 3, run av: where the remaining ATen launches of the static step come from (C2, C5)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3av; mkdir -p $O
timeout -k 10 300 python scripts/glue_sources.py C2 > $O/glue_C2.log 2>&1; echo "C2 rc=$?"
timeout -k 10 300 python scripts/glue_sources.py C5 > $O/glue_C5.log 2>&1; echo "C5 rc=$?"
tail -5 $O/glue_C2.log
