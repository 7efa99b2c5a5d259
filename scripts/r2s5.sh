#!/bin/bash
# round 2, run s: kernel-trace profile of the filter-bank (config 3) fixed-weights step
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2s
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o c5 --output-format csv -- python bench.py --cfg C5 --legs fixed --no-extras --steps 8 --warmup 3 > $OUT/c5.log 2>&1
tail -2 $OUT/c5.log
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c5_kernel_stats.csv
rm -rf $OUT/prof
