#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2p; mkdir -p $O
for sp in 1 0; do
HM_DIST_FORCE=1 HM_DP_SPARSE=$sp RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-extras --legs fixed > $O/rccl1_sparse$sp.log 2>&1; echo "single-rank RCCL sparse=$sp rc=$?"; tail -2 $O/rccl1_sparse$sp.log | cut -c1-300
done
