#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2k; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_distributed_gpu.py -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep "rank-0\|largest\|max param\|passed\|failed\|Error" $O/pytest.log | cut -c1-300 | head
for sp in 1 0; do
HM_DP_SPARSE=$sp HM_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 6 --warmup 3 --no-extras --cfg C4 --legs fixed > $O/gloo_n2_sparse$sp.log 2>&1; echo "gloo C4 n=2 sparse=$sp rc=$?"
tail -1 $O/gloo_n2_sparse$sp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['final_loss'], d['config']['sdf_evals_per_step']['mean'])"
done
