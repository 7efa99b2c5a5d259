#!/bin/bash
# round 3, run b: split-operand kernels (tests + kernel timing), the rest of the new tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_split_gpu.py -q -s > $O/pytest_split.log 2>&1; echo "split rc=$?"
grep -E "^\[|passed|failed|StyleMod|fp32 :|bf16x2:" $O/pytest_split.log | cut -c1-330
timeout -k 10 120 python bench.py --only mlp_split --split f16x2 > $O/mlp_f16x2.log 2>&1; tail -1 $O/mlp_f16x2.log | cut -c1-700
timeout -k 10 120 python bench.py --only mlp_split --split bf16x2 > $O/mlp_bf16x2.log 2>&1; tail -1 $O/mlp_bf16x2.log | cut -c1-700
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "zordered or bf16 or distributed or static_exchange or idr_training_steps or nffb" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest_new.log | cut -c1-300
