#!/bin/bash
# round 3, run bj: corner sums of the gather kernels as single DPP adds - encode tests, gather rooflines at C2 / C4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3bj; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_encode_gpu.py tests/test_nffb_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log | cut -c1-200; grep "^FAILED" $O/pytest.log
for cfg in C2 C4; do
timeout -k 10 200 python bench.py --only gather --cfg $cfg > $O/g_$cfg.log 2>&1 && tail -1 $O/g_$cfg.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("'$cfg'", d["achieved"], d["frac"], d["avg_launch_ms"], d["min_launch_ms"])'
done
timeout -k 10 200 python bench.py --only gather_bwd --cfg C2 > $O/gb.log 2>&1 && tail -1 $O/gb.log | cut -c1-300
