#!/bin/bash
# round-2 batch D: MEMCPY-free graph, prefetch64 A/B, no-sync long runs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -rf -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -4 $O/pytest.log
timeout -k 10 120 python bench.py --only mlp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mlp', d['achieved'], d['avg_launch_ms'])" | tee -a $O/mlp_ab.log
grep -q "rc=0" $O/pytest.rc || exit 0
HM_GRAPH_SYNC=0 timeout -k 10 280 python bench.py --no-extras --steps 600 --warmup 5 > $O/bench_nosync600.log 2>&1; echo "nosync rc=$?"; tail -1 $O/bench_nosync600.log | cut -c1-400
HM_GRAPH_SYNC=1 timeout -k 10 280 python bench.py --no-extras --steps 100 --warmup 5 > $O/bench_sync100.log 2>&1; echo "sync rc=$?"; tail -1 $O/bench_sync100.log | cut -c1-400
