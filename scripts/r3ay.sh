#!/bin/bash
# round 3, run ay: closest-approach scan + secant refinement in one launch - tracer / step tests, A/B of the C2 legs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ay; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_raytrace_gpu.py tests/test_graph_step_gpu.py tests/test_idr_step_gpu.py tests/test_sdf_gpu.py -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log | cut -c1-200; grep "^FAILED" $O/pytest.log; grep -A1 "head 0: stats\|head 16: stats" $O/pytest.log | cut -c1-300
for v in 1 0 1 0; do
  HM_TRACE_OVERLAP=$v timeout -k 10 200 python bench.py --cfg C2 --legs both --no-extras --steps 100 --warmup 10 > $O/b_$v.log 2>&1 && echo "overlap=$v $(tail -1 $O/b_$v.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["train_leg"]["ms_per_step"], d["lazy_sampler_leg"]["ms_per_step"])')"
done
