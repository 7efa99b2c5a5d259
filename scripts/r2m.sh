#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -rf > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc; tail -3 $O/pytest.log
timeout -k 10 500 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log > $O/bench_line.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2m/bench_line.json'))
print(d['value'], d['ms_per_step'], d['train_leg']['ms_per_step'])
for k in ('roofline','roofline_c4','roofline_bwd','roofline_mlp','roofline_mlp_bf16'):
    print(k, d[k]['achieved'], d[k]['frac'], d[k].get('traffic'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for k in ('config3_leg','config5_leg','lazy_sampler_leg'): print(k, d[k]['value'], d[k]['ms_per_step'], d[k].get('lazy_sampler',{}).get('ms_per_step'))
PY
