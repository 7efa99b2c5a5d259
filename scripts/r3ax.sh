#!/bin/bash
# round 3, run ax: where the 40 fillBuffer + 21 copyBuffer launches per step of the world-1 RCCL step come from (HIP API trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ax; mkdir -p $O
export HM_DIST_FORCE=1
timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --legs fixed --no-extras --steps 10 --warmup 3 > $O/prof.log 2>&1; echo "rc=$?"
ls $O/prof/*/ | head -20
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r3ax/prof/*/*hip_api_trace.csv')
print(f)
rows=list(csv.DictReader(open(f[0])))
print(rows[0].keys())
c=collections.Counter(r['Function'] for r in rows)
for k,v in c.most_common(40): print(v,k)
PY
