#!/bin/bash
# round 3, run ar: kernel trace of the TRAIN leg in its steady state (300 steps at lr 1e-4; the last step's anatomy)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ar; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python bench.py --cfg C2 --legs train --no-extras --steps 300 --warmup 5 > $O/prof.log 2>&1; echo "prof rc=$?"
python - <<'PY'
import csv,glob,re
from collections import defaultdict
f=glob.glob('gpurun_out/r3ar/prof/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adam_update' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
step=rows[a+1:b+1]
agg=defaultdict(lambda:[0,0.0])
for r in step:
    n=r['Kernel_Name']; n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'at::native::','',n); n=re.sub(r'void ','',n)
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    agg[n[:56]][0]+=1; agg[n[:56]][1]+=d
print('launches',len(step),'kernel us',round(sum(v[1] for v in agg.values()),1),'span us',(int(step[-1]['End_Timestamp'])-int(step[0]['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:16]: print(f"{v[0]:4d} {v[1]:8.1f}  {k}")
# the small kernel / tail / secant durations in order
for r in step:
    n=r['Kernel_Name']
    if any(k in n for k in ('sdf_fwd_small','march_tail','trace_secant','sdf_fwd_kernel')):
        print(re.sub(r'.*::','',n)[:28], round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1))
PY
