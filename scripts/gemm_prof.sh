#!/bin/bash
# true kernel durations of the training GEMM shapes for each small-tile configuration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in 0 1 2; do
  HM_GEMM_CFG=$cfg rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pgemm_$cfg -- python scripts/gemm_bench.py > gpurun_out/pgemm_$cfg.log 2>&1
done
python - <<'PY'
import csv,glob
for cfg in (0,1,2):
    rows=list(csv.DictReader(open(glob.glob(f'gpurun_out/pgemm_{cfg}/*/*kernel_trace.csv')[0])))
    g=[r for r in rows if 'gemm_f32' in r['Kernel_Name']]
    out=[]
    for i in range(0,len(g),23):
        ch=g[i:i+23]; d=sorted((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in ch)
        out.append(round(d[len(d)//2],1))
    print("cfg",cfg,out)
PY
