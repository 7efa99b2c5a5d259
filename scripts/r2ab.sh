#!/bin/bash
# round 2, run ab: per-dispatch durations of the filter-bank embedder kernel inside the config-5 step
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2ab
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/prof -o c5 --output-format csv -- python bench.py --cfg C5 --legs fixed --no-extras --steps 4 --warmup 3 > $OUT/c5.log 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2ab/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last step: take the last 50 nffb dispatches
idx = [i for i, n in enumerate(names) if "nffb_fwd_kernel" in n][-50:]
d = [(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3 for i in idx]
print("nffb_fwd per-dispatch us (one step):", [round(v, 1) for v in d])
idx = [i for i, n in enumerate(names) if "sdf_fwd_small_kernel" in n][-50:]
d = [(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3 for i in idx]
print("sdf_fwd_small per-dispatch us (one step):", [round(v, 1) for v in d])
PY
rm -rf $OUT/prof
