#!/bin/bash
# round-2 batch E: PMC passes (gather C2/C4, 8-byte gather calibration, MFMA busy), grid sweep, distributed GPU test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY[A-Z_]*\|TCC_EA0_RDREQ[A-Z_0-9a-z]*\|TCC_HIT[a-z_]*\|TCC_MISS[a-z_]*\|GRBM_GUI_ACTIVE\|SQ_WAVE_CYCLES\|SQ_INSTS_MFMA\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*" $O/counters.txt | sort -u | tr '\n' ' ' | tee $O/counters_short.txt; echo
timeout -k 10 200 python -m pytest tests/test_distributed_gpu.py -m gpu -q -s -x > $O/pytest_dist.log 2>&1; echo "dist rc=$?"; tail -6 $O/pytest_dist.log | cut -c1-300
pmc() { # name counters -- cmd...
  local name=$1; local ctr=$2; shift 2
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_$name -- "$@" > $O/pmc_$name.log 2>&1
  echo "== $name [$ctr]"; python scripts/pmc_summary.py $O/pmc_$name encode_fwd gather_calib sdf_fwd gemm_f32 | cut -c1-1500
}
for mode in "1,0" "2,64" "2,32" "16,8"; do
  m=${mode/,/_}
  pmc calib_${m}_fetch FETCH_SIZE python bench.py --only gather_calib --calib $mode
  pmc calib_${m}_rdreq "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" python bench.py --only gather_calib --calib $mode
done
for cfg in C2 C4; do
  pmc gather_${cfg}_fetch FETCH_SIZE python bench.py --only gather --cfg $cfg
  pmc gather_${cfg}_write WRITE_SIZE python bench.py --only gather --cfg $cfg
  pmc gather_${cfg}_rdreq "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" python bench.py --only gather --cfg $cfg
done
pmc mlp_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" python bench.py --only mlp
for g in 128 192 256 320; do HM_ENCODE_FLAGS=2 HM_ENCODE_GRID=$g timeout -k 10 120 python bench.py --only gather --cfg C4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('C4 grid=$g flags=2', d['achieved'], d['avg_launch_ms'])" | tee -a $O/gather_sweep.log; done
for g in 256 384 512; do HM_ENCODE_FLAGS=2 HM_ENCODE_GRID=$g timeout -k 10 120 python bench.py --only gather --cfg C2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('C2 grid=$g flags=2', d['achieved'], d['avg_launch_ms'])" | tee -a $O/gather_sweep.log; done
