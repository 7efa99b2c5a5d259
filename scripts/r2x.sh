#!/bin/bash
# round 2, run x: MFMA-busy PMC of the fused SDF kernel (after the softplus change) and of the grad-path GEMM
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2x; mkdir -p $O
timeout -k 10 200 python bench.py --only gemm 2>/dev/null | tail -1 | tee $O/gemm.json | cut -c1-1500
pmc() { local name=$1; local ctr=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_$name -- "$@" > $O/pmc_$name.log 2>&1
  echo "== $name [$ctr]"; python scripts/pmc_summary.py $O/pmc_$name sdf_fwd_kernel gemm_f32 | tee $O/pmc_$name.json | cut -c1-2500; rm -rf $O/pmc_$name; }
pmc mlp_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" python bench.py --only mlp
pmc gemm_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" python bench.py --only gemm
