#!/bin/bash
# round 3, run h: where does the split kernel's time go?  PMC passes over `bench.py --only mlp_split`
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
pmc() { local name=$1; local ctr=$2; shift 2
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/pmc_$name -- "$@" > $O/pmc_$name.log 2>&1
  echo "== $name"; python scripts/pmc_summary.py $O/pmc_$name sdf_fwd_split | cut -c1-900; }
pmc a "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" python bench.py --only mlp_split --split f16x2
pmc b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" python bench.py --only mlp_split --split f16x2
pmc c "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" python bench.py --only mlp_split --split f16x2
pmc d "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" python bench.py --only mlp_split --split f16x2
pmc e "SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS_F32" python bench.py --only mlp_split --split f16x2
