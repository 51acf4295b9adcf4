#!/bin/bash
# PMC A/B of libebm_hip.so variants on one box: scripts/ab_pmc.sh "2x128" R5 SLOT ...  (counters of the chain kernel per variant)
CASES="$1"; shift
cp torchebm_amd/libebm_hip.so /tmp/_keep.so
for v in "$@"; do
  cp ab/$v.so torchebm_amd/libebm_hip.so
  echo "== $v"
  MLP_CASES=$CASES MLP_NO_STEP_ROUTE=1 MLP_K=${MLP_K:-20} bash scripts/pmc_cmd.sh ab_$v \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
    "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR" \
    -- python $PWD/scripts/bench_mlp_dims.py 2>&1 | grep -v "^pass"
done
cp /tmp/_keep.so torchebm_amd/libebm_hip.so
