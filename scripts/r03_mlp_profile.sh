#!/bin/bash
# Round-3 evidence for the wide MLP kernels, one box: A/B against the round-2 (exact-f32) kernels, the dims sweeps, PMC counters,
# phase times.  Needs build/ab/{R2,CUR,PT}.so ("Wide MLP on the bf16 pipe").  Outputs: gpurun_out/r03_*.
O=gpurun_out; mkdir -p $O
bash scripts/ab_mlp.sh 2x128,8x128,32x128,64x128,32x64 R2 CUR > $O/r03_ab_mlp.txt 2>&1
for v in R2 CUR; do cp build/ab/$v.so torchebm_amd/libebm_hip.so; echo "== $v"; python scripts/ab_hmc2d.py 2>&1 | grep case; done >> $O/r03_ab_mlp.txt
cp build/ab/CUR.so torchebm_amd/libebm_hip.so
python scripts/bench_mlp_dims.py 2>&1 | grep case > $O/r03_bench_mlp_dims.jsonl
python scripts/bench_mlp_hmc_wide.py 2>&1 | grep config > $O/r03_bench_mlp_hmc_wide.jsonl
{
for c in chain_mlp_32_128 hmc_mlp_32_128; do
  echo "# $c"
  bash scripts/pmc_case.sh a $c "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" | grep wide
  bash scripts/pmc_case.sh b $c "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" | grep wide
done
} > $O/r03_pmc_mlp_raw.txt 2>&1
cp build/ab/PT.so torchebm_amd/libebm_hip.so
{ python scripts/mlp_phase_times.py 32 128; python scripts/mlp_phase_times.py 2 128; python scripts/mlp_phase_times.py 64 128; } > $O/r03_mlp_phase_times.txt 2>&1
cp build/ab/CUR.so torchebm_amd/libebm_hip.so
python bench.py --steps 20 --warmup 3 > $O/r03_bench_n1.json 2> $O/r03_bench_n1.err
tail -5 $O/r03_ab_mlp.txt; wc -l $O/r03_*
