#!/bin/bash
# Round-6 evidence for config 5's training step (one HIP graph per step, utils.GraphedTrainingStep): per-kernel times of 40 replays and
# the chain / training kernels' issue counters.  Outputs: gpurun_out/r06_c5_*.
O=$PWD/gpurun_out; mkdir -p $O
R=$PWD
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_c5_kt -o kt -- python $R/scripts/c5_graph_profile.py > $O/r06_c5_kt.log 2>&1 )
python - $O/r06_c5_kt <<'PY' > $O/r06_c5_step_kernels_final.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# rocprofv3 --kernel-trace --stats -- python scripts/c5_graph_profile.py  (5 warm-up / capture calls + 40 replays); total kernel time {tot/1e6:.2f} ms")
for r in rows[:28]:
    print(f'{r["Name"][:110]:112s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.2f} total_ms={float(r["TotalDurationNs"])/1e6:8.3f} pct={float(r["Percentage"]):6.2f}')
PY
bash scripts/pmc_cmd.sh r06c5 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR" \
  -- python $R/scripts/c5_graph_profile.py > $O/r06_c5_pmc_raw.txt 2>&1
tail -3 $O/r06_c5_kt.log; head -12 $O/r06_c5_step_kernels_final.txt; grep -c mean $O/r06_c5_pmc_raw.txt
