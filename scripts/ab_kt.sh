#!/bin/bash
# Kernel-trace A/B of libebm_hip.so variants on ONE box (ab/<name>.so from scripts/ab_build.sh):
#   scripts/ab_kt.sh "2x128,32x128" R5 SLOT ...     -> per variant, rocprofv3's average duration of every ebm:: kernel of
#   `MLP_NO_STEP_ROUTE=1 python scripts/bench_mlp_dims.py` (7 calls of 20 steps per case); MLP_K overrides the step count.
CASES="$1"; shift
R=$PWD; O=$PWD/gpurun_out/abkt; mkdir -p $O
cp torchebm_amd/libebm_hip.so /tmp/_keep.so
for v in "$@"; do
  cp ab/$v.so torchebm_amd/libebm_hip.so
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $O/$v && MLP_CASES=$CASES MLP_NO_STEP_ROUTE=1 MLP_K=${MLP_K:-20} rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o kt -- python $R/scripts/bench_mlp_dims.py > $O/$v.log 2>&1 )
  echo "== $v"
  python - $O/$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("  (no stats)"); sys.exit()
for r in csv.DictReader(open(f[0])):
    if "ebm::" in r["Name"] and int(r["Calls"]) >= 5:
        print(f'  {r["Name"][:80]:82s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.2f} min_us={float(r["MinNs"])/1e3:9.2f}')
PY
done
cp /tmp/_keep.so torchebm_amd/libebm_hip.so
