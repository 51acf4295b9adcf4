#!/bin/bash
# A/B of libebm_hip.so builds on one box, interleaved:  scripts/ab_mlp.sh "32x128,64x128" A B [A B ...]   (build/ab/<name>.so)
CASES="$1"; shift
cp torchebm_amd/libebm_hip.so /tmp/_keep.so
for round in 1 2 3; do
  for v in "$@"; do
    cp ab/$v.so torchebm_amd/libebm_hip.so
    echo "== $v (round $round)"
    MLP_CASES=$CASES MLP_NO_STEP_ROUTE=1 MLP_K=${MLP_K:-20} python scripts/bench_mlp_dims.py 2>&1 | grep kernel_ms | sed 's/.*dim=\([0-9]*\) H=\([0-9]*\).*"kernel_ms": \([0-9.]*\).*/  dim \1 H \2: \3 ms/'
  done
done
cp /tmp/_keep.so torchebm_amd/libebm_hip.so
