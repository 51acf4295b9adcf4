#!/bin/bash
# same-box A/B of libebm_hip.so builds on the cost of return_diagnostics=True for the MLP energy:  scripts/ab_mlp_diag.sh A B ...
cp torchebm_amd/libebm_hip.so /tmp/_keep.so
for round in 1 2 3; do
  for v in "$@"; do
    cp build/ab/$v.so torchebm_amd/libebm_hip.so
    echo "== $v (round $round)"
    python scripts/bench_mlp_diag.py 2>&1 | grep langevin | sed 's/.*mlp \([0-9]*\)-128.*"plain_ms": \([0-9.]*\), "diag_thin5_ms": \([0-9.]*\), "diag_thin5_over_plain": \([0-9.]*\).*/  dim \1: plain \2 thin5 \3 ratio \4/'
  done
done
cp /tmp/_keep.so torchebm_amd/libebm_hip.so
