#!/bin/bash
# A/B of libebm_hip.so builds (build/ab/<name>.so), HMC on the MLP energies, interleaved: scripts/ab_hmc.sh A B ...
cp torchebm_amd/libebm_hip.so /tmp/_keep.so
for round in 1 2 3; do
  for v in "$@"; do
    cp build/ab/$v.so torchebm_amd/libebm_hip.so
    echo "== $v (round $round)"
    python scripts/ab_hmc2d.py 2>&1 | grep case | sed 's/.*dim=\([0-9]*\) H=\([0-9]*\).*"kernel_ms": \([0-9.]*\).*/  dim \1 H \2: \3 ms/'
  done
done
cp /tmp/_keep.so torchebm_amd/libebm_hip.so
