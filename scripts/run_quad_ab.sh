#!/bin/bash
# on the GPU box: A/B timing + per-phase clocks of the quad-wave MLP kernel (scripts/build_quad_ab.sh made the libraries)
cp torchebm_amd/libebm_hip.so /tmp/keep.so
cp build/ab/quad.so torchebm_amd/libebm_hip.so; AB_DIMS=${AB_DIMS:-2,8,32} timeout 300 python scripts/ab_quad.py 2>&1 | grep "^{"
for v in times solo; do cp build/ab/quad_$v.so torchebm_amd/libebm_hip.so; echo "== $v"; timeout 100 python scripts/quad_phase_times.py 2>&1 | tail -7; done
cp /tmp/keep.so torchebm_amd/libebm_hip.so
