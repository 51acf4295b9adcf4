#!/bin/bash
# Experiment builds of the quad-wave MLP kernel (scripts/experiments/mlp_quad.hip; docs/design/mlp_wide.md "Round 5"):
# build/ab/quad.so (routes the plain H = 128, dim <= 32 call to it unless EBM_MLP_NO_QUAD=1), quad_times.so (phase stamps),
# quad_solo.so (stamps, tile 1 idle).  Needs an up-to-date build/csrc (make -C torchebm_amd/csrc).
set -e
cd "$(dirname "$0")/../torchebm_amd/csrc"
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-function -Wno-pass-failed -Wno-array-bounds"
mkdir -p ../../build/ab
[ ../../build/ab/mlp_wide_ab.o -nt mlp_wide_body.h ] || $B -DEBM_AB_SWITCHES -DEBM_MLP_QUAD_EXPERIMENT -c mlp_wide.hip -o ../../build/ab/mlp_wide_ab.o &
$B -DEBM_PHASE_TIMES $QUAD_FLAGS -I. -c ../../scripts/experiments/mlp_quad.hip -o ../../build/ab/mlp_quad_times.o &
$B -DEBM_PHASE_TIMES -DEBM_QUAD_SOLO $QUAD_FLAGS -I. -c ../../scripts/experiments/mlp_quad.hip -o ../../build/ab/mlp_quad_solo.o &
$B $QUAD_FLAGS -I. -c ../../scripts/experiments/mlp_quad.hip -o ../../build/ab/mlp_quad_plain.o &
wait
cd ../../build
OTHERS=$(ls csrc/*.o | grep -v "csrc/mlp_wide.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/quad.so $OTHERS ab/mlp_wide_ab.o ab/mlp_quad_plain.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/quad_times.so $OTHERS ab/mlp_wide_ab.o ab/mlp_quad_times.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/quad_solo.so $OTHERS ab/mlp_wide_ab.o ab/mlp_quad_solo.o &
wait
rm -f ab/quad_t0.so
ls -la ab/*.so
