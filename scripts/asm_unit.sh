#!/bin/bash
# scripts/asm_unit.sh <unit.hip> <out.s> [extra flags]: the gfx950 assembly of one translation unit of torchebm_amd/csrc, with the
# library's flags (for scripts/isa_gaps.py / scripts/isa_mix.py).
set -e
unit=$1; out=$2; shift 2
cd "$(dirname "$0")/../torchebm_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-function \
  -Wno-pass-failed -Wno-array-bounds -Wno-unused-command-line-argument "$@" -S --cuda-device-only "$unit" -o "$out"
