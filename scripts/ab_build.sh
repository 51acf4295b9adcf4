#!/bin/bash
# scripts/ab_build.sh NAME "unit1.hip unit2.hip ..." [extra hipcc flags]: an A/B variant of libebm_hip.so -- the named units rebuilt
# from the working tree (with the extra flags) into build/ab_NAME/, everything else taken from build/csrc/ -> ab/NAME.so
# (ab/ travels to the GPU box; scripts/ab_mlp.sh / scripts/ab_run.sh swap the variants in on ONE box).
set -e
name=$1; units=$2; shift 2
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/torchebm_amd/csrc"
mkdir -p "$root/build/ab_$name" "$root/ab"
objs=""
for o in "$root"/build/csrc/*.o; do
  b=$(basename "$o" .o)
  if [[ " $units " == *" $b.hip "* ]]; then objs="$objs $root/build/ab_$name/$b.o"; else objs="$objs $o"; fi
done
pids=""
for u in $units; do
  b=$(basename "$u" .hip)
  extra=""
  case "$b" in mlp*) extra="-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form";; gauss_res*) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-function \
    -Wno-pass-failed -Wno-array-bounds $extra "$@" -c "$u" -o "$root/build/ab_$name/$b.o" &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/ab/$name.so" $objs
ls -la "$root/ab/$name.so"
