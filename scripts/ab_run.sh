#!/bin/bash
# scripts/ab_run.sh "<command>" NAME ...: run <command> once per libebm_hip.so variant ab/NAME.so on ONE box (the tree's own library
# is put back afterwards); output per variant in gpurun_out/ab_NAME.log
CMD="$1"; shift
mkdir -p gpurun_out
cp torchebm_amd/libebm_hip.so /tmp/_keep.so
for v in "$@"; do
  cp ab/$v.so torchebm_amd/libebm_hip.so
  echo "== $v"
  bash -c "$CMD" > gpurun_out/ab_$v.log 2>&1
  grep -v amdgpu.ids gpurun_out/ab_$v.log | cut -c1-${AB_COLS:-110}
done
cp /tmp/_keep.so torchebm_amd/libebm_hip.so
