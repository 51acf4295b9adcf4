#!/bin/bash
# Same-box timing of variants of the tiled Gaussian kernel (gauss_big.hip built with extra -D flags into build/ab/big_<name>.so):
#   scripts/ab_big.sh build name1="-DEBM_BIG_EXP=1" name2="-DEBM_BIG_WAVES1=4 -DEBM_BIG_TWO_WG=1" ...     (here: cross-compiles)
#   scripts/ab_big.sh run name1 name2 ...                                                                (on the GPU box)
MODE="$1"; shift
cd "$(dirname "$0")/.."
if [ "$MODE" = build ]; then
  mkdir -p build/ab
  OBJS=$(ls build/csrc/*.o | grep -v "gauss_big.o\|gauss_res.o")
  CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-pass-failed"
  for spec in "$@"; do
    v="${spec%%=*}"; flags="${spec#*=}"; [ "$flags" = "$spec" ] && flags=""
    ( $CC $flags -c torchebm_amd/csrc/gauss_big.hip -o build/ab/gauss_big_$v.o && \
      $CC $flags -c torchebm_amd/csrc/gauss_res.hip -o build/ab/gauss_res_$v.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/big_$v.so $OBJS build/ab/gauss_big_$v.o build/ab/gauss_res_$v.o && echo "built $v" ) &
  done
  wait
else
  cp torchebm_amd/libebm_hip.so /tmp/_keep.so
  for round in 1 2; do
    for v in "$@"; do
      cp build/ab/big_$v.so torchebm_amd/libebm_hip.so
      echo "== $v (round $round)"
      BIG_NO_EAGER=1 BIG_DIMS=${BIG_DIMS:-160,256,512} python scripts/bench_gauss_big.py 2>&1 | grep '"ms"' | sed 's/.*"dim": \([0-9]*\),.*"ms": \([0-9.]*\),.*/  dim \1: \2 ms/' | tr '\n' ' '; echo
    done
  done
  cp /tmp/_keep.so torchebm_amd/libebm_hip.so
fi
