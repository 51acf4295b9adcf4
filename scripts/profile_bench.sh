#!/usr/bin/env bash
# Profile `python bench.py` on the GPU box with rocprofv3 (run through gpurun).
#   1) --kernel-trace --stats : per-kernel durations (must agree with bench.py's event timing)
#   2) --pmc FETCH_SIZE       : HBM read traffic  (its own pass; TCC slots don't fit both)
#   3) --pmc WRITE_SIZE       : HBM write traffic (its own pass)
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG="${1:-r01}"
ARGS="${2:---steps 20 --warmup 3 --no-cpu-baseline}"          # the kernel trace: the driver's own command line minus the CPU baseline
PMC_ARGS="${3:---steps 5 --warmup 1 --no-cpu-baseline}"  # counter passes: headline kernel AND the extras (the HBM-bound step kernel is one of them)
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- python "$REPO/bench.py" $PMC_ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- python "$REPO/bench.py" $PMC_ARGS > "$OUT/pmc_write.log" 2>&1
echo "pmc write rc=$?"
find "$OUT" -name '*.csv' | head -20
