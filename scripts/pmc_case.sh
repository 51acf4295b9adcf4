#!/usr/bin/env bash
# rocprofv3 PMC pass over one scripts/bench_kernels.py case:  scripts/pmc_case.sh <tag> <case> "<counters>"
set -u
TAG="$1"; CASE="$2"; CTRS="$3"
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -o pmc -- python "$REPO/scripts/bench_kernels.py" "$CASE" > "$OUT/run.log" 2>&1
echo "rc=$?"
python - "$OUT/pmc_counter_collection.csv" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if "ebm::" not in name:
        continue
    agg[(name[:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:72s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
