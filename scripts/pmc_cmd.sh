#!/usr/bin/env bash
# rocprofv3 PMC passes over one command, one pass per counter group:  scripts/pmc_cmd.sh <tag> "<group1>" "<group2>" ... -- <cmd...>
set -u
TAG="$1"; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$REPO/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for G in "${GROUPS_[@]}"; do
  rocprofv3 --pmc $G --output-format csv -d "$OUT/p$i" -o pmc -- "$@" > "$OUT/run$i.log" 2>&1
  echo "pass $i ($G) rc=$?"
  i=$((i+1))
done
python - "$OUT" <<'PY'
import csv, sys, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if "ebm::" not in name:
            continue
        agg[(name[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:62s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
