#!/bin/bash
# scripts/ab_kt_all.sh A B ...: rocprofv3 kernel averages of ${AB_SCRIPT:-scripts/ab_mlp_all.py} per libebm_hip.so variant (ab/<name>.so), one box;
# prints a table kernel | grid | avg us per variant.
R=$PWD; O=$PWD/gpurun_out/abkt_${AB_TAG:-all}; mkdir -p $O
cp torchebm_amd/libebm_hip.so /tmp/_keep.so
for v in "$@"; do
  cp ab/$v.so torchebm_amd/libebm_hip.so
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $O/$v && rocprofv3 --kernel-trace --output-format csv -d $O/$v -o kt -- python $R/${AB_SCRIPT:-scripts/ab_mlp_all.py} > $O/$v.log 2>&1 )
done
cp /tmp/_keep.so torchebm_amd/libebm_hip.so
python - $O "$@" <<'PY'
import csv, glob, sys, collections
O, names = sys.argv[1], sys.argv[2:]
tab = collections.OrderedDict()
for v in names:
    f = glob.glob(f"{O}/{v}/**/*kernel_trace.csv", recursive=True)
    if not f:
        print("no trace for", v); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "ebm::" not in k: continue
        key = (k.replace("void ebm::", "").split("(")[0][:64], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for key, ts in agg.items():
        ts.sort()
        tab.setdefault(key, {})[v] = (ts[len(ts) // 2], len(ts))
print(f'{"kernel":66s} {"grid":>9s} {"lds":>7s} ' + " ".join(f"{v:>10s}" for v in names) + "   ratio(last/first)")
for key, d in tab.items():
    if max(x[0] for x in d.values()) < 20: continue
    row = " ".join(f"{d[v][0]:10.1f}" if v in d else f'{"-":>10s}' for v in names)
    ratio = d[names[-1]][0] / d[names[0]][0] if names[0] in d and names[-1] in d else float("nan")
    print(f"{key[0]:66s} {key[1]:>9s} {key[2]:>7s} {row}   {ratio:.3f}")
PY
