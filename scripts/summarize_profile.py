#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/prof_<tag>/) into small tracked files under profiles/:

  profiles/<tag>_kernel_stats.csv   per-kernel calls / total / avg / min / max (names shortened)
  profiles/<tag>_pmc.json           FETCH_SIZE / WRITE_SIZE per kernel, raw and corrected
  profiles/traffic.json             HBM bytes per launch of the dominant kernel (read by bench.py)

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request on
wide coalesced streams, i.e. exactly half the bytes read -> doubled here; WRITE_SIZE is taken
as is (it equals the state size to the byte for this kernel).  Both counters are in KiB.

    python scripts/summarize_profile.py <tag> [dominant-kernel-substring]
"""
import collections
import csv
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
dominant = sys.argv[2] if len(sys.argv) > 2 else "langevin_chain_elem_kernel"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^()]{0,40}>)?)", name)
    return (m.group(1) if m else name)[:90]


rows = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,total_ms,avg_ms,min_ms,max_ms,percent\n")
    for r in rows:
        f.write(
            f"{short(r['Name'])},{r['Calls']},{float(r['TotalDurationNs'])/1e6:.4f},{float(r['AverageNs'])/1e6:.4f},"
            f"{float(r['MinNs'])/1e6:.4f},{float(r['MaxNs'])/1e6:.4f},{float(r['Percentage']):.3f}\n"
        )

# The default bench.py run also measures the other BASELINE configs (`extra`), some of them with the SAME kernel at
# another shape (config 4's shard uses the lean chain kernel at dim 128, k 500).  The plain per-name statistics above
# mix them; this second table splits the package's kernels by launch grid, from the kernel trace.
trace = os.path.join(src, "trace", "bench_kernel_trace.csv")
if os.path.exists(trace):
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if "ebm::" not in r["Kernel_Name"]:
            continue
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["LDS_Block_Size"]),
               int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["Scratch_Size"]))
        groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    with open(os.path.join(dst, f"{tag}_kernel_stats_by_grid.csv"), "w") as f:
        f.write("kernel,workgroups,lds_bytes,vgprs,agprs,scratch_bytes,calls,avg_ms,min_ms,max_ms\n")
        for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            f.write(",".join(str(k) for k in key) + f",{len(v)},{sum(v)/len(v):.4f},{min(v):.4f},{max(v):.4f}\n")
    print(open(os.path.join(dst, f"{tag}_kernel_stats_by_grid.csv")).read())

# keyed by kernel AND launch grid (workgroups): the default run uses some kernels at several shapes (the lean chain kernel at
# config 2 and at config 4's shard; the per-step kernel at 2^26 elements and at config 5's 2^17), and a mean over all of
# them describes none
pmc = {}
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    path = os.path.join(src, sub, "bench_counter_collection.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "ebm::" in r["Kernel_Name"]:
            wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            agg[f"{short(r['Kernel_Name'])} @ {wgs} workgroups"].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc.setdefault(k, {})[counter + "_KiB_mean"] = sum(v) / len(v)
        pmc[k][counter + "_launches"] = len(v)
for k, d in pmc.items():
    fetch = d.get("FETCH_SIZE_KiB_mean", 0.0) * 1024 * 2  # gfx950: x2, see docstring
    write = d.get("WRITE_SIZE_KiB_mean", 0.0) * 1024
    d["hbm_read_bytes_corrected"] = fetch
    d["hbm_write_bytes"] = write
    d["hbm_bytes_per_launch"] = fetch + write
json.dump(pmc, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)

# the dominant kernel at the headline shape = its most frequent grid in the run
dom = sorted((k for k in pmc if dominant in k), key=lambda k: -pmc[k].get("FETCH_SIZE_launches", 0))[:1]
for k in dom:
    d = pmc[k]
    if True:
        json.dump(
            {
                "kernel": k,
                "hbm_bytes_per_launch": d["hbm_bytes_per_launch"],
                "hbm_read_bytes_corrected": d["hbm_read_bytes_corrected"],
                "hbm_write_bytes": d["hbm_write_bytes"],
                "source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; "
                          "FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md)",
            },
            open(os.path.join(dst, "traffic.json"), "w"),
            indent=1,
        )
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read())
print(json.dumps(pmc, indent=1)[:1500])
