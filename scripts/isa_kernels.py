#!/usr/bin/env python3
"""One line per kernel of a hipcc -S listing: MFMAs / VALU / packed in the biggest loop, the gap histogram, the issue-model estimate.
   scripts/isa_kernels.py file.s [name-filter]"""
import collections, re, subprocess, sys
src = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [i for i, l in enumerate(src) if l.startswith("_Z") and ":" in l.split(";")[0] and l.split(":")[0].endswith("E") or (l.startswith("_Z") and re.match(r"^_Z\S+:\s", l))]
names = subprocess.run(["c++filt"], input="\n".join(src[i].split(":")[0] for i in starts), capture_output=True, text=True).stdout.split("\n")
for i, nm in zip(starts, names):
    short = re.sub(r"^void ", "", re.sub(r"\(.*", "", nm.replace("(anonymous namespace)::", ""))).replace("ebm::", "")
    if flt not in short:
        continue
    end = next((j for j in range(i, len(src)) if src[j].startswith(".Lfunc_end")), None)
    if end is None:
        continue
    body = src[i:end]
    labels = {m.group(1): k for k, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = (0, 0, 0)
    for k, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k and k - labels[m.group(1)] > best[0]:
            best = (k - labels[m.group(1)], labels[m.group(1)], k)
    gaps, cur, total = [], None, 0
    for l in body[best[1]:best[2]]:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            continue
        op = l.split()[0]
        total += 1
        if op.startswith("v_mfma"):
            cur = collections.Counter(); gaps.append(cur)
        elif cur is not None:
            if op.startswith("v_pk_"): cur["pk"] += 1
            elif op.startswith("v_"): cur["valu"] += 1
            elif op.startswith("scratch_"): cur["scr"] += 1
            else: cur["other"] += 1
    if not gaps:
        continue
    hist = collections.Counter()
    for g in gaps:
        n = g["valu"] + g["pk"]
        hist["0" if n == 0 else "1-2" if n <= 2 else "3-5" if n <= 5 else "6-8" if n <= 8 else "9-16" if n <= 16 else ">16"] += 1
    pk_in = sum(g["pk"] for g in gaps if g["valu"] + g["pk"] <= 16)
    scr = sum(g["scr"] for g in gaps)
    mf = "32x32" if any("32x32" in l for l in body if "v_mfma" in l) else "16x16"
    print(f"{short[:70]:70s} loop {total:5d} mfma {len(gaps):4d} ({mf}) gaps " + " ".join(f"{k}:{hist[k]}" for k in ("0", "1-2", "3-5", "6-8", "9-16", ">16") if hist[k]) + f"  pk-in-gaps {pk_in} scratch-ops {scr}")
