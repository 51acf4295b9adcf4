#!/usr/bin/env python3
"""MFMA gap histogram of a kernel's biggest loop in a hipcc -S listing: how many non-MFMA instructions sit behind each MFMA.
  scripts/isa_gaps.py file.s <mangled-name-substring> [--blocks N]   (--blocks: print the class mix of MFMA-free runs > N)
A gfx950 wave hides ~5 plain VALU instructions (3 transcendentals; 0 packed-f32 ones: +20 cycles) behind a 32-cycle
v_mfma_f32_32x32x16_bf16 (profiles/r06_mfma_valu_overlap.txt): gaps with more expose the excess at ~5 cycles each, gaps with fewer
leave issue slots empty."""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l and ":" in l.split(";")[0])
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
body = src[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
best = (0, 0, 0)
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
        best = (i - labels[m.group(1)], labels[m.group(1)], i)
loop = body[best[1]:best[2]]
ins = []
for l in loop:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
        continue
    ins.append(l.split()[0])


def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_pk_"): return "pk"
    if op in ("v_exp_f32_e32", "v_rcp_f32_e32", "v_log_f32_e32", "v_sqrt_f32_e32", "v_sin_f32_e32", "v_cos_f32_e32", "v_rsq_f32_e32"): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "mem"


gaps = []  # per MFMA: Counter of what follows it up to the next MFMA
cur = None
head = collections.Counter()
for op in ins:
    k = kind(op)
    if k == "mfma":
        cur = collections.Counter()
        gaps.append(cur)
    elif cur is None:
        head[k] += 1
    else:
        cur[k] += 1
n_mfma = len(gaps)
tot = collections.Counter()
for g in gaps:
    tot.update(g)
print(f"loop: {len(ins)} instructions, {n_mfma} MFMAs; before the first MFMA: {dict(head)}")
print("all gaps:", dict(tot))
hist = collections.Counter()
for g in gaps:
    n = g["valu"] + g["pk"] + g["trans"]
    b = "0" if n == 0 else "1-2" if n <= 2 else "3-5" if n <= 5 else "6-8" if n <= 8 else "9-16" if n <= 16 else "17-64" if n <= 64 else ">64"
    hist[b] += 1
print("VALU instructions per gap:", {k: hist[k] for k in ("0", "1-2", "3-5", "6-8", "9-16", "17-64", ">64") if hist[k]})
print("packed-f32 ops in gaps of <= 16 VALU:", sum(g["pk"] for g in gaps if g["valu"] + g["pk"] + g["trans"] <= 16))
# exposed-issue estimate: per gap max(32, 10 + 5 valu + 8 trans + (20 + 5 pk if pk) + 4 lds + 2 salu)
est = 0
for g in gaps:
    issue = 10 + 5 * g["valu"] + 8 * g["trans"] + (20 + 5 * g["pk"] if g["pk"] and g["valu"] + g["pk"] + g["trans"] <= 16 else 5 * g["pk"]) + 4 * g["lds"] + 2 * (g["salu"] + g["wait"] + g["nop"])
    est += max(32 + 0.5 * (g["valu"] + g["trans"]), issue)
print(f"issue-model estimate: {est:.0f} cycles per trip ({est / n_mfma:.1f} per MFMA; matrix pipe busy {32 * n_mfma / est:.0%})")
if "--blocks" in sys.argv:
    n0 = int(sys.argv[sys.argv.index("--blocks") + 1])
    for i, g in enumerate(gaps):
        n = sum(g.values())
        if n > n0:
            print(f"  run behind MFMA {i}: {n} instructions {dict(g)}")
