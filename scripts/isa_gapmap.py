#!/usr/bin/env python3
"""One character per MFMA gap of a kernel's biggest loop: the number of VALU instructions behind that MFMA (0-9, then a-z = 10-35,
'#' beyond), 48 gaps per line; 'p' marks in upper case a gap that holds a packed-f32 instruction.  scripts/isa_gapmap.py file.s <name>"""
import re, sys
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
gaps, cur = [], None
for l in body[best[1]:best[2]]:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
        continue
    op = l.split()[0]
    if op.startswith("v_mfma"):
        cur = [0, False]; gaps.append(cur)
    elif cur is not None and op.startswith("v_"):
        cur[0] += 1
        cur[1] |= op.startswith("v_pk_")
chars = "0123456789abcdefghijklmnopqrstuvwxyz"
s = "".join(("#" if n >= 36 else chars[n]).upper() if pk else ("#" if n >= 36 else chars[n]) for n, pk in gaps)
for i in range(0, len(s), 48):
    print(f"{i:4d} {s[i:i + 48]}")
