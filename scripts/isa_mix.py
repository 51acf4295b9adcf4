#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: scripts/isa_mix.py file.s <mangled-name-substring> [--loop]
Counts by class over the whole kernel body, and over the biggest backward-branch loop with --loop."""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l and ":" in l.split(";")[0])
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
body = src[start:end]
if "--loop" in sys.argv:
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = (0, 0, 0)
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    body = body[best[1]:best[2]]
    print("loop lines", best)


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds:" + op
    if op.startswith("scratch_"): return op
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    if op in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rsq_f32"): return "valu:trans"
    if op.startswith("v_accvgpr") : return "valu:accvgpr"
    if op.startswith("v_cvt_pk_bf16"): return "valu:cvt_pk_bf16"
    if op.startswith("v_pk_"): return "valu:pk"
    if op.startswith("v_mov") : return "valu:mov"
    if op.startswith("v_"): return "valu:other"
    return "other"


c = collections.Counter()
ops = collections.Counter()
for l in body:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
        continue
    op = l.split()[0]
    c[cls(op)] += 1
    ops[op] += 1
tot_valu = sum(v for k, v in c.items() if k.startswith("valu"))
print("VALU total", tot_valu)
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v}")
if "--ops" in sys.argv:
    for k, v in ops.most_common(40):
        print(f"    {k:28s} {v}")
