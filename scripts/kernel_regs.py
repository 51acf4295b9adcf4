#!/usr/bin/env python3
"""VGPR / AGPR / spill / scratch / SGPR / static LDS of every kernel in the given objects (code-object metadata).
    scripts/kernel_regs.py build/csrc/mlp_wide.o [filter-substring]"""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
objs = [a for a in sys.argv[1:] if a.endswith(".o") or a.endswith(".so")]
flt = [a for a in sys.argv[1:] if a not in objs]
for o in objs:
    with tempfile.NamedTemporaryFile(suffix=".co") as t, tempfile.NamedTemporaryFile(suffix=".fat") as fat:
        subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat.name, o], check=True)
        r = subprocess.run([LLVM + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat.name,
                            "--output=" + t.name, "--unbundle"], capture_output=True, text=True)
        if r.returncode:
            print(o, r.stderr.strip())
            continue
        txt = subprocess.run([LLVM + "llvm-readelf", "--notes", t.name], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        def g(k):
            m = re.search(r"\." + k + r":\s*(\S+)", blk)
            return m.group(1) if m else "?"
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")))
        if flt and not all(f in name for f in flt):
            continue
        print(f"{name[-78:]:78s} vgpr {g('vgpr_count'):>4} agpr {blk.split()[0]:>4} spill {g('vgpr_spill_count'):>4} "
              f"scratch {g('private_segment_fixed_size'):>5} sgpr {g('sgpr_count'):>4}")
