"""Round 6: WHERE a kernel's vector instructions sit relative to its MFMAs is a property the compiler can silently undo -- round 5's
config-5 kernel had both of its tile-major tails (96 of 384 MFMA gaps) emptied by LLVM's code sinking and a third of its vector work
running with the matrix pipe idle, with every numerical test green.  This test compiles the unit that holds config 5's chain kernel to
gfx950 assembly (hipcc cross-compiles without a GPU; ~30 s) and checks the gap histogram scripts/isa_gaps.py prints: the MFMA count, no
packed-f32 instruction behind an MFMA (+20 cycles each there: profiles/r06_mfma_valu_overlap.txt), few empty gaps, no gap so full that
its excess is exposed.  A failure here is a performance regression to look at with scripts/isa_gapmap.py, not a wrong result."""

import ast
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_config5_chain_kernel_keeps_its_epilogues_behind_its_mfmas(tmp_path):
    out = tmp_path / "thin.s"
    # the flags csrc/Makefile gives this unit
    subprocess.run([os.path.join(ROOT, "scripts", "asm_unit.sh"), "mlp_wide_thin.hip", str(out), "-fno-slp-vectorize", "-mllvm",
                    "-amdgpu-mfma-vgpr-form"], check=True, capture_output=True, timeout=600)
    rep = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_gaps.py"), str(out), "mlp_wide_chain_kernelILi4ELi1ELi4ELi1E"],
                         check=True, capture_output=True, text=True).stdout
    n_mfma = int(re.search(r"(\d+) MFMAs", rep).group(1))
    hist = ast.literal_eval(re.search(r"VALU instructions per gap: (\{.*\})", rep).group(1))
    packed = int(re.search(r"packed-f32 ops in gaps of <= 16 VALU: (\d+)", rep).group(1))
    assert n_mfma == 384, rep                       # 2 contractions x 4 tiles x 8 K-blocks x 6 terms
    assert packed == 0, rep                         # element-wise arithmetic behind MFMAs
    assert hist.get("0", 0) <= 40, rep              # round 5: 94 (both tails empty); now the tails' nine trailing aux gaps + a few
    assert hist.get("9-16", 0) + hist.get("17-64", 0) <= 8, rep   # no slot so coarse that it overflows its gap
    assert hist.get("3-5", 0) + hist.get("6-8", 0) >= 280, rep    # the bulk: ~5 per gap


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_gaussian_hmc_stream_kernels_keep_their_state_in_registers():
    """The one-launch dense-Gaussian HMC kernels (dims 164 ... 256) were bound by their own spill traffic until round 6 (1 045
    spilled values per lane at eight tiles, 41 GB per launch): position, momentum and a whole force array were live together.  With
    the force evaluated in pieces (mfma_hmc_body.h PW) the allocator's problem fits -- a property the next refactoring can lose with
    every numerical test green.  Reads the code-object metadata of the built unit (build() leaves it under build/csrc/)."""
    obj = os.path.join(ROOT, "build", "csrc", "gauss_hmc_stream.o")
    if not os.path.exists(obj):
        pytest.skip("build/csrc/gauss_hmc_stream.o not there (python -c 'import __graft_entry__ as g; g.build()')")
    rep = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "kernel_regs.py"), obj], check=True, capture_output=True, text=True).stdout
    spills = {}
    for line in rep.splitlines():
        m = re.search(r"gauss_hmc_mfma_kernel<(\d), (true|false), ebm::GaussStreamE<\d>.*spill\s+(\d+)", line)
        if m:
            spills[(int(m.group(1)), m.group(2))] = int(m.group(3))
    assert len(spills) == 6, rep
    for (nt, mass), n in spills.items():
        bound = {6: 0, 7: 150, 8: 400}[nt]   # measured: 0 / 54-65 / ~150 (round 5: 0-250 / 700 / 1 045)
        assert n <= bound, (nt, mass, n, rep)
