"""The C-ABI library loads (no GPU needed) and exports every symbol include/ebm_hip.h declares."""

import ctypes
import os
import re

import pytest

from torchebm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ebm_hip.h")).read()
    return sorted(set(re.findall(r"EBM_API\s+[\w\s\*]+?\b(ebm_\w+)\s*\(", text)))


def test_header_declares_what_the_binding_lists():
    assert _declared_symbols() == sorted(_lib.EXPORTS)


def test_library_is_built_and_exports_every_symbol():
    assert _lib.is_built(), f"{_lib.LIB_PATH} missing: run __graft_entry__.build()"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(handle, name), name


def test_library_reads_no_environment():
    """The kernel-selection switches (EBM_GAUSS_ROWS ...) are compiled out of the shipped library (-DEBM_AB_SWITCHES builds
    only): no getenv import, no debug entry points -- include/ebm_hip.h promises no process-global state."""
    import subprocess

    out = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    undefined = {line.split()[-1].split("@")[0] for line in out.splitlines() if " U " in line}
    assert "getenv" not in undefined and "secure_getenv" not in undefined
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert not any(name.startswith("ebm_debug") for name in exported), exported


def test_version_and_error_string():
    lib = _lib.lib()
    assert lib.ebm_version() == _lib.ABI_VERSION
    assert isinstance(lib.ebm_last_error_string(), bytes)


def test_argument_errors_do_not_need_a_gpu():
    """Validation happens before any launch: NULL state -> EBM_EINVAL -> ValueError."""
    desc = _lib.EnergyDesc()
    desc.kind = _lib.ENERGY_DOUBLE_WELL
    with pytest.raises(ValueError, match="state pointer is NULL"):
        _lib.call("ebm_langevin_chain_f32", desc, None, 8, 4, 3, 0.1, 0.3, 1.0, None, 0, 0.0, 0.0, 1, None, None, None, 0, 0, None)
    desc.kind = 77
    with pytest.raises(RuntimeError, match="unknown energy kind"):
        _lib.call("ebm_langevin_chain_f32", desc, 16, 8, 4, 3, 0.1, 0.3, 1.0, None, 0, 0.0, 0.0, 1, None, None, None, 0, 0, None)
    with pytest.raises(ValueError, match="16-byte aligned"):
        _lib.call("ebm_noise_fill_f32", 4, 8, 0, 0, 0, None)


@pytest.mark.gpu
def test_plain_hip_program_drives_the_library_without_python(cuda_device, tmp_path):
    """examples/c_abi_demo.cpp: hipMalloc'ed buffers, ebm_energy_t by value, no torch anywhere -- links
    libebm_hip.so, runs the fused chain with native and with injected noise and finds them bit-identical."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "c_abi_demo")
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    build = subprocess.run(
        [hipcc, "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.cpp"),
         "-L", lib_dir, "-lebm_hip", f"-Wl,-rpath,{lib_dir}", "-o", exe],
        capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "c_abi_demo: OK" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]


def test_image_sizes_are_host_arithmetic():
    """ebm_*_image_bytes (ABI version 4) need no GPU: 0 = this shape has no image."""
    lib = _lib.lib()
    assert lib.ebm_mlp_w1_image_bytes(128, 128) == lib.ebm_mlp_w1_image_bytes(128, 65) == 4 * 3 * 32 * 128 * 2
    assert lib.ebm_mlp_w1_image_bytes(128, 64) == 0 and lib.ebm_mlp_w1_image_bytes(64, 128) == 0 and lib.ebm_mlp_w1_image_bytes(256, 128) == 0
    assert lib.ebm_gauss_prec_image_bytes(128) == 0 and lib.ebm_gauss_prec_image_bytes(130) == 0 and lib.ebm_gauss_prec_image_bytes(516) == 0
    assert lib.ebm_gauss_prec_image_bytes(256) == 2 * 8 * 3 * 1024 * 16          # tiled kernel's copy + resident kernel's copy
    assert lib.ebm_gauss_prec_image_bytes(512) == 2 * 16 * 3 * 1024 * 16         # two slices x 16 stages
    assert lib.ebm_gauss_prec_image_bytes(132) == 5 * 3 * 640 * 16 + 5 * 3 * 640 * 16
