"""ctypes binding of ``libebm_hip.so`` (C ABI: ``include/ebm_hip.h``).

This is the only place the package touches the shared library.  Everything is
pointer-and-size: tensors are passed as ``data_ptr()``, the stream as
``torch.cuda.current_stream().cuda_stream``.  There is no fallback of any kind:
if the library is missing or a call fails, a ``RuntimeError`` is raised.
"""

from __future__ import annotations

import ctypes as C
import os
from collections import Counter
from typing import Optional

import torch  # imported first on purpose: it loads the HIP runtime (libamdhip64.so.7) that our library links against

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libebm_hip.so")

ABI_VERSION = 8

# energy kinds / enums: keep in sync with include/ebm_hip.h
ENERGY_DOUBLE_WELL, ENERGY_HARMONIC, ENERGY_GAUSSIAN, ENERGY_GMM, ENERGY_MLP = 0, 1, 2, 3, 4
CHAIN_CLAMP, CHAIN_CONTRACTED = 1, 2  # the flag word `clamp_on` of the chain entries (ABI 8)
NOISE_NORMAL, NOISE_UNIFORM, NOISE_RAW_U32 = 0, 1, 2
MASS_NONE, MASS_SCALAR, MASS_DIAG = 0, 1, 2
DIAG_LANGEVIN, DIAG_LANGEVIN_HEUN, DIAG_HMC = 0, 1, 2

#: entry points declared in include/ebm_hip.h (tests check the library exports every one)
EXPORTS = (
    "ebm_version",
    "ebm_last_error_string",
    "ebm_langevin_step_f32",
    "ebm_langevin_step_diffusion_f32",
    "ebm_langevin_step_dev_f32",
    "ebm_langevin_chain_f32",
    "ebm_langevin_heun_chain_f32",
    "ebm_hmc_chain_f32",
    "ebm_hmc_chain_audit_f32",
    "ebm_leapfrog_kick_drift_f32",
    "ebm_leapfrog_kick_f32",
    "ebm_hmc_accept_f32",
    "ebm_hmc_accept_dev_f32",
    "ebm_descent_chain_f32",
    "ebm_descent_step_f32",
    "ebm_lookahead_f32",
    "ebm_pcd_gather_f32",
    "ebm_pcd_scatter_f32",
    "ebm_langevin_chain_dev_f32",
    "ebm_pcd_gather_dev_f32",
    "ebm_pcd_scatter_dev_f32",
    "ebm_pcd_start_points_f32",
    "ebm_cd_loss_work_bytes",
    "ebm_cd_loss_f32",
    "ebm_cd_loss_backward_f32",
    "ebm_energy_grad_f32",
    "ebm_mlp_backward_acts_f32",
    "ebm_mlp_param_grads_work_f32",
    "ebm_mlp_param_grads_f32",
    "ebm_chain_stats_f32",
    "ebm_noise_fill_f32",
    "ebm_noise_fill_dev_f32",
    "ebm_diag_layout",
    "ebm_diag_finish_f32",
    "ebm_probe_valu_f32",
    "ebm_probe_issue_f32",
    "ebm_mlp_w1_image_bytes",
    "ebm_mlp_w1_image_f32",
    "ebm_gauss_prec_image_bytes",
    "ebm_gauss_prec_image_f32",
    "ebm_gmm_active_columns_i32",
)

#: number of calls made through each entry point in this process (tests use it to
#: prove that a GPU test really went through the HIP library)
call_counts: Counter = Counter()


class EnergyDesc(C.Structure):
    """Mirror of ``ebm_energy_t``."""

    _fields_ = [
        ("kind", C.c_int32),
        ("n_comp", C.c_int32),
        ("s", C.c_float * 4),
        ("dev0", C.c_void_p),
        ("dev1", C.c_void_p),
        ("aux", C.c_void_p),
    ]


_f, _i32, _i64, _u64, _p, _d = C.c_float, C.c_int32, C.c_int64, C.c_uint64, C.c_void_p, C.c_double
_ENERGY_P = C.POINTER(EnergyDesc)

_PROTOTYPES = {
    "ebm_version": (C.c_int, []),
    "ebm_last_error_string": (C.c_char_p, []),
    "ebm_langevin_step_f32": (C.c_int, [_p, _p, _p, _p, _i64, _f, _f, _f, _i32, _f, _f, _u64, _u64, _p]),
    "ebm_langevin_step_diffusion_f32": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _f, _f, _u64, _u64, _p]),
    "ebm_langevin_step_dev_f32": (C.c_int, [_p, _p, _p, _i64, _f, _f, _f, _i32, _f, _f, _p, _p]),
    "ebm_langevin_chain_f32": (
        C.c_int,
        [_ENERGY_P, _p, _i64, _i32, _i32, _f, _f, _f, _p, _i32, _f, _f, _i32, _p, _p, _p, _u64, _u64, _p],
    ),
    "ebm_langevin_heun_chain_f32": (
        C.c_int,
        [_ENERGY_P, _p, _i64, _i32, _i32, _f, _f, _f, _p, _i32, _f, _f, _i32, _p, _p, _p, _u64, _u64, _p],
    ),
    "ebm_hmc_chain_f32": (
        C.c_int,
        [_ENERGY_P, _p, _i64, _i32, _i32, _i32, _f, _p, _i32, _d, _p, _i32, _p, _p, _p, _p, _p, _p, _u64, _u64, _p],
    ),
    "ebm_hmc_chain_audit_f32": (
        C.c_int,
        [_ENERGY_P, _p, _i64, _i32, _i32, _i32, _f, _p, _i32, _d, _p, _i32, _p, _p, _p, _p, _p, _u64, _u64, _p],
    ),
    "ebm_diag_layout": (C.c_int, [_ENERGY_P, _i32, _i64, _i32, _i32, _i32, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32)]),
    "ebm_diag_finish_f32": (C.c_int, [_p, _i32, _i64, _i32, _i32, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "ebm_leapfrog_kick_drift_f32": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i32, _f, _i32, _d, _p, _i32, _p]),
    "ebm_leapfrog_kick_f32": (C.c_int, [_p, _p, _p, _p, _i64, _f, _i32, _p]),
    "ebm_hmc_accept_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _u64, _u64, _p]),
    "ebm_hmc_accept_dev_f32": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i32, _p, _u64, _p]),
    "ebm_descent_chain_f32": (C.c_int, [_ENERGY_P, _p, _i64, _i32, _i32, _f, _p, _i32, _f, _i32, _p, _p]),
    "ebm_descent_step_f32": (C.c_int, [_p, _p, _p, _p, _i64, _f, _f, _p]),
    "ebm_lookahead_f32": (C.c_int, [_p, _p, _p, _i64, _f, _p]),
    "ebm_pcd_gather_f32": (C.c_int, [_p, _i64, _i32, _p, _i64, _i64, _p, _p, _u64, _u64, _p]),
    "ebm_pcd_scatter_f32": (C.c_int, [_p, _i64, _i32, _p, _i64, _i64, _p]),
    "ebm_langevin_chain_dev_f32": (
        C.c_int,
        [_ENERGY_P, _p, _i64, _i32, _i32, _f, _f, _f, _p, _i32, _f, _f, _i32, _p, _p, _u64, _p],
    ),
    "ebm_pcd_gather_dev_f32": (C.c_int, [_p, _i64, _i32, _p, _i64, _i64, _p, _p, _u64, _p]),
    "ebm_pcd_scatter_dev_f32": (C.c_int, [_p, _i64, _i32, _p, _i64, _p, _p]),
    "ebm_cd_loss_work_bytes": (C.c_int64, []),
    "ebm_cd_loss_f32": (C.c_int, [_p, _i64, C.c_float, _p, _p, _p, _p]),
    "ebm_cd_loss_backward_f32": (C.c_int, [_p, _i64, C.c_float, _p, _p, _p, _p]),
    "ebm_pcd_start_points_f32": (C.c_int, [_p, _i64, _i32, _p, _i64, _i64, _i64, C.c_float, _u64, _u64, _p, _p]),
    "ebm_energy_grad_f32": (C.c_int, [_ENERGY_P, _p, _i64, _i32, _p, _p, _p]),
    "ebm_mlp_backward_acts_f32": (C.c_int, [_ENERGY_P, _p, _i64, _i32, _p, _p, _p, _p, _p]),
    "ebm_mlp_param_grads_work_f32": (C.c_int64, [_i32, _i32, _i64]),
    "ebm_mlp_param_grads_f32": (C.c_int, [_p, _i64, _i32, _p, _i32, _p, _p, _p, _i64, _p, _p]),
    "ebm_chain_stats_f32": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _p]),
    "ebm_noise_fill_f32": (C.c_int, [_p, _i64, _i32, _u64, _u64, _p]),
    "ebm_noise_fill_dev_f32": (C.c_int, [_p, _i64, _i32, _p, _u64, _p]),
    "ebm_probe_valu_f32": (C.c_int, [_p, _i32, _i32, _p]),
    "ebm_probe_issue_f32": (C.c_int, [_p, _i32, _i32, _i32, _p]),
    "ebm_mlp_w1_image_bytes": (C.c_size_t, [_i32, _i32]),
    "ebm_mlp_w1_image_f32": (C.c_int, [_p, _i32, _i32, _p, _p]),
    "ebm_gauss_prec_image_bytes": (C.c_size_t, [_i32]),
    "ebm_gauss_prec_image_f32": (C.c_int, [_p, _i32, _p, _p]),
    "ebm_gmm_active_columns_i32": (C.c_int, [_p, _i32, _i32, _p, _p]),
}

_lib: Optional[C.CDLL] = None


def is_built() -> bool:
    return os.path.exists(LIB_PATH)


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"torchebm_amd: HIP library not found at {LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C torchebm_amd/csrc`). "
            "There is no CPU/PyTorch fallback for CUDA-device sampling."
        )
    handle = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _PROTOTYPES.items():
        fn = getattr(handle, name)
        fn.restype = restype
        fn.argtypes = argtypes
    got = handle.ebm_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"torchebm_amd: libebm_hip.so ABI version {got}, expected {ABI_VERSION}; rebuild it")
    _lib = handle
    return handle


def check(rc: int, name: str) -> None:
    if rc == 0:
        return
    msg = lib().ebm_last_error_string()
    text = msg.decode("utf-8", "replace") if msg else ""
    if rc == -1:
        raise ValueError(f"{name}: {text}")
    raise RuntimeError(f"{name} failed (code {rc}): {text}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device address of a tensor, or NULL."""
    return None if t is None else t.data_ptr()


class _StreamArg(int):
    """A ``hipStream_t`` value that remembers which device it belongs to (see ``call``)."""

    device_index: int = -1


def stream_handle(device: torch.device) -> int:
    """The current stream of ``device`` as the ``void* stream`` argument of the entry points."""
    h = _StreamArg(torch.cuda.current_stream(device).cuda_stream)
    h.device_index = device.index if device.index is not None else torch.cuda.current_device()
    return h


def dense_f32(t: torch.Tensor) -> torch.Tensor:
    """fp32, contiguous, 16-byte aligned view or copy of ``t`` (what the ABI requires)."""
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def diag_layout(energy: "EnergyDesc", sampler: int, n_chains: int, dim: int, injected_noise: bool = False,
                with_traj: bool = False) -> Optional[tuple]:
    """``(n_blocks, slots, block_elems)`` of the in-kernel diagnostics records for this chain call, or ``None``
    when the configuration has no in-kernel form (``ebm_diag_layout`` returned EBM_EDIM / EBM_EKIND)."""
    nb, sl, be = C.c_int64(0), C.c_int32(0), C.c_int32(0)
    rc = lib().ebm_diag_layout(energy, sampler, n_chains, dim, int(injected_noise), int(with_traj), C.byref(nb), C.byref(sl),
                               C.byref(be))
    if rc in (-2, -3):
        return None
    check(rc, "ebm_diag_layout")
    return nb.value, sl.value, be.value


#: when an entry-point name is a key here, every call to it is bracketed by a pair of
#: timing events recorded on the launch stream (bench.py reads kernel durations from it)
timed_events: dict = {}


def call(name: str, *args) -> None:
    """Invoke an int-returning entry point and raise on a non-zero status.

    The launch is made with the stream's own device current: a kernel enqueued on a stream of ``cuda:1``
    while ``cuda:0`` is current would go to a foreign-device stream, and ``hipGetLastError`` /
    the timing events would look at the wrong device."""
    fn = getattr(lib(), name)
    call_counts[name] += 1
    st = args[-1] if args else None
    if isinstance(st, _StreamArg) and st.device_index != torch.cuda.current_device():
        with torch.cuda.device(st.device_index):
            _call_here(fn, name, args)
        return
    _call_here(fn, name, args)


def _call_here(fn, name: str, args) -> None:
    pairs = timed_events.get(name)
    if pairs is None:
        check(fn(*args), name)
        return
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()  # current stream of the current device == the stream handed to the kernel (stream_handle)
    check(fn(*args), name)
    stop.record()
    pairs.append((start, stop))
