#!/usr/bin/env python3
"""Kernel-level timings through the C ABI (one JSON line per case): used to pick kernel variants
and to fill the per-kernel roofline table in DESIGN.md.  Run on the GPU box:

    python scripts/bench_kernels.py [case-substring ...]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torchebm_amd as ta  # noqa: E402
from torchebm_amd import _lib  # noqa: E402
from torchebm_amd.samplers.langevin import em_coefficients  # noqa: E402

dev = torch.device("cuda")
want = sys.argv[1:]


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def report(name, ms, best, units, unit_name, algo_bytes, **extra):
    rec = {
        "case": name, "ms_median": round(ms, 4), "ms_min": round(best, 4),
        unit_name + "_per_s": units / (ms * 1e-3),
        "algo_GBps": algo_bytes / (ms * 1e-3) / 1e9, "frac_of_8TBps": algo_bytes / (ms * 1e-3) / 8e12,
    }
    rec.update(extra)
    print(json.dumps(rec), flush=True)


def selected(name):
    return not want or any(w in name for w in want)


def chain_case(name, model, n, dim, k, clamp=None, thin=1, traj=False, table=False):
    if not selected(name):
        return
    spec = model.fused_spec()
    x0 = torch.randn(n, dim, device=dev).clamp_(-3, 3)
    x = x0.clone()
    a, sq, coef = em_coefficients(0.01, 1.0)
    tab = None
    if table:
        tab = torch.tensor([(a, sq, coef, 0.0)] * k, dtype=torch.float32, device=dev)
    tr = torch.empty(n, k // thin, dim, device=dev) if traj else None
    c_on, cmin, cmax = (0, 0.0, 0.0) if clamp is None else (1, clamp[0], clamp[1])
    st = _lib.stream_handle(dev)
    c = spec.to_c()

    def run():
        _lib.call("ebm_langevin_chain_f32", c, x.data_ptr(), n, dim, k, a, sq, coef, _lib.ptr(tab), c_on, cmin, cmax,
                  thin, _lib.ptr(tr), None, None, 1, 0, st)

    ms, best = timeit(run)
    report(name, ms, best, n * k, "chain_steps", n * k * 8 * dim, n=n, dim=dim, k=k)


def step_case(name, n_elem, noise_ptr=False):
    if not selected(name):
        return
    x = torch.randn(n_elem, device=dev)
    g = torch.randn(n_elem, device=dev)
    out = torch.empty_like(x)
    nz = torch.randn(n_elem, device=dev) if noise_ptr else None
    a, sq, coef = em_coefficients(0.01, 1.0)
    st = _lib.stream_handle(dev)

    def run():
        _lib.call("ebm_langevin_step_f32", x.data_ptr(), g.data_ptr(), out.data_ptr(), _lib.ptr(nz), n_elem, a, sq, coef,
                  0, 0.0, 0.0, 1, 0, st)

    ms, best = timeit(run, reps=20)
    bytes_ = n_elem * (16 if noise_ptr else 12)
    report(name, ms, best, n_elem, "elements", bytes_, n_elem=n_elem)


def hmc_case(name, model, n, dim, T, L, eps, mass=None):
    if not selected(name):
        return
    from torchebm_amd.integrators.symplectic import _mass_args

    spec = model.fused_spec()
    x = torch.randn(n, dim, device=dev).clamp_(-3, 3)
    kind, ms_, md = _mass_args(mass, x)
    st = _lib.stream_handle(dev)
    c = spec.to_c()

    def run():
        _lib.call("ebm_hmc_chain_f32", c, x.data_ptr(), n, dim, T, L, eps, None, kind, ms_, _lib.ptr(md), 1, None, None, None,
                  None, None, None, 1, 0, st)

    ms, best = timeit(run, reps=5, warm=1)
    report(name, ms, best, n * T, "mh_steps", n * T * 8 * dim, n=n, dim=dim, T=T, L=L,
           leapfrog_steps_per_s=n * T * L / (ms * 1e-3), grad_evals_per_s=n * T * (L + 1) / (ms * 1e-3))


def misc_cases():
    if selected("noise_fill"):
        n = 1 << 26
        out = torch.empty(n, device=dev)
        st = _lib.stream_handle(dev)
        ms, best = timeit(lambda: _lib.call("ebm_noise_fill_f32", out.data_ptr(), n, _lib.NOISE_NORMAL, 1, 0, st), reps=20)
        report("noise_fill_normal_2^26", ms, best, n, "elements", n * 4)
    if selected("chain_stats"):
        n, dim = 1 << 20, 64
        x = torch.randn(n, dim, device=dev)
        mean, var = torch.empty(dim, device=dev), torch.empty(dim, device=dev)
        work = torch.zeros(2 * dim + 1, dtype=torch.float64, device=dev)
        st = _lib.stream_handle(dev)

        def run():
            _lib.call("ebm_chain_stats_f32", x.data_ptr(), n, dim, mean.data_ptr(), var.data_ptr(), work.data_ptr(), st)

        ms, best = timeit(run, reps=20)
        report("chain_stats_2^20x64", ms, best, n, "rows", n * dim * 4)
    if selected("energy_grad"):
        n, dim = 1 << 20, 64
        x = torch.randn(n, dim, device=dev)
        e, g = torch.empty(n, device=dev), torch.empty(n, dim, device=dev)
        c = ta.DoubleWellModel(device=dev).fused_spec().to_c()
        st = _lib.stream_handle(dev)
        ms, best = timeit(lambda: _lib.call("ebm_energy_grad_f32", c, x.data_ptr(), n, dim, e.data_ptr(), g.data_ptr(), st), reps=20)
        report("energy_grad_dw_2^20x64", ms, best, n, "rows", 2 * n * dim * 4 + n * 4)


dw = ta.DoubleWellModel(device=dev)
chain_case("chain_dw_c2_lean", dw, 1 << 20, 64, 200)
chain_case("chain_dw_c2_clamp", dw, 1 << 20, 64, 200, clamp=(-10.0, 10.0))
chain_case("chain_dw_c2_table", dw, 1 << 20, 64, 200, table=True)
chain_case("chain_dw_c2_traj_thin50", dw, 1 << 20, 64, 200, thin=50, traj=True)
chain_case("chain_har_c2_lean", ta.HarmonicModel(device=dev), 1 << 20, 64, 200)
chain_case("chain_dw_c4shard_dim128", dw, 1 << 20, 128, 500)
gm = torch.Generator().manual_seed(0)
A = torch.randn(64, 64, generator=gm)
gauss64 = ta.GaussianModel(torch.zeros(64), (A @ A.t() / 64 + torch.eye(64)), device=dev)
chain_case("chain_gauss_dim64", gauss64, 1 << 18, 64, 50)
gauss2 = ta.GaussianModel(torch.zeros(2), torch.tensor([[1.0, 0.8], [0.8, 1.0]]), device=dev)
chain_case("chain_gauss_dim2", gauss2, 1 << 22, 2, 100)
chain_case("chain_gmm8_dim2", ta.core.ring_mixture(8, 2, device=dev), 1 << 22, 2, 100)
chain_case("chain_gmm8_dim32", ta.core.ring_mixture(8, 32, device=dev), 1 << 18, 32, 50)
step_case("step_native_rng_2^26", 1 << 26)
step_case("step_noise_ptr_2^26", 1 << 26, noise_ptr=True)
torch.manual_seed(0)
chain_case("chain_mlp_2_128", ta.MLPEnergy(2, 128, device=dev), 1 << 16, 2, 20)
chain_case("chain_mlp_32_128", ta.MLPEnergy(32, 128, device=dev), 1 << 16, 32, 20)
chain_case("chain_mlp_32_256", ta.MLPEnergy(32, 256, device=dev), 1 << 16, 32, 20)
hmc_case("hmc_mlp_32_128", ta.MLPEnergy(32, 128, device=dev), 1 << 16, 32, 10, 10, 0.05)
hmc_case("hmc_gmm8_c3", ta.core.ring_mixture(8, 32, device=dev), 1 << 18, 32, 10, 20, 0.1)
gd = torch.Generator().manual_seed(7)
hmc_case("hmc_gmm8_dense", ta.GaussianMixtureModel(torch.randn(8, 32, generator=gd) * 2.0, sigma=1.0, device=dev), 1 << 18, 32, 10, 20, 0.1)
hmc_case("hmc_dw_dim32", dw, 1 << 18, 32, 10, 20, 0.05)
hmc_case("hmc_dw_dim128", dw, 1 << 16, 128, 10, 10, 0.03)
hmc_case("hmc_gauss_dim64", gauss64, 1 << 16, 64, 10, 10, 0.1)
misc_cases()
