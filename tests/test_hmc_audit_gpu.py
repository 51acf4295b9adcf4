"""the AUDIT instantiation of the HMC transition kernel (ebm_hmc_chain_audit_f32; csrc/hmc_kernel.h
leapfrog_literal): the reference's safe-mode leapfrog step literally -- separate half kicks, no fused multiply-add, the drift
divided by max(m, 1e-10) on every step, both scrubs (torchebm/integrators/leapfrog.py:156-185, samplers/hmc.py:243-312).
For the element-wise energies the state is the reference's BIT FOR BIT: torch.equal on every recorded quartic-well / harmonic
HMC fixture, the sha256 of the whole final state included.  What the fast body of ebm_hmc_chain_f32 trades for its speed is then
a measured quantity: its distance from these very states, and its time next to the audit kernel's."""

import time

import pytest
import torch

import torchebm_amd as ta
from helpers import grid_inputs, grid_names, load_grid, mass_to, package_model, sha16
from torchebm_amd import _lib
from torchebm_amd.integrators.symplectic import _mass_args

pytestmark = pytest.mark.gpu

ELEMENTWISE = [n for n in grid_names("hmc") if "_dw_" in n or "_har_" in n]


def _run(entry, fx, dev):
    x0, p, u = grid_inputs(fx)
    n, dim, T, L, thin = fx["n"], fx["dim"], fx["T"], fx["L"], fx["thin"]
    desc = package_model(fx["energy"], dev).fused_spec().to_c()
    eps = fx["eps"]
    table = torch.tensor(eps, dtype=torch.float32, device=dev) if len(set(eps)) > 1 else None
    p_d, u_d = p.to(dev), u.to(dev)
    x = x0.to(dev)
    mask = torch.empty(T, n, dtype=torch.uint8, device=dev)
    kind, ms, md = _mass_args(mass_to(fx["mass"], dev), x)
    args = [desc, x.data_ptr(), n, dim, T, L, eps[0], _lib.ptr(table), kind, ms, _lib.ptr(md), thin, None]
    if entry == "ebm_hmc_chain_f32":
        args.append(None)  # diag_partials
    args += [mask.data_ptr(), None, p_d.data_ptr(), u_d.data_ptr(), 0, 0, _lib.stream_handle(dev)]
    _lib.call(entry, *args)
    return x.cpu(), mask.cpu().bool()


def _oracle_with_exact_sqrt(fx):
    """The oracle's run of the fixture with sqrt(mass) CORRECTLY ROUNDED.  torch's CPU float32 sqrt is faithful, not exact: on
    near-halfway cases it lands on the wrong side (0.505 ulp), and on which inputs depends on the host (2 of the 32 masses of
    hmc20_har_32 where the fixtures were recorded, 6 of 32 on the GPU box's CPU) -- so for a diagonal mass the recorded momenta
    p = n sqrt(m) are the recording host's, not IEEE's.  The kernel's sqrtf is the correctly rounded one; this is its yardstick."""
    from unittest import mock

    import oracle
    from helpers import oracle_energy

    real = torch.sqrt
    exact = lambda t, *a, **k: real(t.double()).to(t.dtype) if t.dtype == torch.float32 else real(t, *a, **k)  # noqa: E731
    x0, p, u = grid_inputs(fx)
    with mock.patch.object(torch, "sqrt", exact):
        return oracle.hmc_chain(oracle_energy(fx["energy"]), x0, p, u, fx["eps"], fx["L"], mass=fx["mass"])


@pytest.mark.parametrize("name", ELEMENTWISE)
def test_audit_kernel_reproduces_the_reference_states_bit_for_bit(cuda_device, name):
    fx = load_grid(name)
    got, mask = _run("ebm_hmc_chain_audit_f32", fx, cuda_device)
    assert torch.equal(mask, fx["accepted"])
    if torch.is_tensor(fx["mass"]):
        # diagonal mass: against the oracle with IEEE sqrt(mass) -- all n x dim values; the RECORDED states differ from it only in
        # the columns whose sqrt(mass) the recording host rounded the other way (none for three of the four such fixtures)
        want = _oracle_with_exact_sqrt(fx)
        assert torch.equal(mask, want["accepted"]) and torch.equal(got, want["x"])
        off = (want["x"][:256] != fx["ref"]["x_rows"]).any(dim=0).nonzero().flatten().tolist()
        exact_sqrt = torch.sqrt(fx["mass"].double()).float()
        assert len(off) <= 4, off  # (hmc20_har_32: columns 1 and 14 where the fixtures were recorded)
        assert torch.equal((got[:256] != fx["ref"]["x_rows"]).any(dim=0).nonzero().flatten(), torch.tensor(off, dtype=torch.long))
        del exact_sqrt
    else:
        assert torch.equal(got[:256], fx["ref"]["x_rows"])
        assert sha16(got) == fx["ref"]["sha_x"]  # all n x dim values, not just the stored rows
    # the fast body on the same inputs: same decisions, states within the tolerance tier -- and HOW far, against the exact ones
    fast, mask_f = _run("ebm_hmc_chain_f32", fx, cuda_device)
    assert torch.equal(mask_f, fx["accepted"])
    rel = ((fast - got).abs() / got.abs().clamp(min=1.0)).amax(dim=1)
    assert (rel <= 5e-4).float().mean().item() >= 0.99 and rel.max().item() <= 5e-3, (name, rel.max().item())


def test_audit_kernel_native_rng_equals_the_oracle_on_the_materialised_field(cuda_device):
    """With its own draws the audit kernel is bit-identical to the oracle fed the same Philox field (the fast body is not)."""
    import oracle
    from torchebm_amd import _rng

    n, dim, T, L, eps, seed = 777, 24, 6, 7, 0.07, 4242
    x0 = (torch.randn(n, dim, generator=torch.Generator().manual_seed(1)) * 0.9)
    desc = ta.DoubleWellModel(device=cuda_device).fused_spec().to_c()
    x = x0.to(cuda_device)
    mask = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
    ks = _rng.kernel_seed(seed)
    _lib.call("ebm_hmc_chain_audit_f32", desc, x.data_ptr(), n, dim, T, L, eps, None, _lib.MASS_SCALAR, 1.7, None, 1, None,
              mask.data_ptr(), None, None, None, ks, 10, _lib.stream_handle(cuda_device))
    ps, us = [], []
    for t in range(T):
        pt, ut = torch.empty(n * dim, device=cuda_device), torch.empty(n, device=cuda_device)
        _lib.call("ebm_noise_fill_f32", pt.data_ptr(), n * dim, _lib.NOISE_NORMAL, ks, 10 + 2 * t, _lib.stream_handle(cuda_device))
        _lib.call("ebm_noise_fill_f32", ut.data_ptr(), n, _lib.NOISE_UNIFORM, ks, 10 + 2 * t + 1, _lib.stream_handle(cuda_device))
        ps.append(pt.view(n, dim).cpu())
        us.append(ut.cpu())
    want = oracle.hmc_chain(oracle.DoubleWell(), x0, torch.stack(ps), torch.stack(us), [eps] * T, L, mass=1.7)
    if want["margin"] > 1e-5:
        assert torch.equal(mask.cpu().bool(), want["accepted"])
        assert torch.equal(x.cpu(), want["x"])


def test_audit_kernel_refuses_what_it_is_not_built_for(cuda_device):
    x = torch.zeros(64, 8, device=cuda_device)
    mask = torch.empty(1, 64, dtype=torch.uint8, device=cuda_device)
    gm = ta.core.ring_mixture(8, 8, device=cuda_device).fused_spec().to_c()
    with pytest.raises(RuntimeError, match="element-wise"):
        _lib.call("ebm_hmc_chain_audit_f32", gm, x.data_ptr(), 64, 8, 1, 2, 0.1, None, 0, 0.0, None, 1, None, mask.data_ptr(), None, None, None,
                  0, 0, _lib.stream_handle(cuda_device))
    wide = torch.zeros(8, 300, device=cuda_device)
    dw = ta.DoubleWellModel(device=cuda_device).fused_spec().to_c()
    with pytest.raises(RuntimeError, match="dim 300"):
        _lib.call("ebm_hmc_chain_audit_f32", dw, wide.data_ptr(), 8, 300, 1, 2, 0.1, None, 0, 0.0, None, 1, None, None, None, None, None,
                  0, 0, _lib.stream_handle(cuda_device))


def test_audit_kernel_cost_next_to_the_fast_body(cuda_device, capsys):
    """The trade, measured: quartic well, 2^18 chains x 32, L = 20, 10 transitions (config 3's shape on an element-wise energy)."""
    n, dim, T, L = 1 << 18, 32, 10, 20
    desc = ta.DoubleWellModel(device=cuda_device).fused_spec().to_c()
    x0 = torch.randn(n, dim, device=cuda_device).clamp_(-2.0, 2.0)
    out = {}
    for entry in ("ebm_hmc_chain_f32", "ebm_hmc_chain_audit_f32"):
        extra = [None] if entry == "ebm_hmc_chain_f32" else []
        ts = []
        for rep in range(4):
            x = x0.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.call(entry, desc, x.data_ptr(), n, dim, T, L, 0.05, None, 0, 0.0, None, 1, None, *extra, None, None, None, None, 7, 0,
                      _lib.stream_handle(cuda_device))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out[entry] = (min(ts) * 1e3, x)
    fast_ms, audit_ms = out["ebm_hmc_chain_f32"][0], out["ebm_hmc_chain_audit_f32"][0]
    rel = ((out["ebm_hmc_chain_f32"][1] - out["ebm_hmc_chain_audit_f32"][1]).abs() / out["ebm_hmc_chain_audit_f32"][1].abs().clamp(min=1.0)).amax(dim=1)
    with capsys.disabled():
        print(f"\n[hmc audit] fast body {fast_ms:.3f} ms, literal body {audit_ms:.3f} ms ({audit_ms / fast_ms:.2f}x) per {T} transitions of "
              f"2^18 x 32, L = {L}; chains whose states differ by more than 5e-4: {(rel > 5e-4).float().mean().item():.2e}")
    assert audit_ms > fast_ms  # the literal sequence evaluates the gradient 2 L + 2 times per transition; the fast body L times


def test_sampler_exact_option_runs_the_literal_kernel(cuda_device):
    """`HamiltonianMonteCarlo(...).exact = True` -- the reference's operation sequence reachable from the API the
    reference's users call.  Element-wise energies: the literal kernel (the same states as the ABI entry, seed for seed; diagnostics and
    trajectories through the same launches); any other energy: the per-transition route on the reference's own torch operations."""
    from torchebm_amd import _rng

    n, dim, T, L, eps, seed = 1000, 32, 6, 9, 0.06, 99
    model = ta.DoubleWellModel(device=cuda_device)
    x0 = torch.randn(n, dim, generator=torch.Generator().manual_seed(3)).to(cuda_device)
    s = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L, mass=1.3, device=cuda_device)
    s.exact = True  # (an attribute: the constructor keeps the reference's signature -- its own tests pin `integrator` as the last parameter)
    assert s._route(x0, {})[0] == "fused"
    gen = torch.Generator(device=cuda_device).manual_seed(seed)
    got = s.sample(x=x0, n_steps=T, generator=gen)
    # the ABI entry on the same coordinates
    x = x0.clone()
    _lib.call("ebm_hmc_chain_audit_f32", model.fused_spec().to_c(), x.data_ptr(), n, dim, T, L, eps, None, _lib.MASS_SCALAR, 1.3, None, 1, None,
              None, None, None, None, _rng.kernel_seed(seed), 0, _lib.stream_handle(cuda_device))
    assert torch.equal(got, x)
    # ... and it is not the fast body's state (same draws, other arithmetic), though close
    fast = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L, mass=1.3, device=cuda_device).sample(
        x=x0, n_steps=T, generator=torch.Generator(device=cuda_device).manual_seed(seed))
    assert not torch.equal(fast, got) and (fast - got).abs().median().item() < 1e-5
    # diagnostics / trajectory with exact=True: same final state, acceptance in (0, 1]
    traj, diag = s.sample(x=x0, n_steps=T, thin=2, return_trajectory=True, return_diagnostics=True,
                          generator=torch.Generator(device=cuda_device).manual_seed(seed))
    assert torch.equal(traj[:, -1], got) and diag["acceptance_rate"].shape == (T // 2,)
    assert 0.0 < diag["acceptance_rate"].min().item() <= 1.0
    # no literal kernel for a dense Gaussian: the reference's torch operations per transition (never the fast fused body)
    g = ta.GaussianModel(torch.zeros(8), torch.eye(8), device=cuda_device)
    sg = ta.HamiltonianMonteCarlo(g, step_size=0.1, n_leapfrog_steps=5, device=cuda_device)
    sg.exact = True
    assert sg._route(torch.zeros(64, 8, device=cuda_device), {})[0] == "step"
    assert torch.isfinite(sg.sample(x=torch.zeros(64, 8, device=cuda_device), n_steps=3)).all()


def test_exact_and_fast_bodies_full_size_differing_accept_decisions(cuda_device, capsys):
    """The same call through both bodies at 2^18 x 32 (L = 20, 10 transitions, the kernels' own draws, one seed): REPORTED, not
    bounded -- the fraction of accept decisions that differ, per transition and overall, and of chains whose final states differ.
    (A decision differs where |u - a| is below what the two bodies' energies differ by; from the first differing decision on the
    two chains are different chains.)  The line goes to the test log and to gpurun_out/ when that exists."""
    import os

    n, dim, T, L, eps, seed = 1 << 18, 32, 10, 20, 0.05, 7
    desc = ta.DoubleWellModel(device=cuda_device).fused_spec().to_c()
    x0 = torch.randn(n, dim, generator=torch.Generator(device=cuda_device).manual_seed(5), device=cuda_device).clamp_(-2.0, 2.0)
    masks, states = {}, {}
    for entry in ("ebm_hmc_chain_f32", "ebm_hmc_chain_audit_f32"):
        extra = [None] if entry == "ebm_hmc_chain_f32" else []
        x = x0.clone()
        mask = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
        _lib.call(entry, desc, x.data_ptr(), n, dim, T, L, eps, None, 0, 0.0, None, 1, None, *extra, mask.data_ptr(), None, None, None,
                  seed, 0, _lib.stream_handle(cuda_device))
        masks[entry], states[entry] = mask.bool(), x
    differ = masks["ebm_hmc_chain_f32"] != masks["ebm_hmc_chain_audit_f32"]
    per_t = differ.float().mean(dim=1).tolist()
    first = differ.float().cumsum(dim=0).clamp_(max=1.0)  # chains that have diverged by transition t
    rel = ((states["ebm_hmc_chain_f32"] - states["ebm_hmc_chain_audit_f32"]).abs()
           / states["ebm_hmc_chain_audit_f32"].abs().clamp(min=1.0)).amax(dim=1)
    acc = masks["ebm_hmc_chain_audit_f32"].float().mean().item()
    line = (f"[hmc exact vs fast] 2^18 x {dim}, L = {L}, {T} transitions, eps = {eps}: acceptance {acc:.4f}; accept decisions that differ: "
            f"{differ.float().mean().item():.3e} overall, per transition {['%.1e' % v for v in per_t]}; chains with at least one differing "
            f"decision {first[-1].mean().item():.3e}; chains whose final states differ by more than 5e-4: {(rel > 5e-4).float().mean().item():.3e}, "
            f"by more than 1e-6: {(rel > 1e-6).float().mean().item():.3e}")
    with capsys.disabled():
        print("\n" + line)
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/hmc_exact_vs_fast.txt", "w") as f:
            f.write(line + "\n")
    assert differ.float().mean().item() < 0.05  # sanity only: the two bodies are the same sampler
