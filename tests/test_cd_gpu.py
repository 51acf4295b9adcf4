"""BASELINE config 5 on the GPU: persistent CD training of an MLP energy on two-moons.  The
gradient is autograd (PyTorch-ROCm), every Langevin step is one launch of the fused per-step HIP
kernel.  Parity: the negatives of one CD step are reproduced on the CPU by the oracle's
Euler-Maruyama restatement driven by autograd on the same MLP and fed the very noise field the
kernel drew (materialised with ebm_noise_fill_f32); the CD loss then agrees too."""

import copy

import numpy as np

import pytest
import torch
from torch import nn

import oracle
import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib, _rng
from torchebm_amd.utils.synthetic import two_moons

pytestmark = pytest.mark.gpu


class MLPEnergy(ta.core.BaseModel):
    def __init__(self, width=128):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(2, width), nn.SiLU(), nn.Linear(width, width), nn.SiLU(), nn.Linear(width, 1))

    def forward(self, x):
        return self.net(x).squeeze(-1)


def test_cd_negatives_and_loss_match_cpu_restatement(cuda_device):
    torch.manual_seed(0)
    model_cpu = MLPEnergy()
    model = copy.deepcopy(model_cpu).to(cuda_device)
    n, k, eta, sigma = 4096, 20, 0.1, 1.0
    data = two_moons(n, 0.05, seed=0)
    sampler = ta.LangevinDynamics(model, step_size=eta, noise_scale=sigma, device=cuda_device)
    cd = ta.ContrastiveDivergence(model, sampler, k_steps=k, persistent=False, device=cuda_device)
    gen = torch.Generator(device=cuda_device).manual_seed(2024)
    off = gen.get_offset() // 4
    before, dev0 = hip_calls("ebm_langevin_step_f32"), hip_calls("ebm_langevin_step_dev_f32")
    loss, neg = cd(data.to(cuda_device), generator=gen)
    # the HIP per-step kernel did the updates -- replayed from a HIP graph by default (the first, eager step + the
    # captured one pass through the binding, the k - 1 replays do not); same Philox field as eager launches
    assert hip_calls("ebm_langevin_step_dev_f32") == dev0 + 2 and hip_calls("ebm_langevin_step_f32") == before
    assert neg.is_cuda and not neg.requires_grad and neg.shape == (n, 2)

    # the same chain on the CPU, with the kernel's own noise
    x = data.clone()
    for i in range(k):
        eps = torch.empty(n, 2, device=cuda_device)
        _lib.call("ebm_noise_fill_f32", eps.data_ptr(), eps.numel(), _lib.NOISE_NORMAL, _rng.kernel_seed(2024), off + i,
                  _lib.stream_handle(cuda_device))
        x = oracle.em_step(x, model_cpu.gradient(x), eps.cpu(), eta, sigma)
    # 20 steps of an MLP gradient: rocBLAS vs CPU GEMM round-off, amplified by the dynamics
    torch.testing.assert_close(neg.cpu(), x, rtol=2e-3, atol=2e-3)
    cd_cpu = ta.ContrastiveDivergence(model_cpu, sampler=None, k_steps=k)
    want = cd_cpu.compute_loss(data, x)
    torch.testing.assert_close(loss.cpu(), want, rtol=2e-3, atol=2e-3)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_pcd_training_config5_shape(cuda_device):
    """Config 5's shape: n_chains = buffer = batch = 65 536, k = 20, a few optimiser steps; the
    replay buffer persists, the loss stays finite, parameters move."""
    torch.manual_seed(1)
    model = MLPEnergy().to(cuda_device)
    n, k = 65536, 20
    data = two_moons(n, 0.05, seed=0, device=cuda_device)
    sampler = ta.LangevinDynamics(model, step_size=0.1, noise_scale=1.0, device=cuda_device)
    pcd = ta.ContrastiveDivergence(model, sampler, k_steps=k, persistent=True, buffer_size=n, init_steps=0,
                                   device=cuda_device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in model.parameters()]
    calls0, dev0 = hip_calls("ebm_langevin_step_f32"), hip_calls("ebm_langevin_step_dev_f32")
    losses = []
    for _ in range(3):
        loss, neg = pcd(data)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    # one capture, then every training step replays it (the optimiser's in-place updates are seen by the replays)
    assert hip_calls("ebm_langevin_step_dev_f32") == dev0 + 2 and hip_calls("ebm_langevin_step_f32") == calls0
    assert all(torch.isfinite(l) for l in losses)
    assert pcd.replay_buffer.shape == (n, 2) and torch.isfinite(pcd.replay_buffer).all()
    assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))
    assert "replay_buffer" in pcd.state_dict()


# ---------------------------------------------------------------------------------------
# replay-buffer traffic in one launch each
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("cap,batch,dim", [(64, 16, 2), (1000, 37, 5), (4096, 4096, 3)])
def test_pcd_gather_injected_offsets_equals_torch_indexing(cuda_device, cap, batch, dim):
    buf = torch.randn(cap, dim, device=cuda_device)
    stride = cap // batch
    offs = torch.randint(0, stride, (batch,), device=cuda_device)
    out = torch.empty(batch, dim, device=cuda_device)
    rows = torch.empty(batch, dtype=torch.int64, device=cuda_device)
    _lib.call("ebm_pcd_gather_f32", buf.data_ptr(), cap, dim, out.data_ptr(), batch, stride, offs.data_ptr(), rows.data_ptr(),
              0, 0, _lib.stream_handle(cuda_device))
    want_rows = (torch.arange(batch, device=cuda_device) * stride + offs) % cap  # core/base_loss.py:306-312
    assert torch.equal(rows, want_rows) and torch.equal(out, buf[want_rows])


def test_pcd_gather_native_offsets_are_stratified_and_uniform(cuda_device):
    cap, batch, dim = 1 << 16, 1 << 12, 2
    stride = cap // batch
    buf = torch.arange(cap, device=cuda_device, dtype=torch.float32)[:, None].expand(cap, dim).contiguous()
    out = torch.empty(batch, dim, device=cuda_device)
    rows = torch.empty(batch, dtype=torch.int64, device=cuda_device)
    hist = torch.zeros(stride, device=cuda_device)
    for step in range(50):
        _lib.call("ebm_pcd_gather_f32", buf.data_ptr(), cap, dim, out.data_ptr(), batch, stride, None, rows.data_ptr(), 77, step,
                  _lib.stream_handle(cuda_device))
        lo = torch.arange(batch, device=cuda_device) * stride
        assert ((rows >= lo) & (rows < lo + stride)).all()  # one row per stratum
        assert torch.equal(out[:, 0], rows.float())
        hist += torch.bincount(rows - lo, minlength=stride).float()
    freq = hist / hist.sum()
    assert (freq - 1.0 / stride).abs().max().item() < 0.005  # 204 800 draws over 16 bins
    # reproducible per (seed, step); different steps differ
    r2 = torch.empty_like(rows)
    _lib.call("ebm_pcd_gather_f32", buf.data_ptr(), cap, dim, out.data_ptr(), batch, stride, None, r2.data_ptr(), 77, 49,
              _lib.stream_handle(cuda_device))
    assert torch.equal(rows, r2)


@pytest.mark.parametrize("cap,batch,pos", [(12, 5, 0), (12, 9, 5), (12, 12, 7), (1000, 333, 900)])
def test_pcd_scatter_equals_torch_fifo(cuda_device, cap, batch, pos):
    dim = 3
    buf = torch.randn(cap, dim, device=cuda_device)
    want = buf.clone()
    samples = torch.randn(batch, dim, device=cuda_device)
    end = (pos + batch) % cap  # core/base_loss.py:412-424
    if batch == cap:
        idx = (pos + torch.arange(batch, device=cuda_device)) % cap
        want[idx] = samples
    elif end > pos:
        want[pos:end] = samples
    else:
        first = cap - pos
        want[pos:] = samples[:first]
        want[:end] = samples[first:]
    _lib.call("ebm_pcd_scatter_f32", buf.data_ptr(), cap, dim, samples.data_ptr(), batch, pos, _lib.stream_handle(cuda_device))
    assert torch.equal(buf, want)


def test_pcd_loss_uses_the_buffer_kernels(cuda_device):
    model = MLPEnergy(16).to(cuda_device)
    sampler = ta.LangevinDynamics(model, step_size=0.1, device=cuda_device)
    pcd = ta.ContrastiveDivergence(model, sampler, k_steps=2, persistent=True, buffer_size=256, init_steps=0,
                                   new_sample_ratio=0.0, device=cuda_device)
    data = two_moons(64, 0.05, seed=1, device=cuda_device)
    g0, s0 = hip_calls("ebm_pcd_gather_f32"), hip_calls("ebm_pcd_scatter_f32")
    gen = torch.Generator(device=cuda_device).manual_seed(3)
    for _ in range(5):  # 5 x 64 rows into a 256-row FIFO: wraps once
        loss, neg = pcd(data, generator=gen)
    assert hip_calls("ebm_pcd_gather_f32") == g0 + 5 and hip_calls("ebm_pcd_scatter_f32") == s0 + 5
    assert pcd._write_pos == (5 * 64) % 256 and pcd.buffer_ptr.item() == pcd._write_pos
    assert torch.equal(pcd.replay_buffer[:64], neg)  # the last batch landed at rows 0..63 after the wrap
    assert torch.isfinite(loss)


# ------------------------------------------------------------------------------------------
# Round 5 (ABI 7): the chain starts with the exploration noise in ONE launch -- ebm_pcd_start_points_f32
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cap,batch,dim,n_noise", [(5000, 1000, 3, 50), (65536, 65536, 2, 3276), (4096, 1000, 1, 1), (777, 259, 5, 259),
                                                   (64, 3, 2, 1), (1 << 20, 1 << 17, 2, 6553), (100, 1, 4, 1), (4096, 1024, 7, 0)])
def test_pcd_start_points_equal_the_oracle(cuda_device, cap, batch, dim, n_noise):
    """Against the numpy restatement (oracle/pcd.py): the gathered rows and the noisy subset bit for bit (integer work), the noise
    itself to a few ulps of the normals (hardware log / sin / cos), and the device-coordinate form equal to the by-value form."""
    from oracle import pcd as oracle_pcd

    g = torch.Generator().manual_seed(cap + batch)
    buf = torch.randn(cap, dim, generator=g)
    buf_d = buf.to(cuda_device)
    stride, seed, step = cap // batch, 0x1234567 + cap, 40 + batch
    out = torch.empty(batch, dim, device=cuda_device)
    _lib.call("ebm_pcd_start_points_f32", buf_d.data_ptr(), cap, dim, out.data_ptr(), batch, stride, n_noise, 0.01, seed, step, None,
              _lib.stream_handle(cuda_device))
    want, rows, noisy = oracle_pcd.start_points(buf.numpy(), batch, stride, n_noise, 0.01, seed, step)
    assert int(noisy.sum()) == n_noise
    got = out.cpu().numpy()
    clean = torch.empty(batch, dim, device=cuda_device)
    rows_d = torch.empty(batch, dtype=torch.int64, device=cuda_device)
    _lib.call("ebm_pcd_gather_f32", buf_d.data_ptr(), cap, dim, clean.data_ptr(), batch, stride, None, rows_d.data_ptr(), seed, step,
              _lib.stream_handle(cuda_device))
    assert np.array_equal(rows_d.cpu().numpy(), rows)  # the same stratified rows as the plain gather at the same step
    changed = (got != clean.cpu().numpy()).any(axis=1)
    assert np.array_equal(changed | ~noisy, np.ones(batch, dtype=bool)) and not (changed & ~noisy).any()  # exactly the oracle's subset
    assert np.array_equal(got[~noisy], want[~noisy])
    np.testing.assert_allclose(got[noisy], want[noisy], rtol=0, atol=0.01 * 4e-6 * 8)  # the normals: a few fp32 ulps of |z| <= ~6
    # device-resident coordinates: {seed, step0} + offset
    rng = torch.tensor([seed, step - 7], dtype=torch.int64, device=cuda_device)
    out_dev = torch.empty_like(out)
    _lib.call("ebm_pcd_start_points_f32", buf_d.data_ptr(), cap, dim, out_dev.data_ptr(), batch, stride, n_noise, 0.01, 0, 7, rng.data_ptr(),
              _lib.stream_handle(cuda_device))
    assert torch.equal(out_dev, out)


def test_pcd_start_points_subset_is_uniform_over_rows_and_steps(cuda_device):
    """The LAW of the reference's ``randperm(batch)[:n_new]`` (core/base_loss.py:318-321): every row equally likely (chi-square over rows
    across 400 steps), pairs of rows independent up to the fixed subset size (adjacent and far pairs: co-selection rate n(n-1)/(b(b-1))),
    the noise N(0, 0.01^2) (KS), and a different subset at every step."""
    import scipy.stats as st

    cap = batch = 2048
    dim, n_noise, steps = 2, 205, 400
    buf = torch.zeros(cap, dim, device=cuda_device)
    out = torch.empty(batch, dim, device=cuda_device)
    hits = np.zeros(batch)
    pair_adj = pair_far = 0
    noise = []
    prev = None
    for s in range(steps):
        _lib.call("ebm_pcd_start_points_f32", buf.data_ptr(), cap, dim, out.data_ptr(), batch, 1, n_noise, 0.01, 99, 3 * s, None,
                  _lib.stream_handle(cuda_device))
        o = out.cpu().numpy()
        sel = (o != 0).any(axis=1)
        assert sel.sum() == n_noise
        assert prev is None or (sel != prev).any()
        prev = sel
        hits += sel
        pair_adj += (sel[:-1] & sel[1:]).sum()
        pair_far += (sel[: batch // 2] & sel[batch // 2:]).sum()
        if s < 40:
            noise.append(o[sel].ravel() / 0.01)
    p = n_noise / batch
    chi2 = ((hits - steps * p) ** 2 / (steps * p * (1 - p))).sum()  # ~ chi-square(batch - 1) (the fixed subset size costs one degree)
    assert st.chi2.sf(chi2, batch - 1) > 1e-4 and st.chi2.cdf(chi2, batch - 1) > 1e-4, chi2
    pp = n_noise * (n_noise - 1) / (batch * (batch - 1))
    for count, n_pairs in ((pair_adj, (batch - 1) * steps), (pair_far, batch // 2 * steps)):
        z = (count - n_pairs * pp) / np.sqrt(n_pairs * pp * (1 - pp))
        assert abs(z) < 4.5, (count, n_pairs * pp, z)
    assert st.kstest(np.concatenate(noise), "norm").pvalue > 1e-4


@pytest.mark.parametrize("n", [1, 5, 300, 65536, 100003])
@pytest.mark.parametrize("reg", [0.0, 0.01])
def test_cd_loss_kernels_equal_the_fp64_loss(cuda_device, n, reg):
    """ebm_cd_loss_f32 / ebm_cd_loss_backward_f32 (losses.cd._PairedCDLossHip) against the op-by-op loss of the reference
    (contrastive_divergence.py:141-155) evaluated in fp64: the value to one fp32 rounding of each mean (the kernel adds in fp64), the
    per-row gradient to two roundings; the same bits on every call (fixed-order partial sums; the workspace is shared by the calls)."""
    from torchebm_amd.losses.cd import _PairedCDLossHip

    g = torch.Generator(device=cuda_device).manual_seed(n)
    e = (torch.randn(2 * n, device=cuda_device, generator=g) * 3 + 1).requires_grad_(True)
    work = torch.zeros(int(_lib.lib().ebm_cd_loss_work_bytes()) // 8 + 1, dtype=torch.float64, device=cuda_device)
    vals, grads = [], []
    for _ in range(3):
        loss = _PairedCDLossHip.apply(e, n, reg, work)
        (ge,) = torch.autograd.grad(loss * 1.7, e)
        vals.append(loss.detach().clone())
        grads.append(ge)
    assert torch.equal(vals[0], vals[1]) and torch.equal(vals[1], vals[2]) and torch.equal(grads[0], grads[2])
    e64 = e.detach().double().requires_grad_(True)
    e2 = e64.view(2, n)
    want = e2[0].mean() - e2[1].mean() + reg * ((e2[0] ** 2).mean() + (e2[1] ** 2).mean())
    (g_want,) = torch.autograd.grad(want * 1.7, e64)
    scale = e2.abs().mean().item() + reg * (e2 ** 2).mean().item()
    assert abs(vals[0].item() - want.item()) <= 4 * 6e-8 * scale + 1e-12
    torch.testing.assert_close(grads[0].double(), g_want, rtol=4e-7, atol=1e-12)
    assert (work == 0).all() or int(work.view(torch.int32)[-4:].abs().sum()) == 0  # the ticket is back at zero (the partials are scratch)


def test_cd_loss_kernel_guard_on_a_non_finite_loss(cuda_device):
    from torchebm_amd.losses.cd import _PairedCDLossHip

    work = torch.zeros(int(_lib.lib().ebm_cd_loss_work_bytes()) // 8 + 1, dtype=torch.float64, device=cuda_device)
    bad = torch.randn(10, device=cuda_device)
    bad[2] = float("nan")
    bad.requires_grad_(True)
    out = _PairedCDLossHip.apply(bad, 5, 0.01, work)
    assert out.item() == pytest.approx(0.1)
    (gb,) = torch.autograd.grad(out, bad)
    assert (gb[torch.arange(10, device=cuda_device) != 2] == 0).all()  # the constant sends no gradient (contrastive_divergence.py:150-155)
    inf = torch.full((8,), 3e38, device=cuda_device, requires_grad=True)  # E^2 overflows fp32: the regulariser is inf
    out = _PairedCDLossHip.apply(inf, 4, 0.01, work)
    assert out.item() == pytest.approx(0.1)
    good = torch.randn(8, device=cuda_device, requires_grad=True)  # ... and the workspace is fine for the next call
    assert torch.isfinite(_PairedCDLossHip.apply(good, 4, 0.01, work))


def test_reference_subset_opt_in_draws_the_exploration_rows_as_the_reference_does(cuda_device):
    """Round 6: `loss.reference_subset = True` -- the exploration subset of a PCD step from
    `randperm(batch)[:n_new]` + `randn` on the caller's torch generator (core/base_loss.py:316-332) instead of the one-launch keyed
    bijection.  Given the same start rows the perturbed rows are then EXACTLY torch's: replayed here from a clone of the generator."""
    from torchebm_amd.utils import GraphedTrainingStep

    model = MLPEnergy(2).to(cuda_device)
    sampler = ta.LangevinDynamics(model, step_size=0.1, device=cuda_device)
    pcd = ta.ContrastiveDivergence(model, sampler, k_steps=1, persistent=True, buffer_size=2048, init_steps=0,
                                   new_sample_ratio=0.05, device=cuda_device)
    pcd.reference_subset = True
    data = two_moons(512, 0.05, seed=1, device=cuda_device)
    gen = torch.Generator(device=cuda_device).manual_seed(5)
    pcd(data, generator=gen)  # initialises the buffer
    c0, g0 = hip_calls("ebm_pcd_start_points_f32"), hip_calls("ebm_pcd_gather_f32")
    before = gen.clone_state() if hasattr(gen, "clone_state") else None
    starts = pcd.get_start_points(data, generator=gen)
    assert hip_calls("ebm_pcd_start_points_f32") == c0 and hip_calls("ebm_pcd_gather_f32") == g0 + 1  # gather: still one HIP launch
    n_new = max(1, int(512 * 0.05))
    # the rows that moved off their buffer values are n_new distinct rows, each by ~0.01 sigma
    stride = 2048 // 512
    moved = ((starts.view(512, 1, -1) - pcd.replay_buffer.view(512, stride, -1)).abs().amin(dim=1).amax(dim=1) > 0)
    assert int(moved.sum()) == n_new
    d = (starts.view(512, 1, -1) - pcd.replay_buffer.view(512, stride, -1)).abs().amin(dim=1)[moved]
    assert 0.0 < d.mean().item() < 0.05
    # ... and a captured training step refuses the torch-side sort
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
    with pytest.raises(ValueError, match="reference_subset"):
        GraphedTrainingStep(pcd, opt, generator=gen)
    del before
