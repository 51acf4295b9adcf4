"""The pre-split weight images (ABI version 4: ebm_gauss_prec_image_f32, ebm_mlp_w1_image_f32; handed to the kernels in
ebm_energy_t.aux).  They are HINTS: a kernel given the image moves ready-made bf16 slabs by LDS-direct loads, a kernel given
NULL loads, splits and stores fp32 rows itself (Gaussian) or runs the exact-f32 matrix instruction (MLP) -- same chains to the
tolerance both are held to against the oracle.  Checked here, through the C ABI:
  * the image's documented layout, decoded on the host: hi + mid + lo is the fp32 matrix (zero in the padding);
  * with and without the image: injected noise against the oracle, native draws against the materialised field (bit for bit),
    the energy / gradient entry;
  * a stale image is what the header says it is: the OLD matrix's chains."""

import numpy as np
import pytest
import torch

import oracle
import torchebm_amd as ta
from torchebm_amd import _lib
from torchebm_amd.samplers.langevin import em_coefficients

pytestmark = pytest.mark.gpu


def _gauss(dim, device, seed=0):
    g = torch.Generator().manual_seed(seed + dim)
    a = torch.randn(dim, dim, generator=g)
    cov = a @ a.t() / dim + 0.5 * torch.eye(dim)
    mean = torch.randn(dim, generator=g) * 0.5
    return ta.GaussianModel(mean, cov, device=device), oracle.Gaussian(mean, cov)


def _chain(desc, x, k, coef, noise=None, seed=0, step=0):
    n, dim = x.shape
    a, sq, c = coef
    _lib.call("ebm_langevin_chain_f32", desc, x.data_ptr(), n, dim, k, a, sq, c, None, 0, 0.0, 0.0, 1, None, None, _lib.ptr(noise), seed, step,
              _lib.stream_handle(x.device))


def _bf16_sum(words):
    """int32 words holding two bf16 each -> float32 values, in memory order."""
    return words.view(torch.bfloat16).float()


def _tiled_shape(dim):
    tiles = (dim + 31) // 32
    return (tiles, 1) if tiles <= 8 else ((tiles + 1) // 2, 2)


@pytest.mark.parametrize("dim", [132, 192, 256, 260, 448, 512])
def test_gaussian_image_layout_decodes_to_the_matrix(cuda_device, dim):
    model, _ = _gauss(dim, cuda_device)
    spec = model.fused_spec()
    assert spec.aux is not None and spec.aux.numel() * 4 == _lib.lib().ebm_gauss_prec_image_bytes(dim)
    P = spec.dev1.cpu()
    ot, ns = _tiled_shape(dim)
    kbs, d32 = 2, 32 * ((dim + 31) // 32)
    n_stage, units = d32 // (16 * kbs), ot * 64 * kbs
    tiled_words = ns * n_stage * 3 * units * 4
    img = spec.aux.cpu()
    t = _bf16_sum(img[:tiled_words]).view(ns, n_stage, 3, units, 8).sum(dim=2)          # hi + mid + lo
    want = torch.zeros(ns, n_stage, units, 8)
    u = torch.arange(units)
    it, kb2, ul = u // (64 * kbs), (u // 64) % kbs, u % 64
    for sl in range(ns):
        rows = sl * 32 * ot + 32 * it + (ul % 32)
        for s in range(n_stage):
            cols = (16 * kbs * s + 16 * kb2 + 8 * (ul // 32))[:, None] + torch.arange(8)[None, :]
            ok = (rows[:, None] < dim) & (cols < dim)
            want[sl, s] = torch.where(ok, P[rows.clamp(max=dim - 1)[:, None], cols.clamp(max=dim - 1)], torch.zeros(()))
    assert (t - want).abs().max().item() <= 2.0 ** -22 * P.abs().max().item()
    assert torch.equal(t[want == 0], torch.zeros_like(t[want == 0]))
    if ns == 1:  # the resident kernel's copy behind it: [stage][piece][tile][K-block][K-half][rotated slot]
        slabu = ot * 128
        r = _bf16_sum(img[tiled_words:]).view(ot, 3, slabu, 8).sum(dim=1)
        u = torch.arange(slabu)
        j, kb2, hh, slot = u // 128, (u // 64) % 2, (u // 32) % 2, u % 32
        rows = 32 * j + ((slot - 2 * (2 * kb2 + hh)) % 32)
        for s in range(ot):
            k0 = 32 * s + 16 * kb2 + 4 * hh
            cols = k0[:, None] + torch.tensor([0, 1, 2, 3, 8, 9, 10, 11])[None, :]
            ok = (rows[:, None] < dim) & (cols < dim)
            w = torch.where(ok, P[rows.clamp(max=dim - 1)[:, None], cols.clamp(max=dim - 1)], torch.zeros(()))
            assert (r[s] - w).abs().max().item() <= 2.0 ** -22 * P.abs().max().item(), s


@pytest.mark.parametrize("dim,n", [(164, 200), (224, 130), (256, 300), (320, 129), (512, 140)])
def test_gaussian_chain_with_and_without_the_image(cuda_device, dim, n):
    model, ref = _gauss(dim, cuda_device, seed=1)
    k, coef = 5, em_coefficients(0.02, 1.0)
    gen = torch.Generator().manual_seed(dim)
    x0 = torch.randn(n, dim, generator=gen)
    noise = torch.randn(k, n, dim, generator=gen)
    want, _, _ = oracle.langevin_chain(ref, x0, noise, [0.02] * k, [1.0] * k)
    spec = model.fused_spec()
    with_img, without = spec.to_c(), spec.to_c()
    without.aux = None
    for desc in (with_img, without):
        x = x0.to(cuda_device)
        _chain(desc, x, k, coef, noise=noise.to(cuda_device))
        torch.testing.assert_close(x.cpu(), want, rtol=2e-5, atol=2e-5)
        # native draws == the materialised field, whichever way the slabs arrive
        seed, step0 = 0x5EED0000 + dim, 17
        field = torch.empty(k, n, dim, device=cuda_device)
        for i in range(k):
            _lib.call("ebm_noise_fill_f32", field[i].data_ptr(), n * dim, _lib.NOISE_NORMAL, seed, step0 + i, _lib.stream_handle(cuda_device))
        xa, xb = x0.to(cuda_device), x0.to(cuda_device)
        _chain(desc, xa, k, coef, seed=seed, step=step0)
        _chain(desc, xb, k, coef, noise=field)
        assert torch.equal(xa, xb)
    # energy / gradient entry (one contraction of the tiled kernel)
    xq = x0.to(cuda_device)
    outs = []
    for desc in (with_img, without):
        e, g = torch.empty(n, device=cuda_device), torch.empty(n, dim, device=cuda_device)
        _lib.call("ebm_energy_grad_f32", desc, xq.data_ptr(), n, dim, e.data_ptr(), g.data_ptr(), _lib.stream_handle(cuda_device))
        outs.append((e.cpu(), g.cpu()))
    d = x0.double() - ref.mean.double()
    g64 = d @ ref.cov_inv.double()
    for e, g in outs:
        torch.testing.assert_close(g.double(), g64, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(e.double(), 0.5 * (d * g64).sum(dim=1), rtol=1e-5, atol=1e-4)


def test_gaussian_stale_image_runs_the_old_matrix(cuda_device):
    dim, n, k = 256, 128, 3
    old, _ = _gauss(dim, cuda_device, seed=2)
    new, ref_new = _gauss(dim, cuda_device, seed=3)
    new.mean.copy_(old.mean)
    gen = torch.Generator().manual_seed(9)
    x0, noise = torch.randn(n, dim, generator=gen), torch.randn(k, n, dim, generator=gen)
    coef = em_coefficients(0.02, 1.0)
    s_old, s_new = old.fused_spec(), new.fused_spec()
    mixed = s_new.to_c()
    mixed.aux = _lib.ptr(s_old.aux)         # the new matrix's fp32 rows, the OLD matrix's image
    xs = []
    for desc in (s_old.to_c(), mixed, s_new.to_c()):
        x = x0.to(cuda_device)
        _chain(desc, x, k, coef, noise=noise.to(cuda_device))
        xs.append(x.cpu())
    assert torch.equal(xs[0], xs[1]) and not torch.allclose(xs[1], xs[2], atol=1e-3)
    # a model whose matrix is modified in place rebuilds its image (the spec is keyed on the tensor's version)
    with torch.no_grad():
        old.cov_inv.copy_(new.cov_inv)
    x = x0.to(cuda_device)
    _chain(old.fused_spec().to_c(), x, k, coef, noise=noise.to(cuda_device))
    assert torch.equal(x.cpu(), xs[2])


@pytest.mark.parametrize("dim", [65, 96, 100, 128])
def test_mlp_w1_image_holds_the_first_layer(cuda_device, dim):
    model = ta.MLPEnergy(dim, 128, device=cuda_device)
    spec = model.fused_spec()
    assert spec.aux is not None and spec.aux.numel() * 4 == _lib.lib().ebm_mlp_w1_image_bytes(128, dim) == 4 * 3 * 32 * 128 * 2
    w1 = model.net[0].weight.detach().cpu()
    img = _bf16_sum(spec.aux.cpu()).view(4, 3, 32 * 128).sum(dim=1)   # slab, hi + mid + lo, [32 rows x 128 columns] in swizzled order
    for s in range(4):   # a slab holds exactly the rows 32 s .. 32 s + 31 (the order inside is the LDS swizzle: compare as multisets)
        want = torch.zeros(32, 128)
        want[:, :dim] = w1[32 * s : 32 * s + 32]
        got, _ = img[s].sort()
        ref, _ = want.reshape(-1).sort()
        assert (got - ref).abs().max().item() <= 2.0 ** -22 * w1.abs().max().item(), s


@pytest.mark.parametrize("dim,n", [(65, 300), (96, 257), (100, 129), (128, 515)])
def test_mlp_with_and_without_the_image(cuda_device, dim, n):
    torch.manual_seed(dim)
    model = ta.MLPEnergy(dim, 128, device=cuda_device)
    spec = model.fused_spec()
    with_img, without = spec.to_c(), spec.to_c()
    without.aux = None
    x0 = torch.randn(n, dim, device=cuda_device)
    # one evaluation against autograd in fp64
    net64 = ta.MLPEnergy(dim, 128, device=cuda_device, dtype=torch.float64)
    net64.load_state_dict({k_: v.double() for k_, v in model.state_dict().items()})
    xr = x0.double().requires_grad_(True)
    e64 = net64(xr)
    (g64,) = torch.autograd.grad(e64.sum(), xr)
    for desc in (with_img, without):
        e, g = torch.empty(n, device=cuda_device), torch.empty(n, dim, device=cuda_device)
        _lib.call("ebm_energy_grad_f32", desc, x0.data_ptr(), n, dim, e.data_ptr(), g.data_ptr(), _lib.stream_handle(cuda_device))
        torch.testing.assert_close(e.double(), e64.detach(), rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(g.double(), g64, rtol=2e-5, atol=2e-5)
    # a chain on its own draws: same field, same chains to the tolerance of the two contractions
    k, coef = 8, em_coefficients(0.05, 1.0)
    xa, xb = x0.clone(), x0.clone()
    _chain(with_img, xa, k, coef, seed=77, step=3)
    _chain(without, xb, k, coef, seed=77, step=3)
    torch.testing.assert_close(xa, xb, rtol=1e-4, atol=1e-4)
    assert not torch.equal(xa, x0)
    # ... and on injected noise with a kept LAST step's energy pass (the general kernel, records): drained slab in flight
    s = ta.LangevinDynamics(model, step_size=0.05, device=cuda_device)
    traj, d = s.sample(x=x0, n_steps=6, thin=3, return_trajectory=True, return_diagnostics=True,
                       generator=torch.Generator(device=cuda_device).manual_seed(5))
    plain = s.sample(x=x0, n_steps=6, generator=torch.Generator(device=cuda_device).manual_seed(5))
    assert torch.equal(traj[:, -1], plain)
    torch.testing.assert_close(d["energy"][-1], model(traj[:, -1]).mean(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dim,mass_kind", [(96, 0), (128, 0), (100, 2)])
def test_mlp_hmc_with_and_without_the_image(cuda_device, dim, mass_kind):
    """The transition kernel at H = 128 beyond dim 64: MODE 3 (image) and MODE 0 (aux = NULL: exact-f32 MFMA) on the same draws --
    the same accept decisions but for chains within rounding of a tie, the same positions to the tolerance of the two contractions."""
    torch.manual_seed(dim)
    model = ta.MLPEnergy(dim, 128, device=cuda_device)
    spec = model.fused_spec()
    n, T, L, eps = 300, 4, 5, 0.05
    x0 = torch.randn(n, dim, device=cuda_device)
    mass = (torch.rand(dim, device=cuda_device) + 0.5) if mass_kind == 2 else None
    outs = []
    for with_image in (True, False):
        d = spec.to_c()
        if not with_image:
            d.aux = None
        x = x0.clone()
        mask = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
        _lib.call("ebm_hmc_chain_f32", d, x.data_ptr(), n, dim, T, L, eps, None, mass_kind, 0.0, _lib.ptr(mass), 1, None, None, mask.data_ptr(),
                  None, None, None, 11, 4, _lib.stream_handle(cuda_device))
        outs.append((x.cpu(), mask.cpu().bool()))
    (xa, ma), (xb, mb) = outs
    same = (ma == mb).all(dim=0)
    assert same.float().mean().item() >= 0.98
    assert 0.3 < ma.float().mean().item() <= 1.0
    torch.testing.assert_close(xa[same], xb[same], rtol=2e-4, atol=2e-4)
    assert torch.isfinite(xa).all()
