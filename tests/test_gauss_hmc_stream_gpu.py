"""HMC on dense Gaussians at dims 164 .. 256 in ONE ebm_hmc_chain_f32 launch (csrc/gauss_hmc_stream.hip, gauss_stream_e.h: the
matrix-layout transition body with an evaluation that streams the pre-split precision image through LDS).  Before round 4 these
widths ran per transition as library GEMMs.  Checked: its own Philox draws against the oracle fed with the materialised field
(no mass, scalar, diagonal; ragged last tiles and workgroups), injected momenta / uniforms with accept masks bit for bit where
the decision is not a rounding away from flipping, the records, the literal safe-mode path (taken per WORKGROUP here: the
evaluation has barriers inside), and that without the image the call still answers (the lane-group kernel)."""

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib, _rng

pytestmark = pytest.mark.gpu


def _field(shape, seed, steps, device, kind=None):
    kind = _lib.NOISE_NORMAL if kind is None else kind
    rows = []
    for st in steps:
        buf = torch.empty(shape, device=device)
        _lib.call("ebm_noise_fill_f32", buf.data_ptr(), buf.numel(), kind, seed, st, _lib.stream_handle(device))
        rows.append(buf)
    return torch.stack(rows)


def _gauss(dim, device, seed=0):
    g = torch.Generator().manual_seed(seed + dim)
    a = torch.randn(dim, dim, generator=g)
    mean, cov = torch.randn(dim, generator=g) * 0.5, a @ a.t() / dim + 0.5 * torch.eye(dim)
    return ta.GaussianModel(mean, cov, device=device), oracle.Gaussian(mean, cov), g


@pytest.mark.parametrize("dim", [164, 192, 200, 224, 256])
@pytest.mark.parametrize("mass", [None, 1.6, "diag"])
def test_native_draws_match_the_oracle_on_the_same_field(cuda_device, dim, mass):
    model, en, g = _gauss(dim, cuda_device)
    if mass == "diag":
        mass = torch.rand(dim, generator=g) + 0.5
    n, T, L, thin, eps = 150, 4, 5, 2, 0.06          # 150 chains: a full workgroup of 128 and a ragged one
    s = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L,
                                 mass=mass.to(cuda_device) if torch.is_tensor(mass) else mass, device=cuda_device)
    x0 = torch.randn(n, dim, generator=g).clamp_(-1.5, 1.5)
    seed = 7000 + dim
    c0 = hip_calls("ebm_hmc_chain_f32")
    traj = s.sample(x=x0.to(cuda_device), n_steps=T, thin=thin, return_trajectory=True,
                    generator=torch.Generator(device=cuda_device).manual_seed(seed))
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
    p = _field((n, dim), _rng.kernel_seed(seed), range(0, 2 * T, 2), cuda_device).cpu()
    u = _field((n,), _rng.kernel_seed(seed), range(1, 2 * T, 2), cuda_device, kind=_lib.NOISE_UNIFORM).cpu()
    want = oracle.hmc_chain(en, x0, p, u, [eps] * T, L, mass=mass, thin=thin, want_traj=True)
    err = ((traj.cpu() - want["trajectory"]).abs() / want["trajectory"].abs().clamp(min=1.0)).reshape(n, -1).amax(dim=1)
    if want["margin"] > 1e-4:
        assert (err <= 5e-4).all(), err.max().item()
    else:
        assert (err <= 5e-4).float().mean().item() >= 0.9


def _run(desc, x0, T, L, eps, dev, p=None, u=None, seed=0, offset=0):
    n, dim = x0.shape
    x = x0.to(dev)
    mask = torch.empty(T, n, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(T, dtype=torch.int32, device=dev)
    _lib.call("ebm_hmc_chain_f32", desc, x.data_ptr(), n, dim, T, L, eps, None, 0, 0.0, None, 1, None, None, mask.data_ptr(), cnt.data_ptr(),
              _lib.ptr(p), _lib.ptr(u), seed, offset, _lib.stream_handle(dev))
    torch.cuda.synchronize()
    return x.cpu(), mask.cpu().bool(), cnt.cpu()


@pytest.mark.parametrize("dim,n", [(164, 1), (192, 129), (256, 300)])
def test_injected_draws_accept_masks_and_the_general_kernel(cuda_device, dim, n):
    model, en, g = _gauss(dim, cuda_device, seed=1)
    T, L, eps = 5, 6, 0.07
    x0 = torch.randn(n, dim, generator=g)
    p, u = torch.randn(T, n, dim, generator=g), torch.rand(T, n, generator=g)
    want = oracle.hmc_chain(en, x0, p, u, [eps] * T, L, want_margins=True)
    spec = model.fused_spec()   # (kept alive: the descriptor points into its tensors)
    x, mask, cnt = _run(spec.to_c(), x0, T, L, eps, cuda_device, p=p.to(cuda_device), u=u.to(cuda_device))
    clear = want["margins"] > 2e-4
    assert torch.equal(mask[clear], want["accepted"][clear])
    assert torch.equal(cnt, mask.sum(dim=1).to(torch.int32))
    if bool(clear.all()):
        assert ((x - want["x"]).abs() / want["x"].abs().clamp(min=1.0)).max().item() <= 5e-4
    # without the image the same call runs the lane-group kernel: same decisions where they are clear
    plain = spec.to_c()
    plain.aux = None
    xg, mg, _ = _run(plain, x0, T, L, eps, cuda_device, p=p.to(cuda_device), u=u.to(cuda_device))
    assert torch.equal(mg[clear], want["accepted"][clear])
    # its own draws are the field ebm_noise_fill_f32 materialises
    seed, step0 = 99, 12
    pf = torch.stack([_field((n, dim), seed, [step0 + 2 * t], cuda_device)[0] for t in range(T)])
    uf = torch.stack([_field((n,), seed, [step0 + 2 * t + 1], cuda_device, kind=_lib.NOISE_UNIFORM)[0] for t in range(T)])
    a = _run(spec.to_c(), x0, T, L, eps, cuda_device, seed=seed, offset=step0)
    b = _run(spec.to_c(), x0, T, L, eps, cuda_device, p=pf, u=uf)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("dim", [192, 256])
def test_records_are_one_launch_and_the_same_chains(cuda_device, dim):
    model, _, _ = _gauss(dim, cuda_device, seed=2)
    n, T, L, eps = 515, 6, 5, 0.06
    s = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L, device=cuda_device)
    x0 = torch.randn(n, dim, device=cuda_device)
    c0 = hip_calls("ebm_hmc_chain_f32")
    traj, d = s.sample(x=x0, n_steps=T, thin=2, return_trajectory=True, return_diagnostics=True,
                       generator=torch.Generator(device=cuda_device).manual_seed(9))
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
    traj2 = s.sample(x=x0, n_steps=T, thin=2, return_trajectory=True, generator=torch.Generator(device=cuda_device).manual_seed(9))
    assert torch.equal(traj, traj2)
    for j in range(T // 2):
        torch.testing.assert_close(d["mean"][j], traj[:, j].mean(dim=0), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(d["var"][j], traj[:, j].var(dim=0, unbiased=False), rtol=1e-3, atol=1e-5)
        torch.testing.assert_close(d["energy"][j], model(traj[:, j]).mean(), rtol=1e-4, atol=1e-3)
    assert 0.3 < d["acceptance_rate"].mean().item() <= 1.0


def test_safe_mode_literal_path_is_taken_by_the_whole_workgroup(cuda_device):
    """A chain whose force leaves the +-1e6 clamp or whose coordinates are not finite sends ITS WORKGROUP through the literal
    sequence (integrators/leapfrog.py:165-185) -- every wave makes the same number of evaluations, whatever its own chains
    need.  Known answers: the oracle; the tame chains of the same workgroup are unaffected."""
    dim, n, T, L, eps = 192, 300, 3, 4, 0.05
    model, en, g = _gauss(dim, cuda_device, seed=3)
    x0 = torch.randn(n, dim, generator=g)
    x0[0, 9] = 5e7             # force beyond the clamp
    x0[1, 100] = 1e30          # the energy overflows to +inf (x_j P_jj x_j), the force is beyond the clamp
    x0[40, 0] = float("inf")   # another wave of the first workgroup
    x0[70, 191] = float("nan")
    x0[200, 5] = -4e7          # second workgroup
    p, u = torch.randn(T, n, dim, generator=g), torch.rand(T, n, generator=g)
    want = oracle.hmc_chain(en, x0, p, u, [eps] * T, L, want_margins=True)
    spec = model.fused_spec()
    x, mask, _ = _run(spec.to_c(), x0, T, L, eps, cuda_device, p=p.to(cuda_device), u=u.to(cuda_device))
    wild = torch.zeros(n, dtype=torch.bool)
    wild[[0, 1, 40, 70, 200]] = True
    clear = want["margins"] > 2e-4
    assert torch.equal(mask[clear], want["accepted"][clear])
    tame_ok = clear.all(dim=0) & ~wild
    assert torch.isfinite(x[~wild]).all()
    assert ((x[tame_ok] - want["x"][tame_ok]).abs() / want["x"][tame_ok].abs().clamp(min=1.0)).max().item() <= 5e-4
    assert torch.equal(torch.isnan(x[wild]), torch.isnan(want["x"][wild]))
    fin = torch.isfinite(want["x"][wild])
    assert torch.equal(torch.isfinite(x[wild]), fin)
    assert ((x[wild][fin] - want["x"][wild][fin]).abs() / want["x"][wild][fin].abs().clamp(min=1.0)).max().item() <= 1e-3


@pytest.mark.parametrize("dim,thin", [(192, 1), (224, 2), (256, 2), (128, 1), (160, 2)])
def test_mid_call_hand_over_to_the_literal_body(cuda_device, dim, thin):
    """(round 6) The one-launch kernel evaluates the force in pieces with merged kicks -- the fast sequence -- and a WORKGROUP that
    meets a non-finite energy or momentum hands the rest of its call, from the transition at hand, to the literal body (out of line,
    gauss_hmc_fallback; at dims 100 ... 160 -- the LDS-resident contraction, no barriers inside -- a single WAVE does).  Here one chain of the first workgroup draws an absurd momentum in transition 2 and one of the third in
    transition 4: accept decisions, final states and the kept trajectory rows of ALL chains -- before and after the hand-over, with
    the thinning counter resumed in mid-call -- are the oracle's (integrators/leapfrog.py:165-185, samplers/hmc.py:243-312)."""
    n, T, L, eps = 300, 6, 4, 0.05
    model, en, g = _gauss(dim, cuda_device, seed=5)
    x0 = torch.randn(n, dim, generator=g)
    p, u = torch.randn(T, n, dim, generator=g), torch.rand(T, n, generator=g)
    p[2, 5] *= 1e37      # x overflows in the first drift: E = inf (a later transition than the first: tr0 > 0)
    p[4, 290] *= 1e25    # the third workgroup, and at a kept / not kept boundary of thin = 2
    want = oracle.hmc_chain(en, x0, p, u, [eps] * T, L, thin=thin, want_traj=True, want_margins=True)
    spec = model.fused_spec()
    x = x0.to(cuda_device)
    mask = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
    cnt = torch.zeros(T, dtype=torch.int32, device=cuda_device)
    traj = torch.full((n, T // thin, dim), float("nan"), device=cuda_device)
    pd, ud = p.to(cuda_device), u.to(cuda_device)
    _lib.call("ebm_hmc_chain_f32", spec.to_c(), x.data_ptr(), n, dim, T, L, eps, None, 0, 0.0, None, thin, traj.data_ptr(), None,
              mask.data_ptr(), cnt.data_ptr(), pd.data_ptr(), ud.data_ptr(), 0, 0, _lib.stream_handle(cuda_device))
    torch.cuda.synchronize()
    x, mask, traj = x.cpu(), mask.cpu().bool(), traj.cpu()
    clear = want["margins"] > 2e-4
    assert torch.equal(mask[clear], want["accepted"][clear])
    assert not mask[2, 5] and not mask[4, 290]          # the absurd proposals are rejected, as in the reference
    assert torch.equal(cnt.cpu(), mask.sum(dim=1).to(torch.int32))
    ok = clear.all(dim=0)
    assert ok.float().mean().item() > 0.9
    assert torch.isfinite(x).all() and torch.isfinite(traj).all()
    assert ((x[ok] - want["x"][ok]).abs() / want["x"][ok].abs().clamp(min=1.0)).max().item() <= 5e-4
    wt = want["trajectory"]
    assert ((traj[ok] - wt[ok]).abs() / wt[ok].abs().clamp(min=1.0)).max().item() <= 5e-4
