"""Shared test helpers: golden fixtures, oracle/package energy construction."""

import glob
import os

import torch

import oracle
import torchebm_amd as ta

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.pt")))


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=False)


def oracle_energy(spec):
    kind = spec["kind"]
    if kind == "double_well":
        return oracle.DoubleWell(spec["h"], spec["b"])
    if kind == "harmonic":
        return oracle.Harmonic(spec["k"])
    if kind == "gaussian":
        return oracle.Gaussian(spec["mean"], spec["cov"])
    if kind == "gmm":
        return oracle.GaussianMixture(spec["means"], spec["sigma"])
    raise ValueError(kind)


def to64(energy):
    """The same oracle energy with its parameters upcast EXACTLY (the fp32 precision matrix / means / log-weights an fp32 run
    uses): what an fp64 referee run of the oracle evaluates (tests/golden/make_referee.py, the yardstick tests)."""
    for attr in ("mean", "cov_inv", "means", "log_weights"):
        if hasattr(energy, attr) and torch.is_tensor(getattr(energy, attr)):
            setattr(energy, attr, getattr(energy, attr).double())
    return energy


def yardstick(got, ref32, ref64, *, k_med, k_max=None, k_chain=None, k_q90=None, what=""):
    """`got` (the kernel, fp32) may be no further from the fp64 run than the REFERENCE's own fp32 run is, up to the factors given:
    population median, 90th percentile and maximum of the per-chain error, and per chain against that chain's own reference error
    (floored at the population's median reference error -- a chain whose fp32 run lands on the fp64 one to the last bit is no
    yardstick).  The maximum and the per-chain ratio of a few hundred chains are heavy-tailed where single chains amplify
    round-off (a mixture chain near a tie between two components): bars on them are for populations that do not do that."""
    err_ref = (ref32.double() - ref64).abs().amax(dim=1)
    err_hip = (got.double() - ref64).abs().amax(dim=1)
    floor = err_ref.median().clamp(min=1e-30)
    ratio = (err_hip / torch.maximum(err_ref, floor)).max().item()
    stats = {"what": what, "hip_med": err_hip.median().item(), "ref_med": err_ref.median().item(),
             "hip_q90": err_hip.quantile(0.9).item(), "ref_q90": err_ref.quantile(0.9).item(),
             "hip_max": err_hip.max().item(), "ref_max": err_ref.max().item(), "chain_ratio_max": ratio}
    assert stats["hip_med"] <= k_med * stats["ref_med"], stats
    if k_q90 is not None:
        assert stats["hip_q90"] <= k_q90 * stats["ref_q90"], stats
    if k_max is not None:
        assert stats["hip_max"] <= k_max * stats["ref_max"], stats
    if k_chain is not None:
        assert ratio <= k_chain, stats
    return stats


def package_model(spec, device=None):
    kind = spec["kind"]
    if kind == "double_well":
        return ta.DoubleWellModel(barrier_height=spec["h"], b=spec["b"], device=device)
    if kind == "harmonic":
        return ta.HarmonicModel(k=spec["k"], device=device)
    if kind == "gaussian":
        return ta.GaussianModel(spec["mean"], spec["cov"], device=device)
    if kind == "gmm":
        return ta.GaussianMixtureModel(spec["means"], sigma=spec["sigma"], device=device)
    raise ValueError(kind)


def mass_to(mass, device):
    return mass.to(device) if torch.is_tensor(mass) else mass


def hip_calls(name):
    from torchebm_amd import _lib

    return _lib.call_counts[name]


# ---- SURVEY 8c fixture grid (tests/golden/grid, made by tests/golden/make_grid.py) ---------------------------
GRID = os.path.join(GOLDEN, "grid")


def grid_names(prefix=""):
    return sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GRID, prefix + "*.pt")))


def load_grid(name):
    return torch.load(os.path.join(GRID, name + ".pt"), weights_only=False)


def grid_inputs(fx):
    """x0 and the draws the reference consumed, replayed from the fixture's seeds in the reference's draw order
    (Langevin: one randn per step; HMC: momentum normal_ then torch.rand per transition)."""
    n, dim = fx["n"], fx["dim"]
    x0 = torch.randn(n, dim, generator=torch.Generator().manual_seed(fx["seed"])) * fx["x0_scale"]
    g = torch.Generator().manual_seed(fx["run_seed"])
    if fx["sampler"] == "langevin":
        return x0, torch.stack([torch.randn(n, dim, generator=g) for _ in range(fx["k"])])
    ps, us = [], []
    for _ in range(fx["T"]):
        ps.append(torch.empty(n, dim).normal_(generator=g))
        us.append(torch.rand(n, generator=g))
    return x0, torch.stack(ps), torch.stack(us)


def sha16(t):
    import hashlib

    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()[:16]
