#!/usr/bin/env python3
"""SURVEY.md section 8c fixture grid, recorded from the REAL reference (authoring container only; see make_golden.py):

    energy in {DoubleWell, Gaussian, Harmonic, 8-mode mixture (+ a dense-means mixture at dim 32)}
    x  sampler in {Langevin k = 16, HMC L = 5, HMC L = 20 (T = 8 transitions)}
    x  dim in {2, 32, 64, 100},  n = 1000 chains (not a multiple of 64),
    with scheduled step sizes and scalar / diagonal masses spread over the HMC cases.

Compact files (tests/golden/grid/*.pt): the noise is NOT stored -- the tests replay the seeded CPU generator in the
reference's draw order, exactly as this script does -- and of the reference's final state only the first 256 rows are
kept, next to the sha256 of the whole tensor, the population diagnostics (all 1000 chains) and the accept masks.

    python tests/golden/make_grid.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference, defines RefGMM / make_energy / sched_values / sha)

import torch  # noqa: E402
from torchebm.core import LinearScheduler  # noqa: E402
from torchebm.samplers import HamiltonianMonteCarlo, LangevinDynamics  # noqa: E402

OUT = os.path.join(HERE, "grid")
os.makedirs(OUT, exist_ok=True)
N, K, T, KEEP = 1000, 16, 8, 256


def energies(dim):
    g = torch.Generator().manual_seed(900 + dim)
    out = {"dw": {"kind": "double_well", "h": 2.0, "b": 1.0}, "har": {"kind": "harmonic", "k": 1.5}}
    if dim == 2:
        out["gauss"] = {"kind": "gaussian", "mean": torch.tensor([0.5, -0.25]), "cov": torch.tensor([[1.0, 0.8], [0.8, 1.0]])}
    else:
        a = torch.randn(dim, dim, generator=g)
        out["gauss"] = {"kind": "gaussian", "mean": torch.randn(dim, generator=g) * 0.5, "cov": a @ a.t() / dim + 0.5 * torch.eye(dim)}
    out["gmm8"] = {"kind": "gmm", "means": mg.ring_means(8, dim), "sigma": 1.0}
    if dim == 32:
        out["gmmd"] = {"kind": "gmm", "means": torch.randn(8, dim, generator=g) * 1.5, "sigma": 1.1}
    return out


def langevin(name, energy, dim, eta, sigma, seed, x0_scale):
    model = mg.make_energy(energy)
    x0 = torch.randn(N, dim, generator=torch.Generator().manual_seed(seed)) * x0_scale
    run_seed = seed + 1000
    s = LangevinDynamics(model, step_size=eta, noise_scale=sigma)
    x, diag = s.sample(x=x0.clone(), n_steps=K, thin=4, return_diagnostics=True, generator=torch.Generator().manual_seed(run_seed))
    fx = {"sampler": "langevin", "name": name, "energy": energy, "n": N, "dim": dim, "k": K, "thin": 4, "seed": seed,
          "run_seed": run_seed, "x0_scale": x0_scale, "etas": mg.sched_values(eta, K), "sigmas": mg.sched_values(sigma, K),
          "ref": {"x_rows": x[:KEEP].clone(), "sha_x": mg.sha(x), "diagnostics": diag}}
    torch.save(fx, os.path.join(OUT, name + ".pt"))
    print(f"{name:22s} sha {fx['ref']['sha_x']}")


def hmc(name, energy, dim, L, eps, seed, x0_scale, mass=None, min_margin=5e-5):
    import oracle
    from tests.helpers import oracle_energy

    for attempt in range(400):
        sd = seed + 1000 * attempt
        model = mg.make_energy(energy)
        x0 = torch.randn(N, dim, generator=torch.Generator().manual_seed(sd)) * x0_scale
        run_seed = sd + 500
        s = HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L, mass=mass)
        x, diag = s.sample(x=x0.clone(), n_steps=T, thin=2, return_diagnostics=True, generator=torch.Generator().manual_seed(run_seed))
        replay = torch.Generator().manual_seed(run_seed)
        ps, us = [], []
        for _ in range(T):
            ps.append(torch.empty(N, dim).normal_(generator=replay))
            us.append(torch.rand(N, generator=replay))
        eps_vals = mg.sched_values(eps, T)
        o = oracle.hmc_chain(oracle_energy(energy), x0, torch.stack(ps), torch.stack(us), eps_vals, L, mass=mass, thin=2, want_diag=True)
        assert torch.equal(o["x"], x), name  # the accept masks come from the oracle: it must BE the reference here
        if o["margin"] < min_margin:
            continue
        fx = {"sampler": "hmc", "name": name, "energy": energy, "n": N, "dim": dim, "T": T, "L": L, "thin": 2, "seed": sd,
              "run_seed": run_seed, "x0_scale": x0_scale, "mass": mass, "eps": eps_vals, "accepted": o["accepted"], "margin": o["margin"],
              "ref": {"x_rows": x[:KEEP].clone(), "sha_x": mg.sha(x), "diagnostics": diag}}
        torch.save(fx, os.path.join(OUT, name + ".pt"))
        print(f"{name:22s} sha {fx['ref']['sha_x']} margin {o['margin']:.1e} acc {diag['acceptance_rate'].tolist()}")
        return
    raise RuntimeError(f"{name}: no seed with margin > {min_margin}")


def main():
    seed = 7000
    for dim in (2, 32, 64, 100):
        for tag, en in energies(dim).items():
            seed += 1
            scale = {"dw": 0.8, "har": 1.0, "gauss": 1.0, "gmm8": 3.0, "gmmd": 1.5}[tag]
            eta = {"dw": 0.01, "har": 0.05, "gauss": 0.02, "gmm8": 0.05, "gmmd": 0.03}[tag]
            # one scheduled case per dim: the harmonic energy
            step = LinearScheduler(0.05, 0.01, 12) if tag == "har" else eta
            langevin(f"ld_{tag}_{dim}", en, dim, step, 1.0 if tag != "gauss" else 0.8, seed, scale)
            e5 = {"dw": 0.05, "har": 0.2, "gauss": 0.15, "gmm8": 0.15, "gmmd": 0.15}[tag]
            e20 = {"dw": 0.03, "har": 0.1, "gauss": 0.1, "gmm8": 0.1, "gmmd": 0.1}[tag]
            hmc(f"hmc5_{tag}_{dim}", en, dim, 5, e5, seed + 100, scale)
            # L = 20: scalar mass at dim 2, diagonal mass at dim 32 / 64, scheduled step size at dim 64 / 100
            mass = None
            if dim == 2:
                mass = 1.7
            elif dim in (32, 64):
                mass = torch.rand(dim, generator=torch.Generator().manual_seed(seed)) + 0.5
            step20 = LinearScheduler(e20, 0.6 * e20, 6) if dim in (64, 100) else e20
            hmc(f"hmc20_{tag}_{dim}", en, dim, 20, step20, seed + 200, scale, mass=mass)


def extra():
    """Round 3: dense Gaussians at widths the matrix-layout kernel does not take as is (below 20, not a multiple of 4) --
    it runs them as PACKED rows (csrc/gauss_mfma.hip: gauss_pack_factor).  Langevin only; new files, the grid above is
    not rewritten."""
    seed = 9000
    for dim in (5, 8, 12, 30, 50):
        seed += 1
        en = energies(dim)["gauss"]
        langevin(f"ld_gauss_{dim}", en, dim, 0.02, 0.8, seed, 1.0)


def extra_wide():
    """Round 3: dense Gaussians above 128 dims (csrc/gauss_big.hip: Ps streamed through LDS -- register-resident state at 160,
    tiled per step at 256) and HMC on a Gaussian at a width that is not a multiple of 4 / above 128 (lane-group kernels)."""
    seed = 9100
    for dim in (160, 256):
        seed += 1
        langevin(f"ld_gauss_{dim}", energies(dim)["gauss"], dim, 0.02, 0.8, seed, 1.0)
    for dim in (30, 256):
        seed += 1
        hmc(f"hmc5_gauss_{dim}", energies(dim)["gauss"], dim, 5, 0.15 if dim == 30 else 0.08, seed + 100, 1.0)


if __name__ == "__main__":
    if "--extra-wide" in sys.argv:
        extra_wide()
    elif "--extra" in sys.argv:
        extra()
    else:
        main()
