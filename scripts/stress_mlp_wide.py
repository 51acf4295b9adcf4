#!/usr/bin/env python3
"""Randomised parity sweep of the wide-MLP kernels (csrc/mlp_wide.hip, mlp_stream.hip, mlp_wide_hmc.hip) against the CPU
autograd network through the oracle's loops: random hidden width / input width (tile boundaries included) / batch /
mass form / thinning.  Prints one line per case and a summary; exit status 1 on any failure.
    python scripts/stress_mlp_wide.py [n_cases] [seed]"""
import copy, os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import torchebm_amd as ta  # noqa: E402
from torchebm_amd import _lib  # noqa: E402
from torchebm_amd.integrators.symplectic import _mass_args  # noqa: E402
from torchebm_amd.samplers.langevin import em_coefficients  # noqa: E402

dev = torch.device("cuda")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
EDGE = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 65, 80, 95, 96, 97, 100, 127, 128]


class CpuE:
    def __init__(self, m):
        self.m = m

    def energy(self, x):
        return self.m(x).detach()

    def grad(self, x):
        return self.m.gradient(x)


bad = 0
for case in range(n_cases):
    hidden = rng.choice([64, 128, 256])
    kind = rng.choice(["langevin", "hmc", "grad"])
    dims = [d for d in EDGE if kind != "hmc" or d <= ta.MLPEnergy.HMC_MAX_DIM[hidden]]
    dim = rng.choice(dims)
    n = rng.choice([1, 31, 32, 33, 100, 128, 129, 257, 515])
    torch.manual_seed(case)
    cpu = ta.MLPEnergy(dim, hidden)
    with torch.no_grad():
        for p in cpu.parameters():
            p.mul_(rng.choice([0.7, 1.0, 1.4]))
    gpu = copy.deepcopy(cpu).to(dev)
    spec = gpu.fused_spec()
    g = torch.Generator().manual_seed(1000 + case)
    x0 = torch.randn(n, dim, generator=g) * rng.choice([0.5, 1.0, 2.0])
    st = _lib.stream_handle(dev)
    ok, detail = True, ""
    try:
        if kind == "grad":
            e, gr = torch.empty(n, device=dev), torch.empty(n, dim, device=dev)
            xd = x0.to(dev)
            _lib.call("ebm_energy_grad_f32", spec.to_c(), xd.data_ptr(), n, dim, e.data_ptr(), gr.data_ptr(), st)
            we, wg = cpu(x0).detach(), cpu.gradient(x0)
            sc = max(wg.abs().max().item(), 1.0)
            ok = torch.allclose(e.cpu(), we, rtol=3e-5, atol=3e-5) and torch.allclose(gr.cpu(), wg, rtol=3e-4, atol=3e-5 * sc)
            detail = f"dE {float((e.cpu() - we).abs().max()):.2e} dg {float((gr.cpu() - wg).abs().max()):.2e}"
        elif kind == "langevin":
            k, thin = rng.choice([(6, 1), (8, 2), (9, 3)])
            eta, sigma = 0.04, rng.choice([0.0, 0.5, 1.0])
            noise = torch.randn(k, n, dim, generator=g)
            want, rows = x0, []
            for i in range(k):
                want = oracle.em_step(want, cpu.gradient(want), noise[i], eta, sigma)
                if (i + 1) % thin == 0:
                    rows.append(want)
            x = x0.to(dev)
            a, sq, coef = em_coefficients(eta, sigma)
            traj = torch.empty(n, k // thin, dim, device=dev)
            nz = noise.to(dev)
            _lib.call("ebm_langevin_chain_f32", spec.to_c(), x.data_ptr(), n, dim, k, a, sq, coef, None, 0, 0.0, 0.0, thin, traj.data_ptr(),
                      None, nz.data_ptr(), 0, 0, st)
            ok = torch.allclose(x.cpu(), want, rtol=1e-3, atol=1e-3) and torch.allclose(traj.cpu(), torch.stack(rows, 1), rtol=1e-3, atol=1e-3)
            detail = f"dx {float((x.cpu() - want).abs().max()):.2e}"
        else:
            T, L, eps = rng.choice([(2, 3), (4, 5), (3, 8)]) + (0.05,)
            thin = rng.choice([1, 2])
            mass = rng.choice([None, 1.6, "diag"])
            p = torch.randn(T, n, dim, generator=g)
            u = torch.rand(T, n, generator=g)
            if mass == "diag":
                mass = torch.rand(dim, generator=g) + 0.5
            want = oracle.hmc_chain(CpuE(cpu), x0, p, u, [eps] * T, L, mass=mass, thin=thin, want_traj=True)
            x = x0.to(dev).clone()
            mk, ms, md = _mass_args(mass.to(dev) if torch.is_tensor(mass) else mass, x)
            nk = T // thin
            traj = torch.empty(n, max(nk, 1), dim, device=dev)
            mask = torch.empty(T, n, dtype=torch.uint8, device=dev)
            pd, ud = p.to(dev).contiguous(), u.to(dev).contiguous()
            _lib.call("ebm_hmc_chain_f32", spec.to_c(), x.data_ptr(), n, dim, T, L, eps, None, mk, ms, _lib.ptr(md), thin,
                      traj.data_ptr() if nk else None, None, mask.data_ptr(), None, pd.data_ptr(), ud.data_ptr(), 0, 0, st)
            agree = (mask.cpu().bool() == want["accepted"]).all(dim=0)
            frac = agree.float().mean().item()
            ok = frac >= (0.98 if n >= 100 else 0.9)
            if nk:
                err = ((traj.cpu() - want["trajectory"]).abs() / want["trajectory"].abs().clamp(min=1.0)).reshape(n, -1).amax(dim=1)
                ok = ok and bool((err[agree] <= 3e-3).all())
            detail = f"agree {frac:.3f}"
    except Exception as exc:  # noqa: BLE001
        ok, detail = False, f"{type(exc).__name__}: {exc}"[:200]
    bad += not ok
    print(f"{'ok  ' if ok else 'FAIL'} case {case:3d} {kind:8s} H={hidden:3d} dim={dim:3d} n={n:3d} {detail}", flush=True)
print(f"{n_cases - bad} / {n_cases} passed")
sys.exit(1 if bad else 0)
