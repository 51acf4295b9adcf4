"""Repeat an HMC grid fixture many times and count the calls whose accept mask / state differs from the first call's and
from the reference's (a race or an uninitialised read shows up as a nonzero count).  usage: stress_hmc_grid.py name [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import grid_inputs, load_grid, mass_to, package_model
from torchebm_amd import _lib
from torchebm_amd.integrators.symplectic import _mass_args
dev = torch.device("cuda")
name = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fx = load_grid(name)
x0, p, u = grid_inputs(fx)
n, dim, T, L, thin = fx["n"], fx["dim"], fx["T"], fx["L"], fx["thin"]
model = package_model(fx["energy"], dev)
desc = model.fused_spec().to_c()
eps = fx["eps"]
table = torch.tensor(eps, dtype=torch.float32, device=dev) if len(set(eps)) > 1 else None
p_d, u_d = p.to(dev), u.to(dev)
mass = mass_to(fx["mass"], dev)
layout = _lib.diag_layout(desc, _lib.DIAG_HMC, n, dim, True, False)
nb, S, E = layout
bad = {"mask_rec": 0, "mask_plain": 0, "x_rec": 0, "x_plain": 0}
first = {}
for rep in range(reps):
    for records in (True, False):
        junk = torch.full((1 << 22,), float("nan"), device=dev); del junk  # dirty the allocator's blocks
        x = x0.to(dev)
        mask = torch.empty(T, n, dtype=torch.uint8, device=dev)
        rec = torch.empty((T // thin) * nb * (2 * S + 8), device=dev) if records else None
        kind, ms, md = _mass_args(mass, x)
        _lib.call("ebm_hmc_chain_f32", desc, x.data_ptr(), n, dim, T, L, eps[0], _lib.ptr(table), kind, ms, _lib.ptr(md), thin, None,
                  _lib.ptr(rec), mask.data_ptr(), None, p_d.data_ptr(), u_d.data_ptr(), 0, 0, _lib.stream_handle(dev))
        m = mask.cpu().bool(); xc = x.cpu()
        key = "rec" if records else "plain"
        if not torch.equal(m, fx["accepted"]):
            bad["mask_" + key] += 1
            if bad["mask_" + key] <= 3:
                d = (m != fx["accepted"]).nonzero()
                print(name, key, "rep", rep, "mask differs at", d[:8].tolist(), "count", len(d))
        if key not in first: first[key] = xc
        elif not torch.equal(xc, first[key]):
            bad["x_" + key] += 1
            if bad["x_" + key] <= 3:
                d = (xc != first[key]).any(dim=1).nonzero().flatten()
                print(name, key, "rep", rep, "state differs in chains", d[:8].tolist(), "count", len(d))
print(name, "reps", reps, bad)
