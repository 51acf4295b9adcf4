#!/usr/bin/env python3
"""End-to-end timings of BASELINE.json configs 3, 4 (one GPU's shard) and 5 through the public API
(sampler.sample / ContrastiveDivergence), one JSON line each.  Config 2 is bench.py itself.

    python scripts/bench_configs.py [c3] [c4] [c5]
"""
import json
import os
import sys
import time

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torchebm_amd as ta  # noqa: E402
from torchebm_amd.utils.synthetic import two_moons  # noqa: E402

dev = torch.device("cuda")
want = sys.argv[1:] or ["c3", "c4", "c5"]


def wall(fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


if "c3" in want:
    # HMC, L=20, 8-mode mixture, n=2^18, dim=32, eps=0.1, 50 MH steps
    n, dim, T, L = 1 << 18, 32, 50, 20
    model = ta.core.ring_mixture(8, dim, device=dev)
    s = ta.HamiltonianMonteCarlo(model, step_size=0.1, n_leapfrog_steps=L, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1234)
    x0 = torch.randn(n, dim, device=dev, generator=gen)
    t = wall(lambda: s.sample(x=x0, n_steps=T, generator=gen), reps=5, warm=1)
    _, d = s.sample(x=x0, n_steps=T, thin=T, return_diagnostics=True, generator=gen)
    flops_per_mh = (L + 1) * 1100 + 2 * 0 + L * 6 * dim  # ~1.1 kflop per mixture gradient+energy eval
    print(json.dumps({
        "config": "c3 HMC L=20 GMM-8 n=2^18 dim=32, 50 MH steps/call", "s_per_call": t, "mh_steps_per_s": n * T / t,
        "leapfrog_steps_per_s": n * T * L / t, "grad_evals_per_s": n * T * (L + 1) / t,
        "approx_fp32_GFLOPs": n * T * flops_per_mh / t / 1e9, "algo_GBps": n * T * 8 * dim / t / 1e9,
        "acceptance_rate_last": d["acceptance_rate"][-1].item(),
    }), flush=True)

if "c4" in want:
    # one GPU's shard of config 4: Langevin DoubleWell, 2^20 chains, dim=128, k=500
    n, dim, k = 1 << 20, 128, 500
    s = ta.LangevinDynamics(ta.DoubleWellModel(device=dev), step_size=0.01, noise_scale=1.0, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1234)
    x0 = torch.randn(n, dim, device=dev, generator=gen)
    t = wall(lambda: s.sample(x=x0, n_steps=k, generator=gen), reps=5, warm=1)
    print(json.dumps({
        "config": "c4 shard: Langevin DoubleWell n=2^20 (of 2^23 on 8 GPUs) dim=128 k=500", "s_per_call": t,
        "chain_steps_per_s_per_gpu": n * k / t, "algo_GBps": n * k * 8 * dim / t / 1e9,
        "frac_of_8TBps": n * k * 8 * dim / t / 8e12, "allgather_bytes_per_rank": n * dim * 4,
    }), flush=True)

if "c5" in want:
    # PCD training, MLP 2-128-128-1 SiLU on two-moons, n_chains = batch = buffer = 65536, k=20, eta=0.1
    class MLPEnergy(ta.core.BaseModel):
        def __init__(self):
            super().__init__()
            self.net = nn.Sequential(nn.Linear(2, 128), nn.SiLU(), nn.Linear(128, 128), nn.SiLU(), nn.Linear(128, 1))

        def forward(self, x):
            return self.net(x).squeeze(-1)

    torch.manual_seed(0)
    n, k = 65536, 20
    model = MLPEnergy().to(dev)
    data = two_moons(n, 0.05, seed=0, device=dev)
    s = ta.LangevinDynamics(model, step_size=0.1, noise_scale=1.0, device=dev)
    pcd = ta.ContrastiveDivergence(model, s, k_steps=k, persistent=True, buffer_size=n, init_steps=0, device=dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    def train_step():
        loss, _ = pcd(data)
        opt.zero_grad()
        loss.backward()
        opt.step()

    t = wall(train_step, reps=10, warm=3)
    t_sample = wall(lambda: s.sample(x=data, n_steps=k), reps=10, warm=2)
    # the same route with one iteration captured in a HIP graph and replayed k times per call
    s.capture_graph = True
    t_graph = wall(train_step, reps=10, warm=3)
    t_graph_sample = wall(lambda: s.sample(x=data, n_steps=k), reps=10, warm=2)
    s.capture_graph = False
    # the fused update alone (same 20 launches, gradient precomputed)
    from torchebm_amd import _lib
    x = data.clone()
    g = model.gradient(x)
    out = torch.empty_like(x)
    st = _lib.stream_handle(dev)

    def updates():
        for i in range(k):
            _lib.call("ebm_langevin_step_f32", x.data_ptr(), g.data_ptr(), out.data_ptr(), None, x.numel(), 0.1, 0.1**0.5,
                      2.0**0.5, 0, 0.0, 0.0, 1, i, st)

    t_upd = wall(updates, reps=20, warm=3)

    # the same training loop with the fused MLP-energy sampler (ta.MLPEnergy)
    torch.manual_seed(0)
    fmodel = ta.MLPEnergy(2, device=dev)
    fs = ta.LangevinDynamics(fmodel, step_size=0.1, noise_scale=1.0, device=dev)
    fpcd = ta.ContrastiveDivergence(fmodel, fs, k_steps=k, persistent=True, buffer_size=n, init_steps=0, device=dev)
    fopt = torch.optim.Adam(fmodel.parameters(), lr=1e-3)

    def ftrain_step():
        loss, _ = fpcd(data)
        fopt.zero_grad()
        loss.backward()
        fopt.step()

    tf = wall(ftrain_step, reps=10, warm=3)
    tf_sample = wall(lambda: fs.sample(x=data, n_steps=k), reps=10, warm=2)
    mlp_flops = n * k * 2 * (2 * 128 * 128 + 2 * 2 * 128)  # two HxH contractions + the two thin ones, per chain-step
    print(json.dumps({
        "config": "c5 PCD MLP 2-128-128-1 two-moons n=65536 k=20", "s_per_training_step": t, "training_steps_per_s": 1 / t,
        "chain_steps_per_s": n * k / t, "sampler_only_s": t_sample, "hip_update_kernels_only_s": t_upd,
        "update_share_of_sampler": t_upd / t_sample,
        "hip_graph": {"s_per_training_step": t_graph, "training_steps_per_s": 1 / t_graph, "sampler_only_s": t_graph_sample},
        "fused_mlp": {"s_per_training_step": tf, "training_steps_per_s": 1 / tf, "sampler_only_s": tf_sample,
                      "sampler_speedup_vs_autograd_route": t_sample / tf_sample,
                      "sampler_fp32_TFLOPs": mlp_flops / tf_sample / 1e12},
    }), flush=True)

if "ref" in want:
    # the reference's own benchmark scales (benchmarks/conftest.py:35-39, registry.py:141-148,368-370,679-684):
    # batch x dim x n_steps = 64x8x50 / 256x32x100 / 1024x128x200, DoubleWell(h=2), step 1e-3, HMC L=10.
    # These are launch-latency sized on an MI355X: what counts is microseconds per sample() call.
    for n, dim, k in ((64, 8, 50), (256, 32, 100), (1024, 128, 200)):
        model = ta.DoubleWellModel(barrier_height=2.0, device=dev)
        x0 = torch.randn(n, dim, device=dev)
        ld = ta.LangevinDynamics(model, step_size=1e-3, noise_scale=1.0, device=dev)
        hm = ta.HamiltonianMonteCarlo(model, step_size=1e-3, n_leapfrog_steps=10, device=dev)
        t_ld = wall(lambda: ld.sample(x=x0, n_steps=k), reps=50, warm=5)
        t_hm = wall(lambda: hm.sample(x=x0, n_steps=k), reps=20, warm=3)
        print(json.dumps({"config": f"reference benchmark scale {n}x{dim}x{k} (DoubleWell, step 1e-3)",
                          "langevin_us_per_call": t_ld * 1e6, "langevin_chain_steps_per_s": n * k / t_ld,
                          "hmc_L10_us_per_call": t_hm * 1e6, "hmc_mh_steps_per_s": n * k / t_hm}), flush=True)
