#!/usr/bin/env python3
"""HMC on the reference's benchmark MLP energy (benchmarks/registry.py:372-387) beyond the 2-D kernel: MLPEnergy (the
transition kernel of csrc/mlp_wide_hmc.hip / mlp_stream_hmc.hip: H 64 / 128 / 256 at dim <= 128) against the per-transition route on autograd (a subclass with
its own forward), eager and as a replayed HIP graph."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta

dev = torch.device("cuda")


class Sub(ta.MLPEnergy):
    def forward(self, x):
        return super().forward(x)


def wall(fn, reps=5, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


n, T, L = 65536, 10, 10
for dim, hidden in ((8, 128), (32, 128), (64, 128), (128, 128), (32, 64), (32, 256), (64, 256), (128, 256)):
    torch.manual_seed(0)
    fast = ta.MLPEnergy(dim, hidden, device=dev)
    slow = Sub(dim, hidden, device=dev)
    slow.load_state_dict(fast.state_dict())
    x = torch.randn(n, dim, device=dev)
    row = {"config": f"HMC on MLP {dim}-{hidden}-{hidden}-1, n={n}, L={L}, {T} transitions per call"}
    for name, model in (("mlp_energy", fast), ("autograd", slow)):
        for graph in (False, True):
            s = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=L, device=dev)
            s.capture_graph = graph
            row[f"{name}_{'graph' if graph else 'eager'}_ms_per_call"] = wall(lambda: s.sample(x=x, n_steps=T)) * 1e3
    s = ta.HamiltonianMonteCarlo(slow, step_size=0.05, n_leapfrog_steps=L, device=dev)
    s.capture_graph, s.carry_force = True, True   # opt-in: L + 1 gradient evaluations per transition instead of 2 L
    row["autograd_graph_carry_force_ms_per_call"] = wall(lambda: s.sample(x=x, n_steps=T)) * 1e3
    row["speedup_graph_routes"] = row["autograd_graph_ms_per_call"] / row["mlp_energy_graph_ms_per_call"]
    row["useful_TFLOPs"] = n * T * (L + 1) * 2 * (2 * hidden * hidden + 2 * dim * hidden) / (min(row["mlp_energy_graph_ms_per_call"], row["mlp_energy_eager_ms_per_call"]) * 1e-3) / 1e12
    row["mh_steps_per_s"] = n * T / (min(row["mlp_energy_graph_ms_per_call"], row["mlp_energy_eager_ms_per_call"]) * 1e-3)
    print(json.dumps(row), flush=True)
