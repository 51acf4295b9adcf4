#!/usr/bin/env python3
"""Benchmark of the Langevin hot path (BASELINE.json metric: MCMC chain-steps/sec).

A "step" is one ``LangevinDynamics.sample()`` call over the per-GPU batch of BASELINE
config 2: DoubleWell(h=2, b=1), n_chains = 2^20, dim = 64, k = 200 Euler-Maruyama steps,
eta = 0.01, sigma = 1, fp32, initial state already resident in HBM.  One call = one launch
of the fused kernel ``ebm_langevin_chain_f32`` = n_chains * k chain-steps.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The K timed steps continue the same chains (each call starts from the previous call's state).

N > 1: one process per GPU, chains sharded by rank (weak scaling: 2^20 chains per GPU, seed
base+rank, no collective inside the k steps).  The only exchange on the path is the read-back
of the final state: an RCCL all-gather of the [2^20, 64] shards, inside the timed region and
pipelined with the last step (four row blocks; block i's gather is in flight while block i+1 is
sampled -- utils.sample_and_gather).  It also runs once after the warm-up steps, untimed, so that
communicator set-up is not timed.

Output: one JSON line on rank 0 (see DESIGN.md §Measurement for the roofline accounting).
"""

import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: RCCL across processes needs this before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torchebm_amd as ta  # noqa: E402
from torchebm_amd import _lib  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
ETA, SIGMA = 0.01, 1.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    # workload overrides (tests / experiments); the defaults are BASELINE config 2
    ap.add_argument("--n-chains", type=int, default=1 << 20, help="chains PER GPU")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"], help="cpu = plumbing dry-run (gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(dim: int, k_full: int):
    """The reference's CPU path, restated (oracle/), timed on this box's host cores on a bounded
    sample of the same workload: DoubleWell, dim=64, n = 2^16 chains, k sized to ~6 s per run, per-step
    torch.randn + autograd gradient + the eager update ops (what the reference executes).
    torch's intra-op thread pool does not scale to hundreds of cores on ops this small, so a
    short probe picks the fastest thread count first; ``cores`` reports the one used."""
    import oracle

    n, k = 1 << 16, 10
    en = oracle.DoubleWell(2.0, 1.0)
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(n, dim, generator=g)

    def run(steps):
        x = x0
        for _ in range(steps):
            eps = torch.randn(n, dim, generator=g)
            x = oracle.em_step(x, en.grad_autograd(x), eps, ETA, SIGMA)
        return x

    ncpu = os.cpu_count() or 1
    best_threads, best_t = 1, float("inf")
    for th in sorted({t for t in (4, 8, 16, 32, 64, ncpu) if t <= ncpu}):
        torch.set_num_threads(th)
        run(1)
        t0 = time.perf_counter()
        run(2)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best_threads, best_t = th, dt
        if dt > 5.0:
            break
    torch.set_num_threads(best_threads)
    run(1)
    # size the timed sample to ~6 s per run (3 runs) from the probe's rate
    k = max(5, min(400, int(6.0 / (best_t / 2))))
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        run(k)
        times.append(time.perf_counter() - t0)
        if sum(times) > 20.0:
            break
    t = sorted(times)[len(times) // 2]
    return {
        "value": n * k / t,
        "unit": "chain-steps/s",
        "cores": best_threads,
        "kind": "port",
        "sample": f"oracle (torch CPU restatement of the reference loop: randn + autograd gradient + eager update), "
                  f"DoubleWell n=2^16 dim={dim} k={k}, median of {len(times)} runs ({t:.2f} s each) at the fastest of the "
                  f"probed torch thread counts ({best_threads} of {ncpu} cores); rate is per chain-step",
    }


def copy_ceiling_gbs(device, nbytes=1 << 28, reps=10):
    """Measured device-to-device copy rate (read + write bytes / time) on this box: the practical HBM
    ceiling quoted next to the 8 TB/s spec peak."""
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize(device)
    return 2 * nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9


def read_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), if present."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = args.device == "cuda"
    if world > 1:
        import torch.distributed as dist

        if on_gpu:
            torch.cuda.set_device(local_rank)
            # device_id binds the communicator to this rank's GPU up front (no lazy guess at the first collective)
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            except (TypeError, RuntimeError):
                if dist.is_initialized():
                    raise
                dist.init_process_group("nccl")
        else:
            dist.init_process_group("gloo")
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    device = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")

    n, dim, k = args.n_chains, args.dim, args.k
    model = ta.DoubleWellModel(barrier_height=2.0, b=1.0, device=device)
    sampler = ta.LangevinDynamics(model, step_size=ETA, noise_scale=SIGMA, device=device)
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    x0 = torch.randn(n, dim, device=device, generator=gen)

    gathered = None
    pieces = 4 if n % 4 == 0 else 1
    readback = ("none (single process)" if world == 1 else
                f"final state all-gathered inside the timed region, pipelined with the last step: {pieces} row blocks, "
                "the gather of block i in flight while block i+1 is sampled (utils.sample_and_gather)")
    state = x0

    def one_step():
        # the K steps form one sharded run: every call continues the rank's chains
        nonlocal state
        state = sampler.sample(x=state, n_steps=k, generator=gen)

    def last_step_with_readback():
        """The last step of the run together with the path's only collective -- the all-gather of the
        final [n, dim] shards -- overlapped block by block (N > 1).  Falls back to a plain step if the
        collective fails, and says so in config.readback."""
        nonlocal gathered, readback, state
        if world == 1 or readback.startswith("disabled"):
            one_step()
            return
        from torchebm_amd.utils import sample_and_gather

        try:
            state, gathered = sample_and_gather(sampler, state, k, pieces=pieces, generator=gen)
        except Exception as exc:  # report it, keep measuring the sharded compute
            readback = f"disabled after error: {type(exc).__name__}: {exc}"[:300]
            one_step()

    def fence():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    last_step_with_readback()  # untimed: also creates the RCCL communicator outside the timed region
    fence()
    state = x0
    if on_gpu:
        _lib.timed_events["ebm_langevin_chain_f32"] = []
    t0 = time.perf_counter()
    for _ in range(args.steps - 1):
        one_step()
    last_step_with_readback()  # step K
    fence()
    elapsed = time.perf_counter() - t0

    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = None
    if on_gpu:
        pairs = _lib.timed_events.pop("ebm_langevin_chain_f32")
        if pairs:  # GPU time of the chain kernel per step (the pipelined last step of N > 1 is several launches)
            kernel_ms = sum(a.elapsed_time(b) for a, b in pairs) / args.steps

    if rank == 0:
        chain_steps = world * n * k * args.steps
        value = chain_steps / elapsed
        algo_bytes = n * k * 8 * dim  # read x_t + write x_{t+1}, fp32, per chain-step
        roof = None
        if kernel_ms:
            achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
            traffic = read_traffic()
            roof = {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "measured_copy_ceiling": copy_ceiling_gbs(device),
                "traffic": None if not traffic else traffic.get("hbm_bytes_per_launch"),
                "kernel": "langevin_chain_lean_kernel<DoubleWell> (ebm_langevin_chain_f32)",
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": algo_bytes,
            }
        line = {
            "metric": "MCMC chain-steps/sec",
            "value": value,
            "unit": "chain-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"LangevinDynamics.sample on DoubleWell(h=2,b=1): n_chains={n} per GPU, dim={dim}, "
                            f"k={k} steps per call, eta={ETA}, sigma={SIGMA} "
                            + ("(BASELINE configs[1])" if (n, dim, k) == (1 << 20, 64, 200) else "(shape overridden on the command line)"),
                "n_chains_per_gpu": n,
                "dim": dim,
                "k_steps": k,
                "parallelism": f"chains sharded x{world}",
                "readback": readback,
                "device": args.device,
            },
            "roofline": roof,
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(dim, k),
        }
        print(json.dumps(line), flush=True)

    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
