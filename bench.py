#!/usr/bin/env python3
"""Benchmark of the Langevin hot path (BASELINE.json metric: MCMC chain-steps/sec).

A "step" is one ``LangevinDynamics.sample()`` call over the per-GPU batch of BASELINE
config 2: DoubleWell(h=2, b=1), n_chains = 2^20, dim = 64, k = 200 Euler-Maruyama steps,
eta = 0.01, sigma = 1, fp32, initial state already resident in HBM.  One call = one launch
of the fused kernel ``ebm_langevin_chain_f32`` = n_chains * k chain-steps.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

``--gpus N`` with N > 1 and no torchrun environment (``WORLD_SIZE`` unset) launches the N ranks
ITSELF: the process re-executes this file under ``torch.distributed.run`` (127.0.0.1 rendezvous, a
free port) and relays rank 0's JSON line; it fails loudly when fewer than N GPUs are visible.

The K timed steps continue the same chains (each call starts from the previous call's state).

N > 1: one process per GPU, chains sharded by rank (weak scaling: 2^20 chains per GPU, seed
base+rank, no collective inside the k steps).  The only exchange on the path is the read-back
of the final state: an RCCL all-gather of the [2^20, 64] shards, inside the timed region and
pipelined with the last step (four row blocks; block i's gather is in flight while block i+1 is
sampled -- utils.sample_and_gather).  It also runs once after the warm-up steps, untimed, so that
communicator set-up is not timed.

Output: one JSON line on rank 0 (see DESIGN.md §Measurement for the roofline accounting).  At N = 1
the line also carries ``extra``: the other BASELINE configs and the genuinely HBM-bound per-step
kernel, measured in the same run (``--no-extra`` skips them).
"""

import argparse
import json
import os
import socket
import subprocess
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

N_CUS, CLOCK_GHZ = 256, 2.4  # MI355X_MICROARCH.md, chip-level parameters (4 SIMD-32 per CU, max clock)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); the copy ceiling is measured live next to it
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X fp32 vector (non-matrix) peak
FP32_MATRIX_PEAK_TFLOPS = 157.3  # exact-fp32 MFMA peak (v_mfma_f32_32x32x2_f32)
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF headline is 2:1 sparse)
ETA, SIGMA = 0.01, 1.0
# Issue cost of one Euler-Maruyama step of one float4 group in the lean DoubleWell loop, in units of one plain
# full-rate VALU op: 18 v_mad_u64_u32 x 2.6 + 20 v_bitop3_b32 + 8 transcendentals x 3.2 + 10 plain + 20 packed-f32 x 1.8
# (static ISA count of langevin_chain_lean_kernel<DoubleWell>; per-class costs measured by scripts/ubench/valu_rates.hip,
# profiles/r01_valu_rates.txt)
LEAN_LOOP_ISSUE_UNITS = 140.0  # round-1 constant, kept as the fallback and printed next to the value measured in the run
# static instruction mix of the lean loop per float4 group and step (scripts/isa_mix.py on langevin.hip)
LEAN_LOOP_MIX = {"mad_u64_u32": 18, "bitop3": 20, "transcendental": 8, "packed_f32": 20, "plain": 10}  # 76: SQ_INSTS_VALU says 76.1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)  # 0.7 s timed at config 2: long enough for an outside GPU-busy sampler
    ap.add_argument("--warmup", type=int, default=3)
    # workload overrides (tests / experiments); the defaults are BASELINE config 2
    ap.add_argument("--n-chains", type=int, default=None, help="chains PER GPU (default 2^20)")
    ap.add_argument("--dim", type=int, default=None, help="default 64 (--config 4: 128)")
    ap.add_argument("--k", type=int, default=None, help="default 200 (--config 4: 500)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 4],
                    help="4 = BASELINE configs[3]'s per-GPU shard as the timed step (2^20 chains, dim 128, k 500: at 8 ranks "
                         "the timed run IS configs[3]); default 2 = configs[1]")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"], help="cpu = plumbing dry-run (gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE configs (the `extra` array)")
    ap.add_argument("--no-traffic", action="store_true", help="do not re-run two launches under rocprofv3 --pmc for roofline.traffic "
                                                              "(the committed profiles/traffic.json is used instead)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)  # the process rocprofv3 wraps: three launches, no output
    args = ap.parse_args()
    # --config 4 = BASELINE configs[3]'s per-GPU shard: args.n_chains, args.dim, args.k = 1 << 20, 128, 500 unless given explicitly
    # (a reduced shard lets the 8-rank code path run on CPU ranks: tests/test_distributed.py)
    preset = (1 << 20, 128, 500) if args.config == 4 else (1 << 20, 64, 200)
    args.n_chains = preset[0] if args.n_chains is None else args.n_chains
    args.dim = preset[1] if args.dim is None else args.dim
    args.k = preset[2] if args.k is None else args.k
    return args


# ---------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without torchrun
# ---------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """Re-execute this file as N ranks under torch.distributed.run (what the driver's own N > 1 command
    line does) and relay the children's output.  Returns the launcher's exit code."""
    if args.device == "cuda":
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible "
                  f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')!r}); refusing to report a "
                  f"{args.gpus}-GPU number from fewer devices", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, EBM_BENCH_SELF_LAUNCHED="1")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only)
# ---------------------------------------------------------------------------------------
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
    probed = {}
    for th in sorted({t for t in (4, 8, 16, 32, 64, ncpu) if t <= ncpu}):
        torch.set_num_threads(th)
        run(1)
        t0 = time.perf_counter()
        run(2)
        dt = time.perf_counter() - t0
        probed[th] = n * 2 / dt
        if dt < best_t:
            best_threads, best_t = th, dt
        if dt > 5.0:
            break
    if ncpu not in probed:  # the whole machine is always reported (BASELINE.md section 4 names os.cpu_count() threads)
        torch.set_num_threads(ncpu)
        t0 = time.perf_counter()
        run(1)
        probed[ncpu] = n / (time.perf_counter() - t0)
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
        "all_cores": {"cores": ncpu, "value": probed[ncpu], "unit": "chain-steps/s",
                      "note": "torch.set_num_threads(os.cpu_count()), 2-step probe of the same loop"},
        "thread_probe": {str(th): round(v) for th, v in sorted(probed.items())},
        "sample": f"oracle (torch CPU restatement of the reference loop: randn + autograd gradient + eager update), "
                  f"DoubleWell n=2^16 dim={dim} k={k}, median of {len(times)} runs ({t:.2f} s each) at the fastest of the "
                  f"probed torch thread counts ({best_threads} of {ncpu} cores); rate is per chain-step.  BASELINE.md section 4 "
                  f"sketched os.cpu_count() threads at n=2^17: torch's intra-op pool is SLOWER at {ncpu} threads than at "
                  f"{best_threads} on ops this small (the probe above measures it), and n=2^16 keeps the default run short -- "
                  f"the rate per chain-step does not depend on n at these sizes",
    }


# ---------------------------------------------------------------------------------------
# live probes of the box
# ---------------------------------------------------------------------------------------
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


def plain_valu_rate(device, blocks=256 * 8, iters=4096, reps=5):
    """Plain-VALU issue rate of this chip at its current clock, in wave-instructions per second (whole chip):
    ``ebm_probe_valu_f32`` issues blocks * 4 waves * 8 * iters independent v_fma_f32."""
    out = torch.empty(blocks * 256, dtype=torch.float32, device=device)
    st = _lib.stream_handle(device)
    _lib.call("ebm_probe_valu_f32", out.data_ptr(), blocks, iters, st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        _lib.call("ebm_probe_valu_f32", out.data_ptr(), blocks, iters, st)
    b.record()
    torch.cuda.synchronize(device)
    return blocks * 4 * 8 * iters * reps / (a.elapsed_time(b) * 1e-3)


def issue_costs(device, blocks=256 * 8, iters=2048, reps=3):
    """Issue cost of the instruction classes of the lean loop, in units of one plain VALU op, measured NOW on this box
    (``ebm_probe_issue_f32``: the same dependent-free stream per class) -> the loop's price per float4 group and step."""
    out = torch.empty(blocks * 256, dtype=torch.float32, device=device)
    st = _lib.stream_handle(device)

    def t_of(kind):
        _lib.call("ebm_probe_issue_f32", out.data_ptr(), blocks, iters, kind, st)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            _lib.call("ebm_probe_issue_f32", out.data_ptr(), blocks, iters, kind, st)
        b.record()
        torch.cuda.synchronize(device)
        return a.elapsed_time(b) / reps

    t0 = t_of(0)                      # 8 plain ops per iteration
    per = t0 / 8.0
    units = {
        "mad_u64_u32": t_of(1) / 8.0 / per - 1.0,      # slot = v_mad_u64_u32 + one plain v_xor
        "transcendental": t_of(2) / 8.0 / per - 1.0,   # slot = v_log_f32 + one plain v_add
        "packed_f32": t_of(3) / 8.0 / per,             # v_pk_fma_f32 (two fused multiply-adds), SGPR-pair multiplicand
        "packed_f32_vgpr": t_of(5) / 8.0 / per, "packed_mul_f32": t_of(6) / 8.0 / per,
        "bitop3": t_of(4) / 8.0 / per,
        "plain": 1.0,
    }
    loop = sum(LEAN_LOOP_MIX[k] * units[k] for k in LEAN_LOOP_MIX)
    # kind 7: the loop's own static mix (76 instructions per trip), dependency-free at eight waves per SIMD: seconds per
    # wave-trip, whole chip -- the ceiling of a kernel made of exactly this mix
    # eight rounds of workgroups per launch (the kernel's launch is 32 rounds deep: a single round pays the ramp and the tail in
    # full), the fastest of three: a ceiling, not an average
    mix_blocks = blocks * 8
    mix_out = torch.empty(mix_blocks * 256, dtype=torch.float32, device=device)

    def t_mix_once():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.call("ebm_probe_issue_f32", mix_out.data_ptr(), mix_blocks, iters, 7, st)
        b.record()
        torch.cuda.synchronize(device)
        return a.elapsed_time(b)

    t_mix_once()
    t_mix = min(t_mix_once() for _ in range(3)) / 8.0  # per round of `blocks` workgroups
    mixed_s_per_wave_trip = t_mix * 1e-3 / (blocks * 4 * iters)
    return {"units_per_class": {k: round(v, 3) for k, v in units.items()}, "static_mix_per_group_step": LEAN_LOOP_MIX,
            "issue_units_per_float4_group_step": loop, "round1_constant": LEAN_LOOP_ISSUE_UNITS,
            "mixed_stream_s_per_wave_trip": mixed_s_per_wave_trip,
            "mixed_stream_units_per_trip": t_mix / per if per > 0 else None}


def pick_threads(run_once, candidates=(8, 16, 32, 64)):
    """Fastest torch intra-op thread count for ``run_once`` among a few candidates (bounded: stops at a 5 s probe; the
    whole-machine count is not probed -- torch's intra-op pool is slower at 256 threads than at 64 on ops this small)."""
    ncpu = os.cpu_count() or 1
    best, best_t = 1, float("inf")
    for th in sorted({min(t, ncpu) for t in candidates}):
        torch.set_num_threads(th)
        run_once()
        t0 = time.perf_counter()
        run_once()
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = th, dt
        if dt > 5.0:
            break
    torch.set_num_threads(best)
    return best, best_t, ncpu


def cpu_config3():
    """Config 3 on the host: the oracle's HMC (restatement of samplers/hmc.py:243-312 + leapfrog.py:116-187) driving autograd
    on the mixture forward as the reference's BaseModel.gradient does, n = 2^14 chains, 2 MH transitions of L = 20."""
    import oracle

    n, dim, T, L = 1 << 14, 32, 2, 20
    model = ta.core.ring_mixture(8, dim, device="cpu")
    en = oracle.GaussianMixture(model.means.detach().clone(), float(model.sigma))
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(n, dim, generator=g)
    p = torch.randn(T, n, dim, generator=g)
    u = torch.rand(T, n, generator=g)
    run = lambda: oracle.hmc_chain(en, x0, p, u, [0.1] * T, L)  # noqa: E731
    th, _, ncpu = pick_threads(run)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    return {"value": n * T / t, "unit": "MH-steps/s", "cores": th, "kind": "port",
            "sample": f"oracle.hmc_chain (torch CPU restatement), ring mixture K=8 dim=32, n=2^14, {T} transitions of L={L}, "
                      f"median of 3 ({t:.2f} s each), fastest of the probed thread counts ({th} of {ncpu} cores)"}


def cpu_config5():
    """Config 5's sampler call on the host: 20 Langevin steps of 65 536 chains through the autograd gradient of the
    2-128-128-1 SiLU network (what the reference's ContrastiveDivergence does per training step)."""
    import oracle
    from torch import nn

    n, k = 65536, 20
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(2, 128), nn.SiLU(), nn.Linear(128, 128), nn.SiLU(), nn.Linear(128, 1))
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(n, 2, generator=g)

    def grad(x):
        x = x.detach().requires_grad_(True)
        (gx,) = torch.autograd.grad(net(x).sum(), x)
        return gx

    def run(steps=k):
        x = x0
        for _ in range(steps):
            x = oracle.em_step(x, grad(x), torch.randn(n, 2, generator=g), 0.1, 1.0)
        return x

    th, _, ncpu = pick_threads(lambda: run(2))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    return {"value": n * k / t, "unit": "chain-steps/s", "sampler_call_ms": t * 1e3, "cores": th, "kind": "port",
            "sample": f"autograd gradient of the 2-128-128-1 network + the oracle's eager update, n=65536, k=20, median of 3 "
                      f"({t:.2f} s each), fastest of the probed thread counts ({th} of {ncpu} cores)"}


def read_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json): counters cannot be
    read from inside the timed process, so this figure is labelled with its source."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def traffic_child(args) -> int:
    """What `measure_traffic` runs under rocprofv3: the timed workload's launch, three times, nothing else."""
    device = torch.device("cuda", 0)
    s = ta.LangevinDynamics(ta.DoubleWellModel(device=device), step_size=ETA, noise_scale=SIGMA, device=device)
    s.donate_input = True
    x = torch.randn(args.n_chains, args.dim, device=device, generator=torch.Generator(device=device).manual_seed(1234))
    for _ in range(3):
        x = s.sample(x=x, n_steps=args.k)
    torch.cuda.synchronize(device)
    return 0


def measure_traffic(args, kernel_substring="langevin_chain_lean_kernel"):
    """HBM bytes per launch of the dominant kernel, measured NOW: this file re-run under `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (separate passes: the TCC slots do not hold both; counters only, no trace domain), three launches each.
    FETCH_SIZE is doubled (gfx950: 64 B tallied per 128-B request on wide coalesced reads, MI355X_MICROARCH.md, HBM section;
    confirmed on a pure copy in profiles/r02_pmc.json); both counters are in KiB.  None when rocprofv3 is not on PATH or a pass
    fails -- the caller then falls back to the committed profiles/traffic.json and says so."""
    import csv
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    got = {}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", "t", "--", sys.executable, os.path.abspath(__file__),
                   "--traffic-child", "--n-chains", str(args.n_chains), "--dim", str(args.dim), "--k", str(args.k)]
            try:
                subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=240, check=True)
            except Exception:
                return None
            vals = []
            for root, _, files in os.walk(tmp):
                for name in files:
                    if name.endswith("counter_collection.csv"):
                        with open(os.path.join(root, name)) as f:
                            for r in csv.DictReader(f):
                                if r.get("Counter_Name") == counter and kernel_substring in r.get("Kernel_Name", ""):
                                    vals.append(float(r["Counter_Value"]))
            if not vals:
                return None
            got[counter] = sum(vals) / len(vals) * 1024.0
    read = 2.0 * got["FETCH_SIZE"]
    return {"hbm_bytes_per_launch": read + got["WRITE_SIZE"], "hbm_read_bytes_corrected": read, "hbm_write_bytes": got["WRITE_SIZE"],
            "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of three launches each "
                      f"(FETCH_SIZE x 2, gfx950 correction); {time.perf_counter() - t0:.0f} s"}


def timed(fn, reps, warm, device):
    """Wall seconds per call, device-synchronised around each timed block: the MEDIAN of three blocks of ceil(reps / 2) calls.
    (One block is at the mercy of whatever else the box does in those milliseconds: a round-5 run printed config 3 at 2.6 ms per call
    between two runs at 1.15 -- same binary, same kernel time.  The headline's own timed region is exactly K steps and is not this.)"""
    for _ in range(warm):
        fn()
    per_block = max(1, (reps + 1) // 2)
    blocks = []
    for _ in range(3):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(per_block):
            fn()
        torch.cuda.synchronize(device)
        blocks.append((time.perf_counter() - t0) / per_block)
    return sorted(blocks)[1]


def kernel_ms_of(entry, fn, reps, device):
    """Mean GPU duration of the `entry` launches made by `fn` (events on the launch stream), per call of fn."""
    _lib.timed_events[entry] = []
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(device)
    pairs = _lib.timed_events.pop(entry)
    return sum(a.elapsed_time(b) for a, b in pairs) / reps if pairs else None


# ---------------------------------------------------------------------------------------
# the other BASELINE configs, measured in the same run (N = 1, rank 0): `extra`
# ---------------------------------------------------------------------------------------
def extra_measurements(device, valu_rate, loop_units=LEAN_LOOP_ISSUE_UNITS, cpu=True, mixed_s_per_wave_trip=None):
    out = []

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            out.append(fn())
        except Exception as exc:  # one failing extra must not take the headline line with it
            out.append({"name": name, "error": f"{type(exc).__name__}: {exc}"[:300]})
        out[-1]["wall_s_of_this_measurement"] = round(time.perf_counter() - t0, 1)

    def c3():
        # BASELINE configs[2]: HMC, L = 20, 8-mode mixture on a radius-4 ring, n = 2^18, dim = 32, eps = 0.1, 50 MH steps per call
        n, dim, T, L = 1 << 18, 32, 50, 20
        model = ta.core.ring_mixture(8, dim, device=device)
        s = ta.HamiltonianMonteCarlo(model, step_size=0.1, n_leapfrog_steps=L, device=device)
        gen = torch.Generator(device=device).manual_seed(1234)
        x0 = torch.randn(n, dim, device=device, generator=gen)
        fn = lambda: s.sample(x=x0, n_steps=T, generator=gen)  # noqa: E731
        # (a 1.2 ms call: ten of them untimed first -- this leg may start on a GPU that sat idle through a CPU baseline,
        #  and three calls do not bring its clocks back: 1.22 ms per launch against 1.11 in steady state)
        t = timed(fn, reps=20, warm=10, device=device)
        kms = kernel_ms_of("ebm_hmc_chain_f32", fn, 10, device)
        _, d = s.sample(x=x0, n_steps=T, thin=T, return_diagnostics=True, generator=gen)
        # Work actually executed.  The ring's modes differ in columns 0..1 only; the kernel (csrc/hmc_ring.hip:
        # hmc_slot1_kernel<2>) runs the two K x 2 passes over those columns and treats the other 30 as the shared quadratic
        # (kick and drift: two FMAs per column and step), and it carries energy / active force across transitions: L
        # evaluations per transition, not L + 1.
        evals = n * T * L
        flops = evals * (2 * 2 * 8 * 2) + n * T * L * 4 * dim
        # the same call on a mixture whose means differ in EVERY column (the general body: two K x dim passes)
        gd = torch.Generator().manual_seed(7)
        dense = ta.GaussianMixtureModel(torch.randn(8, dim, generator=gd) * 2.0, sigma=1.0, device=device)
        sd = ta.HamiltonianMonteCarlo(dense, step_size=0.1, n_leapfrog_steps=L, device=device)
        fd = lambda: sd.sample(x=x0, n_steps=T, generator=gen)  # noqa: E731
        td = timed(fd, reps=6, warm=3, device=device)
        kd = kernel_ms_of("ebm_hmc_chain_f32", fd, 4, device)
        dense_flops = n * T * (L + 1) * (2 * 2 * 8 * dim + 4 * dim) + n * T * L * 4 * dim  # L + 1 evaluations (csrc/hmc_gmm32.hip)
        return {
            "name": "config3_hmc_gmm8", "workload": "HamiltonianMonteCarlo.sample, L=20, 8-mode GaussianMixture, n_chains=2^18, dim=32, "
            "eps=0.1, 50 MH steps per call (BASELINE configs[2])", "metric": "MH-steps/s", "value": n * T / t,
            "ms_per_call": t * 1e3, "kernel_ms_per_call": kms, "leapfrog_steps_per_s": n * T * L / t,
            "grad_evals_per_s": evals / t, "executed_fp32_TFLOPs": flops / (kms * 1e-3 if kms else t) / 1e12,
            "bound": "valu", "body": "hmc_slot1_kernel<2> (csrc/hmc_ring.hip): two active columns (the ring's means differ in columns 0..1 only), "
                                     "four waves per SIMD, 0 scratch; profiles/r04_pmc_hmc_c3.txt: VALU busy 92.7 %, counter traffic 1.006 x algorithmic",
            "step_equivalent_GBps": n * T * 8 * dim / t / 1e9,
            "acceptance_rate": float(d["acceptance_rate"][-1]),
            "cpu_baseline": cpu_config3() if cpu else None,
            "dense_means": {
                "workload": "same call, 8 random means that differ in every column (general mixture body)",
                "value": n * T / td, "ms_per_call": td * 1e3, "kernel_ms_per_call": kd,
                "executed_fp32_TFLOPs": dense_flops / (kd * 1e-3 if kd else td) / 1e12,
                "frac_of_fp32_vector_peak": dense_flops / (kd * 1e-3 if kd else td) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                "note": "hmc_gmm32_kernel (csrc/hmc_gmm32.hip), four waves per SIMD: 84 % of the SIMD cycles execute VALU instructions, 398 per "
                        "evaluation of which 256 are the packed FMAs of the two K x dim passes, the rest of the time waits on the scalar loads "
                        "that stream the means (profiles/r04_pmc_hmc_c3.txt); the 157 TF/s spec figure is not reachable by FMA streams at the "
                        "clock the part holds (profiles/r03_probe_packed.jsonl: 116 - 123 TFLOP/s)",
            },
        }

    def c4():
        # one GPU's shard of BASELINE configs[3]: Langevin DoubleWell, 2^20 of 2^23 chains, dim = 128, k = 500
        n, dim, k = 1 << 20, 128, 500
        s = ta.LangevinDynamics(ta.DoubleWellModel(device=device), step_size=ETA, noise_scale=SIGMA, device=device)
        gen = torch.Generator(device=device).manual_seed(1234)
        x0 = torch.randn(n, dim, device=device, generator=gen)
        fn = lambda: s.sample(x=x0, n_steps=k, generator=gen)  # noqa: E731
        t = timed(fn, reps=3, warm=1, device=device)
        kms = kernel_ms_of("ebm_langevin_chain_f32", fn, 2, device)
        ceiling_ms = (n * dim / 4 / 64) * k * mixed_s_per_wave_trip * 1e3 if mixed_s_per_wave_trip else None
        return {
            "name": "config4_shard", "workload": "LangevinDynamics.sample on DoubleWell, one GPU's shard of BASELINE configs[3]: "
            "n_chains=2^20 (of 2^23 over 8 GPUs), dim=128, k=500", "metric": "chain-steps/s per GPU", "value": n * k / t,
            "ms_per_call": t * 1e3, "kernel_ms_per_call": kms, "bound": "valu",
            "frac_of_mixed_ceiling": (ceiling_ms / kms) if (kms and ceiling_ms) else None,  # see roofline.valu.mixed_ceiling_definition
            "step_equivalent_GBps": n * k * 8 * dim / t / 1e9, "step_equivalent_frac_of_8TBps": n * k * 8 * dim / t / 8e12,
            "allgather_bytes_per_rank": n * dim * 4,
        }

    def c5():
        # BASELINE configs[4]: PCD training of MLP 2-128-128-1 (SiLU) on two-moons, n_chains = batch = buffer = 65536, k = 20
        from torch import nn

        from torchebm_amd.utils.synthetic import two_moons

        class AutogradMLP(ta.core.BaseModel):
            def __init__(self):
                super().__init__()
                self.net = nn.Sequential(nn.Linear(2, 128), nn.SiLU(), nn.Linear(128, 128), nn.SiLU(), nn.Linear(128, 1))

            def forward(self, x):
                return self.net(x).squeeze(-1)

        n, k = 65536, 20
        data = two_moons(n, 0.05, seed=0, device=device)
        res = {"name": "config5_pcd_mlp", "workload": "ContrastiveDivergence(persistent=True) training step, MLP 2-128-128-1 SiLU energy "
               "on two-moons, n_chains=batch=buffer=65536, k=20, eta=0.1 (BASELINE configs[4])", "metric": "training steps/s"}

        def train_loop(model, sampler):
            pcd = ta.ContrastiveDivergence(model, sampler, k_steps=k, persistent=True, buffer_size=n, init_steps=0, device=device)
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)

            def step():
                loss, _ = pcd(data)
                opt.zero_grad()
                loss.backward()
                opt.step()

            return step

        torch.manual_seed(0)
        model = AutogradMLP().to(device)
        s = ta.LangevinDynamics(model, step_size=0.1, noise_scale=1.0, device=device)
        # as BASELINE prescribes it: autograd gradient + HIP fused update per Langevin step.  `capture_graph` toggles the
        # HIP-graph replay of one such iteration (the default whenever the configuration is eligible).
        default_graph = s.capture_graph is not False and s._use_graph({}, k)  # None = replay whenever eligible
        s.capture_graph = False
        t_eager = timed(train_loop(model, s), reps=5, warm=2, device=device)
        t_eager_sample = timed(lambda: s.sample(x=data, n_steps=k), reps=5, warm=1, device=device)
        s.capture_graph = True
        t_graph = timed(train_loop(model, s), reps=10, warm=3, device=device)
        t_graph_sample = timed(lambda: s.sample(x=data, n_steps=k), reps=10, warm=2, device=device)
        res["autograd_plus_hip_update"] = {
            "default_route": "hip-graph replay" if default_graph else "eager launches",
            "eager_launches": {"training_steps_per_s": 1 / t_eager, "sampler_ms": t_eager_sample * 1e3},
            "hip_graph_replay": {"training_steps_per_s": 1 / t_graph, "sampler_ms": t_graph_sample * 1e3},
        }
        # the same energy as the packaged MLPEnergy: forward + input gradient fused, the four contractions on the bf16 matrix
        # pipe with three-way split operands (fp32 accuracy, 6 bf16 MFMAs per K-block of 16; csrc/mlp_b16.h)
        torch.manual_seed(0)
        fmodel = ta.MLPEnergy(2, device=device)
        fs = ta.LangevinDynamics(fmodel, step_size=0.1, noise_scale=1.0, device=device)
        tf = timed(train_loop(fmodel, fs), reps=10, warm=3, device=device)
        fn = lambda: fs.sample(x=data, n_steps=k)  # noqa: E731
        tf_sample = timed(fn, reps=10, warm=2, device=device)
        kms = kernel_ms_of("ebm_langevin_chain_f32", fn, 5, device)
        mlp_flops = n * k * 2 * (2 * 128 * 128 + 2 * 2 * 128)  # two HxH contractions + the two thin ones, per chain-step
        issued = n * k * 2 * (2 * 128 * 128) * 6    # bf16 products issued: the two HxH contractions, six terms (dim 2: W1's run on the vector unit, round 5)
        tk = kms * 1e-3 if kms else tf_sample
        res["fused_mlp_kernel"] = {
            "training_steps_per_s": 1 / tf, "sampler_ms": tf_sample * 1e3, "kernel_ms": kms,
            "bound": "instruction issue at one wave per SIMD (DESIGN.md section 4, Wide MLP); matrix pipe: bf16 split operands",
            "useful_fp32_TFLOPs": mlp_flops / tk / 1e12,
            "vs_fp32_matrix_peak": mlp_flops / tk / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
            "issued_bf16_TFLOPs": issued / tk / 1e12, "frac_of_bf16_matrix_peak": issued / tk / 1e12 / BF16_MATRIX_PEAK_TFLOPS,
        }
        # the WHOLE training step -- start points, the fused chain, the buffer write, loss, backward, Adam -- captured once and
        # replayed as one HIP graph (utils.GraphedTrainingStep, opt-in: the ABI-6 entry points read their RNG coordinates from
        # device memory); same recipe, Adam(capturable=True)
        from torchebm_amd.utils import GraphedTrainingStep

        torch.manual_seed(0)
        gmodel = ta.MLPEnergy(2, device=device)
        gsampler = ta.LangevinDynamics(gmodel, step_size=0.1, noise_scale=1.0, device=device)
        gpcd = ta.ContrastiveDivergence(gmodel, gsampler, k_steps=k, persistent=True, buffer_size=n, init_steps=0, device=device)
        gstep = GraphedTrainingStep(gpcd, torch.optim.Adam(gmodel.parameters(), lr=1e-3, capturable=True))
        tg = timed(lambda: gstep(data), reps=30, warm=5, device=device)
        res["whole_step_hip_graph"] = {
            "training_steps_per_s": 1 / tg, "ms_per_step": tg * 1e3, "replays": gstep.replays,
            "what": "GraphedTrainingStep: one graph launch per training step (start points with the exploration noise, fused MLP chain, "
                    "FIFO write, loss, forward + activation planes, parameter gradients in one pass, Adam); bit-identical to the eager "
                    "loop for the same seed (tests/test_graphed_step_gpu.py)",
        }
        # the same step with torch's single-kernel Adam (fused=True: an implementation choice of the optimiser, outside this package)
        try:
            torch.manual_seed(0)
            fm = ta.MLPEnergy(2, device=device)
            fsm = ta.LangevinDynamics(fm, step_size=0.1, noise_scale=1.0, device=device)
            fcd = ta.ContrastiveDivergence(fm, fsm, k_steps=k, persistent=True, buffer_size=n, init_steps=0, device=device)
            fstep = GraphedTrainingStep(fcd, torch.optim.Adam(fm.parameters(), lr=1e-3, capturable=True, fused=True))
            tfa = timed(lambda: fstep(data), reps=30, warm=5, device=device)
            res["whole_step_hip_graph"]["with_torch_fused_adam"] = {"training_steps_per_s": 1 / tfa, "ms_per_step": tfa * 1e3}
        except Exception as exc:  # (an optimiser option of the installed torch, not of this package)
            res["whole_step_hip_graph"]["with_torch_fused_adam"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        res["cpu_baseline"] = cpu_config5() if cpu else None
        res["value"] = 1 / (t_graph if default_graph else t_eager)
        res["chain_steps_per_s"] = n * k * res["value"]
        return res

    def matrix_pipe():
        # the row-coupled energies on the bf16 matrix pipe with split operands (csrc/gauss_bf16x3.h, gmm_bf16x3.h)
        n, k = 1 << 18, 50
        gd = torch.Generator().manual_seed(64)
        a64 = torch.randn(64, 64, generator=gd)
        gauss = ta.GaussianModel(torch.zeros(64), a64 @ a64.t() / 64 + 0.5 * torch.eye(64), device=device)
        mix = ta.GaussianMixtureModel(torch.randn(16, 32, generator=gd) * 2.0, sigma=1.0, device=device)
        out = {"name": "matrix_pipe_energies", "workload": "dense Gaussian (dim 64) and a 16-component mixture (dim 32), 2^18 chains: "
               "Langevin k=50 and HMC L=20 x 10 transitions through the public samplers", "bound": "valu + bf16 mfma"}
        for tag, model, dim in (("gaussian_dim64", gauss, 64), ("mixture16_dim32", mix, 32)):
            x0 = torch.randn(n, dim, device=device)
            ld = ta.LangevinDynamics(model, step_size=0.01, device=device)
            f1 = lambda: ld.sample(x=x0, n_steps=k)  # noqa: E731
            f1()  # first launch of a kernel: code upload, LDS opt-in
            kms = kernel_ms_of("ebm_langevin_chain_f32", f1, 3, device)
            hm = ta.HamiltonianMonteCarlo(model, step_size=0.1, n_leapfrog_steps=20, device=device)
            f2 = lambda: hm.sample(x=x0, n_steps=10)  # noqa: E731
            f2()
            hms = kernel_ms_of("ebm_hmc_chain_f32", f2, 3, device)
            out[tag] = {
                "langevin_kernel_ms": kms, "langevin_chain_steps_per_s": n * k / (kms * 1e-3),
                "langevin_step_equivalent_frac_of_8TBps": n * k * 8 * dim / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "hmc_kernel_ms": hms, "hmc_mh_steps_per_s": n * 10 / (hms * 1e-3),
            }
        return out

    def gaussian_widths():
        # dense Gaussian Langevin over widths: packed rows (8 / 30 / 50), resident in registers + LDS (64 / 128), Ps streamed
        # (160 / 256 / 512) -- csrc/gauss_mfma.hip, csrc/gauss_big.hip; VERDICT r2 item 5
        out = {"name": "gaussian_langevin_widths", "workload": "LangevinDynamics.sample on GaussianModel, k = 20 steps per call, widths "
               "8 / 30 / 50 (packed rows), 64 / 128 / 160 (register-resident, Ps in LDS), 192 / 224 / 256 (register-resident, Ps streamed from its "
               "pre-split image by LDS-direct loads), 384 / 512 (tiled, same image)",
               "bound": "valu + bf16 mfma", "metric": "chain-steps/s; step-equivalent fraction of 8 TB/s = n k 8 dim / t / 8e12", "dims": {}}
        k = 20
        for dim, n in ((8, 1 << 18), (30, 1 << 18), (50, 1 << 18), (64, 1 << 18), (128, 1 << 18), (160, 1 << 17), (192, 1 << 17), (224, 1 << 17),
                       (256, 1 << 17), (384, 1 << 16), (512, 1 << 16)):
            gd = torch.Generator().manual_seed(dim)
            a = torch.randn(dim, dim, generator=gd)
            model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=device)
            ld = ta.LangevinDynamics(model, step_size=0.01, device=device)
            x0 = torch.randn(n, dim, device=device)
            fn = lambda: ld.sample(x=x0, n_steps=k)  # noqa: E731
            fn()
            kms = kernel_ms_of("ebm_langevin_chain_f32", fn, 3, device)
            out["dims"][str(dim)] = {"n_chains": n, "kernel_ms": kms, "chain_steps_per_s": n * k / (kms * 1e-3),
                                     "step_equivalent_frac_of_8TBps": n * k * 8 * dim / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "useful_TFLOPs": 2 * n * k * dim * dim / (kms * 1e-3) / 1e12}
        return out

    def step_kernel():
        # the genuinely HBM-bound kernel of the path: one Euler-Maruyama step with an external gradient, 2^26 elements
        n = 1 << 26
        x = torch.randn(n, device=device)
        g = torch.randn(n, device=device)
        o = torch.empty_like(x)
        st = _lib.stream_handle(device)
        step = [0]

        def fn():
            step[0] += 1
            _lib.call("ebm_langevin_step_f32", x.data_ptr(), g.data_ptr(), o.data_ptr(), None, n, ETA, ETA**0.5,
                      (2.0 * SIGMA**2) ** 0.5, 0, 0.0, 0.0, 7, step[0], st)

        for _ in range(3):
            fn()
        kms = kernel_ms_of("ebm_langevin_step_f32", lambda: [fn() for _ in range(10)], 3, device) / 10
        gbs = 12 * n / (kms * 1e-3) / 1e9
        return {"name": "langevin_step_kernel", "workload": "ebm_langevin_step_f32 (Integrator.step with an external gradient, in-kernel "
                "Philox noise), 2^26 fp32 elements: reads x and grad, writes x'", "metric": "GB/s", "bound": "hbm",
                "algorithmic_bytes_per_launch": 12 * n, "kernel_ms": kms, "value": gbs, "peak": HBM_PEAK_GBS,
                "frac": gbs / HBM_PEAK_GBS}

    def wide(hidden, kind, n=65536, dim=32):
        # the same network at another hidden width (256: weights streamed from L2), Langevin k = 20 or HMC L = 10, 10 transitions
        if hidden not in ta.MLPEnergy.FUSED_HIDDEN:
            return {"skipped": f"hidden {hidden} is not in this build (make H256=1; MLPEnergy.FUSED_HIDDEN = {ta.MLPEnergy.FUSED_HIDDEN})"}
        torch.manual_seed(0)
        m = ta.MLPEnergy(dim, hidden, device=device)
        x0 = torch.randn(n, dim, device=device)
        if kind == "langevin":
            s, sym, evals = ta.LangevinDynamics(m, step_size=0.05, device=device), "ebm_langevin_chain_f32", 20
            fn = lambda: s.sample(x=x0, n_steps=20)  # noqa: E731
        else:
            s, sym, evals = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=10, device=device), "ebm_hmc_chain_f32", 110
            fn = lambda: s.sample(x=x0, n_steps=10)  # noqa: E731
        timed(fn, reps=2, warm=2, device=device)
        kms = kernel_ms_of(sym, fn, 3, device)
        tf = n * evals * 2 * (2 * hidden * hidden + 2 * dim * hidden) / (kms * 1e-3) / 1e12
        res = {"kernel_ms": kms, "evaluations_per_launch": evals, "useful_fp32_TFLOPs": tf, "vs_fp32_matrix_peak": tf / FP32_MATRIX_PEAK_TFLOPS,
               "contraction": "exact-f32 MFMA, weights streamed from L2" if hidden == 256 else "bf16 MFMA, three-way split operands"}
        if hidden != 256:
            res["frac_of_bf16_matrix_peak"] = 6 * tf / BF16_MATRIX_PEAK_TFLOPS
        return res

    def mlp_bench_net():
        # the reference's benchmark network (benchmarks/registry.py:372-387) at dim 32: Langevin chain fused on the bf16 matrix pipe
        n, k, dim, hidden = 65536, 20, 32, 128
        torch.manual_seed(0)
        m = ta.MLPEnergy(dim, hidden, device=device)
        s = ta.LangevinDynamics(m, step_size=0.05, device=device)
        x0 = torch.randn(n, dim, device=device)
        fn = lambda: s.sample(x=x0, n_steps=k)  # noqa: E731
        timed(fn, reps=2, warm=2, device=device)
        kms = kernel_ms_of("ebm_langevin_chain_f32", fn, 5, device)
        flops = n * k * 2 * (2 * hidden * hidden + 2 * dim * hidden)
        return {"name": "mlp_benchmark_network_dim32", "workload": "LangevinDynamics.sample on MLPEnergy Linear(32,128)-SiLU-Linear(128,128)-"
                "SiLU-Linear(128,1) (the reference's benchmarks/registry.py network), n_chains=65536, k=20: forward + input gradient + "
                "update fused, four contractions on v_mfma_f32_32x32x16_bf16 with three-way split operands (six products per K-block, fp32 "
                "accuracy)", "metric": "useful fp32-equivalent TFLOP/s", "bound": "instruction issue at one wave per SIMD; bf16 mfma",
                "kernel_ms": kms, "value": flops / (kms * 1e-3) / 1e12, "fp32_matrix_peak": FP32_MATRIX_PEAK_TFLOPS,
                "vs_fp32_matrix_peak": flops / (kms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
                "issued_bf16_TFLOPs": 6 * flops / (kms * 1e-3) / 1e12, "peak": BF16_MATRIX_PEAK_TFLOPS,
                "frac": 6 * flops / (kms * 1e-3) / 1e12 / BF16_MATRIX_PEAK_TFLOPS, "chain_steps_per_s": n * k / (kms * 1e-3),
                "hidden_256": wide(256, "langevin"), "hmc_hidden_128": wide(128, "hmc"), "hmc_hidden_256": wide(256, "hmc"),
                # the same network at the reference's widest benchmark input (dim 128): W1's split image streamed through LDS (round 4)
                "dim_128": wide(128, "langevin", dim=128)}

    def c2_fused_arithmetic():
        # the headline call with the opt-in contracted update (LangevinDynamics.fused_arithmetic = True, EBM_CHAIN_CONTRACTED, ABI 8):
        # not the reference's rounding, hence an `extra` and never `value`; same workload, same timing protocol as the headline's kernel figure
        n, dim, k = 1 << 20, 64, 200
        gen = torch.Generator(device=device).manual_seed(1234)
        x0 = torch.randn(n, dim, device=device, generator=gen)
        res = {}
        for name, flag in (("default", False), ("fused_arithmetic", True)):
            s = ta.LangevinDynamics(ta.DoubleWellModel(device=device), step_size=ETA, noise_scale=SIGMA, device=device)
            s.fused_arithmetic = flag
            s.donate_input = True
            x = x0.clone()
            fn = lambda: s.sample(x=x, n_steps=k, generator=gen)  # noqa: E731
            timed(fn, reps=4, warm=3, device=device)
            kms = kernel_ms_of("ebm_langevin_chain_f32", fn, 10, device)
            res[name] = {"kernel_ms_per_call": kms, "chain_steps_per_s": n * k / (kms * 1e-3)}
        return {"name": "config2_fused_arithmetic", "workload": "BASELINE configs[1] (DoubleWell, n_chains=2^20, dim=64, k=200) through "
                "LangevinDynamics.fused_arithmetic = True: x^2 - b^2 and x - eta g as FMAs, noise_scale * sqrt(step_size) folded into the "
                "Box-Muller radius; opt-in, NOT the reference's rounding (the default is, bit for bit); kernel time by events, same run",
                "metric": "chain-steps/s (kernel)", "value": res["fused_arithmetic"]["chain_steps_per_s"],
                "default_same_protocol": res["default"], "fused": res["fused_arithmetic"],
                "speedup": res["default"]["kernel_ms_per_call"] / res["fused_arithmetic"]["kernel_ms_per_call"]}

    def reference_scales():
        # The reference's OWN benchmark scales (benchmarks/conftest.py:35-39: small 64 x 8 x 50, medium 256 x 32 x 100, large
        # 1024 x 128 x 200; registry.py:141-148, 368-370, 679-717: DoubleWellModel(barrier_height=2), step_size 1e-3,
        # LangevinDynamics(noise_scale=1) and HamiltonianMonteCarlo(n_leapfrog_steps=10), timed through sampler.sample(x=x0, n_steps=...)).
        # At these sizes the number is the HOST path: wall per call, the kernel's own time (events around the launch), and the
        # Python time before / around the launch = wall - kernel; the oracle (CPU restatement of the same call) beside each.
        import oracle
        rows = []
        for scale, (bs, dim, k) in (("small", (64, 8, 50)), ("medium", (256, 32, 100)), ("large", (1024, 128, 200))):
            model = ta.DoubleWellModel(barrier_height=2.0, device=device)
            x0 = torch.randn(bs, dim, device=device)
            for sampler_name in ("LangevinDynamics", "HamiltonianMonteCarlo"):
                if sampler_name == "LangevinDynamics":
                    s = ta.LangevinDynamics(model, step_size=1e-3, noise_scale=1.0, device=device)
                    entry = "ebm_langevin_chain_f32"
                else:
                    s = ta.HamiltonianMonteCarlo(model, step_size=1e-3, n_leapfrog_steps=10, device=device)
                    entry = "ebm_hmc_chain_f32"
                fn = lambda: s.sample(x=x0, n_steps=k, n_samples=bs, dim=dim)  # noqa: E731  (the reference's call, registry.py:716-717)
                wall = timed(fn, reps=200, warm=30, device=device)
                kms = kernel_ms_of(entry, fn, 50, device)
                # the Python side alone: the time sample() takes to RETURN (routing, RNG reservation, allocation, the launch call) with
                # the stream drained before every call, so that nothing of it hides behind a running kernel
                host_only = []
                for _ in range(60):
                    torch.cuda.synchronize(device)
                    t0 = time.perf_counter()
                    fn()
                    host_only.append(time.perf_counter() - t0)
                host_only.sort()
                py_us = host_only[len(host_only) // 2] * 1e6
                # the same call on the host cores: the oracle with pre-drawn noise (drawing it is part of the reference's call too,
                # so it is inside the timed region), one thread count for all (these are 10 - 100 ms problems)
                en = oracle.DoubleWell(2.0, 1.0)
                xc = x0.cpu()
                if sampler_name == "LangevinDynamics":
                    def cpu_call():
                        noise = torch.randn(k, bs, dim)
                        return oracle.langevin_chain(en, xc, noise, [1e-3] * k, [1.0] * k)
                else:
                    def cpu_call():
                        return oracle.hmc_chain(en, xc, torch.randn(k, bs, dim), torch.rand(k, bs), [1e-3] * k, 10)
                cpu_call()
                reps_cpu = 3 if scale == "large" else 5
                t0 = time.perf_counter()
                for _ in range(reps_cpu):
                    cpu_call()
                cpu_s = (time.perf_counter() - t0) / reps_cpu
                rows.append({"scale": scale, "sampler": sampler_name, "batch_size": bs, "dim": dim, "n_steps": k,
                             "wall_us_per_call": wall * 1e6, "kernel_us_per_call": None if kms is None else kms * 1e3,
                             # (calls are queued back to back: where the kernel is longer than the Python around it the host share is hidden -- 0)
                             "host_us_per_call": None if kms is None else max(0.0, wall * 1e6 - kms * 1e3),
                             "python_us_until_sample_returns": py_us,
                             "chain_steps_per_s": bs * k / wall, "cpu_oracle_ms_per_call": cpu_s * 1e3,
                             "cpu_oracle_chain_steps_per_s": bs * k / cpu_s, "cpu_threads": torch.get_num_threads()})
        return {"name": "reference_benchmark_scales", "workload": "the reference's own benchmark scales (benchmarks/conftest.py:35-39, "
                "registry.py:141-148,679-717): DoubleWellModel(barrier_height=2), step_size=1e-3, LangevinDynamics(noise_scale=1) and "
                "HamiltonianMonteCarlo(n_leapfrog_steps=10) through sampler.sample(x=x0, n_steps=...); wall per call (median of 3 blocks of "
                "100 calls), the kernel's time by events, host = wall - kernel in a back-to-back loop, python_us_until_sample_returns = the call's own time on an idle stream (Python routing, RNG "
                "reservation, the output allocation, the launch); cpu_oracle = the oracle's restatement of the same call on this box's host cores (noise drawn inside the call)",
                "metric": "us per sample() call", "rows": rows}

    guarded("config2_fused_arithmetic", c2_fused_arithmetic)
    guarded("reference_benchmark_scales", reference_scales)
    guarded("config3_hmc_gmm8", c3)
    guarded("config4_shard", c4)
    guarded("config5_pcd_mlp", c5)
    guarded("mlp_benchmark_network_dim32", mlp_bench_net)
    guarded("langevin_step_kernel", step_kernel)
    guarded("matrix_pipe_energies", matrix_pipe)
    guarded("gaussian_langevin_widths", gaussian_widths)

    def gaussian_hmc_widths():
        # HMC on dense Gaussians beyond the LDS-resident widths: one ebm_hmc_chain_f32 launch (csrc/gauss_hmc_stream.hip) -- VERDICT r3 item 7
        out = {"name": "gaussian_hmc_widths", "workload": "HamiltonianMonteCarlo.sample on GaussianModel, 5 transitions of 10 leapfrog steps, "
               "n_chains = 2^17: dims 160 (precision images resident in LDS), 192 / 256 (streamed from the pre-split image)",
               "bound": "instruction issue at one wave per SIMD; bf16 mfma", "metric": "ms per call; useful fp32-equivalent TFLOP/s", "dims": {}}
        n, T, L = 1 << 17, 5, 10
        for dim in (160, 192, 256):
            gd = torch.Generator().manual_seed(dim)
            a = torch.randn(dim, dim, generator=gd)
            model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=device)
            hm = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=L, device=device)
            x0 = torch.randn(n, dim, device=device)
            fn = lambda: hm.sample(x=x0, n_steps=T)  # noqa: E731
            fn()
            before = _lib.call_counts["ebm_hmc_chain_f32"]
            kms = kernel_ms_of("ebm_hmc_chain_f32", fn, 3, device)
            launches = (_lib.call_counts["ebm_hmc_chain_f32"] - before) / 3
            out["dims"][str(dim)] = {"kernel_ms": kms, "launches_per_call": launches, "mh_steps_per_s": n * T / (kms * 1e-3),
                                     "useful_TFLOPs": 2 * n * dim * dim * T * (L + 1) / (kms * 1e-3) / 1e12}
        return out

    guarded("gaussian_hmc_widths", gaussian_hmc_widths)
    return out


def main():
    args = parse()
    if args.traffic_child:
        return traffic_child(args)
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = args.device == "cuda"
    if on_gpu and torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} (local {local_rank}) has no GPU: {torch.cuda.device_count()} visible", file=sys.stderr)
        return 2
    backend, backend_world = None, None
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
        backend = dist.get_backend()
        backend_world = dist.get_world_size()
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; reporting n_gpus={world}",
              file=sys.stderr)
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
    elapsed_local = elapsed
    timed_pairs = _lib.timed_events.pop("ebm_langevin_chain_f32") if on_gpu else []  # the K timed steps' launches only

    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N > 1, behind the timed region: WHY the N-rank number is what it is.  The same step three ways, each fenced
    #      (barrier + device sync) on both sides, median of 3, maximum over ranks: plain (no collective), with the
    #      pipelined read-back (what the timed run's last step does), and the all-gather alone (not overlapped).
    multi = None
    if world > 1:
        import torch.distributed as dist

        from torchebm_amd.utils import all_gather_cat

        def fenced_ms(fn, reps=3):
            ts = []
            for _ in range(reps):
                fence()
                a = time.perf_counter()
                fn()
                fence()
                ts.append((time.perf_counter() - a) * 1e3)
            v = torch.tensor([sorted(ts)[len(ts) // 2]], dtype=torch.float64, device=device)
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            return float(v.item())

        plain_ms = fenced_ms(one_step)
        pipelined_ms = None if readback.startswith("disabled") else fenced_ms(last_step_with_readback)
        gather_ms, gather_err = None, None
        try:
            gather_ms = fenced_ms(lambda: all_gather_cat(state))
        except Exception as exc:
            gather_err = f"{type(exc).__name__}: {exc}"[:200]
        shard_bytes = n * dim * 4
        # per-rank time of the timed run (wall per step; on the GPU also the chain kernel's event time): min / max over ranks
        mine = [elapsed_local / args.steps * 1e3, -1.0]
        if timed_pairs:
            mine[1] = sum(a.elapsed_time(b) for a, b in timed_pairs) / args.steps
        every = torch.zeros(world, 2, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(every.view(-1), torch.tensor(mine, dtype=torch.float64, device=device))
        every = every.cpu()
        multi = {
            "bytes_per_rank": shard_bytes,
            "gathered_bytes_per_rank": world * shard_bytes,
            "plain_step_ms": plain_ms,
            "step_with_pipelined_readback_ms": pipelined_ms,
            "exposed_ms": None if pipelined_ms is None else max(0.0, pipelined_ms - plain_ms),
            "allgather_alone_ms": gather_ms,
            # bytes a rank RECEIVES over the links / time of the collective alone; bus bandwidth in nccl-tests' sense is the same
            # figure for an all-gather ((N-1)/N of the gathered size per rank)
            "algbw_GBps": None if not gather_ms else (world - 1) * shard_bytes / (gather_ms * 1e-3) / 1e9,
            "allgather_error": gather_err,
            "per_rank_ms_per_step": {"min": float(every[:, 0].min()), "max": float(every[:, 0].max()),
                                     "all": [round(float(v), 4) for v in every[:, 0]]},
            "per_rank_kernel_ms": None if float(every[:, 1].min()) < 0 else
                {"min": float(every[:, 1].min()), "max": float(every[:, 1].max()), "all": [round(float(v), 4) for v in every[:, 1]]},
            "how": "behind the timed region; each figure = median of 3 fenced runs, maximum over ranks",
        }

    kernel_ms, kernel_dist = None, None
    if on_gpu:
        pairs = timed_pairs
        if pairs:  # GPU time of the chain kernel per step (the pipelined last step of N > 1 is several launches)
            each = [a.elapsed_time(b) for a, b in pairs]
            kernel_ms = sum(each) / args.steps
            if len(each) == args.steps:  # one launch per step: the per-step distribution
                srt = sorted(each)
                kernel_dist = {"min": srt[0], "median": srt[len(srt) // 2], "mean": kernel_ms, "max": srt[-1], "launches": len(each)}

    if rank == 0:
        chain_steps = world * n * k * args.steps
        value = chain_steps / elapsed
        algo_bytes = n * k * 8 * dim  # read x_t + write x_{t+1}, fp32, per chain-step
        roof = None
        valu_rate = None
        if kernel_ms:
            achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
            traffic = None if (args.no_traffic or world > 1) else measure_traffic(args)
            if traffic is None:
                traffic = read_traffic()
            valu_rate = plain_valu_rate(device)
            costs = issue_costs(device)  # per-class issue costs measured in THIS run -> the loop's price
            loop_units = costs["issue_units_per_float4_group_step"]
            physical = 2 * n * dim * 4  # what the k-fused launch has to move: the state in, the state out
            issue_limit_ms = (n * dim / 4 / 64) * k * loop_units / valu_rate * 1e3
            # the bound: the same number of wave-trips of the loop's exact static instruction mix, issued dependency-free
            mixed_ceiling_ms = (n * dim / 4 / 64) * k * costs["mixed_stream_s_per_wave_trip"] * 1e3
            # An INDEPENDENT ceiling next to the measured one -- no probe, only the instruction count and the pipe's documented rates:
            # wave-trips per SIMD x cycles per trip / clock.  MI355X_MICROARCH.md gives v_fma_f32 (wave64, SIMD-32) 2 cycles, 256 CUs x
            # 4 SIMDs, 2.4 GHz; the other classes at their ISA rate classes relative to it (packed f32: two results per lane = 2x;
            # transcendental and the 32 x 32 -> 64 multiply-add: quarter rate = 4x; v_bitop3_b32: full rate).
            trips_per_simd = (n * dim / 4 / 64) * k / (N_CUS * 4)
            rate_class_cycles = {"plain": 2, "bitop3": 2, "packed_f32": 4, "transcendental": 8, "mad_u64_u32": 8}
            cycles_all_full_rate = 2 * sum(LEAN_LOOP_MIX.values())
            cycles_rate_classes = sum(LEAN_LOOP_MIX[c] * rate_class_cycles[c] for c in LEAN_LOOP_MIX)
            independent = {
                "instructions_per_float4_group_step": sum(LEAN_LOOP_MIX.values()), "mix": LEAN_LOOP_MIX, "cycles_per_class": rate_class_cycles,
                "wave_trips_per_simd": trips_per_simd, "clock_GHz": CLOCK_GHZ, "simds": N_CUS * 4,
                "cycles_per_trip_measured": kernel_ms * 1e-3 * CLOCK_GHZ * 1e9 / trips_per_simd,
                "cycles_per_trip_if_every_instruction_issued_like_v_fma": cycles_all_full_rate,
                "cycles_per_trip_at_rate_classes": cycles_rate_classes,
                "floor_ms_all_full_rate": trips_per_simd * cycles_all_full_rate / (CLOCK_GHZ * 1e9) * 1e3,
                "ceiling_ms_at_rate_classes": trips_per_simd * cycles_rate_classes / (CLOCK_GHZ * 1e9) * 1e3,
                "kernel_over_rate_class_ceiling": trips_per_simd * cycles_rate_classes / (CLOCK_GHZ * 1e9) * 1e3 / kernel_ms,
                "reading": "76 instructions per group-step cost 152 cycles if each issued like v_fma_f32 and 348 at their pipe rate classes; the "
                           "kernel's measured cycles per trip sit between the two, within a few percent of the rate-class figure (the nominal "
                           "2.4 GHz over-counts cycles when the chip clocks lower): the instruction count, not waiting, is the time",
            }
            roof = {
                # The kernel keeps the state in registers for all k steps, so it is NOT memory-shaped: the physical
                # limiter is VALU issue (Philox-10 + Box-Muller).  `achieved`/`peak`/`frac` keep BASELINE.json's
                # definition -- step-equivalent bytes (8*dim per chain-step) over the kernel time against HBM peak --
                # which for a k-fused kernel exceeds 1; the physical figures are in `hbm_physical` and `valu`.
                "bound": "valu",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "definition": "step-equivalent: n_chains*k*8*dim bytes / kernel time (BASELINE.md section 3)",
                # flat copies of the figures that ARE bounds (a consumer that keeps only the top level of this object still gets them):
                "bound_frac": min(1.0, mixed_ceiling_ms / kernel_ms),          # the one fraction that is <= 1: measured VALU issue ceiling / kernel
                "valu_frac": mixed_ceiling_ms / kernel_ms,
                "valu_insts_per_group_step": sum(LEAN_LOOP_MIX.values()),
                "valu_rate_class_ceiling_frac": independent["kernel_over_rate_class_ceiling"],
                "hbm_physical_frac": physical / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "valu": {
                    "issue_units_per_float4_group_step": loop_units,
                    "issue_costs_measured_in_run": costs,
                    "plain_valu_wave_instr_per_s": valu_rate,
                    "mixed_ceiling_ms": mixed_ceiling_ms,
                    "frac_of_mixed_ceiling": mixed_ceiling_ms / kernel_ms,
                    "mixed_ceiling_definition": "ebm_probe_issue_f32 kind 7: the lean loop's static mix (18 v_mad_u64_u32, 20 v_bitop3, "
                                                "8 transcendentals, 20 packed-f32, 10 plain per float4 group and step = 76; SQ_INSTS_VALU: 76.1), dependency-free, "
                                                "eight waves per SIMD, timed in this run; x the launch's wave-trips / kernel time: <= 1 by construction",
                    "per_class_sum_ms": issue_limit_ms,
                    "per_class_sum_note": "sum of per-class costs x static counts: a calibration that over-prices a mixed stream "
                                          "(classes overlap), NOT a ceiling -- kept for continuity with rounds 1-3",
                    "independent_of_the_probe": independent,
                },
                "hbm_physical": {
                    "bytes_per_launch": physical,
                    "GBps": physical / (kernel_ms * 1e-3) / 1e9,
                    "frac_of_peak": physical / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                },
                "measured_copy_ceiling": copy_ceiling_gbs(device),
                "traffic": None if not traffic else traffic.get("hbm_bytes_per_launch"),
                "traffic_source": None if not traffic else traffic.get("source"),
                "kernel": "langevin_chain_lean_kernel<DoubleWell> (ebm_langevin_chain_f32)",
                "kernel_ms": kernel_ms,
                "kernel_ms_per_step": kernel_dist,
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
                            + ("(BASELINE configs[1])" if (n, dim, k) == (1 << 20, 64, 200) else
                               (f"(BASELINE configs[3]'s per-GPU shard; {world} of its 8 shards run here)" if (n, dim, k) == (1 << 20, 128, 500)
                                else "(shape overridden on the command line)")),
                "n_chains_per_gpu": n,
                "dim": dim,
                "k_steps": k,
                "parallelism": f"chains sharded x{world}",
                "ranks": world,
                "backend_world_size": backend_world,  # torch.distributed's own count (None: single process)
                "backend": backend,
                "launcher": ("bench.py self-launch (torch.distributed.run)" if os.environ.get("EBM_BENCH_SELF_LAUNCHED")
                             else ("torch.distributed.run" if world > 1 else "single process")),
                "readback": readback,
                "device": args.device,
            },
            "readback": multi,
            "roofline": roof,
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(dim, k),
        }
        if on_gpu and world == 1 and not args.no_extra:
            line["extra"] = extra_measurements(device, valu_rate or plain_valu_rate(device),
                                               (roof or {}).get("valu", {}).get("issue_units_per_float4_group_step") or LEAN_LOOP_ISSUE_UNITS,
                                               cpu=not args.no_cpu_baseline,
                                               mixed_s_per_wave_trip=((roof or {}).get("valu", {}).get("issue_costs_measured_in_run") or {}).get("mixed_stream_s_per_wave_trip"))
        print(json.dumps(line), flush=True)

    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
