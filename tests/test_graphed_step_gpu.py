"""BASELINE config 5's whole training step (PCD start points, the k-fused chain, the FIFO write, loss,
backward, optimiser) as ONE HIP graph -- `torchebm_amd.utils.GraphedTrainingStep` over the ABI-6 entry points that read
their RNG coordinates / write position from device memory.  Reference: torchebm/losses/contrastive_divergence.py:82-155,
core/base_loss.py:266-337,390-426.  The bar: the same seed gives the same losses, chains, weights, generator state and
FIFO position as the eager loop (bit for bit: the graph replays the very kernels the eager step launches)."""

import copy

import pytest
import torch

import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib, _rng
from torchebm_amd.utils import GraphedTrainingStep
from torchebm_amd.utils.synthetic import two_moons

pytestmark = pytest.mark.gpu


def _setup(dev, n, k, buffer_size, ratio, seed=0, enabled=True, capturable=True, model=None, gen_seed=11):
    torch.manual_seed(seed)
    model = ta.MLPEnergy(2, device=dev) if model is None else model
    sampler = ta.LangevinDynamics(model, step_size=0.1, noise_scale=1.0, device=dev)
    cd = ta.ContrastiveDivergence(model, sampler, k_steps=k, persistent=True, buffer_size=buffer_size, init_steps=0,
                                  new_sample_ratio=ratio, device=dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=capturable)
    gen = torch.Generator(device=dev).manual_seed(gen_seed)
    step = GraphedTrainingStep(cd, opt, generator=gen, enabled=enabled)
    return model, cd, opt, gen, step


def _run(step, data, steps, keep=(0, 2, 3, 17, 49)):
    losses, negs = [], {}
    for i in range(steps):
        loss, neg = step(data)
        losses.append(loss)
        if i in keep:
            negs[i] = neg.clone()
    return torch.stack(losses), negs


@pytest.mark.parametrize("ratio", [0.0, 0.05])
@pytest.mark.parametrize("n,buffer_size", [(4096, 8192), (4096, 4096), (1000, 5000)])
def test_graphed_step_equals_the_eager_loop(cuda_device, n, buffer_size, ratio):
    """No torch-side draw in the step: identical losses, negatives, weights, generator, FIFO position -- with the loss's default
    exploration noise too (new_sample_ratio = 0.05: the random subset and its normals come from ebm_pcd_start_points_f32, i.e. from
    the kernels' own field, since ABI 7).
    buffer = 2 n: stratified gather with in-kernel offsets + FIFO scatter through the device pointer; buffer = n: whole-buffer
    overwrite; buffer = 5 n with n off the workgroup size: the FIFO wraps ten times in 50 steps.
    The eager loop runs to its end first: EAGER steps of a second Adam(capturable=True) between the replays of a captured one
    make the captured one drift -- in plain PyTorch too (scripts/probes/torch_graph_adam_interference.py), not a property of
    this package's kernels."""
    k, steps = 5, 50
    data = two_moons(n, 0.05, seed=0, device=cuda_device)
    m_e, cd_e, _, g_e, eager = _setup(cuda_device, n, k, buffer_size, ratio, enabled=False)
    losses_e, negs_e = _run(eager, data, steps)
    torch.cuda.synchronize()
    m_g, cd_g, _, g_g, graphed = _setup(cuda_device, n, k, buffer_size, ratio, enabled=True)
    c0 = hip_calls("ebm_langevin_chain_dev_f32")
    losses_g, negs_g = _run(graphed, data, steps)
    assert graphed.replays == steps - graphed.eager_steps
    assert hip_calls("ebm_langevin_chain_dev_f32") == c0 + 1  # captured once, replayed without Python
    for i in negs_e:
        assert torch.equal(negs_e[i], negs_g[i]), f"negatives differ at step {i}"
    assert torch.equal(losses_e, losses_g)
    for pe, pg in zip(m_e.parameters(), m_g.parameters()):
        assert torch.equal(pe, pg)
    assert torch.equal(cd_e.replay_buffer, cd_g.replay_buffer)
    assert cd_e._write_pos == cd_g._write_pos and int(cd_e.buffer_ptr) == int(cd_g.buffer_ptr) == cd_g._write_pos
    assert g_e.get_offset() == g_g.get_offset() and g_e.initial_seed() == g_g.initial_seed()
    # training did something, and the sampler saw the moving weights: the loss curve is not constant
    assert losses_g.std().item() > 0


def test_graphed_step_follows_a_reseeded_generator_and_an_eager_call_in_between(cuda_device):
    n, k = 2048, 4
    data = two_moons(n, 0.05, seed=1, device=cuda_device)

    def run(enabled):
        model, cd, _, gen, step = _setup(cuda_device, n, k, 2 * n, 0.0, enabled=enabled)
        losses, extra = [], None
        for i in range(12):
            if i == 6:  # the user re-seeds: the device coordinates are rewritten before the next replay
                gen.manual_seed(99)
            if i == 9:  # ... or draws from the same generator outside the step (a sample() call of their own)
                extra = cd.sampler.sample(x=data[:64], n_steps=3, generator=gen)
            losses.append(step(data)[0])
        torch.cuda.synchronize()
        return torch.stack(losses), extra, [p.detach().clone() for p in model.parameters()], gen.get_offset()

    le, xe, we, oe = run(False)
    lg, xg, wg, og = run(True)
    assert torch.equal(le, lg) and torch.equal(xe, xg) and oe == og
    for a, b in zip(we, wg):
        assert torch.equal(a, b)


def test_graphed_step_with_torch_side_draws_in_the_step(cuda_device):
    """A torch-side draw in the step (add_noise_to_real: randn_like on the data) runs inside the graph on torch's graph-safe
    generator state -- the same law at other offsets than the eager loop's, so the bar is statistical: finite, training moves,
    fresh draws on every replay, and the kernels' coordinates advance past torch's share too."""
    n, k = 65536, 20
    data = two_moons(n, 0.05, seed=0, device=cuda_device)
    model, cd, opt, gen, step = _setup(cuda_device, n, k, n, 0.05)
    cd.add_noise_to_real = True
    negs, losses = [], []
    for i in range(8):
        off0 = gen.get_offset()
        loss, neg = step(data)
        losses.append(loss.item())
        negs.append(neg.clone())
        assert torch.isfinite(neg).all() and gen.get_offset() > off0
    assert step.replays == 6 and all(map(lambda v: v == v, losses))
    assert not torch.equal(negs[-1], negs[-2])
    g = step._g
    # the kernels' share of a step: three steps of the field for the start points (offsets, subset keys, normals) + k for the chain
    assert g["torch_steps"] > 0 and int(g["advance"]) == g["kernel_steps"] + g["torch_steps"] == (k + 3) + g["torch_steps"]
    # the device coordinates are where the generator is
    torch.cuda.synchronize()
    assert int(g["coords"].tensor[1]) == gen.get_offset() // 4
    assert _rng.DeviceCoords.as_i64(_rng.kernel_seed(gen.initial_seed())) == int(g["coords"].tensor[0])
    # eight calls of 20 steps at eta = 0.1 on a barely trained energy: a diffusion of a few units, not a blow-up
    assert 0.1 < negs[-1].std().item() < 20.0


def test_graphed_step_refusals(cuda_device):
    n, k = 512, 3
    data = two_moons(n, 0.05, seed=0, device=cuda_device)
    with pytest.raises(ValueError, match="capturable"):
        _setup(cuda_device, n, k, n, 0.0, capturable=False)
    # a scheduled step size advances on the host
    torch.manual_seed(0)
    model = ta.MLPEnergy(2, device=cuda_device)
    sampler = ta.LangevinDynamics(model, step_size=ta.core.LinearScheduler(0.1, 0.01, 100), device=cuda_device)
    cd = ta.ContrastiveDivergence(model, sampler, k_steps=k, persistent=True, buffer_size=n, device=cuda_device)
    with pytest.raises(ValueError, match="scheduled"):
        GraphedTrainingStep(cd, torch.optim.Adam(model.parameters(), capturable=True))
    # an energy whose chain kernels take the coordinates by value (here: the step route of a plain nn.Module energy)
    class Net(ta.core.BaseModel):
        def __init__(self):
            super().__init__()
            self.net = torch.nn.Sequential(torch.nn.Linear(2, 16), torch.nn.SiLU(), torch.nn.Linear(16, 1))

        def forward(self, x):
            return self.net(x).squeeze(-1)

    net = Net().to(cuda_device)
    s2 = ta.LangevinDynamics(net, step_size=0.1, device=cuda_device)
    cd2 = ta.ContrastiveDivergence(net, s2, k_steps=k, persistent=True, buffer_size=n, init_steps=0, new_sample_ratio=0.0, device=cuda_device)
    step = GraphedTrainingStep(cd2, torch.optim.Adam(net.parameters(), capturable=True))
    step(data)
    step(data)  # the eager warm-up steps run on any model
    with pytest.raises(ValueError, match="fused chain launch on an MLPEnergy"):
        step(data)
    # and the C ABI says the same for a non-MLP energy
    spec = ta.DoubleWellModel(device=cuda_device).fused_spec()
    x = torch.zeros(64, 4, device=cuda_device)
    rng = torch.zeros(2, dtype=torch.int64, device=cuda_device)
    with pytest.raises(RuntimeError, match="EBM_ENERGY_MLP"):
        _lib.call("ebm_langevin_chain_dev_f32", spec.to_c(), x.data_ptr(), 64, 4, 2, 0.01, 0.1, 1.0, None, 0, 0.0, 0.0, 1, None,
                  rng.data_ptr(), 0, _lib.stream_handle(cuda_device))


def test_dev_entry_points_equal_the_by_value_ones(cuda_device):
    """ABI 6 through the C ABI: coordinates / position read from device memory give exactly what the by-value calls give."""
    dev = cuda_device
    torch.manual_seed(3)
    model = ta.MLPEnergy(8, device=dev)
    spec = model.fused_spec()
    n, dim, k = 777, 8, 6
    x0 = torch.randn(n, dim, device=dev)
    seed, step = _rng.kernel_seed(1234), 40
    a = x0.clone()
    _lib.call("ebm_langevin_chain_f32", spec.to_c(), a.data_ptr(), n, dim, k, 0.05, 0.05 ** 0.5, 2 ** 0.5, None, 0, 0.0, 0.0, 1, None, None,
              None, seed, step + 3, _lib.stream_handle(dev))
    rng = torch.tensor([_rng.DeviceCoords.as_i64(seed), step], dtype=torch.int64, device=dev)
    b = x0.clone()
    _lib.call("ebm_langevin_chain_dev_f32", spec.to_c(), b.data_ptr(), n, dim, k, 0.05, 0.05 ** 0.5, 2 ** 0.5, None, 0, 0.0, 0.0, 1, None,
              rng.data_ptr(), 3, _lib.stream_handle(dev))
    assert torch.equal(a, b)
    # gather: in-kernel offsets at (seed, step + 1)
    buf = torch.randn(5000, 3, device=dev)
    out_a, out_b = torch.empty(1000, 3, device=dev), torch.empty(1000, 3, device=dev)
    rows_a, rows_b = torch.empty(1000, dtype=torch.int64, device=dev), torch.empty(1000, dtype=torch.int64, device=dev)
    _lib.call("ebm_pcd_gather_f32", buf.data_ptr(), 5000, 3, out_a.data_ptr(), 1000, 5, None, rows_a.data_ptr(), seed, step + 1,
              _lib.stream_handle(dev))
    _lib.call("ebm_pcd_gather_dev_f32", buf.data_ptr(), 5000, 3, out_b.data_ptr(), 1000, 5, rows_b.data_ptr(), rng.data_ptr(), 1,
              _lib.stream_handle(dev))
    assert torch.equal(out_a, out_b) and torch.equal(rows_a, rows_b) and rows_a.unique().numel() == 1000
    # scatter: position from device memory, with wrap
    rows = torch.randn(1000, 3, device=dev)
    buf_a, buf_b = buf.clone(), buf.clone()
    pos = torch.tensor(4500, dtype=torch.int64, device=dev)
    _lib.call("ebm_pcd_scatter_f32", buf_a.data_ptr(), 5000, 3, rows.data_ptr(), 1000, 4500, _lib.stream_handle(dev))
    _lib.call("ebm_pcd_scatter_dev_f32", buf_b.data_ptr(), 5000, 3, rows.data_ptr(), 1000, pos.data_ptr(), _lib.stream_handle(dev))
    assert torch.equal(buf_a, buf_b) and torch.equal(buf_a[4500:], rows[:500]) and torch.equal(buf_a[:500], rows[500:])


@pytest.mark.parametrize("in_dim,hidden,persistent,reg,ratio,n,buffer", [
    (2, 64, True, 0.001, 0.05, 4096, 4096),      # the thin-input kernels at H = 64
    (1, 128, True, 0.001, 0.05, 4096, 4096),     # a 1-D energy: general chain kernel, thin training forward
    (3, 128, True, 0.0, 0.05, 4096, 4096),       # no energy regulariser
    (32, 128, True, 0.001, 0.05, 4096, 12288),   # the benchmark network's width, buffer = 3 batches
    (2, 128, False, 0.001, 0.0, 4096, 4096),     # CD-k: the chains start at the data
    (64, 64, True, 0.01, 0.1, 1000, 5000),       # a ragged batch, dim 64
])
def test_graphed_step_equals_the_eager_loop_across_shapes(cuda_device, in_dim, hidden, persistent, reg, ratio, n, buffer):
    """The whole-step graph against the eager loop (losses and final weights torch.equal after 12 steps) across the kernel variants a
    step can route to: MODE 4 / MODE 2 chain kernels, the training forward at several widths, PCD with and without exploration noise,
    plain CD-k, a batch off the tile size."""
    k, steps = 5, 12
    runs = []
    for enabled in (False, True):
        torch.manual_seed(0)
        m = ta.MLPEnergy(in_dim, hidden, device=cuda_device)
        s = ta.LangevinDynamics(m, step_size=0.05, noise_scale=1.0, device=cuda_device)
        cd = ta.ContrastiveDivergence(m, s, k_steps=k, persistent=persistent, buffer_size=buffer, init_steps=0, new_sample_ratio=ratio,
                                      energy_reg_weight=reg, device=cuda_device)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
        step = GraphedTrainingStep(cd, opt, generator=torch.Generator(device=cuda_device).manual_seed(3), enabled=enabled)
        data = torch.randn(n, in_dim, device=cuda_device, generator=torch.Generator(device=cuda_device).manual_seed(1))
        losses = torch.stack([step(data)[0] for _ in range(steps)])
        torch.cuda.synchronize()
        runs.append((losses, [p.detach().clone() for p in m.parameters()]))
        assert not enabled or step.replays == steps - step.eager_steps
    assert torch.isfinite(runs[0][0]).all() and torch.equal(runs[0][0], runs[1][0])
    assert all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))


def test_graphed_step_refuses_capture_behind_a_stale_autograd_graph(cuda_device):
    """An autograd graph through the parameters, built on the caller's stream and kept alive (a diagnostic tensor), makes torch's
    engine cross streams in the captured backward -- which aborts the process.  The warm-up step sees torch's warning about it and the
    capture is refused with a RuntimeError instead.  (Run in a child process: a missed detection would take the test run down.)"""
    import os
    import subprocess
    import sys
    import textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys, torch
        sys.path.insert(0, {root!r})
        import torchebm_amd as ta
        from torchebm_amd.utils import GraphedTrainingStep
        dev = torch.device("cuda")
        torch.manual_seed(0)
        m = ta.MLPEnergy(2, device=dev)
        s = ta.LangevinDynamics(m, step_size=0.1, device=dev)
        cd = ta.ContrastiveDivergence(m, s, k_steps=5, persistent=True, buffer_size=1024, device=dev)
        st = GraphedTrainingStep(cd, torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True))
        x = torch.randn(512, 2, device=dev)
        st(x)
        keep = m(x).mean()  # gradients enabled, default stream, kept alive across the next calls
        st(x)
        try:
            st(x)
            print("CAPTURED")
        except RuntimeError as exc:
            print("REFUSED" if "another stream" in str(exc) else "OTHER: " + str(exc)[:200])
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert "REFUSED" in out.stdout, (out.returncode, out.stdout[-500:], out.stderr[-1500:])


def test_graphed_step_follows_a_changed_learning_rate(cuda_device):
    """a float lr is a launch constant of the captured optimiser kernels -- the capture key holds every host-side scalar
    (optimiser groups, the sampler's step size / noise scale / clamp, the loss's noise scale), so an LR-scheduler step or a manual
    `param_groups[0]["lr"] = ...` re-captures instead of being silently ignored.  Same schedule on the eager loop: same weights."""
    n, k, steps = 2048, 3, 24
    data = two_moons(n, 0.05, seed=0, device=cuda_device)

    def run(enabled):
        model, cd, opt, gen, step = _setup(cuda_device, n, k, 4096, 0.05, enabled=enabled)
        losses = []
        for i in range(steps):
            if i == 8:
                opt.param_groups[0]["lr"] = 3e-3   # what torch.optim.lr_scheduler does to a float lr
            if i == 16:
                cd.sampler.schedulers["step_size"].start_value = cd.sampler.schedulers["step_size"].current_value = 0.05
            losses.append(step(data)[0])
        return model, torch.stack(losses), step

    m_e, l_e, _ = run(False)
    torch.cuda.synchronize()
    m_g, l_g, step = run(True)
    assert step.recaptures == 2
    assert torch.equal(l_e, l_g)
    for pe, pg in zip(m_e.parameters(), m_g.parameters()):
        assert torch.equal(pe, pg)
