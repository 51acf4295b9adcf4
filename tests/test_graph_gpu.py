"""The per-step Langevin route replayed from a HIP graph (``sampler.capture_graph = True``).

One iteration -- autograd gradient of an arbitrary nn.Module energy, ``ebm_langevin_step_dev_f32`` in
place, the device-side Philox step counter += 1 -- is captured once and replayed n_steps times.  The
noise field is a function of (seed, step, element) only, so the graph route must reproduce the eager
step route BIT FOR BIT under the same generator, and must keep the generator contract (fresh noise on
every call, offset advanced by 4 per step)."""

import pytest
import torch
from torch import nn

import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd.core.schedules import LinearScheduler
from torchebm_amd.utils.synthetic import two_moons

pytestmark = pytest.mark.gpu


class Net(ta.BaseModel):
    def __init__(self, dim=2, hidden=64):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.SiLU(), nn.Linear(hidden, hidden), nn.SiLU(), nn.Linear(hidden, 1))

    def forward(self, x):
        return self.net(x).squeeze(-1)


def _pair(cuda_device, **kw):
    torch.manual_seed(0)
    model = Net().to(cuda_device)
    eager = ta.LangevinDynamics(model, step_size=0.05, noise_scale=1.0, device=cuda_device, **kw)
    graph = ta.LangevinDynamics(model, step_size=0.05, noise_scale=1.0, device=cuda_device, **kw)
    assert eager.capture_graph is None  # the default: replay whenever eligible and the call has >= GRAPH_MIN_STEPS steps
    eager.capture_graph = False
    graph.capture_graph = True
    return model, eager, graph


def _gen(cuda_device, seed):
    return torch.Generator(device=cuda_device).manual_seed(seed)


def test_graph_route_is_bit_identical_to_the_eager_step_route(cuda_device):
    model, eager, graph = _pair(cuda_device, clamp=(-2.5, 2.5))
    x0 = two_moons(4099, 0.05, seed=1, device=cuda_device)
    s0, d0 = hip_calls("ebm_langevin_step_f32"), hip_calls("ebm_langevin_step_dev_f32")
    want = eager.sample(x=x0, n_steps=15, generator=_gen(cuda_device, 7))
    got = graph.sample(x=x0, n_steps=15, generator=_gen(cuda_device, 7))
    assert hip_calls("ebm_langevin_step_f32") == s0 + 15
    # the first step of the call runs eagerly (the warm-up IS a real step) + the captured one; the 14 replays do not pass
    # through the host binding
    assert hip_calls("ebm_langevin_step_dev_f32") == d0 + 2
    assert torch.equal(got, want)
    assert got.abs().max().item() <= 2.5
    # second call re-uses the captured graph (no new launches through the binding) and continues the
    # generator: a different noise field, identical again to the eager route continuing ITS generator
    ge, gg = _gen(cuda_device, 3), _gen(cuda_device, 3)
    e1, g1 = eager.sample(x=x0, n_steps=5, generator=ge), graph.sample(x=x0, n_steps=5, generator=gg)
    e2, g2 = eager.sample(x=x0, n_steps=5, generator=ge), graph.sample(x=x0, n_steps=5, generator=gg)
    assert hip_calls("ebm_langevin_step_dev_f32") == d0 + 2
    assert torch.equal(e1, g1) and torch.equal(e2, g2) and not torch.equal(g1, g2)
    assert ge.get_offset() == gg.get_offset() == 4 * 10
    assert x0.data_ptr() != g2.data_ptr()                       # result is not the graph's static buffer


def test_graph_route_trajectory_diagnostics_and_weight_updates(cuda_device):
    model, eager, graph = _pair(cuda_device)
    x0 = two_moons(1024, 0.05, seed=2, device=cuda_device)
    (tw, dw) = eager.sample(x=x0, n_steps=8, thin=2, return_trajectory=True, return_diagnostics=True, generator=_gen(cuda_device, 1))
    (tg, dg) = graph.sample(x=x0, n_steps=8, thin=2, return_trajectory=True, return_diagnostics=True, generator=_gen(cuda_device, 1))
    assert tg.shape == (1024, 4, 2) and torch.equal(tg, tw)
    for key in ("mean", "var", "energy"):
        torch.testing.assert_close(dg[key], dw[key], rtol=1e-6, atol=1e-6)
    # in-place parameter updates (what an optimiser does) are seen by the next replay
    before = graph.sample(x=x0, n_steps=4, generator=_gen(cuda_device, 5))
    with torch.no_grad():
        model.net[4].weight.mul_(4.0)
    after = graph.sample(x=x0, n_steps=4, generator=_gen(cuda_device, 5))
    assert not torch.equal(before, after)
    assert torch.equal(after, eager.sample(x=x0, n_steps=4, generator=_gen(cuda_device, 5)))
    # a new batch shape re-captures
    d0 = hip_calls("ebm_langevin_step_dev_f32")
    small = graph.sample(x=x0[:100], n_steps=3, generator=_gen(cuda_device, 6))
    assert hip_calls("ebm_langevin_step_dev_f32") == d0 + 2
    assert torch.equal(small, eager.sample(x=x0[:100], n_steps=3, generator=_gen(cuda_device, 6)))


def test_scheduled_step_size_falls_back_to_the_eager_route(cuda_device):
    torch.manual_seed(0)
    model = Net().to(cuda_device)
    s = ta.LangevinDynamics(model, step_size=LinearScheduler(0.05, 0.01, 10), device=cuda_device)
    s.capture_graph = True
    s0 = hip_calls("ebm_langevin_step_f32")
    out = s.sample(x=two_moons(256, 0.05, seed=0, device=cuda_device), n_steps=6)
    assert hip_calls("ebm_langevin_step_f32") == s0 + 6 and torch.isfinite(out).all()


# ------------------------------------------------------------------------------------------
# HMC: one Metropolis transition per replay
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mass", [None, 1.7, "diag"])
def test_hmc_graph_route_is_bit_identical_to_the_eager_step_route(cuda_device, mass):
    torch.manual_seed(0)
    model = Net(dim=3, hidden=32).to(cuda_device)
    if mass == "diag":
        mass = torch.tensor([0.5, 1.0, 2.0], device=cuda_device)
    kw = dict(step_size=0.07, n_leapfrog_steps=4, mass=mass, device=cuda_device)
    eager, graph = ta.HamiltonianMonteCarlo(model, **kw), ta.HamiltonianMonteCarlo(model, **kw)
    eager.capture_graph = False
    graph.capture_graph = True
    x0 = torch.randn(777, 3, device=cuda_device)
    a0, d0 = hip_calls("ebm_hmc_accept_f32"), hip_calls("ebm_hmc_accept_dev_f32")
    want, dw = eager.sample(x=x0, n_steps=6, thin=2, return_diagnostics=True, generator=_gen(cuda_device, 9))
    got, dg = graph.sample(x=x0, n_steps=6, thin=2, return_diagnostics=True, generator=_gen(cuda_device, 9))
    assert hip_calls("ebm_hmc_accept_f32") == a0 + 6
    assert hip_calls("ebm_hmc_accept_dev_f32") == d0 + 2          # the first (eager) transition + the captured one
    assert torch.equal(got, want)
    assert torch.equal(dg["acceptance_rate"], dw["acceptance_rate"]) and 0.3 < dg["acceptance_rate"].mean().item() <= 1.0
    torch.testing.assert_close(dg["energy"], dw["energy"], rtol=1e-6, atol=1e-6)
    # continuing generators: fresh draws per call, same stream as the eager route, offset += 8 per transition
    ge, gg = _gen(cuda_device, 4), _gen(cuda_device, 4)
    e1, g1 = eager.sample(x=x0, n_steps=3, generator=ge), graph.sample(x=x0, n_steps=3, generator=gg)
    e2, g2 = eager.sample(x=x0, n_steps=3, generator=ge), graph.sample(x=x0, n_steps=3, generator=gg)
    assert hip_calls("ebm_hmc_accept_dev_f32") == d0 + 2          # graph re-used
    assert torch.equal(e1, g1) and torch.equal(e2, g2) and not torch.equal(g1, g2)
    assert ge.get_offset() == gg.get_offset() == 4 * 2 * 6
    # in-place weight updates reach the next replay
    with torch.no_grad():
        model.net[0].weight.mul_(2.0)
    assert torch.equal(graph.sample(x=x0, n_steps=2, generator=_gen(cuda_device, 5)),
                       eager.sample(x=x0, n_steps=2, generator=_gen(cuda_device, 5)))


def test_hmc_scheduled_step_size_falls_back_to_the_eager_route(cuda_device):
    torch.manual_seed(0)
    model = Net().to(cuda_device)
    s = ta.HamiltonianMonteCarlo(model, step_size=LinearScheduler(0.05, 0.01, 10), n_leapfrog_steps=3, device=cuda_device)
    s.capture_graph = True
    a0 = hip_calls("ebm_hmc_accept_f32")
    out = s.sample(x=two_moons(256, 0.05, seed=0, device=cuda_device), n_steps=4)
    assert hip_calls("ebm_hmc_accept_f32") == a0 + 4 and torch.isfinite(out).all()


def test_uncapturable_model_falls_back_to_the_eager_step_route(cuda_device):
    """A forward with a host sync cannot be captured: the sampler warns once, remembers THIS configuration as uncapturable
    (the flag is left alone: another model state or batch shape gets its own attempt) and continues with eager launches."""

    class Syncing(ta.BaseModel):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(2, 1)

        def forward(self, x):
            scale = float(x.detach().abs().max().item() > -1.0)   # .item(): a device-to-host sync
            return scale * self.lin(x).squeeze(-1) + 0.5 * (x ** 2).sum(-1)

    torch.manual_seed(0)
    model = Syncing().to(cuda_device)
    s = ta.LangevinDynamics(model, step_size=0.05, device=cuda_device)
    s.capture_graph = True
    x0 = torch.randn(128, 2, device=cuda_device)
    with pytest.warns(UserWarning, match="capture of the step route failed"):
        out = s.sample(x=x0, n_steps=4, generator=_gen(cuda_device, 1))
    assert s.capture_graph is True and len(s._graph_refused) == 1 and s._step_graph["graph"] is None and torch.isfinite(out).all()
    ref = ta.LangevinDynamics(model, step_size=0.05, device=cuda_device)
    ref.capture_graph = False
    assert torch.equal(out, ref.sample(x=x0, n_steps=4, generator=_gen(cuda_device, 1)))
    again = s.sample(x=x0, n_steps=4, generator=_gen(cuda_device, 1))  # no second attempt for the same configuration
    assert torch.equal(again, out) and len(s._graph_refused) == 1


def test_default_is_replay_and_python_state_changes_recapture(cuda_device):
    """replay is the DEFAULT when eligible.  A replay cannot see a changed Python attribute
    of the model, so the model's state key (core.module.graph_state_key) is part of the cache key."""

    class Tempered(ta.BaseModel):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(2, 1)
            self.temperature = 1.0

        def forward(self, x):
            return (self.lin(x).squeeze(-1) + 0.5 * (x ** 2).sum(-1)) / self.temperature

    torch.manual_seed(0)
    model = Tempered().to(cuda_device)
    s = ta.LangevinDynamics(model, step_size=0.05, device=cuda_device)
    e = ta.LangevinDynamics(model, step_size=0.05, device=cuda_device)
    e.capture_graph = False
    x0 = torch.randn(512, 2, device=cuda_device)
    d0, s0 = hip_calls("ebm_langevin_step_dev_f32"), hip_calls("ebm_langevin_step_f32")
    a = s.sample(x=x0, n_steps=10, generator=_gen(cuda_device, 1))
    assert hip_calls("ebm_langevin_step_dev_f32") == d0 + 2 and hip_calls("ebm_langevin_step_f32") == s0  # replayed by default
    assert torch.equal(a, e.sample(x=x0, n_steps=10, generator=_gen(cuda_device, 1)))
    short = s.sample(x=x0, n_steps=3, generator=_gen(cuda_device, 1))  # below GRAPH_MIN_STEPS: eager launches
    assert hip_calls("ebm_langevin_step_dev_f32") == d0 + 2 and torch.equal(short, e.sample(x=x0, n_steps=3, generator=_gen(cuda_device, 1)))
    model.temperature = 4.0  # a plain attribute the captured graph has frozen: must re-capture
    b = s.sample(x=x0, n_steps=10, generator=_gen(cuda_device, 1))
    assert hip_calls("ebm_langevin_step_dev_f32") == d0 + 4
    assert torch.equal(b, e.sample(x=x0, n_steps=10, generator=_gen(cuda_device, 1))) and not torch.equal(a, b)
    model.eval()  # so does train / eval
    s.sample(x=x0, n_steps=10, generator=_gen(cuda_device, 1))
    assert hip_calls("ebm_langevin_step_dev_f32") == d0 + 6


# ------------------------------------------------------------------------------------------
# the default must have no observable side effect
# ------------------------------------------------------------------------------------------
class _NormNet(ta.BaseModel):
    def __init__(self, drop=0.0):
        super().__init__()
        self.lin1, self.bn, self.lin2 = nn.Linear(2, 16), nn.BatchNorm1d(16), nn.Linear(16, 1)
        self.drop = nn.Dropout(drop) if drop else nn.Identity()

    def forward(self, x):
        return self.lin2(self.drop(torch.tanh(self.bn(self.lin1(x))))).squeeze(-1) + 0.5 * (x ** 2).sum(-1)


def test_default_route_leaves_batchnorm_statistics_exactly_as_eager_launches_do(cuda_device):
    """A model in training mode whose forward updates running statistics: the default route evaluates the forward exactly
    as often as the eager route (no extra warm-up passes), notices the buffer writes behind its first eager step and does
    not capture; the statistics -- and the samples -- are those of eager launches."""
    torch.manual_seed(0)
    a, b = _NormNet().to(cuda_device), _NormNet().to(cuda_device)
    b.load_state_dict(a.state_dict())
    x0 = torch.randn(256, 2, device=cuda_device)
    sa = ta.LangevinDynamics(a, step_size=0.05, device=cuda_device)           # default: capture_graph is None
    sb = ta.LangevinDynamics(b, step_size=0.05, device=cuda_device)
    sb.capture_graph = False
    out_a = sa.sample(x=x0, n_steps=12, generator=_gen(cuda_device, 1))
    out_b = sb.sample(x=x0, n_steps=12, generator=_gen(cuda_device, 1))
    assert torch.equal(out_a, out_b)
    assert torch.equal(a.bn.running_mean, b.bn.running_mean) and torch.equal(a.bn.running_var, b.bn.running_var)
    assert int(a.bn.num_batches_tracked) == int(b.bn.num_batches_tracked) == 12
    assert sa._step_graph["graph"] is None and "writes a buffer" in " ".join(next(iter(sa._graph_refused.values()))[1])
    # in eval mode the forward is pure: captured by default, still identical
    a.eval(); b.eval()
    out_a = sa.sample(x=x0, n_steps=12, generator=_gen(cuda_device, 2))
    assert sa._step_graph["graph"] is not None
    assert torch.equal(out_a, sb.sample(x=x0, n_steps=12, generator=_gen(cuda_device, 2)))
    assert int(a.bn.num_batches_tracked) == 12


def test_default_route_does_not_capture_a_forward_that_draws_random_numbers(cuda_device):
    torch.manual_seed(0)
    m = _NormNet(drop=0.25).to(cuda_device).eval()
    m.drop.train()                                                          # dropout active, BatchNorm frozen
    s = ta.LangevinDynamics(m, step_size=0.05, device=cuda_device)
    out = s.sample(x=torch.randn(128, 2, device=cuda_device), n_steps=10, generator=_gen(cuda_device, 1))
    assert torch.isfinite(out).all() and s._step_graph["graph"] is None
    assert "default CUDA generator" in " ".join(next(iter(s._graph_refused.values()))[1])
    h = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=3, device=cuda_device)
    out = h.sample(x=torch.randn(128, 2, device=cuda_device), n_steps=6, generator=_gen(cuda_device, 1))
    assert torch.isfinite(out).all() and h._step_graph["graph"] is None


def test_cache_key_sees_tensor_attributes_and_refuses_what_it_cannot_see(cuda_device):
    """a tensor kept as a plain attribute (replaced, or written in place through .data) re-captures; a model
    holding an object the key cannot hash is not captured by default."""

    class Scaled(ta.BaseModel):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(2, 1)
            self._scale = torch.tensor(1.0)

        def forward(self, x):
            return self._scale.to(x.device) * (self.lin(x).squeeze(-1) + 0.5 * (x ** 2).sum(-1))

    torch.manual_seed(0)
    m = Scaled().to(cuda_device)
    m._scale = m._scale.to(cuda_device)
    s = ta.LangevinDynamics(m, step_size=0.05, device=cuda_device)
    e = ta.LangevinDynamics(m, step_size=0.05, device=cuda_device)
    e.capture_graph = False
    x0 = torch.randn(64, 2, device=cuda_device)
    a = s.sample(x=x0, n_steps=10, generator=_gen(cuda_device, 1))
    assert s._step_graph["graph"] is not None and torch.equal(a, e.sample(x=x0, n_steps=10, generator=_gen(cuda_device, 1)))
    m._scale = torch.tensor(3.0, device=cuda_device)                        # REPLACED: new storage -> new key
    b = s.sample(x=x0, n_steps=10, generator=_gen(cuda_device, 1))
    assert torch.equal(b, e.sample(x=x0, n_steps=10, generator=_gen(cuda_device, 1))) and not torch.equal(a, b)
    m.opaque = object()                                                     # something the key cannot see into
    s.sample(x=x0, n_steps=10, generator=_gen(cuda_device, 1))
    assert s._step_graph["graph"] is None and "cannot see into" in " ".join(next(reversed(s._graph_refused.values()))[1])
    s.capture_graph = True                                                  # the user's call
    c = s.sample(x=x0, n_steps=10, generator=_gen(cuda_device, 1))
    assert s._step_graph["graph"] is not None and torch.equal(c, b)
