r"""Hamiltonian Monte Carlo sampler (reference: torchebm/samplers/hmc.py:19-315).

Per transition: draw ``p ~ N(0, M)``; ``H0 = U(x) + K(p)``; ``L`` leapfrog steps in safe
mode; ``H1``; accept with probability ``min(1, exp(H0 - H1))``.  The RNG draw order per
transition is momentum normals, then the accept uniforms.

Execution routes (chosen once per ``sample()`` call):

``fused``  CUDA fp32 2-D state + analytic energy + default leapfrog: all transitions run
           inside ONE launch of ``ebm_hmc_chain_f32`` (per-chain energies are wavefront
           reductions, the accept decision never leaves the wave).
``step``   CUDA fp32, any other model: Philox momentum fill, ``LeapfrogIntegrator`` on the
           HIP kick kernels around ``model.gradient`` (autograd), HIP Metropolis kernel.
``eager``  CPU state, or configurations outside the hot path: the reference loop in torch.

``RiemannianManifoldHMC`` of the reference is out of scope.
"""

from __future__ import annotations

import math
from typing import Any, Dict, Optional, Tuple, Union

import torch

from .. import _lib, _rng
from ..core.energies import BaseModel, FusedSpec, fused_spec_for
from ..core.integrator_base import BaseSymplecticIntegrator
from ..core.module import graph_state_key, warn_once
from .langevin import _record_scratch, _record_scratch_done, _replay_or_step
from ..core.sampler_base import BaseSampler
from ..core.schedules import BaseScheduler
from ..integrators.registry import resolve_integrator
from ..integrators.symplectic import LeapfrogIntegrator, _mass_args


class HamiltonianMonteCarlo(BaseSampler):
    """Hamiltonian Monte Carlo.

    Args:
        model: energy model.
        step_size: leapfrog step size (float or ``BaseScheduler``).
        n_leapfrog_steps: leapfrog steps per proposal.
        mass: ``None``, a float, or a per-dimension tensor.
        dtype, device: where the chains live.
        integrator: ``None`` (leapfrog), a registry name, or a separable
            ``BaseSymplecticIntegrator`` instance matching the sampler's device/dtype.
    """

    #: Opt-in, an ATTRIBUTE (``sampler.exact = True``; the constructor keeps the reference's signature, whose own tests pin
    #: ``integrator`` as its last parameter).
    #: ``False`` (default): the fused transition kernel's FAST leapfrog body -- merged half kicks, fused
    #: multiply-adds, ``eps / m`` hoisted: not less accurate than the reference's own fp32 run against an fp64 referee
    #: (medians 0.5 - 0.8x, tests/test_grid_gpu.py), identical accept decisions on every recorded fixture, but not the
    #: reference's operation sequence: after 200 leapfrog steps a quarter of the chains differ from the reference's
    #: states by more than 5e-4 (tests/test_hmc_audit_gpu.py reports the figure).  ``True``: the reference's safe-mode
    #: sequence LITERALLY (torchebm/samplers/hmc.py:243-312, integrators/leapfrog.py:156-185: separate half kicks, every
    #: multiply and add rounded on its own, the drift divided by ``max(m, 1e-10)`` per step, both scrubs, energy and
    #: force re-evaluated at the top of every transition).  For the element-wise energies (``DoubleWellModel``,
    #: ``HarmonicModel``; dim <= 256) that is one launch of the library's literal kernel -- given the same momenta and
    #: uniforms the states are the reference's bit for bit -- at 1.9x the fast body's time (2 L + 2 evaluations per
    #: transition instead of L; 1.18 against 0.61 ms at 2^18 x 32, L = 20, 10 transitions).  For every other energy
    #: ``exact=True`` takes the per-transition route: the reference's own torch operations around the HIP momentum /
    #: accept kernels.  (The in-kernel Philox field is the package's on every GPU route; ``exact`` is about the
    #: arithmetic between the draws.)
    exact: bool = False

    #: energies the literal kernel (``ebm_hmc_chain_audit_f32``) takes, and its widest state
    _EXACT_KINDS = (_lib.ENERGY_DOUBLE_WELL, _lib.ENERGY_HARMONIC)
    _EXACT_MAX_DIM = 256

    def __init__(
        self,
        model: BaseModel,
        step_size: Union[float, BaseScheduler] = 1e-3,
        n_leapfrog_steps: int = 10,
        mass: Optional[Union[float, torch.Tensor]] = None,
        dtype: torch.dtype = torch.float32,
        device: Optional[Union[str, torch.device]] = None,
        integrator: Union[str, BaseSymplecticIntegrator, None] = None,
    ):
        super().__init__(model=model, dtype=dtype, device=device)
        self._register_param("step_size", step_size, positive=True)
        if n_leapfrog_steps <= 0:
            raise ValueError("n_leapfrog_steps must be positive")
        self.n_leapfrog_steps = n_leapfrog_steps
        if mass is not None and not isinstance(mass, float):
            mass = mass.to(self.device)
        self.mass = mass
        integ = resolve_integrator(
            integrator,
            default="leapfrog",
            family=BaseSymplecticIntegrator,
            owner="HamiltonianMonteCarlo",
            device=self.device,
            dtype=self.dtype,
        )
        if not integ.separable:
            raise TypeError(
                "HamiltonianMonteCarlo requires a separable symplectic integrator (drift/mass contract); "
                f"got non-separable {type(integ).__name__}. Use RiemannianManifoldHMC for non-separable "
                "Hamiltonians."
            )
        self.integrator = integ

    # ---------------------------------------------------------------------------------
    # momentum / kinetic energy in torch ops (eager + step routes; hmc.py:92-159)
    # ---------------------------------------------------------------------------------
    def _scale_momentum_(self, p: torch.Tensor) -> torch.Tensor:
        if self.mass is None:
            return p
        if isinstance(self.mass, float):
            return p.mul_(math.sqrt(self.mass))
        return p.mul_(torch.sqrt(self.mass).view((1,) * (p.ndim - 1) + (-1,)))

    def _initialize_momentum(self, shape: torch.Size, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """``p ~ N(0, M)`` with torch's own generator stream (CPU / eager route)."""
        buf = getattr(self, "_momentum_buf", None)
        if buf is None or buf.shape != shape or buf.dtype != self.dtype or buf.device != self.device:
            buf = torch.empty(shape, dtype=self.dtype, device=self.device)
            self._momentum_buf = buf
        return self._scale_momentum_(buf.normal_(generator=generator))

    def _compute_kinetic_energy(self, p: torch.Tensor) -> torch.Tensor:
        """``K(p) = 0.5 p^T M^{-1} p`` per chain."""
        if self.mass is None:
            return 0.5 * torch.sum(p.square(), dim=-1)
        if isinstance(self.mass, float):
            return 0.5 * torch.sum(p.square(), dim=-1) / self.mass
        return 0.5 * torch.sum(p.square() / self.mass.view((1,) * (p.ndim - 1) + (-1,)), dim=-1)

    # ---------------------------------------------------------------------------------
    # routing
    # ---------------------------------------------------------------------------------
    def _route(self, x: torch.Tensor, model_kwargs: Dict[str, Any]) -> Tuple[str, Optional[FusedSpec]]:
        if not x.is_cuda:
            return "eager", None
        plain = type(self.integrator) is LeapfrogIntegrator
        if x.dtype != torch.float32 or not plain or (self.use_mixed_precision and self.autocast_available):
            warn_once(
                "hmc-eager-cuda",
                "torchebm_amd: HamiltonianMonteCarlo with a non-fp32 state, autocast, or a non-default integrator "
                "is not accelerated by the HIP kernels; running the eager torch loop on the GPU.",
                UserWarning,
            )
            return "eager", None
        spec = None
        mass_ok = self.mass is None or isinstance(self.mass, float) or (
            torch.is_tensor(self.mass) and self.mass.ndim == 1 and self.mass.numel() == x.shape[-1]
        )
        if mass_ok:
            spec = fused_spec_for(self.model, x, model_kwargs)
            if spec is not None and not spec.hmc:
                spec = None  # (a wide MLP energy: fused for Langevin only)
            if spec is not None and self.exact and not (spec.kind in self._EXACT_KINDS and x.shape[-1] <= self._EXACT_MAX_DIM):
                spec = None  # exact=True without a literal kernel for this energy: the reference's torch operations, per transition
            if (spec is not None and spec.kind == _lib.ENERGY_GAUSSIAN and spec.dim is not None
                    and spec.dim > 160  # (up to 160: the split operands stay in LDS -- five tiles; widths off multiples of 4 on shifted rows)
                    and not (spec.dim <= 256 and spec.aux is not None)  # (161 .. 256: the slabs stream from the pre-split image -- one per alignment class off multiples of 4, up to 254)
                    and x.shape[0] >= 16384 and self.capture_graph is not False and self._graph_eligible(model_kwargs)):
                # Above 256 dims (255 included) the transition kernel is the lane-group mat-vec (2 TFLOP/s); the per-transition route -- the
                # gradient and the energy as one library GEMM each (GaussianModel), kick / drift / accept kernels on the same
                # random field, replayed from a HIP graph -- is 1.5 - 30x faster there (scripts/bench_gauss_hmc_big.py), for
                # batches that fill the GEMMs and calls that can be replayed; otherwise the one fused launch is kept.
                spec = None
        return ("fused", spec) if spec is not None else ("step", None)

    # ---------------------------------------------------------------------------------
    # public API
    # ---------------------------------------------------------------------------------
    @torch.no_grad()
    def sample(
        self,
        x: Optional[torch.Tensor] = None,
        dim: Optional[int] = None,
        n_steps: int = 100,
        n_samples: int = 1,
        thin: int = 1,
        return_trajectory: bool = False,
        return_diagnostics: bool = False,
        reset_schedulers: bool = True,
        *,
        model_kwargs: Optional[Dict[str, Any]] = None,
        generator: Optional[torch.Generator] = None,
    ) -> Union[torch.Tensor, Tuple[torch.Tensor, Dict[str, torch.Tensor]]]:
        """Generate samples; ``n_steps`` is the number of MH proposals.

        Diagnostics keys: ``"mean"``, ``"var"`` (``[n_kept, dim]``), ``"energy"`` and
        ``"acceptance_rate"`` (``[n_kept]``).  ``dim`` is inferred from ``model.mean`` when
        neither ``x`` nor ``dim`` is given.

        Raises:
            ValueError: ``thin < 1``, or the state dimension cannot be determined.
        """
        if thin < 1:
            raise ValueError("thin must be >= 1")
        if reset_schedulers:
            self.reset_schedulers()
        model_kwargs = self._prepare_model_kwargs(model_kwargs)
        if x is None and dim is None:
            mean = getattr(self.model, "mean", None)
            if not isinstance(mean, torch.Tensor):
                raise ValueError("dim must be provided when x is None and cannot be inferred from model")
            dim = mean.shape[0]
        x = self._init_state(x, dim, n_samples, generator)
        route, spec = self._route(x, model_kwargs)
        if route == "fused":
            return self._sample_fused(x, spec, n_steps, thin, return_trajectory, return_diagnostics, generator)
        return self._sample_stepwise(
            x, model_kwargs, n_steps, thin, return_trajectory, return_diagnostics, generator, hip=(route == "step")
        )

    def _new_outputs(self, n: int, dim: int, n_kept: int, want_traj: bool, want_diag: bool):
        kw = dict(dtype=self.dtype, device=self.device)
        traj = torch.empty((n, n_kept, dim), **kw) if want_traj else None
        diag = None
        if want_diag:
            diag = {
                "mean": torch.empty(n_kept, dim, **kw),
                "var": torch.empty(n_kept, dim, **kw),
                "energy": torch.empty(n_kept, **kw),
                "acceptance_rate": torch.empty(n_kept, **kw),
            }
        return traj, diag

    # ---------------------------------------------------------------------------------
    # route: per-transition loop
    # ---------------------------------------------------------------------------------
    def _sample_stepwise(self, x, model_kwargs, n_steps, thin, want_traj, want_diag, generator, hip: bool):
        n, dim = x.shape[0], x.shape[1]
        n_kept = n_steps // thin
        traj, diag = self._new_outputs(n, dim, n_kept, want_traj, want_diag)
        drift = lambda x_, t_: -self._model_gradient(x_, model_kwargs)  # noqa: E731
        if hip and n > 0 and self._use_graph(model_kwargs, n_steps):
            return self._sample_graph(x, n_steps, thin, traj, diag, want_traj, want_diag, generator)
        if hip:
            x = _lib.dense_f32(x)
            seed, step0 = _rng.reserve(generator, x.device, 2 * n_steps)
            stream = _lib.stream_handle(x.device)
            row_dim = x.numel() // max(n, 1)
        keep = 0
        with self.autocast_context():
            for i in range(n_steps):
                if hip:
                    p = torch.empty_like(x)
                    _lib.call("ebm_noise_fill_f32", _lib.ptr(p), p.numel(), _lib.NOISE_NORMAL, seed, step0 + 2 * i, stream)
                    p = self._scale_momentum_(p)
                else:
                    p = self._initialize_momentum(x.shape, generator)

                h0 = self._model_energy(x, model_kwargs).clamp_(min=-1e10, max=1e10) + self._compute_kinetic_energy(
                    p
                ).clamp_(min=0.0, max=1e10)
                self.integrator.carry_force = bool(self.carry_force)
                prop = self.integrator.integrate(
                    {"x": x, "p": p},
                    step_size=self.get_scheduled_value("step_size"),
                    n_steps=self.n_leapfrog_steps,
                    mass=self.mass,
                    drift=drift,
                    safe=True,
                )
                x_prop, p_prop = prop["x"], prop["p"]
                h1 = self._model_energy(x_prop, model_kwargs).clamp_(min=-1e10, max=1e10) + self._compute_kinetic_energy(
                    p_prop
                ).clamp_(min=0.0, max=1e10)

                if hip:
                    mask = torch.empty(n, dtype=torch.uint8, device=x.device)
                    x_next = x.clone()
                    # locals keep any dense copies alive until the launch that reads them is enqueued
                    xp_d, h0_d, h1_d = _lib.dense_f32(x_prop), _lib.dense_f32(h0), _lib.dense_f32(h1)
                    _lib.call(
                        "ebm_hmc_accept_f32",
                        _lib.ptr(x_next), _lib.ptr(xp_d), _lib.ptr(h0_d), _lib.ptr(h1_d), None, _lib.ptr(mask), None,
                        n, row_dim, seed, step0 + 2 * i + 1, stream,
                    )
                    x = x_next
                    accepted = mask
                else:
                    delta = (h0 - h1).clamp_(min=-50.0, max=50.0)
                    accept_prob = torch.exp(delta).clamp_(max=1.0)
                    u = torch.rand(n, device=self.device, generator=generator)
                    accepted = u < accept_prob
                    x = torch.where(accepted.view(-1, *([1] * (x.ndim - 1))), x_prop, x)

                if (i + 1) % thin == 0:
                    if traj is not None:
                        traj[:, keep, :] = x
                    if diag is not None:
                        diag["mean"][keep] = x.mean(dim=0)
                        diag["var"][keep] = (
                            x.var(dim=0, unbiased=False).clamp_(min=1e-10, max=1e10)
                            if n > 1
                            else torch.zeros(dim, dtype=self.dtype, device=self.device)
                        )
                        diag["energy"][keep] = self._model_energy(x, model_kwargs).clamp_(min=-1e10, max=1e10).mean()
                        diag["acceptance_rate"][keep] = accepted.float().mean()
                    keep += 1
                self.step_schedulers()
        out = traj if want_traj else x
        return (out, diag) if want_diag else out

    # ---------------------------------------------------------------------------------
    # route: one transition of the step route captured in a HIP graph (the default whenever eligible)
    # ---------------------------------------------------------------------------------
    #: Capture ONE Metropolis transition of the step route -- momentum draw, H0, L leapfrog steps (kick /
    #: autograd gradient / kick), H1, accept -- into a HIP graph and replay it n_steps times.  With an
    #: nn.Module energy a transition is ~15 small launches per leapfrog step; a replay is one submission.
    #: The Philox coordinates live in a device buffer the graph advances by 2 per transition
    #: (``ebm_noise_fill_dev_f32`` at +0, ``ebm_hmc_accept_dev_f32`` at +1), so the generator contract and
    #: the noise field are those of the eager step route, bit for bit.  Requirements: constant step
    #: size, no conditioning, a model whose forward is static-shape and free of host-side randomness.
    #: ``None`` (default): replay whenever eligible, the call has at least ``GRAPH_MIN_STEPS`` transitions and the first
    #: (eager) transition showed no side effect of the model's forward; ``True``: whenever eligible; ``False``: never.
    #: See ``LangevinDynamics.capture_graph`` for the rules (first transition eager = the warm-up, thread-local capture,
    #: per-configuration refusal).
    capture_graph: Optional[bool] = None
    GRAPH_MIN_STEPS = 4
    #: Step route only (the fused kernels always do this, bit-identically): reuse the force a leapfrog step ends on as
    #: the force the next one starts from -- L + 1 gradient evaluations per transition instead of the reference's 2 L,
    #: i.e. 1.8x fewer autograd round trips at L = 10.  Off by default because it is observable in two corners: an
    #: energy whose forward is stochastic (dropout in training mode, RNG calls) sees half as many evaluations, and a
    #: chain whose position had to be scrubbed in safe mode keeps the unscrubbed position's force for one step.
    carry_force: bool = False

    def _graph_eligible(self, model_kwargs: Dict[str, Any]) -> bool:
        return not model_kwargs and not self.use_mixed_precision and self.schedulers["step_size"].is_constant()

    def _use_graph(self, model_kwargs: Dict[str, Any], n_steps: int) -> bool:
        if self.capture_graph is False or not self._graph_eligible(model_kwargs):
            return False
        return self.capture_graph is True or n_steps >= self.GRAPH_MIN_STEPS

    def _graph_for(self, x: torch.Tensor):
        eps = self.get_scheduled_value("step_size")
        key = (
            tuple(x.shape), x.device, eps, self.n_leapfrog_steps, id(self.integrator), bool(self.carry_force),
            None if self.mass is None else (self.mass if isinstance(self.mass, float) else self.mass.data_ptr()),
            graph_state_key(self.model),
        )
        cached = getattr(self, "_step_graph", None)
        if cached is not None and cached["key"] == key:
            return cached
        n = x.shape[0]
        row_dim = x.numel() // n
        state = torch.empty_like(x)                                        # static buffers of the graph
        rng = torch.zeros(2, dtype=torch.int64, device=x.device)           # {seed, step} as raw 64-bit patterns
        mask = torch.zeros(n, dtype=torch.uint8, device=x.device)
        drift = lambda x_, t_: -self._model_gradient(x_, {})  # noqa: E731

        def body():
            stream = _lib.stream_handle(x.device)
            p = torch.empty_like(state)
            _lib.call("ebm_noise_fill_dev_f32", _lib.ptr(p), p.numel(), _lib.NOISE_NORMAL, _lib.ptr(rng), 0, stream)
            p = self._scale_momentum_(p)
            h0 = self._model_energy(state, {}).clamp_(min=-1e10, max=1e10) + self._compute_kinetic_energy(p).clamp_(
                min=0.0, max=1e10)
            self.integrator.carry_force = bool(self.carry_force)
            prop = self.integrator.integrate(
                {"x": state, "p": p}, step_size=eps, n_steps=self.n_leapfrog_steps, mass=self.mass, drift=drift, safe=True)
            h1 = self._model_energy(prop["x"], {}).clamp_(min=-1e10, max=1e10) + self._compute_kinetic_energy(
                prop["p"]).clamp_(min=0.0, max=1e10)
            xp_d, h0_d, h1_d = _lib.dense_f32(prop["x"]), _lib.dense_f32(h0), _lib.dense_f32(h1)
            _lib.call(
                "ebm_hmc_accept_dev_f32",
                _lib.ptr(state), _lib.ptr(xp_d), _lib.ptr(h0_d), _lib.ptr(h1_d), _lib.ptr(mask), None,
                n, row_dim, _lib.ptr(rng), 1, stream,
            )
            rng[1:2].add_(2)

        self._step_graph = {"key": key, "graph": None, "state": state, "rng": rng, "mask": mask, "body": body}
        return self._step_graph

    def _sample_graph(self, x, n_steps, thin, traj, diag, want_traj, want_diag, generator):
        x = _lib.dense_f32(x)
        n, dim = x.shape[0], x.shape[1]
        g = self._graph_for(x)
        seed, step0 = _rng.reserve(generator, x.device, 2 * n_steps)
        as_i64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v  # noqa: E731  (bit pattern of a uint64)
        g["state"].copy_(x)
        g["rng"].copy_(torch.tensor([as_i64(seed), as_i64(step0)], dtype=torch.int64), non_blocking=True)
        state, keep = g["state"], 0
        for i in range(n_steps):
            _replay_or_step(self, g, first=(i == 0), more=(i + 1 < n_steps))
            if (i + 1) % thin == 0:
                if traj is not None:
                    traj[:, keep, :] = state
                if diag is not None:
                    diag["mean"][keep] = state.mean(dim=0)
                    diag["var"][keep] = (
                        state.var(dim=0, unbiased=False).clamp_(min=1e-10, max=1e10)
                        if n > 1
                        else torch.zeros(dim, dtype=self.dtype, device=self.device)
                    )
                    diag["energy"][keep] = self._model_energy(state, {}).clamp_(min=-1e10, max=1e10).mean()
                    diag["acceptance_rate"][keep] = g["mask"].float().mean()
                keep += 1
        self.advance_schedulers(n_steps)
        out = traj if want_traj else state.clone()
        return (out, diag) if want_diag else out

    # ---------------------------------------------------------------------------------
    # route: fused HIP kernel
    # ---------------------------------------------------------------------------------
    def _eps_table(self, eps_vals, device) -> Optional[torch.Tensor]:
        if len(eps_vals) == 1:
            return None
        key = (device, tuple(eps_vals))
        cached = getattr(self, "_table_cache", None)
        if cached is None or cached[0] != key:
            cached = (key, torch.tensor(eps_vals, dtype=torch.float32).to(device, non_blocking=True))
            self._table_cache = cached
        return cached[1]

    def _launch_hmc(self, spec_c, state, n, dim, eps_vals, t0, n_mh, thin, traj, counts, seed, step, stream, records=None):
        kind, m_scalar, m_diag = _mass_args(self.mass, state)
        if len(eps_vals) == 1:
            eps, table = eps_vals[0], None
        else:
            eps = eps_vals[t0]
            table = self._eps_table(eps_vals, state.device)[t0 : t0 + n_mh]
        cptr = None if counts is None else counts.data_ptr() + 4 * t0
        if self.exact:  # the literal body (no in-kernel records: `_sample_fused` asks for none)
            assert records is None
            _lib.call(
                "ebm_hmc_chain_audit_f32",
                spec_c, _lib.ptr(state), n, dim, n_mh, self.n_leapfrog_steps, eps, _lib.ptr(table),
                kind, m_scalar, _lib.ptr(m_diag), thin, _lib.ptr(traj), None, cptr, None, None,
                seed, step, stream,
            )
            return
        _lib.call(
            "ebm_hmc_chain_f32",
            spec_c, _lib.ptr(state), n, dim, n_mh, self.n_leapfrog_steps, eps, _lib.ptr(table),
            kind, m_scalar, _lib.ptr(m_diag), thin, _lib.ptr(traj), _lib.ptr(records), None, cptr, None, None,
            seed, step, stream,
        )

    #: see LangevinDynamics.donate_input / DIAG_RECORD_BYTES
    donate_input: bool = False
    DIAG_RECORD_BYTES = 1 << 30

    def _sample_fused(self, x, spec: FusedSpec, n_steps, thin, want_traj, want_diag, generator):
        n, dim = x.shape
        n_kept = n_steps // thin
        state = _lib.dense_f32(x)
        if state.data_ptr() == x.data_ptr() and not self.donate_input:
            state = state.clone()  # the kernel updates in place; never touch the caller's tensor unless it was donated
        traj, diag = self._new_outputs(n, dim, n_kept, want_traj, want_diag)
        sched = self.schedulers["step_size"]
        eps_vals = [sched.get_value()] if sched.is_constant() else sched.preview(n_steps)
        seed, step0 = _rng.reserve(generator, x.device, 2 * n_steps)
        stream = _lib.stream_handle(x.device)
        spec_c = spec.to_c()

        if n_steps > 0 and n > 0:
            layout = _lib.diag_layout(spec_c, _lib.DIAG_HMC, n, dim, False, want_traj) if (want_diag and n_kept > 0 and not self.exact) else None
            if not want_diag or n_kept == 0:
                self._launch_hmc(spec_c, state, n, dim, eps_vals, 0, n_steps, thin, traj, None, seed, step0, stream)
            elif layout is not None:
                self._fused_with_records(spec_c, state, n, dim, eps_vals, n_steps, thin, traj, diag, layout, seed, step0, stream)
            else:
                self._fused_with_state_passes(spec_c, state, n, dim, eps_vals, n_steps, thin, traj, diag, seed, step0, stream)
        self.advance_schedulers(n_steps)
        out = traj if want_traj else state
        return (out, diag) if want_diag else out

    def _fused_with_records(self, spec_c, state, n, dim, eps_vals, n_steps, thin, traj, diag, layout, seed, step0, stream):
        """Diagnostics from inside the transition kernel (include/ebm_hip.h: ``diag_partials``): at every kept
        transition each workgroup stores one record (column sums / M2, the energy of the states its chains hold,
        its accept count); ``ebm_diag_finish_f32`` merges them.  One chain launch + one merge launch per call."""
        n_blocks, slots, block_elems = layout
        n_kept = n_steps // thin
        rec_floats = n_blocks * (2 * slots + 8)
        chunk = max(1, min(n_kept, self.DIAG_RECORD_BYTES // (4 * rec_floats)))
        records, work = _record_scratch(self, state.device, stream, chunk * rec_floats, chunk * (3 * dim + 3))
        done_keep, done = 0, 0
        while done_keep < n_kept:
            kk = min(chunk, n_kept - done_keep)
            n_mh = kk * thin
            if done_keep + kk == n_kept:
                n_mh = n_steps - done  # trailing n_steps % thin transitions ride along
            whole = traj is not None and kk == n_kept
            piece = traj if (traj is None or whole) else torch.empty(n, kk, dim, dtype=torch.float32, device=state.device)
            self._launch_hmc(spec_c, state, n, dim, eps_vals, done, n_mh, thin, piece, None, seed, step0 + 2 * done, stream,
                             records=records)
            if traj is not None and not whole:
                traj[:, done_keep : done_keep + kk] = piece
            sl = slice(done_keep, done_keep + kk)
            _lib.call(
                "ebm_diag_finish_f32", _lib.ptr(records), kk, n_blocks, slots, block_elems, n, dim,
                _lib.ptr(diag["mean"][sl]), _lib.ptr(diag["var"][sl]), _lib.ptr(diag["energy"][sl]),
                _lib.ptr(diag["acceptance_rate"][sl]), _lib.ptr(work), stream,
            )
            done_keep += kk
            done += n_mh
        _record_scratch_done(self)

    def _fused_with_state_passes(self, spec_c, state, n, dim, eps_vals, n_steps, thin, traj, diag, seed, step0, stream):
        """Diagnostics for energies without in-kernel records (the MLP energy's matrix-layout kernel): one launch
        per ``thin`` transitions, then the column-statistics and energy kernels on the state."""
        n_kept = n_steps // thin
        counts = torch.zeros(n_steps, dtype=torch.int32, device=state.device)  # uint32 bit pattern
        work = torch.zeros(2 * dim + 1, dtype=torch.float64, device=state.device)  # the kernel leaves it zeroed
        energy = torch.empty(n, dtype=torch.float32, device=state.device)
        done = 0
        for keep in range(n_kept):
            self._launch_hmc(spec_c, state, n, dim, eps_vals, done, thin, thin, None, counts, seed, step0 + 2 * done, stream)
            done += thin
            if traj is not None:
                traj[:, keep, :] = state
            if n > 1:
                _lib.call(
                    "ebm_chain_stats_f32",
                    _lib.ptr(state), n, dim, _lib.ptr(diag["mean"][keep]), _lib.ptr(diag["var"][keep]),
                    _lib.ptr(work), stream,
                )
            else:
                diag["mean"][keep] = state[0]
                diag["var"][keep].zero_()
            _lib.call("ebm_energy_grad_f32", spec_c, _lib.ptr(state), n, dim, _lib.ptr(energy), None, stream)
            diag["energy"][keep] = energy.clamp_(min=-1e10, max=1e10).mean()
            diag["acceptance_rate"][keep] = counts[done - 1].to(torch.float32) / n
        if done < n_steps:
            self._launch_hmc(spec_c, state, n, dim, eps_vals, done, n_steps - done, thin, None, None, seed,
                             step0 + 2 * done, stream)
