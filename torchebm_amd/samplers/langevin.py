r"""Langevin dynamics sampler (reference: torchebm/samplers/langevin_dynamics.py:16-188).

.. math:: x_{t+1} = x_t - \eta \nabla_x U(x_t) + \sqrt{2\eta}\,\sigma\,\epsilon_t

Three execution routes, chosen once per ``sample()`` call:

``fused``  CUDA fp32 state + one of the analytic energies + default Euler-Maruyama
           integrator: the whole k-step loop (gradient, update, Philox noise, clamp, thinned
           trajectory) is ONE launch of ``ebm_langevin_chain_f32``; schedulers are
           pre-expanded on the host into a per-step coefficient table.
``step``   CUDA fp32 state, any other model (e.g. an MLP energy, conditioning): the gradient
           comes from ``model.gradient`` (autograd) and each step is one launch of
           ``ebm_langevin_step_f32`` (update + noise + clamp fused).
``eager``  CPU state (BASELINE config 1) -- or a configuration outside the hot path
           (non-fp32, non-default integrator): the reference's loop with eager torch ops.

A CUDA state never silently runs on the CPU and never skips the HIP library: ``fused`` and
``step`` raise if ``libebm_hip.so`` is missing.
"""

from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple, Union

import torch

from .. import _lib, _rng
from ..core.energies import BaseModel, FusedSpec, fused_spec_for
from ..core.integrator_base import BaseSDERungeKuttaIntegrator
from ..core.module import ForwardProbe, graph_blind_spots, graph_state_key, warn_once
from ..core.sampler_base import BaseSampler
from ..core.schedules import BaseScheduler
from ..integrators.em import EulerMaruyamaIntegrator, HeunIntegrator
from ..integrators.registry import resolve_integrator


def em_coefficients(eta: float, sigma: float) -> Tuple[float, float, float]:
    """The three scalars of one Euler-Maruyama step, computed in double exactly as the
    reference's Python-float arithmetic does (base_integrator.py:728-729): they are cast to
    fp32 only when they enter the tensor ops / the kernel."""
    return eta, eta**0.5, (2.0 * sigma**2) ** 0.5


_RECORD_SCRATCH_KEEP_BYTES = 16 << 20
WIDE_GAUSSIAN_STEP_ROUTE_MIN_CHAINS = 16384  # below: k x (GEMM + update) launches lose to the one fused launch
_CAPTURE_LOCK = __import__("threading").Lock()  # serialises graph capture: it toggles the process-wide cyclic GC


def _record_scratch(sampler, device: torch.device, stream: int, rec_floats: int, work_doubles: int):
    """The record buffer and the merge's fp64 work row of a diagnostics call, kept on the sampler between calls of the same
    size on the same stream: the merge leaves the work row zeroed (include/ebm_hip.h), so nothing has to be allocated or
    filled per call -- on a 0.6 ms sampler call the two allocations and the fill kernel were a tenth of the records' cost.
    Keyed by the stream: calls on different streams do not share (the zeroed-after-merge contract is per stream order).
    Only SMALL buffers are kept (<= 16 MiB: the sub-millisecond calls this was built for); a large run's records (up to
    1 GiB) are per-call temporaries that go back to the caching allocator, as they did before."""
    if rec_floats * 4 > _RECORD_SCRATCH_KEEP_BYTES:
        sampler.__dict__.pop("_record_scratch_held", None)
        return (torch.empty(rec_floats, dtype=torch.float32, device=device),
                torch.zeros(work_doubles, dtype=torch.float64, device=device))
    key = (device, stream, rec_floats, work_doubles)
    held = sampler.__dict__.get("_record_scratch_held")
    if held is None or held[0] != key or sampler.__dict__.get("_record_scratch_open", False):
        # (open: the previous call did not reach its last merge -- an exception in between -- so the work row may not be zero)
        held = (key, torch.empty(rec_floats, dtype=torch.float32, device=device), torch.zeros(work_doubles, dtype=torch.float64, device=device))
        sampler.__dict__["_record_scratch_held"] = held
    sampler.__dict__["_record_scratch_open"] = True   # closed by the caller after its last merge (_record_scratch_done)
    return held[1], held[2]


def _record_scratch_done(sampler) -> None:
    sampler.__dict__["_record_scratch_open"] = False


def _gaussian_chain_on_matrix_cores(dim: int) -> bool:
    """Widths at which ``ebm_langevin_chain_f32`` runs the dense Gaussian on the matrix cores (csrc/gauss_mfma.hip: up to 128,
    packed rows below 20 included; csrc/gauss_shift.hip, gauss_res_shift.hip: widths off multiples of 4 up to 254 on shifted rows;
    csrc/gauss_big.hip: multiples of 4 up to 512)."""
    return dim <= 254 or (dim % 4 == 0 and dim <= 512)


def _replay_or_step(sampler, g, first: bool, more: bool) -> None:
    """One iteration of the graph route: a replay once the graph exists; until then the step body runs eagerly (a real
    step), and behind the FIRST one of a call the graph is captured -- unless, by default, that step showed side effects
    (``ForwardProbe``) or the model has attributes the cache key cannot see into.  Shared by the Langevin and HMC samplers
    (``sampler.capture_graph``, ``sampler._graph_refused``)."""
    if g["graph"] is not None:
        g["graph"].replay()
        return
    refused = getattr(sampler, "_graph_refused", None)
    if refused is None:
        refused = sampler._graph_refused = {}
    prior = refused.get(g["key"])  # ("probe" | "failed", reasons): a default-mode refusal does not bind capture_graph = True
    decide = first and more and (prior is None or (sampler.capture_graph is True and prior[0] == "probe"))
    probe = ForwardProbe(sampler.model, g["state"].device) if decide and sampler.capture_graph is not True else None
    g["body"]()
    if not decide:
        return
    why = []
    if probe is not None:
        why = probe.side_effects()
        blind = graph_blind_spots(sampler.model)
        if blind:
            why.append("attributes the cache key cannot see into: " + ", ".join(blind[:4]))
    if not why:
        # The cyclic collector must not run inside the capture: finalising a dead sampler's CUDAGraph (or any tensor whose
        # storage goes back to the driver) while this stream records aborts the process.  torch.cuda.graph() collects
        # once on entry; what becomes garbage during the body waits until the capture has ended.
        # The GC switch is process-wide: the lock keeps a second sampler capturing on another thread from re-enabling it in
        # the middle of this capture (and the prior state is read inside the lock).
        import gc

        with _CAPTURE_LOCK:
            gc_was_on = gc.isenabled()
            try:
                graph = torch.cuda.CUDAGraph()
                gc.disable()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    g["body"]()
                g["graph"] = graph
                return
            except RuntimeError as exc:  # the forward cannot be captured (host sync, data-dependent control flow ...)
                if "ebm_" in str(exc):
                    raise
                warn_once("capture-graph-failed", f"torchebm_amd: HIP-graph capture of the step route failed ({exc}); "
                          "continuing with eager launches.", UserWarning)
                probe, why = None, [f"capture failed: {exc}"]
            finally:
                if gc_was_on:
                    gc.enable()
    refused.pop(g["key"], None)
    if len(refused) >= 8:
        refused.pop(next(iter(refused)))
    refused[g["key"]] = ("failed" if probe is None else "probe", why)


class LangevinDynamics(BaseSampler):
    """Langevin dynamics sampler.

    Args:
        model: energy model to sample from.
        step_size: step size, a float or a ``BaseScheduler``.
        noise_scale: noise scale, a float or a ``BaseScheduler``.
        decay: stored, unused (as in the reference).
        clamp: optional ``(min, max)`` applied to the state after every step.
        dtype, device: where the chains live.
        integrator: ``None`` (Euler-Maruyama), a registry name, or a
            ``BaseSDERungeKuttaIntegrator`` instance matching the sampler's device/dtype.
    """

    def __init__(
        self,
        model: BaseModel,
        step_size: Union[float, BaseScheduler] = 1e-3,
        noise_scale: Union[float, BaseScheduler] = 1.0,
        decay: float = 0.0,
        clamp: Optional[Tuple[float, float]] = None,
        dtype: torch.dtype = torch.float32,
        device: Optional[Union[str, torch.device]] = None,
        integrator: Union[str, BaseSDERungeKuttaIntegrator, None] = None,
    ):
        super().__init__(model=model, dtype=dtype, device=device)
        self._register_param("step_size", step_size, positive=True)
        self._register_param("noise_scale", noise_scale, positive=True)
        if clamp is not None and clamp[0] >= clamp[1]:
            raise ValueError(f"clamp min must be < max, got {clamp}")
        self.clamp = clamp
        self.decay = decay
        self.integrator = resolve_integrator(
            integrator,
            default="euler_maruyama",
            family=BaseSDERungeKuttaIntegrator,
            owner="LangevinDynamics",
            device=self.device,
            dtype=self.dtype,
        )

    # ---------------------------------------------------------------------------------
    # routing
    # ---------------------------------------------------------------------------------
    def _route(self, x: torch.Tensor, model_kwargs: Dict[str, Any]) -> Tuple[str, Optional[FusedSpec]]:
        if not x.is_cuda:
            return "eager", None
        plain_em = type(self.integrator) is EulerMaruyamaIntegrator
        heun = type(self.integrator) is HeunIntegrator
        if heun and x.dtype == torch.float32 and not (self.use_mixed_precision and self.autocast_available):
            # Heun has a fused kernel for the analytic energies only; anything else keeps the eager loop
            spec = self._fusable_spec(x, model_kwargs)
            if spec is not None and not spec.langevin_only:
                return "fused", spec
        if x.dtype != torch.float32 or not plain_em or (self.use_mixed_precision and self.autocast_available):
            warn_once(
                "langevin-eager-cuda",
                "torchebm_amd: LangevinDynamics with a non-fp32 state, autocast, or a non-default integrator is "
                "not accelerated by the HIP kernels; running the eager torch loop on the GPU.",
                UserWarning,
            )
            return "eager", None
        spec = self._fusable_spec(x, model_kwargs)
        if (spec is not None and spec.kind == _lib.ENERGY_GAUSSIAN and not _gaussian_chain_on_matrix_cores(spec.dim)
                and x.shape[0] >= WIDE_GAUSSIAN_STEP_ROUTE_MIN_CHAINS and self.capture_graph is not False
                and self._graph_eligible(model_kwargs)):
            # The chain kernels for these widths (not a multiple of 4 above 128, or above 512) are the lane-group mat-vec:
            # 2 TFLOP/s.  The step route -- one library GEMM for the gradient (GaussianModel._hip_gradient) and the fused
            # update kernel on the same random field, replayed from a HIP graph -- is 10 - 40x faster there, for batches
            # that fill the GEMM and calls that can be replayed.  Few chains, capture_graph = False, a scheduled step size
            # or conditioning keep the ONE fused launch (with its in-kernel records and donate_input).
            return "step", None
        return ("fused", spec) if spec is not None else ("step", None)

    def _fusable_spec(self, x: torch.Tensor, model_kwargs: Dict[str, Any]) -> Optional[FusedSpec]:
        # element-wise energies run on the flat kernel, which has no row-width limit
        return fused_spec_for(self.model, x, model_kwargs, cap_elementwise=False)

    # ---------------------------------------------------------------------------------
    # public API
    # ---------------------------------------------------------------------------------
    @torch.no_grad()
    def sample(
        self,
        x: Optional[torch.Tensor] = None,
        dim: Optional[Union[int, Tuple[int, ...]]] = None,
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
        """Generate samples.

        Returns the final state ``[n, *shape]`` (or, with ``return_trajectory``, the kept
        trajectory ``[n, n_steps // thin, *shape]``), optionally with a diagnostics dict
        holding ``"mean"``/``"var"`` (``[n_kept, *shape]``) and ``"energy"`` (``[n_kept]``).

        Raises:
            ValueError: ``thin < 1``, or ``x`` and ``dim`` both ``None``.
        """
        if thin < 1:
            raise ValueError("thin must be >= 1")
        if reset_schedulers:
            self.reset_schedulers()
        x = self._init_state(x, dim, n_samples, generator)
        model_kwargs = self._prepare_model_kwargs(model_kwargs)
        route, spec = self._route(x, model_kwargs)
        if route == "fused":
            return self._sample_fused(x, spec, n_steps, thin, return_trajectory, return_diagnostics, generator)
        return self._sample_stepwise(
            x, model_kwargs, n_steps, thin, return_trajectory, return_diagnostics, generator, hip=(route == "step")
        )

    # ---------------------------------------------------------------------------------
    # shared helpers
    # ---------------------------------------------------------------------------------
    def _new_outputs(self, x: torch.Tensor, n_kept: int, want_traj: bool, want_diag: bool):
        n, shape = x.shape[0], tuple(x.shape[1:])
        traj = torch.empty((n, n_kept, *shape), dtype=self.dtype, device=self.device) if want_traj else None
        diag = None
        if want_diag:
            diag = {
                "mean": torch.empty(n_kept, *shape, dtype=self.dtype, device=self.device),
                "var": torch.empty(n_kept, *shape, dtype=self.dtype, device=self.device),
                "energy": torch.empty(n_kept, dtype=self.dtype, device=self.device),
            }
        return traj, diag

    def _clamp_args(self) -> Tuple[int, float, float]:
        if self.clamp is None:
            return 0, 0.0, 0.0
        return 1, float(self.clamp[0]), float(self.clamp[1])

    # ---------------------------------------------------------------------------------
    # route: per-step loop (eager torch ops, or the per-step HIP kernel)
    # ---------------------------------------------------------------------------------
    def _sample_stepwise(self, x, model_kwargs, n_steps, thin, want_traj, want_diag, generator, hip: bool):
        n = x.shape[0]
        n_kept = n_steps // thin
        traj, diag = self._new_outputs(x, n_kept, want_traj, want_diag)
        keep = 0
        if hip and self._use_graph(model_kwargs, n_steps):
            return self._sample_graph(x, n_steps, thin, traj, diag, want_traj, want_diag, generator)
        if hip:
            x = _lib.dense_f32(x)
            seed, step0 = _rng.reserve(generator, x.device, n_steps)
            stream = _lib.stream_handle(x.device)
            clamp_on, cmin, cmax = self._clamp_args()
        else:
            drift = lambda x_, t_: -self._model_gradient(x_, model_kwargs)  # noqa: E731
        with self.autocast_context():
            for i in range(n_steps):
                eta = self.get_scheduled_value("step_size")
                sigma = self.get_scheduled_value("noise_scale")
                if hip:
                    grad = _lib.dense_f32(self._model_gradient(x, model_kwargs))
                    out = torch.empty_like(x)
                    a, sq, coef = em_coefficients(eta, sigma)
                    _lib.call(
                        "ebm_langevin_step_f32",
                        _lib.ptr(x), _lib.ptr(grad), _lib.ptr(out), None, x.numel(),
                        a, sq, coef, clamp_on, cmin, cmax, seed, step0 + i, stream,
                    )
                    x = out
                else:
                    x = self.integrator.step(
                        state={"x": x}, step_size=eta, noise_scale=sigma, drift=drift, generator=generator
                    )["x"]
                    if self.clamp is not None:
                        x = x.clamp_(*self.clamp)
                self.step_schedulers()

                if (i + 1) % thin == 0:
                    if traj is not None:
                        traj[:, keep] = x
                    if diag is not None:
                        if n > 1:
                            diag["mean"][keep] = x.mean(dim=0)
                            diag["var"][keep] = x.var(dim=0, unbiased=False).clamp_(min=1e-10, max=1e10)
                        else:
                            diag["mean"][keep] = x.squeeze(0)
                            diag["var"][keep].zero_()
                        diag["energy"][keep] = self._model_energy(x, model_kwargs).mean()
                    keep += 1
        out = traj if want_traj else x
        return (out, diag) if want_diag else out

    # ---------------------------------------------------------------------------------
    # route: per-step loop replayed from a HIP graph (the default whenever the configuration is eligible)
    # ---------------------------------------------------------------------------------
    #: Capture one "autograd gradient + fused update" iteration of the step route into a HIP graph and
    #: replay it n_steps times.  The step route is launch-bound (BASELINE config 5: ~13 small
    #: kernels per Langevin step): a replay costs one submission instead of 13.  The Philox
    #: coordinates live in a device buffer advanced inside the graph (``ebm_langevin_step_dev_f32``), so
    #: every replay draws fresh noise and the generator contract is unchanged; the result is bit-identical
    #: to the eager step route.
    #:   ``None`` (default)  replay whenever the call is eligible -- constant step size / noise scale, no conditioning, no
    #:                       autocast, at least ``GRAPH_MIN_STEPS`` steps -- AND capturing has no observable side effect:
    #:                       the first step of the call runs eagerly (it is the warm-up; nothing is evaluated that the
    #:                       eager route would not evaluate), and the graph is captured behind it only if that step
    #:                       wrote no buffer of the model (BatchNorm statistics in training mode), drew nothing from the
    #:                       default CUDA / CPU generators (dropout) and changed no attribute (``core.module.ForwardProbe``),
    #:                       and the model holds no attribute the cache key cannot see into
    #:                       (``core.module.graph_blind_spots``); otherwise the call continues with eager launches.
    #:                       The graph is kept for the next call and re-captured when the batch shape, the coefficients
    #:                       or the model's state key (``core.module.graph_state_key``: parameter / buffer storages, every
    #:                       hashable attribute of every submodule incl. tensor-valued ones, ``training``, autocast)
    #:                       change.  A forward that cannot be captured (host sync, data-dependent control flow) continues
    #:                       with eager launches with a one-time warning; only THAT configuration is remembered as
    #:                       uncapturable.  Capture uses ``capture_error_mode="thread_local"``: work other threads
    #:                       submit meanwhile (a DataLoader's pinning thread) neither breaks it nor is broken by it.
    #:                       Blind spots that remain: state reached through a closure or a global, tensors replaced
    #:                       inside containers the key hashes by storage, side effects outside the model.
    #:   ``True``            capture whenever eligible, without the minimum step count and without the side-effect checks.
    #:   ``False``           always eager launches.
    capture_graph: Optional[bool] = None
    GRAPH_MIN_STEPS = 8

    def _use_graph(self, model_kwargs: Dict[str, Any], n_steps: int) -> bool:
        if self.capture_graph is False or not self._graph_eligible(model_kwargs):
            return False
        return self.capture_graph is True or n_steps >= self.GRAPH_MIN_STEPS

    def _graph_eligible(self, model_kwargs: Dict[str, Any]) -> bool:
        return (
            not model_kwargs
            and not self.use_mixed_precision
            and self.schedulers["step_size"].is_constant()
            and self.schedulers["noise_scale"].is_constant()
        )

    def _graph_for(self, x: torch.Tensor):
        """The replay state for this (batch shape, coefficients, model state): static buffers + the step body; the graph
        itself is captured by ``_sample_graph`` behind the first eager step."""
        a, sq, coef = em_coefficients(self.get_scheduled_value("step_size"), self.get_scheduled_value("noise_scale"))
        key = (
            tuple(x.shape), x.device, (a, sq, coef), self._clamp_args(),
            graph_state_key(self.model),
        )
        cached = getattr(self, "_step_graph", None)
        if cached is not None and cached["key"] == key:
            return cached
        state = torch.empty_like(x)                                        # static buffers of the graph
        rng = torch.zeros(2, dtype=torch.int64, device=x.device)           # {seed, step} as raw 64-bit patterns
        clamp_on, cmin, cmax = self._clamp_args()

        def body():
            grad = _lib.dense_f32(self._model_gradient(state, {}))
            _lib.call(
                "ebm_langevin_step_dev_f32",
                _lib.ptr(state), _lib.ptr(grad), _lib.ptr(state), state.numel(), a, sq, coef,
                clamp_on, cmin, cmax, _lib.ptr(rng), _lib.stream_handle(x.device),
            )
            rng[1:2].add_(1)

        self._step_graph = {"key": key, "graph": None, "state": state, "rng": rng, "body": body}
        return self._step_graph

    def _sample_graph(self, x, n_steps, thin, traj, diag, want_traj, want_diag, generator):
        x = _lib.dense_f32(x)
        n = x.shape[0]
        g = self._graph_for(x)
        seed, step0 = _rng.reserve(generator, x.device, n_steps)
        as_i64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v  # noqa: E731  (bit pattern of a uint64)
        g["state"].copy_(x)
        g["rng"].copy_(torch.tensor([as_i64(seed), as_i64(step0)], dtype=torch.int64), non_blocking=True)
        state, keep = g["state"], 0
        for i in range(n_steps):
            _replay_or_step(self, g, first=(i == 0), more=(i + 1 < n_steps))
            if (i + 1) % thin == 0:
                if traj is not None:
                    traj[:, keep] = state
                if diag is not None:
                    if n > 1:
                        diag["mean"][keep] = state.mean(dim=0)
                        diag["var"][keep] = state.var(dim=0, unbiased=False).clamp_(min=1e-10, max=1e10)
                    else:
                        diag["mean"][keep] = state.squeeze(0)
                        diag["var"][keep].zero_()
                    diag["energy"][keep] = self._model_energy(state, {}).mean()
                keep += 1
        self.advance_schedulers(n_steps)
        out = traj if want_traj else state.clone()
        return (out, diag) if want_diag else out

    # ---------------------------------------------------------------------------------
    # route: k-fused HIP kernel
    # ---------------------------------------------------------------------------------
    def _coef_rows(self, k: int) -> Tuple[bool, List[Tuple[float, float, float]]]:
        """Pre-expand the two schedulers for the next k steps (values are read before
        ``step_schedulers()`` in the reference loop, i.e. at step counts 0..k-1)."""
        s_eta, s_sig = self.schedulers["step_size"], self.schedulers["noise_scale"]
        constant = s_eta.is_constant() and s_sig.is_constant()
        if constant:
            return True, [em_coefficients(s_eta.get_value(), s_sig.get_value())]
        etas, sigmas = s_eta.preview(k), s_sig.preview(k)
        return False, [em_coefficients(e, s) for e, s in zip(etas, sigmas)]

    def _coef_table(self, rows, device) -> Optional[torch.Tensor]:
        """Device table ``float[k][4]`` of a non-constant schedule, kept for the next call with the same schedule
        (a training loop calls ``sample`` with an unchanged scheduler thousands of times)."""
        if len(rows) == 1:
            return None
        key = (device, tuple(rows))
        cached = getattr(self, "_table_cache", None)
        if cached is None or cached[0] != key:
            host = torch.tensor([(r[0], r[1], r[2], 0.0) for r in rows], dtype=torch.float32)
            cached = (key, host.to(device, non_blocking=True))
            self._table_cache = cached
        return cached[1]

    def _launch_chain(self, spec_c, x, n, dim, rows, row0, k, thin, traj, seed, step, stream, table=None, records=None):
        """One ``ebm_langevin_chain_f32`` launch for steps [row0, row0+k) of ``rows``."""
        clamp_on, cmin, cmax = self._clamp_args()
        if len(rows) == 1:  # constant schedule: scalars, no table
            a, sq, coef = rows[0]
            tab = None
        else:
            a, sq, coef = rows[row0]
            tab = (self._coef_table(rows, x.device) if table is None else table)[row0 : row0 + k]
        coords = getattr(self, "_graph_coords", None)
        if coords is not None:  # being captured in a HIP graph: `step` is an offset from the device-resident coordinates
            _lib.call(
                "ebm_langevin_chain_dev_f32",
                spec_c, _lib.ptr(x), n, dim, k, a, sq, coef, _lib.ptr(tab),
                clamp_on, cmin, cmax, thin, _lib.ptr(traj), _lib.ptr(coords.tensor), step, stream,
            )
            return
        entry = "ebm_langevin_heun_chain_f32" if type(self.integrator) is HeunIntegrator else "ebm_langevin_chain_f32"
        flags = clamp_on | (_lib.CHAIN_CONTRACTED if self.fused_arithmetic else 0)  # (ABI 8: a flag word; the bit is a permission)
        _lib.call(
            entry,
            spec_c, _lib.ptr(x), n, dim, k, a, sq, coef, _lib.ptr(tab),
            flags, cmin, cmax, thin, _lib.ptr(traj), _lib.ptr(records), None, seed, step, stream,
        )

    #: Opt-in (an ATTRIBUTE, ``sampler.fused_arithmetic = True``: the constructor keeps the reference's signature, whose own tests pin
    #: ``integrator`` as its last parameter).  ``False``: every kernel rounds each multiply and add of the update on its own, as the
    #: reference's eager torch operations do (core/base_integrator.py:711-731) -- chains are the reference's bit for bit given the
    #: same draws.  ``True`` PERMITS contracted arithmetic where a kernel has that form (the element-wise energies' plain k-step call:
    #: ``x^2 - b^2`` and ``x - eta g`` as fused multiply-adds, ``noise_scale``'s coefficient times ``sqrt(step_size)`` folded into the
    #: Box-Muller radius): the same Philox draws and the same law, states that differ from the default's in the last bits of every
    #: step, 12 % more chain-steps per second on BASELINE config 2 (``EBM_CHAIN_CONTRACTED``, include/ebm_hip.h; bench.py
    #: ``config2_fused_arithmetic``).  Calls without such a kernel run the default arithmetic.
    fused_arithmetic: bool = False

    #: Opt-in: let the fused route update the caller's ``x`` in place and return it (no defensive copy of the
    #: state -- 256 MiB per call at BASELINE config 2).  Off by default: the reference never mutates its input.
    donate_input: bool = False

    #: upper bound on the bytes of per-block diagnostics records held at once; longer runs are cut into several
    #: launches at kept-step boundaries (the state is re-read once per cut)
    DIAG_RECORD_BYTES = 1 << 30

    def _sample_fused(self, x, spec: FusedSpec, n_steps, thin, want_traj, want_diag, generator):
        n, dim = x.shape
        n_kept = n_steps // thin
        state = _lib.dense_f32(x)
        if state.data_ptr() == x.data_ptr() and not self.donate_input:
            state = state.clone()  # the kernel updates in place; never touch the caller's tensor unless it was donated
        traj, diag = self._new_outputs(x, n_kept, want_traj, want_diag)
        _, rows = self._coef_rows(n_steps)
        coords = getattr(self, "_graph_coords", None)
        heun = type(self.integrator) is HeunIntegrator
        if coords is not None:
            # utils.graphed_step is capturing this call: coordinates come from device memory (ebm_langevin_chain_dev_f32)
            if want_diag or heun or spec.kind != _lib.ENERGY_MLP or len(rows) != 1:
                raise RuntimeError(
                    "torchebm_amd: only the plain fused call on an MLPEnergy (constant step size / noise scale, no diagnostics) "
                    "can be captured in a training-step graph"
                )
            seed, step0 = 0, coords.take(n_steps)
        else:
            seed, step0 = _rng.reserve(generator, x.device, n_steps)
        stream = _lib.stream_handle(x.device)
        spec_c = spec.to_c()

        if n_steps > 0 and n > 0:
            layout = None
            if want_diag and n_kept > 0:
                layout = _lib.diag_layout(spec_c, _lib.DIAG_LANGEVIN_HEUN if heun else _lib.DIAG_LANGEVIN, n, dim, False, want_traj)
            if not want_diag or n_kept == 0:
                # one launch for the whole call; thinned rows are stored by the kernel
                self._launch_chain(spec_c, state, n, dim, rows, 0, n_steps, thin, traj, seed, step0, stream)
            elif layout is not None:
                self._fused_with_records(spec_c, state, n, dim, rows, n_steps, thin, traj, diag, layout, seed, step0, stream)
            else:
                self._fused_with_state_passes(spec_c, state, n, dim, rows, n_steps, thin, traj, diag, seed, step0, stream)
        self.advance_schedulers(n_steps)
        out = traj if want_traj else state
        return (out, diag) if want_diag else out

    def _fused_with_records(self, spec_c, state, n, dim, rows, n_steps, thin, traj, diag, layout, seed, step0, stream):
        """Diagnostics from inside the chain launch (include/ebm_hip.h: ``diag_partials``): every workgroup stores
        one record of its chains' partial sums per kept step, ``ebm_diag_finish_f32`` merges them.  One chain
        launch + one merge launch per call (several only when the records of all kept steps would not fit
        ``DIAG_RECORD_BYTES``)."""
        n_blocks, slots, block_elems = layout
        n_kept = n_steps // thin
        # slots > dim: PACKED rows (a Gaussian whose width the matrix-layout kernel does not take as is: csrc/gauss_mfma.hip,
        # gauss_pack_factor) -- `pack` consecutive chains are one row of width pack * dim; the records and the merge are those of
        # n / pack rows, and the `pack` columns of every coordinate are pooled afterwards
        pack = slots // dim if slots > dim else 1
        m_n, m_dim = n // pack, dim * pack
        rec_floats = n_blocks * (2 * slots + 8)
        chunk = max(1, min(n_kept, self.DIAG_RECORD_BYTES // (4 * rec_floats)))
        records, work = _record_scratch(self, state.device, stream, chunk * rec_floats, chunk * (3 * m_dim + 3))
        if pack > 1:
            p_mean = torch.empty(chunk, m_dim, dtype=torch.float32, device=state.device)
            p_var = torch.empty_like(p_mean)
            p_energy = torch.empty(chunk, dtype=torch.float32, device=state.device)
        done_keep, done_steps = 0, 0
        while done_keep < n_kept:
            kk = min(chunk, n_kept - done_keep)
            steps = kk * thin
            if done_keep + kk == n_kept:
                steps = n_steps - done_steps  # the trailing n_steps % thin steps ride along (no kept step among them)
            whole = traj is not None and kk == n_kept
            piece = traj if (traj is None or whole) else torch.empty(n, kk, dim, dtype=torch.float32, device=state.device)
            self._launch_chain(spec_c, state, n, dim, rows, done_steps, steps, thin, piece, seed, step0 + done_steps, stream,
                               records=records)
            if traj is not None and not whole:
                traj[:, done_keep : done_keep + kk] = piece
            sl = slice(done_keep, done_keep + kk)
            if pack == 1:
                _lib.call(
                    "ebm_diag_finish_f32", _lib.ptr(records), kk, n_blocks, slots, block_elems, n, dim,
                    _lib.ptr(diag["mean"][sl]), _lib.ptr(diag["var"][sl]), _lib.ptr(diag["energy"][sl]), None, _lib.ptr(work), stream,
                )
            else:
                _lib.call(
                    "ebm_diag_finish_f32", _lib.ptr(records), kk, n_blocks, slots, block_elems, m_n, m_dim,
                    _lib.ptr(p_mean), _lib.ptr(p_var), _lib.ptr(p_energy), None, _lib.ptr(work), stream,
                )
                # pool the `pack` equally sized groups of every coordinate: within-group variance + variance of the group means
                gm = p_mean[:kk].view(kk, pack, dim).double()
                mean = gm.mean(dim=1)
                var = p_var[:kk].view(kk, pack, dim).double().mean(dim=1) + (gm - mean.unsqueeze(1)).square().mean(dim=1)
                diag["mean"][sl] = mean.to(torch.float32)
                diag["var"][sl] = var.to(torch.float32).clamp_(min=1e-10, max=1e10)
                diag["energy"][sl] = p_energy[:kk] / pack  # a packed row's energy is the sum of its chains'
            done_keep += kk
            done_steps += steps
        _record_scratch_done(self)

    def _fused_with_state_passes(self, spec_c, state, n, dim, rows, n_steps, thin, traj, diag, seed, step0, stream):
        """Diagnostics for the configurations without in-kernel records (the matrix-layout kernels of the MLP
        energy, rows that neither divide nor are divided by the flat kernel's 1024-element blocks): one launch per
        ``thin`` steps, then the column-statistics and energy kernels on the state."""
        n_kept = n_steps // thin
        work = torch.zeros(2 * dim + 1, dtype=torch.float64, device=state.device)  # the kernel leaves it zeroed
        energy = torch.empty(n, dtype=torch.float32, device=state.device)
        # a dense Gaussian above 128 dims: ebm_energy_grad_f32 is the lane-group mat-vec there (2 TFLOP/s: at 2^17 x 256 one
        # energy pass would cost more than the 20 chain steps before it)
        from ..core.energies import GaussianModel
        wide_gaussian = type(self.model) is GaussianModel and dim > GaussianModel.CLOSED_FORM_GRADIENT_ABOVE
        done = 0
        for keep in range(n_kept):
            self._launch_chain(spec_c, state, n, dim, rows, done, thin, thin, None, seed, step0 + done, stream)
            done += thin
            if traj is not None:
                traj[:, keep] = state
            if n > 1:
                _lib.call(
                    "ebm_chain_stats_f32",
                    _lib.ptr(state), n, dim, _lib.ptr(diag["mean"][keep]), _lib.ptr(diag["var"][keep]),
                    _lib.ptr(work), stream,
                )
            else:
                diag["mean"][keep] = state[0]
                diag["var"][keep].zero_()
            if wide_gaussian:
                diag["energy"][keep] = self._model_energy(state, {}).mean()   # one library GEMM (GaussianModel.forward, wide CUDA batches)
            else:
                _lib.call("ebm_energy_grad_f32", spec_c, _lib.ptr(state), n, dim, _lib.ptr(energy), None, stream)
                diag["energy"][keep] = energy.mean()
        if done < n_steps:
            self._launch_chain(spec_c, state, n, dim, rows, done, n_steps - done, thin, None, seed, step0 + done, stream)
