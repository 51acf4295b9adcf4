"""A whole contrastive-divergence TRAINING STEP as one HIP graph (opt-in).

BASELINE config 5's training step is ``loss, neg = cd(x); opt.zero_grad(); loss.backward(); opt.step()``
(reference: torchebm/losses/contrastive_divergence.py:82-155, core/base_loss.py:266-337, the loop of
examples/20-training/01-mcmc-losses/02-persistent-cd/main.py).  With the fused sampler the chain itself is one
0.5 ms launch, and what is left -- the loss forward / backward of a 2-128-128-1 network and Adam, some sixty small
kernels -- is launch-bound: 2.0 of the step's 2.55 ms (profiles/r04_bench_n1.json).  ``GraphedTrainingStep`` captures the
WHOLE step -- start points from the replay buffer, the k-fused chain, the FIFO write, loss, backward, optimiser -- once
and replays it: one graph launch per training step.

What makes the step replayable: everything the host would bake into kernel arguments changes from step to step, so it
lives in device memory that launches inside the graph read and that the graph itself advances --

* the kernels' Philox coordinates ``{seed, step}`` (``_rng.DeviceCoords``; the ``*_dev_f32`` entry points of
  include/ebm_hip.h, ABI 6): every launch draws at ``step + delta`` with the deltas handed out in program order exactly
  as ``_rng.reserve`` advances the generator between the same calls of the eager step, and a one-element add at the
  end of the graph moves ``step`` past the whole training step.  The host mirrors that on the ``torch.Generator``
  (no device read), so the generator ends where the eager loop leaves it and **the same seed gives the same chains,
  losses and weights as the eager loop** as long as no torch-side draw is part of the step -- true of the loss's defaults,
  exploration noise included (``new_sample_ratio > 0``: the random subset of rows and its normals come from
  ``ebm_pcd_start_points_f32``, i.e. from the kernels' own field); with a torch-side draw in the step
  (``add_noise_to_real = True``: ``randn_like`` on the data) that draw comes from torch's own graph-safe generator state -- the same
  law at other offsets than the eager loop's;
* the FIFO write position of the replay buffer (``buffer_ptr``, advanced in the graph; the host copy follows);
* the weights: parameters, gradients and optimiser state are ordinary tensors updated in place by the captured
  optimiser step (``capturable=True`` for Adam-family optimisers), and the fused sampler re-packs the parameters
  inside the graph, so every replay samples from the weights the previous replay left.

One caveat that is not this package's: EAGER steps of a second ``Adam(capturable=True)`` run between the replays of a
captured one make the captured one drift from its eager loop -- plain PyTorch on this stack does the same
(scripts/probes/torch_graph_adam_interference.py); a trainer that owns its GPU stream, as every training loop does, is
bit-identical to its eager loop (tests/test_graphed_step_gpu.py).

A second one, PyTorch's as well: an autograd graph through the model's parameters that was built on ANOTHER stream and is still alive
when the step is captured (a loss tensor kept from an eager call on the default stream, say) leaves gradient-accumulation nodes pinned
to that stream, and the captured backward then has to cross streams -- torch warns ("AccumulateGrad node's stream does not match") and
the capture would abort the process.  The warm-up steps watch for that warning and the capture is then REFUSED (``RuntimeError``):
evaluate diagnostics under ``torch.no_grad()`` or drop their tensors, and build a new ``GraphedTrainingStep``.

Refusals (``ValueError`` at construction or at the first call, each naming its reason): a sampler call that is not the
plain fused call on an ``MLPEnergy`` (the only chain kernels that take device-resident coordinates), scheduled step
sizes, conditioning kwargs, an optimiser that is not capturable, a model holding attributes a replay could not see
change (``core.module.graph_blind_spots`` -- the same rule the samplers' own step-route graphs follow).
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import _lib, _rng
from ..core.loss_base import BaseContrastiveDivergence
from ..core.module import graph_blind_spots, graph_state_key

__all__ = ["GraphedTrainingStep"]


class GraphedTrainingStep:
    """``step = GraphedTrainingStep(cd_loss, optimizer); loss, negatives = step(x)`` -- one training step per call.

    The first ``eager_steps`` calls (default 2) run the ordinary eager step: they initialise the replay buffer, the
    optimiser state and the allocator.  The next call captures the graph, and it and every later one replay it.  A new
    batch shape, or a model / loss whose Python-side state changed (``train()`` / ``eval()``, a replaced parameter),
    re-captures.  ``loss`` is a fresh 0-dim tensor; ``negatives`` is the graph's static output buffer -- valid until the
    next call (``.clone()`` it to keep it).

    Optimiser: any ``torch.optim`` optimiser that can be captured (``capturable=True`` for the Adam family).  With the package's
    kernels a config-5 step is ~0.65 ms of GPU work, and torch's default multi-tensor Adam adds 27 launches (~0.1 ms) to it;
    ``Adam(..., capturable=True, fused=True)`` is one launch (bench.py: ``with_torch_fused_adam``).

    Host-side scalars are part of the capture: the optimiser's float hyper-parameters (``lr``, ``betas``, ``weight_decay`` ...), the
    sampler's constant step size / noise scale / clamp bounds and the loss's ``noise_scale`` are launch constants of the captured
    kernels.  Their VALUES are in the capture key, so changing one (an LR scheduler stepping a float ``lr``, a manual
    ``param_groups[0]["lr"] = ...``) re-captures on the next call -- correct, but a capture costs tens of milliseconds: for a
    per-step LR schedule give the optimiser a TENSOR lr (``Adam(lr=torch.tensor(1e-3, device=...), capturable=True)``), which the
    captured kernels read at replay time.  ``recaptures`` counts the captures after the first.

    Gradients: the captured backward allocates ``p.grad`` in the graph's memory pool and every replay rewrites those buffers.  Read
    ``p.grad`` after a call if you need it, but do not call ``optimizer.zero_grad()`` yourself between calls -- it would detach
    ``p.grad`` from the buffers the replays keep writing (the step zeroes its own gradients).

    ``enabled=False`` makes every call the eager step (same code path as the warm-up): the switch the tests use to
    compare the two.
    """

    def __init__(self, loss_fn: BaseContrastiveDivergence, optimizer: torch.optim.Optimizer, *,
                 generator: Optional[torch.Generator] = None, eager_steps: int = 2, enabled: bool = True, force: bool = False):
        if not isinstance(loss_fn, BaseContrastiveDivergence):
            raise ValueError("GraphedTrainingStep drives a contrastive-divergence loss (BaseContrastiveDivergence)")
        self.loss_fn, self.optimizer, self.generator = loss_fn, optimizer, generator
        self.eager_steps, self.enabled, self.force = max(1, int(eager_steps)), enabled, force
        self.calls = 0
        self.replays = 0
        self.recaptures = 0
        self._g = None
        self._stream = None
        reason = self._static_refusal()
        if reason and enabled:
            raise ValueError(f"GraphedTrainingStep: {reason}")

    # ---------------------------------------------------------------------------------------------------------------
    def _static_refusal(self) -> Optional[str]:
        from ..samplers.langevin import LangevinDynamics

        lf = self.loss_fn
        if torch.device(lf.device).type != "cuda" or lf.dtype != torch.float32:
            return "the loss must live on a CUDA device in float32"
        if getattr(lf, "reference_subset", False) and getattr(lf, "new_sample_ratio", 0.0) > 0.0:
            return "reference_subset=True draws the exploration subset with torch.randperm (a device sort on the caller's generator): not capturable"
        s = lf.sampler
        if type(s) is not LangevinDynamics:
            return "the sampler must be torchebm_amd.LangevinDynamics (its fused chain kernel is the one that takes device-resident RNG coordinates)"
        if not (s.schedulers["step_size"].is_constant() and s.schedulers["noise_scale"].is_constant()):
            return "scheduled step sizes / noise scales advance on the host; a replay would repeat the captured values"
        for group in self.optimizer.param_groups:
            if "capturable" in group and not group["capturable"]:
                return f"{type(self.optimizer).__name__} must be built with capturable=True (its step counter has to live on the device)"
        if not self.force:
            blind = graph_blind_spots(lf.model)
            if blind:
                return ("the model holds attributes a replay could not see change (" + ", ".join(blind[:4]) +
                        "); pass force=True if they never change")
        return None

    def _route_refusal(self, x: torch.Tensor) -> Optional[str]:
        s = self.loss_fn.sampler
        route, spec = s._route(x.detach(), {})
        if route != "fused" or spec is None or spec.kind != _lib.ENERGY_MLP:
            return ("the sampler call must be the fused chain launch on an MLPEnergy (got the '" + route + "' route): the other chain "
                    "kernels take their RNG coordinates by value and would repeat the captured draws")
        return None

    def _eager(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        loss, neg = self.loss_fn(x, generator=self.generator)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward(self._unit(loss))
        self.optimizer.step()
        return loss.detach(), neg

    def _unit(self, like: torch.Tensor) -> torch.Tensor:
        """dL/dL = 1 as a tensor that already exists (``backward()`` without it fills a fresh one: a launch per step)."""
        one = getattr(self, "_one", None)
        if one is None or one.device != like.device or one.dtype != like.dtype:
            one = self._one = torch.ones((), dtype=like.dtype, device=like.device)
        return one

    def _side_stream(self, device: torch.device) -> torch.cuda.Stream:
        if self._stream is None or self._stream.device != device:
            self._stream = torch.cuda.Stream(device)
        return self._stream

    def _warmup(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """A real (eager) training step on the SIDE stream the capture will use.  Autograd pins a parameter's
        gradient-accumulation node to the stream it first ran on; had the warm-up run on the caller's stream, the captured
        backward would hand its gradients across streams (torch warns that this "may break CUDA graph capture").  The
        pattern torch's own documentation prescribes for whole-network capture."""
        if not x.is_cuda:
            return self._eager(x)
        cur, side = torch.cuda.current_stream(x.device), self._side_stream(x.device)
        side.wait_stream(cur)
        import warnings

        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            with torch.cuda.stream(side):
                loss, neg = self._eager(x)
        for w in seen:
            if "AccumulateGrad node's stream does not match" in str(w.message):
                # an autograd graph through the parameters, built on another stream, is still alive: capturing this backward would
                # cross streams inside the capture and abort the process -- remember it and refuse to capture (see the module text)
                self._stale_graph = True
            else:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        cur.wait_stream(side)
        loss.record_stream(cur)
        neg.record_stream(cur)
        return loss, neg

    @staticmethod
    def _host_value(v):
        """How a host-side hyper-parameter enters the capture key: a Python scalar is baked into the captured kernels' arguments (its
        VALUE is the key: a change re-captures); a tensor is read by the kernels at replay time (its identity is the key)."""
        if isinstance(v, torch.Tensor):
            return ("tensor", v.data_ptr(), tuple(v.shape), v.dtype)
        if isinstance(v, (list, tuple)):
            return tuple(GraphedTrainingStep._host_value(e) for e in v)
        if v is None or isinstance(v, (bool, int, float, str)):
            return v
        return repr(v)

    def _optimizer_key(self):
        # every scalar of every parameter group (lr, betas, eps, weight_decay, momentum, ...): the captured optimiser kernels carry
        # float hyper-parameters as launch constants, so an LR scheduler or a manual `param_groups[i]["lr"] = ...` between replays
        # would otherwise be silently ignored.  A tensor lr (torch's capturable form) is live and keys by identity.
        return tuple(tuple((k, self._host_value(g[k])) for k in sorted(g) if k != "params") for g in self.optimizer.param_groups)

    def _key(self, x: torch.Tensor):
        lf, s = self.loss_fn, self.loss_fn.sampler
        sampler = (self._host_value(s.schedulers["step_size"].get_value()), self._host_value(s.schedulers["noise_scale"].get_value()),
                   self._host_value(getattr(s, "clamp", None)), self._host_value(getattr(s, "fused_arithmetic", None)))
        return (tuple(x.shape), x.device, x.dtype, graph_state_key(lf.model), lf.training, lf.persistent, lf.buffer_size,
                lf.k_steps, getattr(lf, "new_sample_ratio", None), getattr(lf, "energy_reg_weight", None),
                getattr(lf, "add_noise_to_real", None), self._host_value(getattr(lf, "noise_scale", None)), sampler,
                self._optimizer_key())

    def _capture(self, x: torch.Tensor, key) -> dict:
        lf, dev = self.loss_fn, x.device
        coords = _rng.DeviceCoords(dev)
        static_x = x.detach().clone()
        advance = torch.zeros(1, dtype=torch.int64, device=dev)  # steps the graph moves the coordinates by (filled below)
        unit = self._unit(static_x.new_zeros(()))
        self.optimizer.zero_grad(set_to_none=True)  # the captured backward allocates the gradients in the graph's pool
        graph = torch.cuda.CUDAGraph()
        if self.generator is not None:
            graph.register_generator_state(self.generator)
        import gc

        from ..samplers.langevin import _CAPTURE_LOCK  # the cyclic GC must not run inside a capture (see samplers.langevin)

        lf._graph_coords = coords
        lf.sampler._graph_coords = coords
        lf._graph_fifo_rows = 0
        with _CAPTURE_LOCK:
            gc_was_on = gc.isenabled()
            try:
                gc.disable()
                with torch.cuda.graph(graph, stream=self._side_stream(dev), capture_error_mode="thread_local"):
                    loss, neg = lf(static_x, generator=self.generator)
                    loss.backward(unit)
                    self.optimizer.step()
                    coords.tensor[1:2].add_(advance)
            finally:
                lf._graph_coords = None
                lf.sampler._graph_coords = None
                if gc_was_on:
                    gc.enable()
        advance.fill_(coords.taken)
        return {"key": key, "graph": graph, "x": static_x, "loss": loss, "neg": neg, "coords": coords, "advance": advance,
                "kernel_steps": coords.taken, "torch_steps": None, "expected": None, "pinned": None}

    def _sync_coords(self, g: dict) -> Tuple[torch.Generator, int]:
        """Make the device coordinates those of the generator (host bookkeeping only; a write happens when the generator
        is not where the previous replay left it: the first replay, or the user re-seeded)."""
        gen = _rng._resolve(self.generator, g["x"].device)
        seed, offset = _rng.kernel_seed(gen.initial_seed()), _rng._get_offset(gen)
        if g["expected"] != (seed, offset):
            host = torch.tensor([_rng.DeviceCoords.as_i64(seed), offset // 4], dtype=torch.int64).pin_memory()
            g["coords"].tensor.copy_(host, non_blocking=True)
            g["pinned"] = host  # alive until the copy has certainly run (replaced at the next re-seed only)
        return gen, offset

    def __call__(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        self.calls += 1
        if not self.enabled:
            return self._eager(x)
        if self.calls <= self.eager_steps:
            return self._warmup(x)
        if not (x.is_cuda and x.dtype == torch.float32):
            raise ValueError("GraphedTrainingStep: the batch must be a CUDA float32 tensor")
        key = self._key(x)
        g = self._g
        if g is None or g["key"] != key:
            if getattr(self, "_stale_graph", False):
                raise RuntimeError(
                    "GraphedTrainingStep: an autograd graph through the model's parameters that was built on another stream is still alive "
                    "(torch warned: \"AccumulateGrad node's stream does not match\") -- a tensor kept from an eager call of the model with "
                    "gradients enabled, typically a diagnostic.  Capturing the step now would abort the process: evaluate diagnostics under "
                    "torch.no_grad() (or drop their tensors), then build a new GraphedTrainingStep.")
            reason = self._static_refusal() or self._route_refusal(x)
            if reason:
                raise ValueError(f"GraphedTrainingStep: {reason}")
            if g is not None:
                self.recaptures += 1
            g = self._g = self._capture(x, key)
        if g["x"].data_ptr() != x.data_ptr():
            g["x"].copy_(x)
        gen, offset = self._sync_coords(g)
        g["graph"].replay()
        self.replays += 1
        # the generator: torch's own replay bookkeeping moved it past the step's torch-side draws; the kernels' steps follow
        after_torch = _rng._get_offset(gen)
        if g["torch_steps"] is None:
            # first replay: now the torch-side share of a step is known -- from here on the graph moves the coordinates past it too
            g["torch_steps"] = (after_torch - offset) // 4
            if g["torch_steps"]:
                g["coords"].tensor[1:2].add_(g["torch_steps"])
                g["advance"].fill_(g["kernel_steps"] + g["torch_steps"])
        _rng._set_offset(gen, after_torch + 4 * g["kernel_steps"])
        g["expected"] = (_rng.kernel_seed(gen.initial_seed()), after_torch + 4 * g["kernel_steps"])
        self.loss_fn._graph_replayed()
        return g["loss"].detach().clone(), g["neg"]
