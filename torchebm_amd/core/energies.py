"""Energy models: ``BaseModel`` and the analytic energies the HIP kernels fuse.

Host-side mirror of the reference's torchebm/core/base_model.py: ``forward`` gives the
energy per sample, the default ``gradient`` is autograd (:62-127) -- no analytic energy
in the reference overrides it, so the closed-form gradients live only inside the kernels
(``csrc/langevin.hip``, ``csrc/rows.hip``), written in autograd's operation order.

Each analytic model advertises itself to the samplers through ``fused_spec()``; a
subclass that overrides ``forward``/``gradient`` is no longer the same function and is
not fused (it takes the per-step kernel + ``gradient()`` route instead).

``GaussianMixtureModel`` does not exist in the reference; it is
defined here so BASELINE config 3 has an energy, and its oracle is the reference's HMC
driving autograd on this ``forward``.
"""

from __future__ import annotations

import math
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Optional, Sequence, Union

import torch

from .. import _lib
from .module import TorchEBMModule


@dataclass
class FusedSpec:
    """What ``ebm_energy_t`` needs (include/ebm_hip.h), with tensors kept alive here."""

    kind: int
    scalars: Sequence[float] = (0.0, 0.0, 0.0, 0.0)
    n_comp: int = 0
    dev0: Optional[torch.Tensor] = None
    dev1: Optional[torch.Tensor] = None
    aux: Optional[torch.Tensor] = None  # device int32 hints (mixture: active-column mask), see include/ebm_hip.h
    elementwise: bool = False  # gradient of coordinate j depends on x_j only
    dim: Optional[int] = None  # the model's own state width (None: any width, e.g. element-wise energies)
    langevin_only: bool = False  # fused for Euler-Maruyama Langevin chains, HMC and energy/gradient evaluation; no Heun / descent kernel
    hmc: bool = True  # ebm_hmc_chain_f32 takes this energy at this shape

    def to_c(self) -> "_lib.EnergyDesc":
        d = _lib.EnergyDesc()
        d.kind = self.kind
        d.n_comp = self.n_comp
        for i, v in enumerate(self.scalars):
            d.s[i] = float(v)
        d.dev0 = _lib.ptr(self.dev0)
        d.dev1 = _lib.ptr(self.dev1)
        d.aux = _lib.ptr(self.aux)
        # the descriptor holds raw device pointers: it keeps their tensors alive itself (``aux`` of a mixture is a fresh
        # temporary of every fused_spec() call -- with ``model.fused_spec().to_c()`` the spec is gone before the launch, the
        # allocator hands the block to the next allocation and the kernel reads whatever was written there)
        d._keepalive = (self.dev0, self.dev1, self.aux)
        return d


class BaseModel(TorchEBMModule, ABC):
    """Unnormalised negative log-density ``E(x)``; ``gradient`` defaults to autograd."""

    #: compute the autograd gradient in fp32 whatever the input dtype (base_model.py:22-27)
    force_fp32_gradient: bool = False

    def __init__(self, dtype: torch.dtype = torch.float32, *args, **kwargs):
        super().__init__(dtype=dtype, *args, **kwargs)

    @abstractmethod
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Energy per sample, shape ``(batch,)``."""

    def gradient(self, x: torch.Tensor, model_kwargs: Optional[dict] = None) -> torch.Tensor:
        r"""``\nabla_x E(x)`` by autograd, detached, in ``x``'s dtype (base_model.py:62-127)."""
        in_dtype = x.dtype
        if self.device and x.device != self.device:
            x = x.to(self.device)
        fast = self._hip_gradient(x, model_kwargs)
        if fast is not None:
            return fast
        work_dtype = torch.float32 if self.force_fp32_gradient else in_dtype
        with torch.enable_grad():
            leaf = x.detach().to(dtype=work_dtype).requires_grad_(True)
            with self.autocast_context():
                energy = self.forward(leaf, **(model_kwargs or {}))
            if energy.shape != (leaf.shape[0],):
                raise ValueError(
                    f"BaseModel forward() output expected shape ({leaf.shape[0]},), but got {energy.shape}."
                )
            if energy.grad_fn is None:
                raise RuntimeError(
                    "Cannot compute gradient: `forward` method did not use the input `x` in a differentiable way."
                )
            (grad,) = torch.autograd.grad(energy, leaf, grad_outputs=torch.ones_like(energy))
        if grad is None:
            raise RuntimeError("Gradient computation failed unexpectedly. Check the forward pass implementation.")
        return grad.to(in_dtype).detach()

    #: ``True`` on energies whose input gradient has a one-launch HIP evaluation that beats autograd (``MLPEnergy``:
    #: forward + backward through both hidden layers on the matrix cores, ``ebm_energy_grad_f32``)
    HIP_GRADIENT = False

    def _hip_gradient(self, x: torch.Tensor, model_kwargs: Optional[dict]) -> Optional[torch.Tensor]:
        """``gradient()`` in one ``ebm_energy_grad_f32`` launch when the energy opts in and the call is the plain
        case (fp32 ``[n, dim]`` CUDA state, no conditioning, no autocast); ``None`` sends the caller to autograd.
        Every per-step route -- HMC on the wide MLP, the integrators' drift, ``Integrator.step`` -- goes through
        ``gradient()``, so they all take it; ``forward()`` (and with it every parameter gradient) stays autograd."""
        if not self.HIP_GRADIENT or model_kwargs or not x.is_cuda or x.dtype != torch.float32 or x.ndim != 2:
            return None
        if torch.is_autocast_enabled() or getattr(self, "use_mixed_precision", False):
            return None
        spec = fused_spec_for(self, x, None)
        if spec is None:
            return None
        state = x.detach().contiguous()
        if state.data_ptr() % 16 != 0 or state.shape[0] == 0:
            return None
        grad = torch.empty_like(state)
        _lib.call("ebm_energy_grad_f32", spec.to_c(), state.data_ptr(), state.shape[0], state.shape[1], None, grad.data_ptr(),
                  _lib.stream_handle(state.device))
        return grad

    # ---- hook for the fused kernels ------------------------------------------------
    def fused_spec(self) -> Optional[FusedSpec]:
        """Descriptor for the fused HIP kernels, or ``None`` (the default: not fusable)."""
        return None

    def _is_exactly(self, cls: type) -> bool:
        mine = type(self)
        return mine.forward is cls.forward and mine.gradient is BaseModel.gradient


class DoubleWellModel(BaseModel):
    r"""``E(x) = h \sum_j (x_j^2 - b^2)^2`` (base_model.py:130-148)."""

    def __init__(self, barrier_height: float = 2.0, b: float = 1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.barrier_height = barrier_height
        self.b = b

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 1:
            x = x.unsqueeze(0)
        wells = (x.pow(2) - self.b**2).pow(2)
        return self.barrier_height * wells.sum(dim=-1)

    def fused_spec(self) -> Optional[FusedSpec]:
        if not self._is_exactly(DoubleWellModel):
            return None
        # b**2 is formed in double and enters the fp32 tensor op as a cast scalar
        return FusedSpec(_lib.ENERGY_DOUBLE_WELL, (self.barrier_height, self.b**2, 0.0, 0.0), elementwise=True)


class HarmonicModel(BaseModel):
    r"""``E(x) = \tfrac12 k \sum_j x_j^2`` (base_model.py:213-229)."""

    def __init__(self, k: float = 1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.k = k

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 1:
            x = x.unsqueeze(0)
        return 0.5 * self.k * x.pow(2).sum(dim=-1)

    def fused_spec(self) -> Optional[FusedSpec]:
        if not self._is_exactly(HarmonicModel):
            return None
        return FusedSpec(_lib.ENERGY_HARMONIC, (0.5 * self.k, 0.0, 0.0, 0.0), elementwise=True)


class RosenbrockModel(BaseModel):
    r"""``E(x) = \sum_{i<n} b (x_{i+1} - x_i^2)^2 + (a - x_i)^2`` (base_model.py:232-264).  A test landscape of the
    reference's ``core``; no fused kernel -- the samplers drive it through the autograd step route."""

    def __init__(self, a: float = 1.0, b: float = 100.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.a, self.b = a, b

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 1:
            x = x.unsqueeze(0)
        if x.shape[-1] < 2:
            raise ValueError(f"Rosenbrock energy function requires at least 2 dimensions, got {x.shape[-1]}")
        head, tail = x[:, :-1], x[:, 1:]
        return ((self.a - head).pow(2) + self.b * (tail - head.pow(2)).pow(2)).sum(dim=-1)


class AckleyModel(BaseModel):
    r"""``E(x) = -a e^{-b \sqrt{\overline{x^2}}} - e^{\overline{\cos(c x)}} + a + e`` (base_model.py:267-294); autograd
    step route."""

    def __init__(self, a: float = 20.0, b: float = 0.2, c: float = 2 * math.pi, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.a, self.b, self.c = a, b, c

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 1:
            x = x.unsqueeze(0)
        n = x.shape[-1]
        radial = -self.a * torch.exp(-self.b * torch.sqrt(torch.sum(x**2, dim=-1) / n))
        ripple = -torch.exp(torch.sum(torch.cos(self.c * x), dim=-1) / n)
        return radial + ripple + self.a + math.e


class RastriginModel(BaseModel):
    r"""``E(x) = a n + \sum_j x_j^2 - a \cos(2 \pi x_j)`` (base_model.py:297-316); autograd step route."""

    def __init__(self, a: float = 10.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.a = a

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 1:
            x = x.unsqueeze(0)
        return self.a * x.shape[-1] + torch.sum(x**2 - self.a * torch.cos(2 * math.pi * x), dim=-1)


class GaussianModel(BaseModel):
    r"""``E(x) = \tfrac12 (x-\mu)^\top \Sigma^{-1} (x-\mu)`` (base_model.py:151-210)."""

    def __init__(self, mean: torch.Tensor, cov: torch.Tensor, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if mean.ndim != 1:
            raise ValueError("Mean must be a 1D tensor.")
        if cov.ndim != 2 or cov.shape[0] != cov.shape[1]:
            raise ValueError("Covariance must be a 2D square matrix.")
        if mean.shape[0] != cov.shape[0]:
            raise ValueError("Mean vector dimension must match covariance matrix dimension.")
        self.register_buffer("mean", mean.to(dtype=self.dtype, device=self.device))
        try:
            precision = torch.inverse(cov)
        except RuntimeError as exc:
            raise ValueError(f"Failed to invert covariance matrix: {exc}. Ensure it is invertible.") from exc
        self.register_buffer("cov_inv", precision.to(dtype=self.dtype, device=self.device))
        self._sym_cache: Optional[tuple] = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 1:
            x = x.unsqueeze(0)
        d = self.mean.shape[0]
        if x.ndim != 2 or x.shape[1] != d:
            raise ValueError(f"Input x expected batch_shape (batch_size, {d}), but got {x.shape}")
        x = x.to(dtype=self.dtype, device=self.device)
        prec = self.cov_inv.to(dtype=self.dtype, device=x.device)  # read LIVE: forward() never goes through the spec cache
        delta = x - self.mean
        if delta.shape[0] > 1 and not (delta.is_cuda and d > self.CLOSED_FORM_GRADIENT_ABOVE):
            # batched form: P (d x d, broadcast) @ delta (d x 1), then delta^T @ that
            p_delta = torch.bmm(prec.unsqueeze(0).expand(delta.shape[0], -1, -1), delta.unsqueeze(-1))
            return 0.5 * torch.bmm(delta.unsqueeze(1), p_delta).squeeze(-1).squeeze(-1)
        # one row, or a wide Gaussian on the GPU: the reference's single-row form (base_model.py:208) -- ONE GEMM instead of n
        # mat-vecs against an expanded P (at dim 512 and 2^15 rows the batched form reads 34 GB); same value up to fp32 rounding
        return 0.5 * torch.sum(delta * torch.matmul(delta, prec), dim=-1)

    def _sampler_energy(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        """The energy for a SAMPLER's own use (the accept step of the per-transition HMC route, diagnostics): one contraction pass
        of the tiled kernel at the widths it covers, ``None`` elsewhere (the caller then calls ``forward``).  Not reachable
        through ``model(x)``: the public forward reads ``cov_inv`` live and rounds the same way with and without autograd --
        this path reads sym(P) from the spec cache (``fused_spec``: keyed on the matrix's storage and version; see
        ``invalidate_cache`` for the one kind of write that key cannot see).  Same guards as ``_hip_gradient`` /
        ``fused_spec_for``: parameters on the state's device, no mixed precision -- and no forward hooks registered, since this
        path does not go through ``Module.__call__``."""
        d = self.mean.shape[0]
        if not (x.is_cuda and x.dtype == torch.float32 and x.ndim == 2 and x.shape[0] > 1 and x.shape[1] == d
                and not torch.is_autocast_enabled() and self._on_matrix_cores(x) and self._is_exactly(GaussianModel)
                and self.cov_inv.dtype == torch.float32 and self.cov_inv.device == x.device and self.mean.device == x.device
                and not getattr(self, "use_mixed_precision", False) and not self._forward_hooks and not self._forward_pre_hooks):
            return None
        spec = self.fused_spec()
        if spec is None:
            return None
        state = x.contiguous()
        energy = torch.empty(state.shape[0], dtype=torch.float32, device=state.device)
        _lib.call("ebm_energy_grad_f32", spec.to_c(), state.data_ptr(), state.shape[0], d, energy.data_ptr(), None,
                  _lib.stream_handle(state.device))
        return energy

    #: widths above which ``gradient()`` is the closed form ``(x - mu) @ sym(P)`` -- ONE library GEMM -- instead of autograd
    #: through ``forward``'s batched form (base_model.py:199-206 expands P to n matrices: at dim 1024 that is n mat-vecs of
    #: 4 MB each).  Same value up to fp32 rounding of the two contraction orders; below, autograd as in the reference.
    CLOSED_FORM_GRADIENT_ABOVE = 128

    def _hip_gradient(self, x: torch.Tensor, model_kwargs: Optional[dict]) -> Optional[torch.Tensor]:
        spec = None
        if (not model_kwargs and x.is_cuda and x.dtype == torch.float32 and x.ndim == 2 and x.shape[0] > 1
                and x.shape[1] > self.CLOSED_FORM_GRADIENT_ABOVE and not torch.is_autocast_enabled()
                and not getattr(self, "use_mixed_precision", False)):
            spec = self.fused_spec()
        if spec is None:
            return None
        if self._on_matrix_cores(x):  # one contraction pass of the tiled kernel: no (x - mu) temporary, a third of the GEMM route's time
            state = x.detach().contiguous()
            grad = torch.empty_like(state)
            _lib.call("ebm_energy_grad_f32", spec.to_c(), state.data_ptr(), state.shape[0], state.shape[1], None, grad.data_ptr(),
                      _lib.stream_handle(state.device))
            return grad
        return torch.mm(x.detach() - spec.dev0, spec.dev1)  # dev1 = sym(P): symmetric, no transpose needed

    @staticmethod
    def _on_matrix_cores(x: torch.Tensor) -> bool:
        """Widths at which ``ebm_energy_grad_f32`` (the streamed-Ps kernel, csrc/gauss_big.hip: multiples of 4 up to 512) beats the
        library GEMM form: up to 256 (HMC route, 2^16 chains: dims 160 / 256 12.0 / 20.0 -> 9.5 / 18.4 ms per 5 transitions; at 512
        the two-slice kernel loses to hipBLAS: 26.6 -> 30.3)."""
        d = x.shape[1]
        return 128 < d <= 256 and d % 4 == 0 and x.data_ptr() % 16 == 0 and x.shape[0] > 0

    def fused_spec(self) -> Optional[FusedSpec]:
        if not self._is_exactly(GaussianModel) or self.cov_inv.dtype != torch.float32:
            return None
        # autograd of 0.5 d^T P d is 0.5 (P + P^T) d: hand the kernel the symmetrised matrix
        key = (self.cov_inv.data_ptr(), self.cov_inv._version, self.cov_inv.device)
        if self._sym_cache is None or self._sym_cache[0] != key:
            sym = (0.5 * (self.cov_inv + self.cov_inv.t())).contiguous()
            # dims 132 .. 512: the matrix pre-split into bf16 triples in the order the tiled kernel's stages consume it
            # (include/ebm_hip.h, ebm_gauss_prec_image_f32) -- built with the symmetrised matrix, i.e. when cov_inv changes
            image = None
            if sym.is_cuda and _lib.is_built():
                n_img = int(_lib.lib().ebm_gauss_prec_image_bytes(int(sym.shape[0])))
                if n_img:
                    image = torch.empty(n_img // 4, dtype=torch.int32, device=sym.device)
                    _lib.call("ebm_gauss_prec_image_f32", sym.data_ptr(), int(sym.shape[0]), image.data_ptr(), _lib.stream_handle(sym.device))
            # the stream the two were built on: a later call on ANOTHER stream must not read them before that work has run
            built = torch.cuda.Event() if sym.is_cuda else None
            if built is not None:
                built.record(torch.cuda.current_stream(sym.device))
            self._sym_cache = (key, sym, image, built, torch.cuda.current_stream(sym.device) if sym.is_cuda else None)
        elif self._sym_cache[3] is not None:
            here = torch.cuda.current_stream(self._sym_cache[1].device)
            if here != self._sym_cache[4]:
                here.wait_event(self._sym_cache[3])
        return FusedSpec(_lib.ENERGY_GAUSSIAN, dev0=self.mean.contiguous(), dev1=self._sym_cache[1], aux=self._sym_cache[2],
                         dim=int(self.mean.shape[0]))

    def invalidate_cache(self) -> None:
        """Drop the symmetrised precision matrix and its pre-split image.  They are rebuilt whenever ``cov_inv``'s storage or
        version counter changes -- every in-place tensor op, ``load_state_dict``, ``.to()`` -- but a write THROUGH ``.data``
        (``model.cov_inv.data.copy_(new)``) changes neither, exactly as it hides from autograd: call this after such a write,
        or the fused kernels keep sampling from the old matrix while ``forward()`` (which reads ``cov_inv`` live) already
        evaluates the new one."""
        self._sym_cache = None


class GaussianMixtureModel(BaseModel):
    r"""Isotropic Gaussian mixture energy

    .. math:: E(x) = -\log \sum_k w_k \exp\!\big(-\lVert x-\mu_k\rVert^2 / (2\sigma^2)\big)

    (up to the additive constant of the normaliser).  ``mean`` exposes the mixture mean so
    that ``HamiltonianMonteCarlo`` can infer ``dim`` the way it does for ``GaussianModel``.

    Args:
        means: ``[K, dim]`` component centres.
        sigma: shared component standard deviation.
        weights: ``[K]`` mixture weights (normalised internally); uniform when ``None``.
    """

    def __init__(
        self,
        means: torch.Tensor,
        sigma: float = 1.0,
        weights: Optional[torch.Tensor] = None,
        *args,
        **kwargs,
    ):
        super().__init__(*args, **kwargs)
        if means.ndim != 2:
            raise ValueError("means must be a [K, dim] tensor.")
        if sigma <= 0:
            raise ValueError("sigma must be positive")
        k = means.shape[0]
        if weights is None:
            weights = torch.full((k,), 1.0 / k)
        if weights.ndim != 1 or weights.shape[0] != k or bool((weights <= 0).any()):
            raise ValueError("weights must be a positive [K] tensor.")
        weights = weights.to(torch.float64)
        log_w = torch.log(weights / weights.sum())
        self.sigma = float(sigma)
        self.register_buffer("means", means.to(dtype=self.dtype, device=self.device).contiguous())
        self.register_buffer("log_weights", log_w.to(dtype=self.dtype, device=self.device))
        mix_mean = (weights[:, None] / weights.sum() * means.to(torch.float64)).sum(dim=0)
        self.register_buffer("mean", mix_mean.to(dtype=self.dtype, device=self.device))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 1:
            x = x.unsqueeze(0)
        if x.ndim != 2 or x.shape[1] != self.means.shape[1]:
            raise ValueError(f"Input x expected shape (batch_size, {self.means.shape[1]}), but got {x.shape}")
        sq_dist = (x.unsqueeze(1) - self.means.unsqueeze(0)).pow(2).sum(dim=-1)  # [B, K]
        logits = self.log_weights - sq_dist / (2.0 * self.sigma**2)
        return -torch.logsumexp(logits, dim=1)

    def _active_column_mask(self) -> Optional[torch.Tensor]:
        """Device int32[1]: bit v set when the component means differ somewhere in columns 4v..4v+3 (the `aux` hint
        of EBM_ENERGY_GMM).  Computed on the device -- no host read -- at EVERY call, by one small launch
        (``ebm_gmm_active_columns_i32``; it was seven tensor ops = 45 us in front of every kernel): a cache keyed on
        the tensor's storage and version would survive a write through ``.data`` (``means.data.copy_(new)`` keeps both),
        and a stale hint sends a mixture whose components now differ elsewhere to the active-column body."""
        m = self.means
        k, d = m.shape
        if not m.is_cuda or d % 4 != 0 or d // 4 > 8 or m.dtype != torch.float32:
            return None
        m = m.detach().contiguous()
        out = torch.empty(1, dtype=torch.int32, device=m.device)
        _lib.call("ebm_gmm_active_columns_i32", m.data_ptr(), int(k), int(d), out.data_ptr(), _lib.stream_handle(m.device))
        return out

    def fused_spec(self) -> Optional[FusedSpec]:
        if not self._is_exactly(GaussianMixtureModel) or self.means.dtype != torch.float32:
            return None
        if self.means.shape[0] > 64:
            return None
        s2 = self.sigma**2
        return FusedSpec(
            _lib.ENERGY_GMM,
            (1.0 / (2.0 * s2), 1.0 / s2, 0.0, 0.0),
            aux=self._active_column_mask(),
            n_comp=int(self.means.shape[0]),
            dev0=self.means,
            dev1=self.log_weights,
            dim=int(self.means.shape[1]),
        )


def _tall_gram(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``a^T b`` for two tall matrices ``[n, p]``, ``[n, q]``: a small ``p x q`` output over K = n.  The library's GEMM for it is
    one 32 x 32 macro tile per workgroup and no split over K -- 16 workgroups of a 256-CU chip, 216 us at n = 65 536, p = q = 128;
    as a batched product over 32 row blocks + a sum it is 28 us (and sums pairwise: ten times closer to the fp64 value), and the
    thin cases (p = 1: 49 -> 26 us; q = 3: 96 + 26 -> 47 us at n = 131 072) beat broadcast product + column reduction too
    (scripts/probes/thin_linear_ops.py)."""
    n = a.shape[0]
    blocks = 32
    if a.is_cuda and n >= 8192 and n % blocks == 0:
        return torch.bmm(a.view(blocks, n // blocks, -1).transpose(1, 2), b.view(blocks, n // blocks, -1)).sum(0)
    return a.t() @ b


class _ThinMLPEnergy(torch.autograd.Function):
    """``E = w3 . silu(W2 silu(W1 x + b1) + b2) + b3`` for inputs that need no gradient: the forward of
    ``Linear - SiLU - Linear - SiLU - Linear`` op for op, the parameter gradients with the thin layers' weight gradients as
    broadcast product + column reduction instead of K = batch GEMMs (``MLPEnergy.forward``)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3):
        import torch.nn.functional as F

        a1 = F.linear(x, w1, b1)
        h1 = F.silu(a1)
        a2 = F.linear(h1, w2, b2)
        h2 = F.silu(a2)
        e = F.linear(h2, w3, b3).squeeze(-1)
        ctx.save_for_backward(x, a1, h1, a2, h2, w2, w3)
        return e

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ge):
        x, a1, h1, a2, h2, w2, w3 = ctx.saved_tensors
        col = ge.unsqueeze(1)
        d_b3 = ge.sum().reshape(1)
        d_w3 = _tall_gram(col, h2)                                   # [1, H]
        dz2 = torch.ops.aten.silu_backward(col * w3, a2)
        d_b2 = dz2.sum(0)
        d_w2 = _tall_gram(dz2, h1)
        dz1 = torch.ops.aten.silu_backward(dz2 @ w2, a1)
        # first layer: weight and bias gradient in one product against [x 1] (a [H, in + 1] output; the bias column rides along)
        g1 = _tall_gram(dz1, torch.cat((x, x.new_ones(x.shape[0], 1)), dim=1))
        d_w1, d_b1 = g1[:, :-1], g1[:, -1]
        return None, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3


def _gram_rows(a_t: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``a_t @ b`` for ``a_t`` given HIDDEN-major ``[p, n]`` (rows contiguous, row stride >= n) and ``b`` ``[n, q]``: the same
    row-block product as ``_tall_gram`` for the layout ``ebm_mlp_backward_acts_f32`` stores its activations in."""
    p, n = a_t.shape
    blocks = 32
    if n >= 8192 and n % blocks == 0:
        nb = n // blocks
        return torch.bmm(a_t.unflatten(1, (blocks, nb)).permute(1, 0, 2), b.view(blocks, nb, -1)).sum(0)
    return a_t @ b


class _FusedMLPTraining(torch.autograd.Function):
    """The training forward / backward of ``MLPEnergy`` through the HIP library (round 5).

    Forward, when a parameter gradient will be asked for: ONE launch (``ebm_mlp_backward_acts_f32`` with a unit seed) evaluates the
    network, writes the energies, runs the backward through the network on the matrix cores and stores the three activation planes the
    parameter gradients are made of (h1, the pre-activation a2, d1) -- where autograd's graph of the same step writes and re-reads some forty ``[n, H]`` arrays.
    Backward: ONE pass over those planes (``ebm_mlp_param_grads_f32``: fp32 MFMA products over K = n, the per-row seed ``dL/dE``
    applied on load, h2 and d2 recomputed from a2, partial records added in a fixed order) yields every parameter gradient.
    Without a gradient to prepare (``no_grad``, frozen parameters) the forward is the energy-only evaluation
    (``ebm_energy_grad_f32``).  Energies and gradients are those of the kernels: fp32-accurate (split-bf16 contractions with fp32
    accumulation in the network, exact fp32 products in the gradient pass), not bit-identical to ``self.net`` -- the tolerance tier
    of every other use of this energy's kernels."""

    @staticmethod
    def forward(ctx, x, packed, hidden, *params):
        n, dim = x.shape
        spec = FusedSpec(_lib.ENERGY_MLP, n_comp=hidden, dev0=packed, langevin_only=True, dim=dim)
        energy = torch.empty(n, dtype=torch.float32, device=x.device)
        ctx.hidden = hidden
        ctx.with_planes = bool(n) and any(ctx.needs_input_grad[3:])
        if ctx.with_planes:
            n_pad = (n + 127) // 128 * 128
            acts = torch.empty(n_pad // 32, 3, hidden, 32, dtype=torch.float32, device=x.device)  # tiles of 32 rows: h1 | a2 | d1
            _lib.call("ebm_mlp_backward_acts_f32", spec.to_c(), x.data_ptr(), n, dim, None, energy.data_ptr(), None, acts.data_ptr(),
                      _lib.stream_handle(x.device))
            ctx.save_for_backward(x, acts, packed)
        else:
            if n:
                _lib.call("ebm_energy_grad_f32", spec.to_c(), x.data_ptr(), n, dim, energy.data_ptr(), None, _lib.stream_handle(x.device))
            ctx.save_for_backward(x)
        return energy

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ge):
        x = ctx.saved_tensors[0]
        n, dim = x.shape
        hidden = ctx.hidden
        sizes = (hidden * dim, hidden, hidden * hidden, hidden, hidden, 1)
        if not ctx.with_planes:  # (an empty batch: every gradient is zero)
            flat = x.new_zeros(sum(sizes))
        else:
            acts, packed = ctx.saved_tensors[1], ctx.saved_tensors[2]
            w3_ptr = packed.data_ptr() + 4 * (hidden * dim + hidden + hidden * hidden + hidden)  # W1 | b1 | W2 | b2 | w3 | b3
            work_floats = int(_lib.lib().ebm_mlp_param_grads_work_f32(hidden, dim, n))
            work = torch.empty(work_floats, dtype=torch.float32, device=x.device)
            flat = torch.empty(sum(sizes), dtype=torch.float32, device=x.device)
            seed = ge.contiguous()
            _lib.call("ebm_mlp_param_grads_f32", acts.data_ptr(), n, hidden, x.data_ptr(), dim, seed.data_ptr(), w3_ptr, work.data_ptr(), work_floats,
                      flat.data_ptr(), _lib.stream_handle(x.device))
        d_w1, d_b1, d_w2, d_b2, d_w3, d_b3 = flat.split(sizes)
        return None, None, None, d_w1.view(hidden, dim), d_b1, d_w2.view(hidden, hidden), d_b2, d_w3.view(1, hidden), d_b3


def _functorch_active() -> bool:
    """A ``torch.func`` transform (grad / vmap / jvp ...) is being traced: custom once-differentiable functions have no rules for it."""
    try:
        return torch._C._functorch.peek_interpreter_stack() is not None
    except Exception:  # noqa: BLE001 -- an internal API: absent means no transform machinery either
        return False


class MLPEnergy(BaseModel):
    r"""Two-hidden-layer SiLU MLP energy ``E(x) = w_3^\top \mathrm{silu}(W_2\,\mathrm{silu}(W_1 x + b_1) + b_2) + b_3``
    -- the trainable energy of the reference's PCD example
    (examples/20-training/01-mcmc-losses/02-persistent-cd/main.py:21-31).

    With ``hidden`` 64 or 128 and ``in_dim <= 128`` on a CUDA device, ``LangevinDynamics`` runs all k steps --
    forward, input-gradient on the matrix cores, update, noise -- in one ``ebm_langevin_chain_f32``
    launch instead of one autograd round trip per step: the reference's benchmark network
    ``Linear(dim, 128) - SiLU - Linear(128, 128) - SiLU - Linear(128, 1)`` at dim 8 / 32 / 128
    (benchmarks/registry.py:372-387) as well as the 2-D two-moons energy of its PCD example.  ``HamiltonianMonteCarlo`` is fused -- all transitions of a call in
    one ``ebm_hmc_chain_f32`` launch -- for the same shapes; configurations the kernels do not take (a non-default
    integrator, conditioning) run the per-transition route with ``gradient()`` as one HIP launch.
    Training is unaffected: the parameters are ordinary ``nn.Linear`` weights and are re-read at every
    ``sample()`` call.
    """

    #: hidden widths with fused kernels in the shipped library (256 -- the streamed-weight family, csrc/mlp_stream*.hip -- only in a
    #: build made with ``make H256=1``; set this to (64, 128, 256) then: the reference's own network is 128 wide, benchmarks/registry.py:370)
    FUSED_HIDDEN = (64, 128)
    HIP_GRADIENT = True
    FUSED_MAX_DIM = 128
    #: HamiltonianMonteCarlo's transition kernels: hidden width -> widest input (csrc/mlp_wide_hmc.hip: state, momentum
    #: and force ride in registers next to the evaluation's own)
    HMC_MAX_DIM = {64: 128, 128: 128, 256: 128}  # (256: see FUSED_HIDDEN)

    def __init__(self, in_dim: int = 2, hidden: int = 128, *args, **kwargs):
        super().__init__(*args, **kwargs)
        from torch import nn

        self.in_dim, self.hidden = in_dim, hidden
        self.net = nn.Sequential(
            nn.Linear(in_dim, hidden), nn.SiLU(), nn.Linear(hidden, hidden), nn.SiLU(), nn.Linear(hidden, 1)
        ).to(device=self._torchebm_probe.device, dtype=self._torchebm_probe.dtype)

    @classmethod
    def from_sequential(cls, net: "torch.nn.Sequential") -> "MLPEnergy":
        """Adopt an existing ``Linear(d, H) - SiLU - Linear(H, H) - SiLU - Linear(H, 1)`` stack (the network
        the reference's example writes by hand) WITHOUT copying it: the returned energy shares ``net``'s
        parameters, so an optimiser built on either sees the same tensors, and sampling from it takes the
        fused route when the shape qualifies (H = 64, 128 or 256, d <= 128, CUDA fp32)."""
        from torch import nn

        layers = list(net)
        ok = (
            len(layers) == 5
            and all(isinstance(layers[i], nn.Linear) for i in (0, 2, 4))
            and all(isinstance(layers[i], nn.SiLU) for i in (1, 3))
            and layers[0].out_features == layers[2].in_features == layers[2].out_features == layers[4].in_features
            and layers[4].out_features == 1
            and all(l.bias is not None for l in (layers[0], layers[2], layers[4]))
        )
        if not ok:
            raise ValueError("expected nn.Sequential(Linear(d, H), SiLU(), Linear(H, H), SiLU(), Linear(H, 1)) with biases")
        w = layers[0].weight
        self = cls(in_dim=layers[0].in_features, hidden=layers[0].out_features, device=w.device, dtype=w.dtype)
        self.net = net
        return self

    #: widest input for which ``forward`` takes the hand-written parameter-gradient backward (every weight gradient is a
    #: tall-K product whatever the input width: ``_tall_gram``)
    THIN_GRAD_MAX_IN = 128
    #: the training forward / backward (inputs that need no gradient) through the HIP library where the shape has the kernels
    #: (hidden 64 / 128, in_dim <= 64): one launch each way instead of autograd's forty passes over ``[n, H]`` arrays; energies and
    #: parameter gradients are then the KERNEL's (fp32-accurate, not bit-identical to ``self.net``).  False: torch ops forward
    #: (bit-identical to ``self.net``) with the hand-written torch backward (``_ThinMLPEnergy``).
    fused_training = True
    #: True: ``forward`` always runs ``self.net`` under autograd -- for anything that differentiates the ENERGY TWICE through the
    #: parameters (gradient penalties with ``create_graph=True``) or applies ``torch.func`` transforms: the hand-written training
    #: paths above are once-differentiable custom functions (a second differentiation raises; it is never silently wrong) and
    #: their energies under grad are the kernel's (fp32-accurate) while ``no_grad`` calls return ``self.net``'s bit for bit.
    #: ``forward`` also switches itself to ``self.net`` while a ``torch.func`` transform is active.
    higher_order = False
    #: a row's energy does not depend on the other rows of the batch (no batch statistics): a loss may evaluate data and negatives
    #: in one call (losses/cd.py)
    ROWS_INDEPENDENT = True

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # The TRAINING forward of config 5's caller (ContrastiveDivergence.compute_loss: energies of the data and of the detached
        # negatives, gradients with respect to the PARAMETERS only).  Same forward ops as ``self.net`` -- the energies are
        # bit-identical -- with a hand-written backward: autograd forms the weight gradients of the two THIN layers
        # (Linear(2, 128), Linear(128, 1)) as library GEMMs with K = batch, for which the library has no kernel worth the name
        # (65 536 rows: 190 + 100 us per call, 1.05 of the 2.4 ms of a whole training step: profiles/r05_c5_step_kernels.txt);
        # as a broadcast product + column reduction they take 68 + 34 us (scripts/probes/thin_linear_ops.py).  Only when the
        # input needs no gradient (the sampler's step route, score-based losses and anything that differentiates twice keep
        # autograd's own graph through ``self.net``).
        if (torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and not x.requires_grad
                and not self.higher_order and self.in_dim <= self.THIN_GRAD_MAX_IN and self._plain_net()
                and self._net_matches(x) and not torch.is_autocast_enabled() and not _functorch_active()):
            n = self.net
            params = (n[0].weight, n[0].bias, n[2].weight, n[2].bias, n[4].weight, n[4].bias)
            if (self.fused_training and self.hidden in (64, 128) and self.in_dim <= 64 and self.hidden in self.FUSED_HIDDEN
                    and x.is_contiguous() and x.data_ptr() % 16 == 0 and _lib.is_built() and all(p.is_cuda for p in params)):
                return _FusedMLPTraining.apply(x, self._packed_parameters(), int(self.hidden), *params)
            return _ThinMLPEnergy.apply(x, *params)
        return self.net(x).squeeze(-1)

    def _net_matches(self, x: torch.Tensor) -> bool:
        """The widths the kernels index the packed parameters by are the widths of ``self.net`` and of ``x`` (a wrong-width batch
        must reach ``self.net`` and raise torch's shape error, not index ``packed`` as W1[H, x.shape[1]]; a ``net`` whose layers
        were resized after construction is not the network ``hidden`` / ``in_dim`` describe)."""
        n = self.net
        return (x.shape[1] == self.in_dim == n[0].in_features and n[0].out_features == self.hidden == n[2].in_features
                and n[2].out_features == self.hidden == n[4].in_features)

    def _plain_net(self) -> bool:
        """``self.net`` is the stack ``fused_spec`` packs, with nothing hooked into it (hooks see module calls; the
        hand-written training forward makes none)."""
        from torch import nn

        n = self.net
        if type(self) is not MLPEnergy or not isinstance(n, nn.Sequential) or len(n) != 5:
            return False
        mods = (self, n, *n)
        if any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None) for m in mods):
            return False
        return (type(n[0]) is nn.Linear and type(n[2]) is nn.Linear and type(n[4]) is nn.Linear and type(n[1]) is nn.SiLU
                and type(n[3]) is nn.SiLU and n[4].out_features == 1 and all(l.bias is not None for l in (n[0], n[2], n[4])))

    #: set by a caller that guarantees the parameters do not change while it is set (``ContrastiveDivergence.forward``: the sampler
    #: call and the loss's model call of ONE step, nothing in between touches the weights): a dict in which the packed copy is kept
    _pack_scope: Optional[dict] = None

    def _packed_parameters(self) -> torch.Tensor:
        """W1[H,in] b1[H] W2[H,H] b2[H] w3[H] b3[1] as one vector, the order include/ebm_hip.h documents -- packed on every call
        (the weights train), or once per ``_pack_scope``."""
        scope = self._pack_scope
        if scope is not None and "packed" in scope:
            return scope["packed"]
        n = self.net
        with torch.no_grad():
            packed = torch.cat([p.detach().reshape(-1) for p in (n[0].weight, n[0].bias, n[2].weight, n[2].bias, n[4].weight, n[4].bias)])
        if scope is not None:
            scope["packed"] = packed
        return packed

    def fused_spec(self) -> Optional[FusedSpec]:
        if not self._is_exactly(MLPEnergy) or self.hidden not in self.FUSED_HIDDEN or self.in_dim > self.FUSED_MAX_DIM:
            return None
        w = self.net[0].weight
        if not w.is_cuda or w.dtype != torch.float32:
            return None
        with torch.no_grad():
            packed = self._packed_parameters()
            # H = 128 beyond dim 64: the pre-split W1 image the chain kernel streams through LDS (include/ebm_hip.h,
            # ebm_mlp_w1_image_f32) -- rebuilt with the parameters, i.e. on every call, like `packed` itself
            image = None
            n_img = int(_lib.lib().ebm_mlp_w1_image_bytes(int(self.hidden), int(self.in_dim)))
            if n_img:
                image = torch.empty(n_img // 4, dtype=torch.int32, device=w.device)
                _lib.call("ebm_mlp_w1_image_f32", packed.data_ptr(), int(self.hidden), int(self.in_dim), image.data_ptr(),
                          _lib.stream_handle(w.device))
        return FusedSpec(_lib.ENERGY_MLP, n_comp=self.hidden, dev0=packed, aux=image, langevin_only=True, dim=int(self.in_dim),
                         hmc=self.in_dim <= self.HMC_MAX_DIM.get(self.hidden, 0))


#: widest chain row the lane-group kernels (csrc/rows.h: pick_geometry) take
FUSED_MAX_ROW = 1024


def fused_spec_for(model, x: torch.Tensor, model_kwargs: Optional[dict], *, cap_elementwise: bool = True) -> Optional[FusedSpec]:
    """The descriptor the fused kernels need for sampling ``model`` on the state ``x`` -- or ``None`` when
    the configuration must take the per-step route (``model.gradient`` + update kernel):

    * conditioning kwargs, a model that carries schedulers, parameters on another device;
    * a state whose trailing width differs from the model's own (``GaussianModel.mean``, the mixture's
      ``means``, the MLP's ``in_dim``): the kernels index the parameters with ``x.shape[1]``, so a mismatch
      would read them with the wrong stride or out of bounds -- on the step route the model's ``forward``
      raises the reference's ``ValueError`` instead (core/base_model.py:185-188);
    * rows wider than ``FUSED_MAX_ROW`` for the row-coupled kernels (``cap_elementwise=False`` lifts the cap
      for element-wise energies where the caller uses the flat kernel, which has no row limit);
    * a state that is not ``[n, dim]``: the analytic energies reduce over the last axis only, so for a
      ``[n, a, b]`` state the reference's ``BaseModel.gradient`` raises ``ValueError`` (energy shape
      ``(n, a)`` is not ``(n,)``, core/base_model.py:95-99) -- and so does the step route here.
    """
    from .schedules import Schedulable

    if model_kwargs or not hasattr(model, "fused_spec") or isinstance(model, Schedulable):
        return None
    spec = model.fused_spec()
    if spec is None:
        return None
    if x.ndim != 2:
        return None
    if any(t is not None and t.device != x.device for t in (spec.dev0, spec.dev1)):
        return None
    width = x.shape[1]
    if spec.dim is not None and width != spec.dim:
        return None
    if width > FUSED_MAX_ROW and (cap_elementwise or not spec.elementwise):
        return None
    return spec


def ring_mixture(
    n_components: int = 8,
    dim: int = 32,
    radius: float = 4.0,
    sigma: float = 1.0,
    device: Union[str, torch.device, None] = None,
) -> GaussianMixtureModel:
    """BASELINE config 3's energy: K equal-weight modes on a circle of ``radius`` in the
    first two coordinates (centres as in the reference's GaussianMixtureDataset,
    datasets/generators.py:179-184), zero elsewhere."""
    ang = torch.arange(n_components, dtype=torch.float64) * (2.0 * math.pi / n_components)
    means = torch.zeros(n_components, dim, dtype=torch.float64)
    means[:, 0] = radius * torch.cos(ang)
    if dim > 1:
        means[:, 1] = radius * torch.sin(ang)
    return GaussianMixtureModel(means.to(torch.float32), sigma=sigma, device=device)
