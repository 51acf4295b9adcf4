"""Euler-Maruyama integrator (reference: torchebm/integrators/euler_maruyama.py:11-65).

A one-stage tableau on top of ``BaseSDERungeKuttaIntegrator``; on a CUDA fp32 state its
``step`` is a single launch of ``ebm_langevin_step_f32`` (update + in-kernel Philox noise).
The backward (implicit) variant of the reference is outside the hot path.
"""

from __future__ import annotations

from ..core.integrator_base import BaseSDERungeKuttaIntegrator


class EulerMaruyamaIntegrator(BaseSDERungeKuttaIntegrator):
    r"""``x_{n+1} = x_n + f(x_n, t_n) h + \sqrt{2 D}\,\Delta W_n``; Euler's method when no
    diffusion is given."""

    @property
    def tableau_a(self):
        return ((),)

    @property
    def tableau_b(self):
        return (1.0,)

    @property
    def tableau_c(self):
        return (0.0,)


class HeunIntegrator(BaseSDERungeKuttaIntegrator):
    r"""Heun (improved Euler) predictor-corrector drift update with the same Euler-order noise term
    (reference: torchebm/integrators/heun.py; ``LangevinDynamics(integrator="heun")`` in the reference's
    tests/samplers/test_langevin_dynamics.py:259-268).  Two drift evaluations per step.  ``step()`` /
    ``integrate()`` here run the generic explicit-tableau path (eager torch ops); the sampler's whole-chain
    route for the analytic energies is fused (``ebm_langevin_heun_chain_f32``, samplers/langevin.py)."""

    @property
    def tableau_a(self):
        return ((), (1.0,))

    @property
    def tableau_b(self):
        return (0.5, 0.5)

    @property
    def tableau_c(self):
        return (0.0, 1.0)
