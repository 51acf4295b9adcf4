"""Integrators on the Langevin/HMC path (reference package: torchebm/integrators)."""

from .em import EulerMaruyamaIntegrator, HeunIntegrator
from .symplectic import LeapfrogIntegrator
from .registry import _integrate_time_grid, get_integrator, resolve_integrator

__all__ = ["EulerMaruyamaIntegrator", "HeunIntegrator", "LeapfrogIntegrator", "get_integrator", "resolve_integrator"]
