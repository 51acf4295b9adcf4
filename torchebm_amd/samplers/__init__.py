"""MCMC samplers (reference package: torchebm/samplers)."""

from .hamiltonian import HamiltonianMonteCarlo
from .langevin import LangevinDynamics

__all__ = ["LangevinDynamics", "HamiltonianMonteCarlo"]
