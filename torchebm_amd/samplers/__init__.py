"""Samplers (reference package: torchebm/samplers)."""

from .descent import GradientDescentSampler, NesterovSampler
from .hamiltonian import HamiltonianMonteCarlo
from .langevin import LangevinDynamics

__all__ = ["LangevinDynamics", "HamiltonianMonteCarlo", "GradientDescentSampler", "NesterovSampler"]
