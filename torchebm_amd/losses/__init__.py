"""Losses that call the sampler hot path (reference package: torchebm/losses)."""

from .cd import ContrastiveDivergence
from .energy_matching import EnergyMatchingContrastive, trimmed_mean

__all__ = ["ContrastiveDivergence", "EnergyMatchingContrastive", "trimmed_mean"]
