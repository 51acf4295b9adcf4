"""Losses that call the sampler hot path (reference package: torchebm/losses)."""

from .cd import ContrastiveDivergence

__all__ = ["ContrastiveDivergence"]
