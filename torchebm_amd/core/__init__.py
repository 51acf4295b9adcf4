"""Core building blocks (reference package: torchebm/core)."""

from .module import TorchEBMModule, warn_once
from .schedules import (
    BaseScheduler,
    ConstantScheduler,
    CosineScheduler,
    ExponentialDecayScheduler,
    LinearScheduler,
    MultiStepScheduler,
    Schedulable,
    TemperatureScheduler,
    WarmupScheduler,
)
from .energies import (
    AckleyModel,
    BaseModel,
    DoubleWellModel,
    FusedSpec,
    GaussianMixtureModel,
    GaussianModel,
    HarmonicModel,
    MLPEnergy,
    RastriginModel,
    RosenbrockModel,
    ring_mixture,
)
from .integrator_base import BaseIntegrator, BaseSDERungeKuttaIntegrator, BaseSymplecticIntegrator
from .sampler_base import BaseSampler
from .loss_base import BaseContrastiveDivergence, BaseLoss

__all__ = [
    "TorchEBMModule", "warn_once",
    "BaseScheduler", "ConstantScheduler", "ExponentialDecayScheduler", "LinearScheduler",
    "CosineScheduler", "MultiStepScheduler", "WarmupScheduler", "TemperatureScheduler", "Schedulable",
    "BaseModel", "DoubleWellModel", "GaussianModel", "HarmonicModel", "GaussianMixtureModel",
    "RosenbrockModel", "AckleyModel", "RastriginModel", "FusedSpec", "MLPEnergy", "ring_mixture",
    "BaseIntegrator", "BaseSDERungeKuttaIntegrator", "BaseSymplecticIntegrator",
    "BaseSampler", "BaseLoss", "BaseContrastiveDivergence",
]
