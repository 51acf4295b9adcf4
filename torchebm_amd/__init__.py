"""torchebm_amd: MI355X-native Langevin / HMC sampling behind torchebm's sampler API.

The package mirrors the import layout of the reference for the pieces on the hot path:
``torchebm_amd.core``, ``.integrators``, ``.samplers``, ``.losses``, ``.utils``, ``.datasets`` (the two generators the path's callers use).  CUDA-device
fp32 sampling runs in hand-written gfx950 kernels reached through ``libebm_hip.so``
(``include/ebm_hip.h``); see DESIGN.md.
"""

__version__ = "0.1.0"

from . import core, datasets, integrators, losses, samplers, utils  # noqa: F401
from .core import (  # noqa: F401
    BaseModel,
    DoubleWellModel,
    GaussianMixtureModel,
    GaussianModel,
    HarmonicModel,
    MLPEnergy,
)
from .integrators import EulerMaruyamaIntegrator, LeapfrogIntegrator  # noqa: F401
from .losses import ContrastiveDivergence, EnergyMatchingContrastive  # noqa: F401
from .samplers import HamiltonianMonteCarlo, LangevinDynamics  # noqa: F401
