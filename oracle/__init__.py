"""CPU oracle for the Langevin / HMC hot path -- TEST INFRASTRUCTURE ONLY.

A restatement, in plain eager torch CPU ops, of the reference's algorithm for the path
(soran-ghaderi/torchebm @ 2026-08-21), each function citing the reference file:line it
follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; nothing under ``torchebm_amd/`` does.

Pinning: ``tests/golden/*.pt`` hold inputs, the exact noise tensors and the outputs of the
REAL reference (generated in the authoring container by ``tests/golden/make_golden.py``,
which imports /root/reference); ``tests/test_oracle_golden.py`` asserts this restatement
reproduces them bit for bit (``torch.equal``), so parity is pinned.
"""

from .energies import (  # noqa: F401
    DoubleWell,
    Gaussian,
    GaussianMixture,
    Harmonic,
)
from .langevin import em_step, heun_step, langevin_chain  # noqa: F401
from .hmc import hmc_chain, leapfrog  # noqa: F401
from .descent import descent_chain  # noqa: F401
from .philox import normal_field, philox4x32_10, raw_field, uniform_field  # noqa: F401
