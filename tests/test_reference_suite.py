"""Drop-in check: the REFERENCE'S OWN test files for the Langevin / HMC path, run unmodified against torchebm_amd.

tests/ref_compat/ref_alias.py makes ``import torchebm...`` resolve to this package and turns every name the package
does not provide (components outside SURVEY.md §8's hot path) into a stand-in that skips the test touching it.  The
test files are read in place from /root/reference/tests (nothing is copied); the container that has no reference
checkout (the GPU box) skips this module.  What must hold:

* nothing fails except the three tests that assert properties of out-of-scope integrators by NAME
  (``generalised_leapfrog`` / ``midpoint`` in the registry, ``GeneralisedLeapfrogIntegrator``'s class hierarchy);
* the files that cover the path pass in (at least) the numbers recorded below -- a drop to "skipped" would mean a
  component went missing, which is how this test tells "not provided" from "passes".
"""

import os
import subprocess
import sys
import tempfile
import xml.etree.ElementTree as ET

import pytest

REF_TESTS = "/root/reference/tests"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

pytestmark = pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="no reference checkout in this container")

# (directory or file under the reference's tests/, minimum number of its tests that must PASS against torchebm_amd)
TARGETS = [
    ("samplers/test_langevin_dynamics.py", 10),   # + 7 CUDA-only tests the CPU container skips
    ("samplers/test_hmc.py", 24),
    ("samplers/test_gradient_descent.py", 7),
    ("samplers/test_api_contract.py", 43),
    ("integrators/test_euler_maruyama.py", 27),
    ("integrators/test_heun.py", 22),
    ("integrators/test_leapfrog.py", 24),
    ("integrators/test_symplectic_base.py", 8),
    ("integrators/test_integrator_utils.py", 20),
    ("core/test_base_model.py", 18),
    ("core/test_base_scheduler.py", 66),
    ("core/test_schedulable.py", 15),
    ("core/test_base_module.py", 8),
    ("core/test_loss.py", 15),
    ("core/test_gpu_first.py", 5),
    ("core/test_energy_function.py", 1),
    ("losses/test_contrastive_divergence.py", 17),
]
# out-of-scope components asserted by name (GeneralisedLeapfrog is RMHMC-only, midpoint is an ODE tableau)
EXPECTED_FAILURES = {
    "test_get_integrator_uses_class_defaults",
    "test_midpoint_registry_name",
    "test_hierarchy_and_separable_flags",
}


def test_reference_test_files_pass_against_this_package():
    with tempfile.TemporaryDirectory() as tmp:
        xml = os.path.join(tmp, "report.xml")
        env = dict(os.environ)
        env["PYTHONPATH"] = os.pathsep.join([os.path.join(HERE, "ref_compat"), "/root/reference", REPO])
        env.pop("PYTEST_ADDOPTS", None)
        dirs = [os.path.join(REF_TESTS, d) for d in ("core", "integrators", "samplers", "losses", "utils", "test_generator.py")]
        cmd = [sys.executable, "-m", "pytest", "-p", "ref_alias", *dirs, "-q", "--no-header", "-p", "no:cacheprovider",
               "--rootdir", tmp, "-c", os.devnull, "-W", "ignore", "--tb=line", f"--junit-xml={xml}", "-o", "junit_family=xunit1"]
        run = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=1500)
        assert os.path.exists(xml), run.stdout[-3000:] + run.stderr[-3000:]
        cases = list(ET.parse(xml).getroot().iter("testcase"))
    passed, failed = {}, []
    for c in cases:
        kinds = {child.tag for child in c}
        where = (c.get("file") or c.get("classname", "")).replace(".py", "")
        if kinds & {"failure", "error"}:
            failed.append((where, c.get("name")))
        elif "skipped" not in kinds:
            passed[where] = passed.get(where, 0) + 1
    unexpected = [f for f in failed if f[1].split("[")[0] not in EXPECTED_FAILURES]
    assert not unexpected, unexpected
    for target, minimum in TARGETS:
        key = target[:-3]
        got = sum(n for where, n in passed.items() if where.endswith(key))
        assert got >= minimum, (target, got, minimum, sorted(passed))
    assert sum(passed.values()) >= 340, sum(passed.values())


def test_reference_distributed_tests_pass_against_this_package():
    """The reference's 2-process gloo tests for the sharded path: the torch.distributed shim, per-rank
    generator streams of the samplers, rank-local PCD buffers + cross-rank mixing + checkpoint round trip, and an
    FSDP2-sharded energy model under the seeded samplers.  Their worker processes get the import alias through
    tests/ref_compat/sitecustomize.py."""
    files = ["test_distributed_shim.py", "test_generator_ranks.py", "test_pcd_buffer_ranks.py", "test_fsdp2_energy_model.py"]
    with tempfile.TemporaryDirectory() as tmp:
        xml = os.path.join(tmp, "report.xml")
        env = dict(os.environ)
        env["PYTHONPATH"] = os.pathsep.join([os.path.join(HERE, "ref_compat"), "/root/reference", REPO])
        env.pop("PYTEST_ADDOPTS", None)
        cmd = [sys.executable, "-m", "pytest", "-p", "ref_alias", *[os.path.join(REF_TESTS, "distributed", f) for f in files],
               "-q", "--no-header", "-p", "no:cacheprovider", "--rootdir", tmp, "-c", os.devnull, "-W", "ignore", "--tb=short",
               f"--junit-xml={xml}", "-o", "junit_family=xunit1"]
        run = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=1500)
        assert os.path.exists(xml), run.stdout[-3000:] + run.stderr[-3000:]
        cases = list(ET.parse(xml).getroot().iter("testcase"))
    bad = [(c.get("file"), c.get("name")) for c in cases if {child.tag for child in c} & {"failure", "error"}]
    ok = [c for c in cases if not len(c)]
    assert not bad, (bad, run.stdout[-2000:])
    assert len(ok) >= 15, (len(ok), run.stdout[-2000:])


def test_reference_examples_on_the_path_run_unmodified():
    """The reference's example scripts that live on the path -- energy landscapes, a custom energy, scheduler anatomy,
    Langevin 101, HMC 101, parallel chains, CD-k and PERSISTENT CD (BASELINE config 5's recipe: MLP 2-128-128-1 on
    two-moons) -- executed unmodified by the reference's own smoke test (TORCHEBM_SMOKE=1), importing this package."""
    slugs = ["01-energy-landscapes", "02-custom-energy", "01-scheduler-anatomy", "01-langevin-101", "02-hmc-101",
             "03-parallel-chains", "01-cd-k", "02-persistent-cd"]
    with tempfile.TemporaryDirectory() as tmp:
        xml = os.path.join(tmp, "report.xml")
        env = dict(os.environ)
        env["PYTHONPATH"] = os.pathsep.join([os.path.join(HERE, "ref_compat"), "/root/reference", REPO])
        env.pop("PYTEST_ADDOPTS", None)
        cmd = [sys.executable, "-m", "pytest", "-p", "ref_alias", os.path.join(REF_TESTS, "examples", "test_examples_smoke.py"),
               "-k", " or ".join(slugs), "-q", "--no-header", "-p", "no:cacheprovider", "--rootdir", tmp, "-c", os.devnull,
               "-W", "ignore", "--tb=short", f"--junit-xml={xml}", "-o", "junit_family=xunit1"]
        run = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=1500)
        assert os.path.exists(xml), run.stdout[-3000:] + run.stderr[-3000:]
        cases = list(ET.parse(xml).getroot().iter("testcase"))
    bad = [c.get("name") for c in cases if len(c)]  # failed, errored or skipped
    assert not bad and len(cases) == len(slugs), (bad, len(cases), run.stdout[-3000:])
