"""Every example script runs to completion (reference: tests/examples/test_examples_smoke.py:60-74):
on CPU here, and on the GPU (HIP routes) under -m gpu."""

import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = sorted(glob.glob(os.path.join(ROOT, "examples", "*.py")))


def _run(script, extra_env):
    env = dict(os.environ, TORCHEBM_SMOKE="1", **extra_env)
    out = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout


@pytest.mark.parametrize("script", SCRIPTS, ids=[os.path.basename(s) for s in SCRIPTS])
def test_example_runs_on_cpu(script):
    assert "cpu" in _run(script, {"CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})


@pytest.mark.gpu
@pytest.mark.parametrize("script", SCRIPTS, ids=[os.path.basename(s) for s in SCRIPTS])
def test_example_runs_on_gpu(script, cuda_device):
    assert "cuda" in _run(script, {})
