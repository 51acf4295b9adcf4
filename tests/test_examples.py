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


@pytest.mark.gpu
def test_pcd_example_whole_step_graph_prints_the_eager_loop_losses(cuda_device):
    """examples/pcd_two_moons.py with TORCHEBM_WHOLE_STEP_GRAPH=1 (utils.GraphedTrainingStep on a stream of changing batches) prints
    the same loss lines as the plain loop."""
    script = os.path.join(ROOT, "examples", "pcd_two_moons.py")
    eager = [line for line in _run(script, {}).splitlines() if line.startswith("step")]
    graphed = [line for line in _run(script, {"TORCHEBM_WHOLE_STEP_GRAPH": "1"}).splitlines() if line.startswith("step")]
    assert eager and eager == graphed
