"""GPU: the 64-pixel form of the per-box convolution kernels (k_conv_patch<..., HALF>, csrc/conv_patch.hip) is not selected by the
default policy any more (MD_PATCH_HALF=0 since the end of round 3) but stays a supported switch: the random-geometry convolution tests
and the fused BatchNorm-backward-reduction cases are run once more in a child process with MD_PATCH_HALF=2 (every layer that can be built
in the 64-pixel form uses it).  The switch is read once per process, hence the child."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_conv_tests_pass_with_the_64_pixel_form_forced():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MD_PATCH_HALF="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_conv_random_gpu.py",
                        "tests/test_conv_pers_gpu.py::test_fused_reduction_in_the_per_box_kernels", "tests/test_shapes_gpu.py"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
