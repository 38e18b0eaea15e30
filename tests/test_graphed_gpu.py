"""Whole-step HIP graph helper (src/utils/graphed.py): runs tools/graphed_check.py in a fresh process (a capture after eager
model steps on the legacy default stream crashes in the runtime on this stack, and every other test here runs eagerly on that
stream) and expects bit-identical losses / gradients between the graph replays and the eager steps."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_graphed_step_matches_eager_in_a_fresh_process():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "graphed_check.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "graphed_check OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
