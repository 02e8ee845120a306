"""Randomised parity sweep through the C ABI (tools/fuzz_parity.py): problem shape, constant poses, loss and trust-region
strategy drawn at random; cost trace, accept / reject sequence and final cost against the CPU oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 19])
def test_random_problem_shapes_match_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "16", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout
