"""The randomised geometry checks (tests/stress_*.py: every transform against the oracle at random window / hop / length /
batch, including the W = 4096 / 8192 band kernels) as collected GPU tests: fixed seeds, at most 20 iterations each."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.parametrize("script,seed,iters", [
    ("stress_random.py", 12345, 20),
    ("stress_random.py", 2024, 20),
    ("stress_random_more.py", 777, 16),
    ("stress_w4096.py", 4096, 20),
    ("stress_w4096.py", 8192, 12),
])
def test_random_geometries_against_the_oracle(script, seed, iters):
    res = subprocess.run([sys.executable, os.path.join(HERE, script), str(seed), str(iters)], capture_output=True, text=True, timeout=900)
    tail = (res.stdout + res.stderr)[-3000:]
    assert res.returncode == 0, tail
    assert f"iterations {iters} done, failures: 0" in res.stdout, tail
