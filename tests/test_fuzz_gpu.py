"""Short runs of the fuzzers (tests/fuzz_energy_grad.py, tests/fuzz_sampler.py, tests/fuzz_transformer.py) with fixed seeds: random
geometries / sampler configurations against the oracle. Each runs in its own process, as on the command line."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,seed,trials", [("fuzz_energy_grad.py", 21, 14), ("fuzz_sampler.py", 21, 10), ("fuzz_transformer.py", 21, 10)])
def test_fuzz(script, seed, trials):
    env = dict(os.environ, FZ_TRIALS=str(trials))
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", script), str(seed)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert "Memory access fault" not in r.stdout + r.stderr, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "failures: 0" in r.stdout
