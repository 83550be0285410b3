"""A slice of the randomised sweeps of tests/fuzz/ inside the driver-run GPU suite: the sweeps found
two races in earlier rounds and used to run by hand only.  Each script is its own process (its own seeds, its own
module-level state) and exits non-zero on the first failing case it counted."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (script, seed, cases): sized so that the four slices together take well under a minute on an MI355X
SLICES = [("fuzz_hot.py", 501, 50), ("fuzz_parity.py", 502, 50), ("fuzz_int.py", 503, 50), ("fuzz_round4.py", 504, 50)]


@pytest.mark.gpu
@pytest.mark.parametrize("script,seed,cases", SLICES)
def test_fuzz_slice(script, seed, cases):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", script), str(seed), str(cases)],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = res.stdout.decode(errors="replace")
    assert res.returncode == 0, out[-3000:]
    assert "0 failures" in out, out[-3000:]
