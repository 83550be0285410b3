import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


class _Golden:
    """Lazy access to the reference outputs stored by tests/golden/gen_golden.py."""

    def __init__(self):
        self._files = {}

    def _get(self, which):
        if which not in self._files:
            self._files[which] = np.load(os.path.join(GOLDEN, which + ".npz"))
        return self._files[which]

    def outputs(self, case, kind="out"):
        f = self._get("big" if case["big"] else "small")
        res, i = [], 0
        while "%s/%s%d" % (case["name"], kind, i) in f.files:
            res.append(f["%s/%s%d" % (case["name"], kind, i)])
            i += 1
        return res

    def filters(self):
        return self._get("filters")


@pytest.fixture(scope="session")
def golden():
    return _Golden()
