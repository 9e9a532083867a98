import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


class _Tuning:
    """Test / tuning switches of the HIP library (diffsal_set_tuning); everything set through the fixture is unset afterwards."""

    def __init__(self):
        self._touched = set()

    def set(self, name, value):
        from diff_sal_amd import _lib

        self._touched.add(name)
        _lib.set_tuning(name, value)

    def reset(self):
        from diff_sal_amd import _lib

        for n in self._touched:
            _lib.set_tuning(n, None)
        self._touched.clear()


@pytest.fixture
def tuning():
    t = _Tuning()
    yield t
    t.reset()
