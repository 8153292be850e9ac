import os
import sys

import pytest

# The library reads its GENPHI_* tuning / test hooks from the environment only under this gate (csrc/planner.h: env_hook); the tests steer
# kernel families, LDS budgets, in-place runs ... through them (monkeypatch.setenv), so the gate is on for the whole session -- and for
# the worker processes the tests start.  Set before the library is loaded: it is read once.
os.environ["GENPHI_ENV_HOOKS"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gen():
    import genlib_jl_amd
    return genlib_jl_amd


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.fit_threads_to_quota()      # (one OpenMP thread per CPU the container may use: 16 on a GPU box, where 256 threads run 2.3 x slower)
    return O
