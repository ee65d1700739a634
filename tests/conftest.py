import os
import sys

import pytest

try:  # torch bundles its own HIP runtime; it must be the first one loaded in a process that also loads ours
    import torch  # noqa: F401
except ImportError:
    pass

# the library reads no environment variable; the suite steers its engine A/Bs through the environment, so the Python wrapper is asked to forward
# the documented switches (vectordb_amd/_lib.py TUNING_NAMES -> eps_set_tuning) before every call
os.environ["EPS_TUNING_FROM_ENV"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libepsilla_ref.so (reference compiled verbatim)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle, build_oracle
    build_oracle()
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.pyoracle import Ref, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref/libepsilla_ref.so not built (needs /root/reference; `make -C oracle ref`)")
    return Ref()
