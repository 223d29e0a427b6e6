import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/radon_oracle.c), compiled on demand with gcc."""
    from oracle import radon_oracle
    radon_oracle.build()
    return radon_oracle


@pytest.fixture(scope="session")
def torch_node():
    """The C++ autograd nodes (ct_pvae_amd/csrc/torch_node.cpp): built by __graft_entry__.build(); compiled on demand (g++,
    under a minute) when a test session starts without them."""
    from ct_pvae_amd import _lib
    if _lib.torch_node() is None:
        _lib.build_torch_node()
        _lib._node = False
    node = _lib.torch_node()
    assert node is not None
    return node


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _reset_library_knobs():
    """Developer knobs of the library (ctpvae_tune_set) never leak from one test into the next."""
    yield
    from ct_pvae_amd import _lib
    if os.path.exists(_lib.LIB_PATH):
        _lib.tune("*")


@pytest.fixture(autouse=True, scope="session")
def _poisoned_outputs():
    """Outputs of the projector calls start as NaN in every test (ct_pvae_amd.forward_functions.POISON_OUTPUTS): an element a
    launch does not write fails its comparison instead of showing the previous call's value out of recycled memory."""
    from ct_pvae_amd import forward_functions
    forward_functions.POISON_OUTPUTS = True
    yield
