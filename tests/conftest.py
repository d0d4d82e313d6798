import os
import sys

import pytest

try:   # a process that uses both torch and libtopolow_relax.so must load torch's HIP runtime FIRST (torch
    import torch  # noqa: F401  ships its own copy; loaded second it reports "No HIP GPUs are available")
except Exception:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_checkers():
    """The CPU checkers (oracle, slab model) are compiled on demand; the HIP library is built
    by __graft_entry__.build() / make -C topolow_amd/csrc and must already exist."""
    import oracle
    oracle.build()
    from tests.models import slab_model
    slab_model.build()
    yield


def layout_call_args(call):
    return (call.initial_positions, call.dissimilarity_matrix, call.threshold_matrix, call.degrees,
            call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.n_iter, call.k0,
            call.cooling_rate, call.c_repulsion, call.relative_epsilon, call.convergence_window,
            call.convergence_check_freq)
