"""The R `.Call` shim (topolow_amd/r/topolow_shim.c) compiled against a test double of R's C API
(tests/fake_r/) and driven the way `useDynLib(topolow, .registration = TRUE)` + `.Call` drive the
reference's glue (src/RcppExports.cpp:16-49): registration by name with arity 16, the 16 arguments in
R/core.R:439-456's order and types, the named 5-element result list (src/optimization.cpp:375-381),
R errors raised after every resource is released, interrupts re-raised.  R itself is absent from the
image; everything below the shim is the real library."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import layout_call_args
from tests.helpers import quickstart_matrix
from topolow_amd import _native, core

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("rshim") / "harness"
    csrc = os.path.join(ROOT, "topolow_amd", "csrc")
    cmd = ["gcc", "-O1", "-I", os.path.join(ROOT, "tests", "fake_r"), "-I", os.path.join(ROOT, "include"),
           "-o", str(out), os.path.join(ROOT, "tests", "fake_r", "fake_r.c"),
           os.path.join(ROOT, "topolow_amd", "r", "topolow_shim.c"), "-L", csrc, "-ltopolow_relax",
           "-Wl,-rpath," + csrc, "-lm"]
    subprocess.run(cmd, check=True, capture_output=True)
    return str(out)


def _fmt(a):
    return " ".join("Inf" if np.isinf(x) else repr(float(x)) for x in np.asarray(a, dtype=np.float64).ravel(order="F"))


def _run(harness, tmp_path, call, options=()):
    n, ndim = call.initial_positions.shape
    lines = [f"opt {name} {kind} {val}" for name, kind, val in options]
    lines.append(f"{n} {ndim} {call.edge_i.size} {call.n_iter} {call.convergence_window} "
                 f"{call.convergence_check_freq} 0")
    lines.append(f"{call.k0!r} {call.cooling_rate!r} {call.c_repulsion!r} {call.relative_epsilon!r}")
    for arr in (call.initial_positions, call.dissimilarity_matrix, call.threshold_matrix, call.degrees,
                call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh):
        lines.append(_fmt(arr))
    path = tmp_path / "call.txt"
    path.write_text("\n".join(lines) + "\n")
    res = subprocess.run([harness, str(path)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    return json.loads(res.stdout)


def test_registration_and_r_error_without_a_device(harness, tmp_path):
    """n = 1: the reference's first guard (src/optimization.cpp:131) fires before any device work."""
    import dataclasses
    call = core.prepare_layout_call(quickstart_matrix(), 2, 10, 5.0, 0.03, 0.7, 1e-4, 5,
                                    None, False, 3, False, np.random.default_rng(0))
    one = dataclasses.replace(call, initial_positions=call.initial_positions[:1],
                              dissimilarity_matrix=call.dissimilarity_matrix[:1, :1],
                              threshold_matrix=call.threshold_matrix[:1, :1], degrees=call.degrees[:1],
                              edge_i=call.edge_i[:0], edge_j=call.edge_j[:0], edge_dist=call.edge_dist[:0],
                              edge_thresh=call.edge_thresh[:0])
    out = _run(harness, tmp_path, one, [("topolow.seed", "int", 1)])
    assert out == {"error": "Need at least 2 points for embedding", "protect_depth": 0, "interrupted": 0}


@pytest.mark.gpu
def test_dot_call_round_trip_equals_the_library(harness, tmp_path):
    call = core.prepare_layout_call(quickstart_matrix(), 2, 1000, 5.0, 0.03, 0.7, 1e-4, 5,
                                    None, False, 3, False, np.random.default_rng(4))
    out = _run(harness, tmp_path, call, [("topolow.seed", "int", 42)])
    assert out["names"] == ["positions", "converged", "iterations", "final_mae", "final_k"]
    assert out["dim"] == [5, 2] and out["types"] == [14, 10, 13, 14, 14] and out["protect_depth"] == 0
    want = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=42)
    got = np.array(out["positions"]).reshape((5, 2), order="F")
    assert np.array_equal(got, want.positions)
    assert out["iterations"] == want.iterations and bool(out["converged"]) == want.converged
    assert out["final_mae"] == want.final_mae and out["final_k"] == want.final_k
    # backend options travel through options(): a forced schedule / precision must be honoured
    slab = _run(harness, tmp_path, call, [("topolow.seed", "int", 42), ("topolow.schedule", "str", "slab")])
    want_slab = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=42, schedule="slab")
    assert np.array_equal(np.array(slab["positions"]).reshape((5, 2), order="F"), want_slab.positions)
    # no topolow.seed: the order stream is seeded from R's RNG (set.seed() reproducibility)
    free = _run(harness, tmp_path, call)
    assert np.isfinite(free["positions"]).all() and free["protect_depth"] == 0


@pytest.mark.gpu
def test_interrupt_is_reraised_after_cleanup(harness, tmp_path):
    call = core.prepare_layout_call(quickstart_matrix(), 2, 400, 5.0, 1e-4, 0.7, 1e-12, 1000,
                                    None, False, 3, False, np.random.default_rng(4))
    # (the single-workgroup GS kernel runs a whole small embedding in one launch; the schedules driven
    #  from the host -- slab, tile GS -- poll between launches)
    opts = [("topolow.seed", "int", 7), ("topolow.schedule", "str", "slab")]
    quiet = _run(harness, tmp_path, call, opts)
    assert quiet["iterations"] > 0 and quiet["interrupt_polls"] >= 400 // 50      # polled every 50 iterations
    out = _run(harness, tmp_path, call, opts + [("fake.interrupt_after", "str", "2")])
    assert out["interrupted"] == 1 and out["protect_depth"] == 0 and "nterrupt" in out["error"]
