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


def _run(harness, tmp_path, call, options=(), mode=None, verbose=0):
    n, ndim = call.initial_positions.shape
    lines = [f"opt {name} {kind} {val}" for name, kind, val in options]
    if mode:
        lines.append("mode " + mode)
    lines.append(f"{n} {ndim} {call.edge_i.size} {call.n_iter} {call.convergence_window} "
                 f"{call.convergence_check_freq} {verbose}")
    lines.append(f"{call.k0!r} {call.cooling_rate!r} {call.c_repulsion!r} {call.relative_epsilon!r}")
    for arr in (call.initial_positions, call.dissimilarity_matrix, call.threshold_matrix, call.degrees,
                call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh):
        lines.append(_fmt(arr))
    path = tmp_path / "call.txt"
    path.write_text("\n".join(lines) + "\n")
    res = subprocess.run([harness, str(path)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    return json.loads(res.stdout)


def one_point(call):
    import dataclasses
    return dataclasses.replace(call, initial_positions=call.initial_positions[:1],
                               dissimilarity_matrix=call.dissimilarity_matrix[:1, :1],
                               threshold_matrix=call.threshold_matrix[:1, :1], degrees=call.degrees[:1],
                               edge_i=call.edge_i[:0], edge_j=call.edge_j[:0], edge_dist=call.edge_dist[:0],
                               edge_thresh=call.edge_thresh[:0])


def test_registration_and_r_error_without_a_device(harness, tmp_path):
    """n = 1: the reference's first guard (src/optimization.cpp:131) fires before any device work."""
    call = core.prepare_layout_call(quickstart_matrix(), 2, 10, 5.0, 0.03, 0.7, 1e-4, 5,
                                    None, False, 3, False, np.random.default_rng(0))
    one = one_point(call)
    out = _run(harness, tmp_path, one, [("topolow.seed", "int", 1)])
    assert out["error"] == "Need at least 2 points for embedding"
    assert out["protect_depth"] == 0 and out["interrupted"] == 0 and out["unif_rand_calls"] == 0


def test_cv_fold_entry_equals_the_library_routine(harness, tmp_path):
    """`.Call("_topolow_cv_fold", row, col, value, code, n, picks, preserve_order, named)`: one fold's
    payload from the non-NA cells (what R/adaptive_sampling.R:2608-2616 + R/core.R:269-436 produce from
    the masked matrix).  No device work: compared with topolow_cv_fold driven through ctypes."""
    rng = np.random.default_rng(5)
    n = 23
    pts = rng.normal(size=(n, 3))
    D = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    M = D.astype(object)
    for a in range(n):
        for b in range(a + 1, n):
            u = rng.random()
            if u < 0.35:
                M[a, b] = M[b, a] = None
            elif u < 0.45:
                M[a, b] = M[b, a] = ">" + repr(float(D[a, b]))
            elif u < 0.5:
                M[a, b] = M[b, a] = "<" + repr(float(D[a, b]))
    m = core.coded_matrix(M)
    rows, cols = np.nonzero(~np.isnan(m.values.T))      # column-major listing: (col, row) pairs
    rows, cols = cols, rows
    vals, codes = m.values[rows, cols], m.codes[rows, cols]
    pos_of = np.full(n * n, -1, np.int64)
    pos_of[rows + cols * n] = np.arange(rows.size)
    cells = _native.CellList(n, rows, cols, vals, codes, pos_of)
    lin = rows + cols * n
    picks = rng.choice(lin[rows != cols], size=20, replace=False)
    for preserve, named in ((0, 0), (1, 0), (0, 1)):
        want = _native.cv_fold(cells, picks, bool(preserve), bool(named))
        lines = ["mode cvfold", f"{n} {rows.size} {picks.size} {preserve} {named}", _fmt(rows), _fmt(cols),
                 _fmt(vals), _fmt(codes), _fmt(picks)]
        path = tmp_path / "fold.txt"
        path.write_text("\n".join(lines) + "\n")
        res = subprocess.run([harness, str(path)], capture_output=True, text=True, timeout=60)
        assert res.returncode == 0, res.stderr
        got = json.loads(res.stdout)
        order, deg, ei, ej, ed, et, hi, hj, ht, vmax = want
        assert got["order"] == ([] if order is None else order.tolist())
        assert got["degrees"] == deg.tolist() and got["edge_i"] == ei.tolist() and got["edge_j"] == ej.tolist()
        assert got["edge_dist"] == ed.tolist() and got["edge_thresh"] == et.tolist()
        assert got["holdout_i"] == hi.tolist() and got["holdout_j"] == hj.tolist() and got["holdout_truth"] == ht.tolist()
        assert got["numeric_max"] == [vmax] and got["protect_depth"] == 0


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
    # no topolow.seed: the order stream is seeded from a hash of .Random.seed READ in place -- the
    # reference draws nothing from R's RNG (src/RcppExports.cpp:19, src/optimization.cpp:153-154), so the
    # caller's stream after set.seed() must be exactly where it was
    free = _run(harness, tmp_path, call, [("fake.random_seed", "int", 3)])
    assert np.isfinite(free["positions"]).all() and free["protect_depth"] == 0
    assert free["unif_rand_calls"] == 0
    untouched = _run(harness, tmp_path, one_point(call), [("fake.random_seed", "int", 3)])   # never reaches the seed
    assert free["random_seed_hash"] == untouched["random_seed_hash"]
    again = _run(harness, tmp_path, call, [("fake.random_seed", "int", 3)])
    other = _run(harness, tmp_path, call, [("fake.random_seed", "int", 4)])
    assert again["positions"] == free["positions"]            # same .Random.seed, same first call: reproducible
    assert other["positions"] != free["positions"]
    bare = _run(harness, tmp_path, call)                      # no RNG state yet: clock / pid
    assert np.isfinite(bare["positions"]).all() and bare["unif_rand_calls"] == 0
    # verbose lines travel through Rprintf (sink()-able), in the reference's format (:183-188, :298-301)
    loud = _run(harness, tmp_path, call, [("topolow.seed", "int", 42)], verbose=1)
    assert loud["positions"] == out["positions"]
    assert "Points: 5, Pairs per iteration: 10" in loud["printed"] and "Parameters: k0=5, cooling=0.03, c_rep=0.7" in loud["printed"]
    assert "Iter " in loud["printed"] and "MAE=" in loud["printed"] and ", k=" in loud["printed"]
    assert ("Converged (plateau) at iter %d" % want.iterations in loud["printed"] or
            "Converged (MAE worsening, best restored) at iter %d" % want.iterations in loud["printed"])
    assert out["printed"] == ""


@pytest.mark.gpu
def test_batch_entry_equals_the_library_batch(harness, tmp_path):
    """`.Call("_topolow_optimize_layout_exact_batch", calls)`: the per-fold loop of likelihood_function
    (R/adaptive_sampling.R:2604-2693) as one launch.  Six copies of one payload, odd ones with NULL
    matrices (the edge list is the matrix), the first three edges scored as hold-out pairs."""
    call = core.prepare_layout_call(quickstart_matrix(), 2, 300, 5.0, 0.03, 0.7, 1e-4, 5,
                                    None, False, 3, False, np.random.default_rng(4))
    out = _run(harness, tmp_path, call, [("topolow.seed", "int", 11)], mode="batch 6 3")
    assert len(out["results"]) == 6 and out["protect_depth"] == 0 and out["unif_rand_calls"] == 0
    est = lambda p: np.sqrt(((p[:, None] - p[None]) ** 2).sum(-1))
    for b, r in enumerate(out["results"]):
        assert r["error"] is None and r["n_names"] == 9 and r["holdout_count"] == 3.0
        pos = np.array(r["positions"]).reshape((5, 2), order="F")
        d = est(pos)
        want = sum(abs(call.edge_dist[q] - d[call.edge_i[q], call.edge_j[q]]) for q in range(3))
        assert r["holdout_sum_abs"] == pytest.approx(want, rel=1e-12)
        assert r["iterations"] > 0 and r["iterations_run"] >= r["iterations"] and np.isfinite(r["final_mae"])
    # different seeds per call (options(topolow.seed) + index, hashed), matrices or edge list alike
    assert len({tuple(r["positions"]) for r in out["results"]}) == 6
    again = _run(harness, tmp_path, call, [("topolow.seed", "int", 11)], mode="batch 6 3")
    assert [r["positions"] for r in again["results"]] == [r["positions"] for r in out["results"]]


@pytest.mark.gpu
def test_cv_sweep_entry_equals_the_library_sweep(harness, tmp_path):
    """`.Call("_topolow_cv_sweep", list(...))`: all folds of a sweep in one call, per-fold scores back -- the same
    numbers as topolow_cv_sweep driven through ctypes."""
    rng = np.random.default_rng(8)
    n = 60
    pts = rng.normal(size=(n, 3))
    D = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    D[rng.random((n, n)) < 0.3] = np.nan
    D = np.where(np.isnan(D) | np.isnan(D.T), np.nan, D)
    np.fill_diagonal(D, 0.0)
    m = core.coded_matrix(D)
    rows, cols = np.nonzero(~np.isnan(m.values.T))
    rows, cols = cols, rows
    vals, codes = m.values[rows, cols], m.codes[rows, cols]
    pos_of = np.full(n * n, -1, np.int64)
    pos_of[rows + cols * n] = np.arange(rows.size)
    cells = _native.CellList(n, rows, cols, vals, codes, pos_of)
    lin = (rows + cols * n)[rows != cols]
    F = 6
    ndim = [2, 3, 2, 4, 3, 2]
    picks = [rng.choice(lin, size=40, replace=False) for _ in range(F)]
    draws = [rng.random((d, n - 1)) for d in ndim]
    k0, cool, crep = [3.0 + f for f in range(F)], [0.02] * F, [0.01] * F
    seeds = [11 + f for f in range(F)]
    want = _native.cv_sweep(cells, False, False, ndim, k0, cool, crep, picks, draws, seeds, 80, 1e-4, 5, 3, "f64")
    p_off = np.concatenate([[0], np.cumsum([p.size for p in picks])])
    d_off = np.concatenate([[0], np.cumsum([u.size for u in draws])])
    lines = ["mode cvsweep", f"{n} {rows.size} {F} 0 0 80 5 3 1e-4 {int(p_off[-1])} {int(d_off[-1])}", _fmt(rows), _fmt(cols),
             _fmt(vals), _fmt(codes), _fmt(ndim), _fmt(k0), _fmt(cool), _fmt(crep), _fmt(np.concatenate(picks)), _fmt(p_off),
             " ".join(repr(float(x)) for u in draws for x in u.ravel()), _fmt(d_off), _fmt(seeds)]
    hsum, hcnt, its, conv, ec, _secs = want
    for mode in ("cvsweep", "cvsweep_named"):      # positional list; the same arguments NAMED, in reverse order
        lines[0] = "mode " + mode
        path = tmp_path / (mode + ".txt")
        path.write_text("\n".join(lines) + "\n")
        res = subprocess.run([harness, str(path)], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        got = json.loads(res.stdout)
        assert got["holdout_sum_abs"] == hsum.tolist() and got["holdout_count"] == hcnt.tolist()
        assert got["iterations"] == its.tolist() and got["converged"] == conv.tolist() and got["error_code"] == ec.tolist()
        assert got["protect_depth"] == 0 and all(c > 0 for c in hcnt)


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
