"""Randomised differential tests on the GPU (-m gpu): many small random problems -- every ndim the
kernels are instantiated for, coincident points, zero targets, huge and tiny targets, thresholds,
fully measured and almost empty matrices, check_freq / window / epsilon corners -- through the C ABI
against the CPU oracle replaying the same pair order (GS kernel, exact) and against the CPU slab
model (slab kernel)."""
import dataclasses

import numpy as np
import pytest

from oracle import topolow_oracle as orc
from tests.conftest import layout_call_args
from tests.models import slab_model
from tests.test_gpu_parity import _decode_rounded, _model_run
from topolow_amd import _native, core

pytestmark = pytest.mark.gpu


def _fuzz_problem(rng, n, dim):
    style = rng.integers(0, 5)
    pts = rng.normal(size=(n, dim)) * rng.choice([0.01, 1.0, 30.0])
    if style == 1:                       # duplicates: coincident points with zero targets
        pts[rng.integers(0, n, size=max(1, n // 3))] = pts[0]
    D = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    if style == 2:
        D *= rng.choice([1e-6, 1e4])
    M = D.astype(object)
    miss = rng.choice([0.0, 0.3, 0.9])
    iu, ju = np.triu_indices(n, 1)
    for a, b in zip(iu, ju):
        u = rng.random()
        if u < miss:
            M[a, b] = M[b, a] = None
        elif u < miss + 0.15 and style in (3, 4):
            sgn = ">" if rng.random() < 0.5 else "<"
            M[a, b] = M[b, a] = sgn + repr(float(D[a, b] * rng.uniform(0.5, 1.5)))
        else:
            M[a, b] = M[b, a] = repr(float(D[a, b]))
    for a in range(n):
        M[a, a] = "0"
    if all(M[a, b] is None for a, b in zip(iu, ju)):
        M[0, 1] = M[1, 0] = "1.5"
    init = rng.normal(size=(n, dim)) * rng.choice([0.1, 5.0])
    if style == 1:
        init[1] = init[0]                # start with coincident points too (dist = 0 branch)
    return core.prepare_layout_call(M, dim, int(rng.integers(1, 40)), float(rng.uniform(0.1, 25)),
                                    float(rng.uniform(0.001, 0.2)), float(rng.uniform(1e-4, 0.5)),
                                    float(rng.choice([1e-2, 1e-4, 1e-10])), int(rng.integers(1, 6)), init, False,
                                    int(rng.integers(1, 8)), bool(rng.integers(0, 2)))


@pytest.mark.parametrize("block", range(6))
def test_gs_fuzz_bit_parity(block):
    rng = np.random.default_rng(9000 + block)
    calls, seeds = [], []
    for _ in range(10):
        n = int(rng.integers(2, 90))
        dim = int(rng.integers(1, 11))
        calls.append(_fuzz_problem(rng, n, dim))
        seeds.append(int(rng.integers(0, 2 ** 62)))
    results, _ = _native.optimize_layout_exact_batch(calls, seeds=seeds, precision="f64")
    for call, seed, got in zip(calls, seeds, results):
        n = call.initial_positions.shape[0]

        def order_fn(it, arr, n=n, seed=seed):
            arr[:] = _native.gs_pair_order(n, seed, it)
        try:
            ref = orc.optimize_layout_exact(*layout_call_args(call), order_mode=orc.ORDER_SUPPLIED,
                                            order_fn=order_fn)
        except orc.OracleError as e:
            assert isinstance(got, _native.NativeError) and str(e) == str(got)
            continue
        assert not isinstance(got, Exception), got
        assert np.array_equal(got.positions, ref.positions) or np.abs(got.positions - ref.positions).max() <= 1e-12 * (
            1 + np.abs(ref.positions).max())
        assert (got.converged, got.iterations) == (ref.converged, ref.iterations)
        assert got.final_k == ref.final_k
        assert got.final_mae == pytest.approx(ref.final_mae, rel=1e-11, abs=1e-300)


@pytest.mark.parametrize("block", range(3))
def test_slab_fuzz_against_model(block):
    rng = np.random.default_rng(7000 + block)
    for _ in range(6):
        n = int(rng.integers(2, 220))
        dim = int(rng.integers(1, 11))
        call = _fuzz_problem(rng, n, dim)
        iters = min(call.n_iter, 5)
        seed = int(rng.integers(0, 2 ** 62))
        s = _native.Session(n, dim, precision="f64")
        s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
        s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        s.set_positions(call.initial_positions)
        s.begin(iters, call.k0, call.cooling_rate, call.c_repulsion, 1e-12, 1000, 2, seed, 0)
        s.run()
        got = s.get_positions()
        s.close()
        want, _k = _model_run(dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call)), seed, 0,
                              iters, "f64")
        if not np.isfinite(want).all():
            continue
        scale = max(1.0, np.abs(want).max())
        # skip problems on which the schedule itself is chaotic over these iterations (a 1e-13
        # nudge of the start moves the model's own answer): there only the summation order decides
        nudged = dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call),
                                     initial_positions=call.initial_positions * (1 + 1e-13))
        want2, _k = _model_run(nudged, seed, 0, iters, "f64")
        if np.abs(want2 - want).max() > 1e-10 * scale:
            continue
        assert np.abs(got - want).max() <= 1e-7 * scale, (n, dim, np.abs(got - want).max())


@pytest.mark.parametrize("block", range(3))
def test_slab_one_shot_fuzz_properties(block, monkeypatch):
    """The production entry on the slab path (random labels, edge-list-built block, fp32, checks fused into
    one-stage sweeps) on random problems of ragged sizes, every instantiated coordinate count, thresholds,
    duplicates, tiny and huge scales: (1) deterministic for a seed; (2) the reported MAE is the oracle's edge MAE of
    the returned positions; (3) the controller fields are consistent; (4) fused and separate checks agree; (5)
    keeping the caller's labels or relabelling changes nothing in kind (finite, same error class)."""
    rng = np.random.default_rng(8100 + block)
    for case in range(8):
        n = int(rng.integers(2, 500))
        dim = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 16]))
        call = _fuzz_problem(rng, n, dim)
        call = dataclasses.replace(call, n_iter=int(rng.integers(1, 120)), relative_epsilon=float(rng.choice([1e-2, 1e-3])))
        seed = int(rng.integers(1, 2 ** 62))
        outs = []
        for fuse, keep in (("1", False), ("1", False), ("0", False), ("1", True)):
            monkeypatch.setenv("TOPOLOW_FUSE_CHECKS", fuse)
            try:
                outs.append(_native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, schedule="slab",
                                                                 keep_labels=keep))
            except _native.NativeError as e:
                outs.append(e)
        a, b, c, d = outs
        if isinstance(a, Exception):
            assert a.code == _native.ERR_NONFINITE and "Numerical instability at iteration" in str(a), (n, dim, a)
            assert isinstance(b, Exception) and str(b) == str(a)
            continue
        assert not isinstance(b, Exception) and np.array_equal(a.positions, b.positions)
        assert (a.converged, a.iterations, a.final_mae, a.final_k) == (b.converged, b.iterations, b.final_mae, b.final_k)
        assert np.isfinite(a.positions).all() and a.positions.shape == (n, dim)
        assert 0 <= a.iterations <= a.info["iterations_run"] <= call.n_iter
        assert a.converged == (a.info["iterations_run"] < call.n_iter) or a.info["iterations_run"] == call.n_iter
        sm, cnt = orc.edge_error(a.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        want = sm / cnt if cnt else 0.0
        assert a.final_mae == pytest.approx(want, rel=5e-5, abs=1e-6 * max(1.0, float(np.abs(a.positions).max()))), (n, dim)
        # separate error pass instead of the fused one: same trajectory, same verdicts
        assert not isinstance(c, Exception), c
        assert np.array_equal(a.positions, c.positions) and (a.converged, a.iterations) == (c.converged, c.iterations)
        assert a.final_mae == pytest.approx(c.final_mae, rel=5e-6, abs=1e-12)
        assert not isinstance(d, Exception) and np.isfinite(d.positions).all()
