"""The symmetric sweep (topolow_amd/csrc/relax_symm.h) against the row-owner stage kernel it replaces on one-stage
iterations of whole-matrix fp32 sessions (run with -m gpu).  Both compute the same update from the same positions
(reference src/optimization.cpp:203-281 applied to all pairs at once); only the fp32 summation order differs."""
import os

import numpy as np
import pytest

from tests import parity_problems as pp
from tests.conftest import layout_call_args
from topolow_amd import _native

pytestmark = pytest.mark.gpu


def session_run(call, n, dim, symmetric, iters, k0, check_freq=3, stages=1, window=10 ** 9):
    old = os.environ.get("TOPOLOW_SYMMETRIC")
    os.environ["TOPOLOW_SYMMETRIC"] = "1" if symmetric else "0"
    try:
        s = _native.Session(n, dim, precision="f32")
    finally:
        if old is None:
            os.environ.pop("TOPOLOW_SYMMETRIC")
        else:
            os.environ["TOPOLOW_SYMMETRIC"] = old
    s.set_relabel(11)
    s.load_coo(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.degrees)
    s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    s.set_positions(call.initial_positions)
    s.begin(iters, k0, 0.01, 0.01, 1e-4, window, check_freq, 5, stages)
    s.run()
    s.sync()
    r, trace = s.finish(), s.check_trace()
    s.close()
    return r, trace


@pytest.mark.parametrize("dim,thresholds", [(2, 0.0), (3, 0.0), (4, 0.1), (5, 0.0), (5, 0.1), (6, 0.0)])
def test_symmetric_sweep_equals_the_row_owner_sweep(dim, thresholds):
    n = 7205                                     # not a multiple of 32: phantom rows and columns in the last tiles
    call, _ = pp.random_problem(n, dim, 0.7, seed=50 + dim, thresholds=0.0, n_iter=10, k0=1.5)
    if thresholds > 0:                           # a fraction of the measured pairs become ">" / "<" targets
        rng = np.random.default_rng(3)
        code = rng.choice([0, 1, -1], size=call.edge_thresh.shape[0], p=[1 - thresholds, thresholds / 2, thresholds / 2])
        call.edge_thresh[:] = code.astype(call.edge_thresh.dtype)
    scale = float(np.abs(call.initial_positions).max())
    for iters in (1, 7):                         # 7: checks at 3 and 6 ride on the sweeps of iterations 4 and 7
        a, ta = session_run(call, n, dim, False, iters, 1.5)
        b, tb = session_run(call, n, dim, True, iters, 1.5)
        assert np.abs(a.positions - b.positions).max() <= 2e-5 * scale * iters
        assert ta.shape == tb.shape and np.array_equal(ta[:, 0], tb[:, 0])
        assert np.allclose(ta[:, 1], tb[:, 1], rtol=2e-6, atol=0)      # the fused MAE, sum and count
        assert b.final_mae == pytest.approx(a.final_mae, rel=2e-6) and a.iterations == b.iterations


def test_production_entry_with_and_without_the_symmetric_sweep():
    """The one-shot entry on a problem large enough to take the symmetric path: same stop, same MAE to rounding."""
    call, _ = pp.cfg3_generator(7400)
    old = os.environ.get("TOPOLOW_SYMMETRIC")
    try:
        os.environ["TOPOLOW_SYMMETRIC"] = "0"
        a = [_native.optimize_layout_exact_arrays(*layout_call_args(call), seed=1 + q) for q in range(3)]
        os.environ["TOPOLOW_SYMMETRIC"] = "1"
        b = [_native.optimize_layout_exact_arrays(*layout_call_args(call), seed=1 + q) for q in range(3)]
    finally:
        if old is None:
            os.environ.pop("TOPOLOW_SYMMETRIC")
        else:
            os.environ["TOPOLOW_SYMMETRIC"] = old
    for x, y in zip(a, b):
        assert x.converged and y.converged
        assert abs(x.iterations - y.iterations) <= 6                     # two checks (rounding can move a plateau)
        assert y.final_mae == pytest.approx(x.final_mae, rel=1e-3)


def test_symmetric_sweep_reports_a_diverging_run():
    """Fixed one stage per iteration at k far above the stable range: the non-finite guard must fire as it does on
    the row-owner path (reference :359-361)."""
    n = 7205
    call, _ = pp.random_problem(n, 3, 0.7, seed=8, n_iter=10, k0=1.5)
    for sym in (False, True):
        with pytest.raises(_native.NativeError, match="Numerical instability"):
            session_run(call, n, 3, sym, 200, 500.0, stages=1)


def test_single_block_engine_run_takes_the_symmetric_sweep_too():
    """topolow_sessions_run_sharded with ONE block (the whole matrix) and the session's own loop run the same
    kernels in the same order: bit-identical positions, same checks."""
    n, dim = 7205, 3
    call, _ = pp.random_problem(n, dim, 0.7, seed=21, n_iter=10, k0=1.5)

    def make():
        s = _native.Session(n, dim, precision="f32")
        s.load_coo(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.degrees)
        s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        return s
    s = make()
    s.set_positions(call.initial_positions)
    s.begin(40, 2.0, 0.01, 0.01, 1e-4, 10 ** 9, 3, 5, 0)
    s.run()
    s.sync()
    a, ta = s.finish(), s.check_trace()
    s.close()
    s = make()
    b = _native.run_sharded([s], call.initial_positions, 40, 2.0, 0.01, 0.01, 1e-4, 10 ** 9, 3, 5, 0)
    tb = s.check_trace()
    s.close()
    assert np.array_equal(a.positions, b.positions)
    assert a.iterations == b.iterations and b.final_mae == pytest.approx(a.final_mae, rel=1e-12)
    assert np.allclose(ta[:, 1], tb[:, 1], rtol=1e-12)
