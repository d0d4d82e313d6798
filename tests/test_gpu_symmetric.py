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


@pytest.mark.parametrize("dim,thresholds", [(2, 0.0), (3, 0.0), (4, 0.1), (5, 0.0), (5, 0.1), (6, 0.0), (6, 0.1)])
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
    kernels in the same order -- row-owner stages while the layout unfolds, two symmetric half sweeps per two-stage
    iteration, the symmetric sweep afterwards: bit-identical positions, same checks."""
    n, dim = 7205, 3
    call, _ = pp.random_problem(n, dim, 0.7, seed=21, n_iter=10, k0=1.5)

    def make():
        s = _native.Session(n, dim, precision="f32")
        s.load_coo(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.degrees)
        s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        return s
    s = make()
    s.set_positions(call.initial_positions)
    s.begin(60, 4.0, 0.01, 0.01, 1e-4, 10 ** 9, 3, 5, 0)       # 8 iterations of 16 stages, 21 of two (half sweeps), 31 of one
    s.run()
    s.sync()
    a, ta = s.finish(), s.check_trace()
    s.close()
    s = make()
    b = _native.run_sharded([s], call.initial_positions, 60, 4.0, 0.01, 0.01, 1e-4, 10 ** 9, 3, 5, 0)
    tb = s.check_trace()
    s.close()
    assert np.array_equal(a.positions, b.positions)
    assert a.iterations == b.iterations and b.final_mae == pytest.approx(a.final_mae, rel=1e-12)
    assert np.allclose(ta[:, 1], tb[:, 1], rtol=1e-12)


# ----------------------------------------------------------------------------------------
# the symmetric sweep against the CPU model of a one-stage iteration and against the oracle's edge error --
# directly, not through the row-owner kernel (size gate lowered with TOPOLOW_SYMMETRIC_MIN_N)
# ----------------------------------------------------------------------------------------
import dataclasses

from oracle import topolow_oracle as orc
from tests.models import slab_model
from tests.test_gpu_parity import _decode_rounded


class _Env:
    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _with_thresholds(call, frac, seed=3):
    if frac > 0:
        rng = np.random.default_rng(seed)
        code = rng.choice([0, 1, -1], size=call.edge_thresh.shape[0], p=[1 - frac, frac / 2, frac / 2])
        et = code.astype(call.edge_thresh.dtype)
        T = call.threshold_matrix.copy()
        T[call.edge_i, call.edge_j] = et
        T[call.edge_j, call.edge_i] = et
        call = dataclasses.replace(call, edge_thresh=et, threshold_matrix=T)
    return call


def _symmetric_session(call, n, dim, iters, k0, cooling, c_rep, check_freq, profile, relabel=0, precision="f32"):
    """A whole-matrix session forced onto the symmetric sweep: ONE stage per iteration, no early stop."""
    with _Env(TOPOLOW_SYMMETRIC="1", TOPOLOW_SYMMETRIC_MIN_N="0"):
        s = _native.Session(n, dim, precision=precision)
    if relabel:
        s.set_relabel(relabel)
    s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
    s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    s.set_positions(call.initial_positions)
    s.set_profiling(profile)
    s.begin(iters, k0, cooling, c_rep, 1e-12, 10 ** 9, check_freq, 5, 1)
    s.run()
    s.sync()
    pos = s.get_positions()
    trace = s.check_trace()
    counts = s.profile_symmetric() if profile else None
    s.close()
    return pos, trace, counts


def _model_iterations(call_r, iters, k0, cooling, c_rep):
    """Positions after 1..iters one-stage iterations of the CPU model (f64; reference update src/optimization.cpp:
    203-281 applied to every ordered pair from the positions the previous iteration left)."""
    n = call_r.initial_positions.shape[0]
    out, pos, k = [], call_r.initial_positions, k0
    for _ in range(iters):
        pos = slab_model.stage(pos, call_r.dissimilarity_matrix, call_r.threshold_matrix, call_r.degrees, [[0, n]], k,
                               c_rep, "f64")
        out.append(pos)
        k *= 1.0 - cooling
    return out


@pytest.mark.parametrize("n", [33, 66, 1000, 7205])
@pytest.mark.parametrize("dim,thr", [(2, 0.0), (3, 0.15), (4, 0.0), (5, 0.0), (5, 0.15), (6, 0.0), (6, 0.15)])
def test_symmetric_sweep_against_the_model_and_the_oracle(n, dim, thr):
    """n: 33 and 66 leave 31 / 30 phantom rows/columns in the last tile (and fewer tiles than waves), 1000 and 7205 have
    n % 32 = 8 / 5; odd n too (the row-owner ERR instance wants an even block, the sweep's does not); every diagonal tile meets its pairs twice at half weight.  One and seven iterations at one stage:
    positions against slab_model.stage in f64 (bands of test_slab_f32_close_to_model: mean 5e-5, max 5e-3 of the
    displacement scale); the checks at iterations 3 and 6 ride on the sweeps of iterations 4 and 7 (ERR instance):
    their (sum / count) against orc.edge_error of the positions those sweeps read (2e-5, fp32)."""
    k0, cooling, c_rep = 1.5, 0.01, 0.01
    call, _ = pp.random_problem(n, dim, 0.7 if n > 100 else 0.3, seed=90 + n % 50 + dim, n_iter=7, k0=k0)
    call = _with_thresholds(call, thr)
    call_r = dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call))
    want = _model_iterations(call_r, 7, k0, cooling, c_rep)
    scale = np.abs(want[-1] - call.initial_positions).max()
    for iters in (1, 7):
        got, trace, counts = _symmetric_session(call, n, dim, iters, k0, cooling, c_rep, 3, profile=True)
        assert counts[1] + counts[3] == iters, counts            # every iteration ran as a symmetric sweep + apply
        err = np.abs(got - want[iters - 1])
        assert err.mean() <= 5e-5 * scale and err.max() <= 5e-3 * scale, (err.mean() / scale, err.max() / scale)
        if iters == 7:
            assert counts[3] == 2                                # ... two of them also reduced a check's MAE
            assert [int(t) for t in trace[:, 0]] == [3, 6, 7]
            for row in trace:
                s, c = orc.edge_error(want[int(row[0]) - 1], call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
                assert row[1] == pytest.approx(s / c, rel=2e-5), (row, s / c)
        # the unprofiled run (checks beside the next iteration on the second stream) gives the same bits
        again, trace2, _ = _symmetric_session(call, n, dim, iters, k0, cooling, c_rep, 3, profile=False)
        assert np.array_equal(again, got) and np.array_equal(trace2, trace)


def test_symmetric_sweep_with_random_labels_against_the_model():
    """The production entry stores the points in a random order (topolow_session_set_relabel): the sweep then tiles
    the relabelled matrix; host-facing positions stay in the caller's labels."""
    n, dim, k0 = 1000, 5, 1.5
    call, _ = pp.random_problem(n, dim, 0.7, seed=5, n_iter=3, k0=k0)
    call_r = dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call))
    want = _model_iterations(call_r, 3, k0, 0.01, 0.01)
    got, _, counts = _symmetric_session(call, n, dim, 3, k0, 0.01, 0.01, 3, profile=True, relabel=77)
    assert counts[1] + counts[3] == 3
    scale = np.abs(want[-1] - call.initial_positions).max()
    err = np.abs(got - want[-1])
    assert err.mean() <= 5e-5 * scale and err.max() <= 5e-3 * scale


def test_symmetric_sweep_at_config3_size_against_the_model():
    """BASELINE config 3 (N = 10 000, ndim 5, 70 % missing): ONE symmetric sweep from the reference's start positions
    against the model's one-stage iteration over all 10^8 ordered pairs, and the fused check of that sweep's input
    against the oracle's edge error (1.5 x 10^7 edges)."""
    from topolow_amd import core, synthetic
    n, dim = 10000, 5
    prob = synthetic.make_problem(n, latent_dim=dim, missing=0.7, seed=12345)
    init = synthetic.initial_positions(prob.dissimilarity, dim, 12345)
    call = core.prepare_layout_call(prob.dissimilarity, dim, 2, 2.0, 0.01, 0.01, 1e-4, 5, init, False, 1, True)
    call_r = dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call))
    want = _model_iterations(call_r, 2, 2.0, 0.01, 0.01)
    with _Env(TOPOLOW_SYMMETRIC="1"):                      # default size gate: 10 000 >= 7 168
        s = _native.Session(n, dim, precision="f32")
    s.set_relabel(3)
    s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
    s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    s.set_positions(call.initial_positions)
    s.set_profiling(True)
    s.begin(2, 2.0, 0.01, 0.01, 1e-12, 10 ** 9, 1, 5, 1)   # a check after every iteration: iteration 2's sweep is the ERR instance
    s.run()
    s.sync()
    got, trace, counts = s.get_positions(), s.check_trace(), s.profile_symmetric()
    s.close()
    assert counts[1] == 1 and counts[3] == 1
    scale = np.abs(want[-1] - call.initial_positions).max()
    err = np.abs(got - want[-1])
    assert err.mean() <= 5e-5 * scale and err.max() <= 5e-3 * scale, (err.mean() / scale, err.max() / scale)
    sm, cnt = orc.edge_error(want[0], call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    assert int(trace[0, 0]) == 1 and trace[0, 1] == pytest.approx(sm / cnt, rel=2e-5)


# ----------------------------------------------------------------------------------------
# the same sweep in f64 (csrc/relax_symm64.h): the reference's arithmetic type, so the CPU model in f64 is matched to
# rounding -- a far tighter statement about the tiling (phantom rows and columns, the diagonal squares swept from both
# sides, units cut anywhere, the swapped column order of odd lane groups) than the fp32 bands allow
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [33, 66, 1000, 2973])
@pytest.mark.parametrize("dim,thr", [(2, 0.0), (2, 0.15), (3, 0.15), (4, 0.0), (5, 0.0), (5, 0.15), (6, 0.0), (6, 0.15)])
def test_symmetric_sweep_f64_equals_the_model_to_rounding(n, dim, thr):
    """One and seven one-stage iterations in f64 against slab_model.stage in f64 (reference update
    src/optimization.cpp:203-281 on every ordered pair): relative 1e-12 of the displacement scale per iteration (the
    two differ in the order of the sums and in (t - r) / (r + 0.01) x 2k / (4 g + k) against the model's grouping);
    the checks -- 3 and 6 ride on the sweeps of 4 and 7, whose ERR instance adds the delta tiles (exact f64 target minus
    the tile's 4-byte word) so that the MAE is the exact one -- against the oracle's edge error, computed from the exact
    f64 edge list, of the model's positions to 1e-11 (from the words alone: 3e-8);
    and against the row-owner f64 stage kernel (TOPOLOW_SYMMETRIC=0) the same band."""
    k0, cooling, c_rep = 1.5, 0.01, 0.01
    call, _ = pp.random_problem(n, dim, 0.7 if n > 100 else 0.3, seed=190 + n % 50 + dim, n_iter=7, k0=k0)
    call = _with_thresholds(call, thr)
    call_r = dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call))
    want = _model_iterations(call_r, 7, k0, cooling, c_rep)
    scale = np.abs(want[-1] - call.initial_positions).max()
    for iters in (1, 7):
        got, trace, counts = _symmetric_session(call, n, dim, iters, k0, cooling, c_rep, 3, profile=True, precision="f64")
        assert counts[1] + counts[3] == iters, counts            # every iteration ran as a symmetric sweep + apply
        assert np.abs(got - want[iters - 1]).max() <= 1e-12 * scale * iters, np.abs(got - want[iters - 1]).max() / scale
        if iters == 7:
            assert counts[3] == 2                                # the checks at 3 and 6 rode on the sweeps of 4 and 7 (ERR instance)
            assert [int(t) for t in trace[:, 0]] == [3, 6, 7]
            for row in trace:
                sm, c = orc.edge_error(want[int(row[0]) - 1], call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
                assert row[1] == pytest.approx(sm / c, rel=1e-11), (row, sm / c)
        again, trace2, _ = _symmetric_session(call, n, dim, iters, k0, cooling, c_rep, 3, profile=False, precision="f64")
        assert np.array_equal(again, got) and np.array_equal(trace2, trace)
    with _Env(TOPOLOW_SYMMETRIC="0"):
        s = _native.Session(n, dim, precision="f64")
    s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
    s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    s.set_positions(call.initial_positions)
    s.begin(7, k0, cooling, c_rep, 1e-12, 10 ** 9, 3, 5, 1)
    s.run()
    s.sync()
    row_owner = s.get_positions()
    s.close()
    assert np.abs(got - row_owner).max() <= 1e-12 * scale * 7


def test_symmetric_sweep_f64_whole_run_at_config3_generator_size():
    """The production entry in f64 on a 7 400-point problem of the config-3 generator (above the size gate): multi-stage
    iterations on the row-owner kernel, one-stage iterations on the f64 symmetric sweep, against the same run with the
    sweep switched off -- the same schedule and arithmetic, sums grouped differently: same stop within two checks, final
    MAE to 1e-6, and the reported MAE is the oracle's edge error of the returned positions."""
    call, _ = pp.cfg3_generator(7400)
    # (two-stage iterations on the row-owner kernel in both runs: their symmetric form is another schedule, tested above)
    with _Env(TOPOLOW_SYMMETRIC="1", TOPOLOW_SYMMETRIC_TWO_STAGE="0"):
        a = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=3, schedule="slab", precision="f64")
    with _Env(TOPOLOW_SYMMETRIC="0"):
        b = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=3, schedule="slab", precision="f64")
    assert a.converged and b.converged and abs(a.iterations - b.iterations) <= 6
    assert a.final_mae == pytest.approx(b.final_mae, rel=1e-6)
    sm, cnt = orc.edge_error(a.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    assert a.final_mae == pytest.approx(sm / cnt, rel=1e-12)


def test_symmetric_sweep_f64_check_falls_back_when_the_edge_list_is_not_the_block():
    """The fused f64 check sums over the measured cells of the tiles, so it may only replace the pass over the edge list
    when that list IS the block's measured upper triangle (verified by fingerprint when the list is set).  A session
    given half of the edges keeps the separate pass (no ERR launches) and reports the MAE of exactly those edges; with
    random labels and the full list the fused check is used and is exact."""
    n, dim, k0 = 1000, 5, 1.5
    call, _ = pp.random_problem(n, dim, 0.7, seed=77, n_iter=7, k0=k0)
    call = _with_thresholds(call, 0.1)
    call_r = dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call))
    want = _model_iterations(call_r, 7, k0, 0.01, 0.01)
    half = np.arange(call.edge_i.shape[0]) % 2 == 0
    part = dataclasses.replace(call, edge_i=call.edge_i[half], edge_j=call.edge_j[half], edge_dist=call.edge_dist[half],
                               edge_thresh=call.edge_thresh[half])
    got, trace, counts = _symmetric_session(part, n, dim, 7, k0, 0.01, 0.01, 3, profile=True, precision="f64")
    assert counts[1] == 7 and counts[3] == 0
    scale = np.abs(want[-1] - call.initial_positions).max()
    assert np.abs(got - want[-1]).max() <= 1e-12 * scale * 7
    for row in trace:
        sm, c = orc.edge_error(want[int(row[0]) - 1], part.edge_i, part.edge_j, part.edge_dist, part.edge_thresh)
        assert row[1] == pytest.approx(sm / c, rel=1e-11)
    got2, trace2, counts2 = _symmetric_session(call, n, dim, 7, k0, 0.01, 0.01, 3, profile=True, relabel=91, precision="f64")
    assert counts2[1] == 5 and counts2[3] == 2
    assert np.abs(got2 - want[-1]).max() <= 1e-12 * scale * 7
    for row in trace2:
        sm, c = orc.edge_error(want[int(row[0]) - 1], call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        assert row[1] == pytest.approx(sm / c, rel=1e-11)
    # the same session given another edge list afterwards: its delta tiles are rebuilt (or dropped), never reused
    with _Env(TOPOLOW_SYMMETRIC="1", TOPOLOW_SYMMETRIC_MIN_N="0"):
        ses = _native.Session(n, dim, precision="f64")
    ses.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
    for edges, fused in ((call, 2), (part, 0), (call, 2)):
        ses.set_edges(edges.edge_i, edges.edge_j, edges.edge_dist, edges.edge_thresh)
        ses.set_positions(call.initial_positions)
        ses.set_profiling(True)
        ses.begin(7, k0, 0.01, 0.01, 1e-12, 10 ** 9, 3, 5, 1)
        ses.run()
        ses.sync()
        assert ses.profile_symmetric()[3] == fused
        for row in ses.check_trace():
            sm, c = orc.edge_error(want[int(row[0]) - 1], edges.edge_i, edges.edge_j, edges.edge_dist, edges.edge_thresh)
            assert row[1] == pytest.approx(sm / c, rel=1e-11)
        ses.set_profiling(False)
    ses.close()


def test_symmetric_sweep_f64_fuzz_through_the_production_entry():
    """Ragged sizes (n not a multiple of 8, 32 or 64; fewer tile-rows than waves), 2..6 coordinates, thresholds, every
    check cadence, the size gate at zero, random labels (the production entry relabels): whole f64 runs with the
    symmetric sweep against the same runs on the row-owner f64 kernel -- same schedule and arithmetic, the sums grouped
    differently, the checks fused (exact through the delta tiles) against separate: positions to 1e-9 of the
    coordinate scale, the same verdict, the final MAE to 1e-10 and equal to the oracle's edge error of the returned
    positions to 1e-11."""
    rng = np.random.default_rng(4711)
    done = 0
    for case in range(10):
        n = int(rng.integers(130, 900))
        dim = int(rng.choice([2, 3, 4, 5, 6]))
        call, _ = pp.random_problem(n, dim, float(rng.choice([0.3, 0.7, 0.9])), seed=2000 + case,
                                    thresholds=float(rng.choice([0.0, 0.2])), n_iter=int(rng.integers(12, 60)),
                                    k0=float(rng.uniform(0.5, 6.0)), cool=0.02, c_rep=0.01, check_freq=int(rng.integers(1, 5)),
                                    window=3, eps=1e-6)
        seed = int(rng.integers(1, 2 ** 62))
        runs = {}
        for sym in ("1", "0"):
            with _Env(TOPOLOW_SYMMETRIC=sym, TOPOLOW_SYMMETRIC_MIN_N="0", TOPOLOW_SYMMETRIC_TWO_STAGE="0"):
                try:
                    runs[sym] = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, schedule="slab",
                                                                    precision="f64")
                except _native.NativeError as e:
                    runs[sym] = str(e)
        a, b = runs["1"], runs["0"]
        if isinstance(a, str) or isinstance(b, str):
            assert a == b
            continue
        scale = max(1.0, float(np.abs(b.positions).max()))
        assert np.abs(a.positions - b.positions).max() <= 1e-9 * scale, (n, dim)
        assert (a.converged, a.iterations) == (b.converged, b.iterations)
        assert a.final_mae == pytest.approx(b.final_mae, rel=1e-10, abs=1e-14)
        sm, cnt = orc.edge_error(a.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        if cnt > 0:
            assert a.final_mae == pytest.approx(sm / cnt, rel=1e-11, abs=1e-14)
        done += 1
    assert done >= 7


# ----------------------------------------------------------------------------------------
# 2-, 4- and 8-stage iterations as symmetric sweeps that split the PAIRS (stage st: slab a meets slab (st - a) mod S)
# ----------------------------------------------------------------------------------------
def _multi_stage_model(call_r, iters, k0, cooling, c_rep, seed, stages):
    """The CPU model of that schedule in f64: in stage st every point of slab c moves by the sum of its own halves of its
    pairs with the points of slab (st - c) mod S, from the positions the previous stage left (slab_model.stage gives
    every row its move from a range of columns)."""
    n = call_r.initial_positions.shape[0]
    bounds = _native.symm_stage_bounds(n, stages)
    out, pos, k = [], call_r.initial_positions, k0
    args = (call_r.dissimilarity_matrix, call_r.threshold_matrix, call_r.degrees)
    for it in range(iters):
        for st in _native.symm_stage_order(seed, it, stages):
            against = [slab_model.stage(pos, *args, [[bounds[q], bounds[q + 1]]], k, c_rep, "f64") for q in range(stages)]
            new = np.empty_like(pos)
            for c in range(stages):
                new[bounds[c]:bounds[c + 1]] = against[(st - c) % stages][bounds[c]:bounds[c + 1]]
            pos = new
        out.append(pos)
        k *= 1.0 - cooling
    return out


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("n,dim,thr,stages", [(300, 2, 0.0, 2), (1000, 5, 0.15, 2), (2973, 3, 0.0, 2), (2973, 6, 0.15, 2),
                                              (1000, 5, 0.15, 4), (2973, 3, 0.0, 8), (2050, 4, 0.15, 8)])
def test_multi_stage_iterations_as_symmetric_sweeps_against_the_model(n, dim, thr, stages, precision):
    """Four S-stage iterations (slab_stages = S) as S symmetric sweeps against the CPU model of exactly that schedule:
    f64 to 1e-12 of the displacement scale per stage, fp32 in the bands of the one-stage test; the separate checks
    against the oracle's edge error of the model's positions; and TOPOLOW_SYMMETRIC_TWO_STAGE=0 gives the row-owner
    stages back (a different, equally valid S-stage schedule: not compared)."""
    k0, cooling, c_rep, seed, iters = 2.0 * stages, 0.01, 0.01, 5, 4
    call, _ = pp.random_problem(n, dim, 0.7 if n > 500 else 0.3, seed=300 + n % 50 + dim, n_iter=iters, k0=k0)
    call = _with_thresholds(call, thr)
    call_r = dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call))
    want = _multi_stage_model(call_r, iters, k0, cooling, c_rep, seed, stages)
    scale = np.abs(want[-1] - call.initial_positions).max()

    def run(symmetric_stages):
        with _Env(TOPOLOW_SYMMETRIC="1", TOPOLOW_SYMMETRIC_MIN_N="0", TOPOLOW_SYMMETRIC_TWO_STAGE=symmetric_stages,
                  TOPOLOW_SYMMETRIC_STAGE_MIN_TILES="0"):      # (production: a stage must give a resident wave ~3 tiles)
            s = _native.Session(n, dim, precision=precision)
        s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
        s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        s.set_positions(call.initial_positions)
        s.begin(iters, k0, cooling, c_rep, 1e-12, 10 ** 9, 2, seed, stages)
        s.run()
        s.sync()
        out = s.get_positions(), s.check_trace(), s.stage_launches
        s.close()
        return out
    got, trace, launches = run("1")
    assert launches == iters * stages
    err = np.abs(got - want[-1])
    if precision == "f64":
        assert err.max() <= 1e-12 * scale * iters * stages, err.max() / scale
    else:
        assert err.mean() <= 5e-5 * scale and err.max() <= 5e-3 * scale, (err.mean() / scale, err.max() / scale)
    assert [int(t) for t in trace[:, 0]] == [2, 4]
    for row in trace:
        sm, c = orc.edge_error(want[int(row[0]) - 1], call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        assert row[1] == pytest.approx(sm / c, rel=1e-11 if precision == "f64" else 2e-5)
    other, _, _ = run("0")
    assert np.abs(other - got).max() > 1e-6 * scale        # the row-owner stages are another schedule
