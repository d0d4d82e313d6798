"""GPU parity tests (run with `-m gpu` on an MI355X): every test calls the HIP path through
the C ABI (libtopolow_relax.so via topolow_amd._native) and checks it against the CPU oracle
(reference semantics) or, for the slab schedule, stage-for-stage against the CPU slab model
and statistically against the oracle.

Tolerances (stated per test):
  * GS schedule, f64: the kernel performs the oracle's operations in an equivalent order ->
    positions equal to 1e-12 absolute (bit-for-bit in practice), identical controller fields.
  * slab stages, f64: vs the model <= 1e-9 of the largest displacement after 6 iterations
    (only the summation order differs; 1e-6 in one dimension where near-coincident points
    make the repulsion term stiff).
  * slab stages, f32: mean |diff| <= 5e-5, max <= 5e-3 of the largest displacement after 9
    iterations (fp32 positions, v_rcp_f32 / v_sqrt_f32, 4-ulp target rounding).
  * end-to-end statistics against the reference schedule (oracle, std::shuffle order): the contract
    mean +- max(3 sd_ref, 1 %) over >= 20 oracle seeds lives in tests/test_gpu_contract.py.
"""
import os

import numpy as np
import pytest

import oracle
from oracle import topolow_oracle as orc
from tests.conftest import layout_call_args
from tests.helpers import numpy_pdist, quickstart_matrix
from tests.models import slab_model
from tests import parity_problems as pp
from topolow_amd import _native, core, synthetic

pytestmark = pytest.mark.gpu


from tests.parity_problems import random_problem as _random_problem  # noqa: E402


def _oracle_with_gs_order(call, seed, arith="f64"):
    n = call.initial_positions.shape[0]

    def order_fn(it, arr):
        arr[:] = _native.gs_pair_order(n, seed, it)

    return orc.optimize_layout_exact(*layout_call_args(call), order_mode=orc.ORDER_SUPPLIED,
                                     order_fn=order_fn, arith=arith)


# ----------------------------------------------------------------------------------------
# GS schedule: exact parity with the oracle replaying the same pair order
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,dim,missing,thr", [(2, 2, 0.0, 0.0), (3, 1, 0.0, 0.0), (5, 2, 0.2, 0.0),
                                               (8, 3, 0.3, 0.3), (33, 5, 0.5, 0.2), (64, 4, 0.7, 0.1),
                                               (101, 5, 0.7, 0.0), (257, 3, 0.9, 0.15), (300, 10, 0.6, 0.0)])
def test_gs_f64_matches_oracle_same_order(n, dim, missing, thr):
    call, _ = _random_problem(n, dim, missing if n > 4 else 0.0, seed=n, thresholds=thr, n_iter=25)
    seed = 1000 + n
    got = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, schedule="gs",
                                               precision="f64")
    ref = _oracle_with_gs_order(call, seed)
    assert got.info["schedule"] == "gs" and got.info["precision"] == "f64"
    assert np.abs(got.positions - ref.positions).max() <= 1e-12
    assert got.converged == ref.converged and got.iterations == ref.iterations
    assert got.final_mae == pytest.approx(ref.final_mae, rel=1e-12)
    assert got.final_k == ref.final_k
    assert got.info["iterations_run"] == ref.iters_run


def test_gs_f32_close_to_oracle_f32_same_order():
    """fp32 positions: compared with the oracle's own float32 arithmetic, same order."""
    call, _ = _random_problem(120, 5, 0.6, seed=7, n_iter=15)
    got = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=5, schedule="gs", precision="f32")
    ref = _oracle_with_gs_order(call, 5, arith="f32")
    scale = np.abs(ref.positions).max()
    assert np.abs(got.positions - ref.positions).max() <= 1e-4 * scale
    assert got.final_mae == pytest.approx(ref.final_mae, rel=1e-4)


def test_gs_convergence_controller_end_to_end_quickstart():
    """README.md:54-87 of the reference on the GPU: converges, restores the best snapshot,
    V1-V2 ~ 2.83; and equals the oracle replay exactly."""
    m = quickstart_matrix()
    call = core.prepare_layout_call(m, 2, 1000, 5.0, 0.03, 0.7, 1e-4, 5, None, False, 3, False,
                                    np.random.default_rng(2))
    vals = []
    for seed in range(6):
        got = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, schedule="gs")
        ref = _oracle_with_gs_order(call, seed)
        assert got.converged and got.iterations == ref.iterations < 1000
        assert np.abs(got.positions - ref.positions).max() <= 1e-12
        est = _native.est_distances(got.positions)
        vals.append(est[call.names.index("V1"), call.names.index("V2")])
    assert 2.6 < np.mean(vals) < 3.1


def test_gs_nonfinite_and_small_n_errors():
    call, _ = _random_problem(20, 2, 0.3, seed=3, n_iter=40)
    bad = call.initial_positions.copy()
    bad[4, 0] = np.nan
    args = list(layout_call_args(call)); args[0] = bad
    with pytest.raises(_native.NativeError, match=r"Numerical instability at iteration 10\. Reduce k0 or c_repulsion\."):
        _native.optimize_layout_exact_arrays(*args, seed=1, schedule="gs")
    with pytest.raises(orc.OracleError, match=r"Numerical instability at iteration 10\."):
        orc.optimize_layout_exact(*args, seed=1)


# ----------------------------------------------------------------------------------------
# slab schedule: stage-for-stage against the CPU model
# ----------------------------------------------------------------------------------------
def _model_run(call, seed, stages_fixed, n_iter, arith):
    n = call.initial_positions.shape[0]
    plans, counts = [], []
    k = call.k0
    for it in range(n_iter):
        st = stages_fixed if stages_fixed else _native.slab_stages_at(it, k, call.initial_positions.shape[1])
        p = _native.slab_plan(n, st, seed, it)
        plans.append(p); counts.append(len(p))
        k *= 1.0 - call.cooling_rate
    mx = max(counts)
    plan = np.zeros((n_iter, mx, 4), np.int32)
    for it, p in enumerate(plans):
        plan[it, : len(p)] = p
    return slab_model.run(call.initial_positions, call.dissimilarity_matrix, call.threshold_matrix,
                          call.degrees, plan, np.array(counts, np.int32), call.k0, call.cooling_rate,
                          call.c_repulsion, arith)


def _decode_rounded(call):
    """The slab path keeps targets as 4-ulp-rounded fp32; give the model the same values."""
    D = call.dissimilarity_matrix.copy()
    fin = np.isfinite(D)
    u = D[fin].astype(np.float32).view(np.uint32)
    mag = ((u & np.uint32(0x7FFFFFFF)) + np.uint32(2)) & np.uint32(0xFFFFFFFC)
    out = ((u & np.uint32(0x80000000)) | mag).view(np.float32).astype(np.float64)
    sample = D[fin][:64]
    assert all(_native.decode_target(_native.encode_target(v, 0))[0] == o for v, o in zip(sample, out[:64]))
    D[fin] = out
    return D


@pytest.mark.parametrize("n,dim,missing,thr,stages", [(256, 5, 0.7, 0.0, 4), (301, 3, 0.5, 0.2, 4),
                                                      (1030, 5, 0.7, 0.1, 8), (2050, 2, 0.9, 0.0, 16),
                                                      (777, 10, 0.6, 0.0, 4), (513, 1, 0.3, 0.0, 4)])
def test_slab_f64_matches_model(n, dim, missing, thr, stages):
    call, _ = _random_problem(n, dim, missing, seed=n + 1, thresholds=thr, n_iter=6, check_freq=3)
    import dataclasses
    call_r = dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call))
    seed = 42
    s = _native.Session(n, dim, precision="f64")
    s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
    s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    s.set_positions(call.initial_positions)
    s.begin(6, call.k0, call.cooling_rate, call.c_repulsion, 1e-12, 1000, 3, seed, stages)
    s.run()
    got = s.get_positions()
    want, _k = _model_run(call_r, seed, stages, 6, "f64")
    scale = np.abs(want - call.initial_positions).max()
    # only the summation order differs; the stiff 1/(r+0.01)^3 term amplifies that noise where
    # points nearly coincide, which in one dimension they routinely do
    assert np.abs(got - want).max() <= (1e-6 if dim == 1 else 1e-9) * max(scale, 1.0)
    res = s.finish()
    sm, cnt = orc.edge_error(got, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    if res.iterations == 6:
        assert res.final_mae == pytest.approx(sm / cnt, rel=1e-12)
    s.close()


# (the fp32 instance that carries the threshold classification is built for fewer waves per SIMD from
#  ndim 7 and again from ndim 9; thr = 0 takes the threshold-free instance; no 1-D case: points pass
#  through each other there and fp32 rounding is amplified without bound, as the fuzz tests note)
@pytest.mark.parametrize("n,dim,stages,thr", [(1024, 5, 4, 0.1), (1500, 3, 0, 0.1), (999, 2, 8, 0.1),
                                              (600, 7, 4, 0.1), (520, 9, 4, 0.15), (500, 10, 4, 0.1),
                                              (610, 8, 4, 0.0), (530, 10, 4, 0.0),
                                              (640, 4, 4, 0.0), (560, 6, 4, 0.1)])
def test_slab_f32_close_to_model(n, dim, stages, thr):
    call, _ = _random_problem(n, dim, 0.7, seed=n + 5, thresholds=thr, n_iter=9, k0=12.0 if stages == 0 else 4.0)
    import dataclasses
    call_r = dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call))
    seed = 9
    s = _native.Session(n, dim, precision="f32")
    s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
    s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    s.set_positions(call.initial_positions)
    s.begin(9, call.k0, call.cooling_rate, call.c_repulsion, 1e-12, 1000, 3, seed, stages)
    s.run()
    got = s.get_positions()
    want, _k = _model_run(call_r, seed, stages, 9, "f64")
    scale = np.abs(want - call.initial_positions).max()
    err = np.abs(got - want)
    assert err.mean() <= 5e-5 * scale and err.max() <= 5e-3 * scale
    s.close()


def test_coo_load_equals_dense_load():
    call, _ = _random_problem(700, 3, 0.8, seed=31, thresholds=0.2, n_iter=4)
    outs = []
    for mode in ("dense", "coo"):
        s = _native.Session(700, 3, precision="f64")
        if mode == "dense":
            s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
        else:
            s.load_coo(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.degrees)
        s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        s.set_positions(call.initial_positions)
        s.begin(4, 3.0, 0.05, 0.02, 1e-4, 5, 2, 77, 4)
        s.run()
        outs.append(s.get_positions())
        s.close()
    assert np.array_equal(outs[0], outs[1])


def test_checks_beside_next_iteration_change_nothing(monkeypatch):
    """The convergence check runs on a second stream beside the next iteration's stages (single
    GPU, slab schedule); TOPOLOW_SERIAL_CHECKS=1 keeps it on the main stream.  Same verdicts, same
    snapshot: converged runs (the stop arrives while later stages are already queued or running),
    check_freq 1 (a check every iteration) and an exhausted run must all be identical."""
    call, _ = _random_problem(700, 3, 0.6, seed=31, thresholds=0.1, n_iter=400, k0=6.0, cool=0.03, c_rep=0.01)
    outs = {}
    for serial in ("1", "0"):
        monkeypatch.setenv("TOPOLOW_SERIAL_CHECKS", serial)
        rows = []
        for n_iter, window, freq in ((400, 3, 3), (400, 2, 1), (7, 5, 2)):
            s = _native.Session(700, 3, precision="f32")
            s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
            s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
            assert s.uses_dense_mae     # the MAE comes from the encoded block in every run
            s.set_positions(call.initial_positions)
            s.begin(n_iter, 6.0, 0.03, 0.01, 1e-3, window, freq, 5, 0)
            s.run()
            r = s.finish()
            rows.append((r.positions.copy(), r.converged, r.iterations, r.final_mae, r.final_k))
            s.close()
        outs[serial] = rows
    assert outs["1"][0][1] and outs["1"][1][1] and not outs["1"][2][1]     # two stops, one exhausted run
    for a, b in zip(outs["1"], outs["0"]):
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]


def test_check_fused_into_the_next_iterations_sweep(monkeypatch):
    """Once k <= 3 an iteration is ONE stage, and that stage meets every pair from the positions the previous
    iteration left -- the positions the previous iteration's convergence check measures.  The kernel then reduces
    the check's MAE on its way (ERR launch) and the separate pass over the block is dropped;
    TOPOLOW_FUSE_CHECKS=0 keeps the separate pass.  Same trajectory and verdicts; the MAE agrees to fp32 rounding
    (it is the oracle's edge MAE of the returned positions either way).  Thresholds, a check every iteration, a
    run that exhausts its iterations (its last check has no next iteration: separate pass) and the caller-paced
    form (enqueue / sync between checks: a pending check is flushed) included."""
    call, _ = _random_problem(1200, 3, 0.6, seed=31, thresholds=0.1, n_iter=400, k0=2.4, cool=0.02, c_rep=0.01)
    outs = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("TOPOLOW_FUSE_CHECKS", fuse)
        rows = []
        for n_iter, window, freq, paced in ((400, 3, 3, False), (400, 2, 1, False), (41, 50, 2, False), (400, 3, 3, True)):
            s = _native.Session(1200, 3, precision="f32")
            s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
            s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
            assert s.uses_dense_mae
            s.set_positions(call.initial_positions)
            s.begin(n_iter, 2.4, 0.02, 0.01, 1e-3, window, freq, 5, 0)
            if paced:
                while s.enqueue(3) > 0:
                    s.sync()
            else:
                s.run()
            r = s.finish()
            trace = s.check_trace()
            rows.append((r.positions.copy(), r.converged, r.iterations, r.final_mae, r.final_k, trace))
            s.close()
        outs[fuse] = rows
    assert outs["1"][0][1] and outs["1"][1][1] and not outs["1"][2][1]     # two stops, one exhausted run
    for a, b in zip(outs["0"], outs["1"]):
        assert np.array_equal(a[0], b[0]) and a[1:3] == b[1:3] and a[4] == b[4]
        assert a[3] == pytest.approx(b[3], rel=2e-6)
        assert a[5].shape == b[5].shape and np.array_equal(a[5][:, 0], b[5][:, 0])
        assert np.allclose(a[5][:, 1], b[5][:, 1], rtol=2e-6, atol=0)
    got = outs["1"][0]
    sm, cnt = orc.edge_error(got[0], call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    assert got[3] == pytest.approx(sm / cnt, rel=2e-5)


# ----------------------------------------------------------------------------------------
# slab schedule end-to-end vs the reference schedule (statistical), plus post metrics
# ----------------------------------------------------------------------------------------
def test_slab_post_metrics_at_the_reference_level():
    """`mae` of the returned object (R/core.R:470-481: all non-NA cells, both triangles, diagonal) for a slab
    run against an oracle run of the same problem; the final-MAE statistics are in test_gpu_contract.py."""
    call, truth = pp.build("syn1500_h3n2params")
    ref = orc.optimize_layout_exact(*layout_call_args(call), seed=0)
    _, mae_ref = oracle.post_metrics(ref.positions, truth)
    for seed in range(3):
        got = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, schedule="slab")
        assert got.info["schedule"] == "slab" and got.info["precision"] == "f32" and got.converged
        _, mae_post = oracle.post_metrics(got.positions, truth)
        assert mae_post == pytest.approx(mae_ref, rel=0.08)
        sm, cnt = orc.edge_error(got.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        assert got.final_mae == pytest.approx(sm / cnt, rel=2e-5)


def test_relabelled_session_equals_plain_session_on_the_permuted_problem():
    """topolow_session_set_relabel stores the points in a random order; every host-facing entry point keeps
    speaking the caller's labels.  A relabelled session on the problem as given must therefore equal, bit for
    bit, a plain session on the problem permuted by hand -- dense and edge-list loading, thresholds, both
    precisions, the MAE from the block and from the edge list."""
    n, dim = 777, 3
    call, _ = _random_problem(n, dim, 0.8, seed=31, thresholds=0.2, n_iter=30, k0=6.0, cool=0.03, c_rep=0.01)
    for precision, load in (("f32", "dense"), ("f32", "coo"), ("f64", "dense")):
        a = _native.Session(n, dim, precision=precision)
        a.set_relabel(12345)
        perm = a.labels().astype(np.int64)          # session label -> caller's label
        assert sorted(perm.tolist()) == list(range(n)) and not np.array_equal(perm, np.arange(n))
        inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
        b = _native.Session(n, dim, precision=precision)
        Dp = call.dissimilarity_matrix[np.ix_(perm, perm)]
        Tp = call.threshold_matrix[np.ix_(perm, perm)]
        ei, ej = inv[call.edge_i], inv[call.edge_j]
        lo, hi = np.minimum(ei, ej).astype(np.int32), np.maximum(ei, ej).astype(np.int32)
        if load == "dense":
            a.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
            b.load_dense(Dp, Tp, call.degrees[perm])
        else:
            a.load_coo(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.degrees)
            b.load_coo(lo, hi, call.edge_dist, call.edge_thresh, call.degrees[perm])
        a.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        b.set_edges(lo, hi, call.edge_dist, call.edge_thresh)
        assert a.uses_dense_mae == b.uses_dense_mae == (precision == "f32")
        outs = []
        for s_, init in ((a, call.initial_positions), (b, call.initial_positions[perm])):
            s_.set_positions(init)
            s_.begin(30, call.k0, call.cooling_rate, call.c_repulsion, 1e-4, 5, 3, 77, 0)
            s_.run()
            outs.append(s_.finish())
            s_.close()
        ra, rb = outs
        assert np.array_equal(ra.positions[perm], rb.positions)
        assert (ra.converged, ra.iterations, ra.final_mae, ra.final_k) == (rb.converged, rb.iterations, rb.final_mae, rb.final_k)


def test_est_distances_matches_numpy():
    rng = np.random.default_rng(0)
    for n, d in ((5, 2), (300, 5), (1111, 3)):
        p = rng.normal(size=(n, d))
        est = _native.est_distances(p)
        assert np.allclose(est, numpy_pdist(p), rtol=1e-14, atol=1e-14)
        assert np.array_equal(est, est.T) and np.all(np.diag(est) == 0)
    # row blocks of the same matrix (large problems stream them instead of holding n x n float64)
    p = rng.normal(size=(70001, 3))
    blk = _native.est_distances_rows(p, 69990, 70001)          # more rows than grid.y allows: rows ride in grid.x
    want = np.sqrt(((p[69990:, None, :] - p[None, :, :]) ** 2).sum(-1))
    assert blk.shape == (11, 70001) and np.allclose(blk, want, rtol=1e-14, atol=1e-14)


def test_euclidean_embedding_drop_in_on_gpu(tmp_path):
    """The public entry point end to end on the device, with the reference's own test cases
    (tests/testthat/test-core.R:90-139, test-edge-cases.R:66-82, test-deprecated.R:4-25)."""
    import topolow_amd
    topolow_amd.set_seed(123)
    m = np.array([["0", ">2", "3"], [">2", "0", "4"], ["3", "4", "0"]], dtype=object)
    m[0, 2] = m[2, 0] = None
    r = topolow_amd.euclidean_embedding(m, 2, 10, 1.0, 0.01, 0.01)
    assert np.isfinite(r.est_distances).all() and r.est_distances[0, 2] == r.est_distances[2, 0]
    tri = core.RMatrix(np.array([[0, 1, 2], [1, 0, 1], [2, 1, 0]], float), ["A", "B", "C"])
    r = topolow_amd.euclidean_embedding(tri, ndim=2, mapping_max_iter=10, k0=1.0, cooling_rate=0.01,
                                        c_repulsion=0.01)
    ix = {nm: q for q, nm in enumerate(r.names)}
    d = lambda p, q: r.est_distances[ix[p], ix[q]]
    assert d("A", "C") > d("A", "B") and d("A", "C") < d("A", "B") + d("B", "C")
    assert r.r_class == "topolow" and "MAE:" in str(r)
    sp = np.full((4, 4), np.nan); sp[0, 1] = sp[1, 0] = 5; np.fill_diagonal(sp, 0)
    r = topolow_amd.euclidean_embedding(sp, 2, 50, 1.0, 0.01, 0.1)
    assert np.isfinite(r.positions).all()
    with pytest.warns(DeprecationWarning, match="was deprecated"):
        r = topolow_amd.create_topolow_map(np.array([[0, 2, 3], [2, 0, 4], [3, 4, 0]], float), ndim=2,
                                           mapping_max_iter=10, k0=1.0, cooling_rate=0.001, c_repulsion=0.01)
    assert "est_distances" in r
    vals = []
    for _ in range(7):   # single runs land in a side minimum now and then (so does the oracle)
        r = topolow_amd.euclidean_embedding(quickstart_matrix(), 2, 1000, 5, 0.03, 0.7,
                                            write_positions_to_csv=True, output_dir=str(tmp_path / "o"))
        vals.append(r.est_distances[r.names.index("V1"), r.names.index("V2")])
    assert (tmp_path / "o" / "Positions_dim_2_k0_5.0000_cooling_0.0300_c_repulsion_0.7000.csv").exists()
    assert 2.6 < np.median(vals) < 3.1


# ----------------------------------------------------------------------------------------
# full BASELINE size (N = 10 000, ndim 5, 70 % missing): size-independent properties
# ----------------------------------------------------------------------------------------
def test_full_size_properties():
    n, dim = 10000, 5
    prob = synthetic.make_problem(n, latent_dim=dim, missing=0.7, seed=12345)
    init = synthetic.initial_positions(prob.dissimilarity, dim, 12345)
    call = core.prepare_layout_call(prob.dissimilarity, dim, 12, 5.0, 0.01, 0.01, 1e-4, 5, init, False, 3,
                                    True)
    s = _native.Session(n, dim, precision="f32")
    s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
    s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)

    def run(init_pos, seed):
        s.set_positions(init_pos)
        s.begin(12, 5.0, 0.01, 0.01, 1e-4, 5, 3, seed, 0)
        s.run()
        return s.finish()

    a = run(call.initial_positions, 5)
    b = run(call.initial_positions, 5)
    assert np.array_equal(a.positions, b.positions)            # deterministic for a fixed seed
    # (1) reported MAE == oracle's edge MAE of the returned positions
    sm, cnt = orc.edge_error(a.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    assert a.final_mae == pytest.approx(sm / cnt, rel=2e-5) and a.iterations == 12
    # (2) translation equivariance: shifting every start point shifts every result point
    shift = np.array([3.0, -2.0, 1.0, 0.5, -4.0])
    c = run(call.initial_positions + shift, 5)
    assert np.abs((c.positions - shift) - a.positions).max() <= 5e-3
    # (3) the error falls monotonically over these first checks and positions stay finite
    assert np.isfinite(a.positions).all() and a.final_mae < 0.9 * (
        orc.edge_error(call.initial_positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)[0] / cnt)
    # (4) one stage of the slab model on a row sample agrees with the device (spot check)
    s.set_positions(call.initial_positions)
    s.begin(1, 5.0, 0.01, 0.01, 1e-4, 5, 1, 11, 4)
    s.run()
    got = s.get_positions()
    plan = _native.slab_plan(n, 4, 11, 0)
    pos = call.initial_positions
    for rg in plan:
        pos = slab_model.stage(pos, call.dissimilarity_matrix, call.threshold_matrix, call.degrees,
                               [r for r in rg.reshape(2, 2) if r[1] > r[0]], 5.0, 0.01, "f64")
    scale = np.abs(pos - call.initial_positions).max()
    assert np.abs(got - pos).max() <= 3e-4 * scale
    s.close()


# ----------------------------------------------------------------------------------------
# edge cases of the slab path (the reference's edge-case tests, forced onto the large-N kernels)
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [2, 3, 5, 9, 64, 65])
def test_slab_tiny_problems_match_model(n):
    call, _ = _random_problem(n, 2, 0.2 if n > 4 else 0.0, seed=50 + n, n_iter=4)
    s = _native.Session(n, 2, precision="f64")
    s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
    s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    s.set_positions(call.initial_positions)
    s.begin(4, call.k0, call.cooling_rate, call.c_repulsion, 1e-12, 1000, 2, 5, 0)
    s.run()
    got = s.get_positions()
    import dataclasses
    want, _k = _model_run(dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call)), 5, 0, 4, "f64")
    assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    s.close()


def test_slab_nonfinite_guard_and_messages():
    call, _ = _random_problem(600, 3, 0.5, seed=8, n_iter=40)
    bad = call.initial_positions.copy()
    bad[17, 1] = np.inf
    args = list(layout_call_args(call)); args[0] = bad
    with pytest.raises(_native.NativeError, match=r"Numerical instability at iteration 10\. Reduce k0 or c_repulsion\.") as ei:
        _native.optimize_layout_exact_arrays(*args, seed=1, schedule="slab")
    assert ei.value.code == _native.ERR_NONFINITE
    with pytest.raises(_native.NativeError) as ei:
        _native.optimize_layout_exact_arrays(np.zeros((4, 65)), np.full((4, 4), np.inf), np.zeros((4, 4), np.int32),
                                             [0] * 4, [], [], [], [], 5, 1.0, 0.1, 0.1, 1e-4, 5, 3, seed=1,
                                             schedule="slab")
    assert ei.value.code == _native.ERR_UNSUPPORTED


@pytest.mark.parametrize("ndim", [11, 13, 16])
def test_more_than_ten_dimensions_run_zero_padded(ndim):
    """The reference accepts any ndim (src/optimization.cpp:129); the kernels are instantiated for 1..10, 12 and
    16 coordinates, and 11 runs as 12, 13..15 as 16 with the extra coordinates held at exactly zero (a zero
    coordinate adds 0 to every distance and receives 0 of every move).  Exact GS against the oracle replay, the
    slab stages against the CPU model, the batch entry, est_distances."""
    call, _ = _random_problem(90, ndim, 0.5, seed=ndim, thresholds=0.2, n_iter=12)
    seed = 7
    got = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, schedule="gs", precision="f64")
    ref = _oracle_with_gs_order(call, seed)
    assert got.positions.shape == (90, ndim)
    assert np.abs(got.positions - ref.positions).max() <= 1e-12 and got.iterations == ref.iterations
    batch, _secs = _native.optimize_layout_exact_batch([call, call], seeds=[seed, seed + 1])
    assert np.abs(batch[0].positions - ref.positions).max() <= 1e-12
    call2, _ = _random_problem(520, ndim, 0.7, seed=ndim + 1, thresholds=0.1, n_iter=6, k0=4.0)
    import dataclasses
    call_r = dataclasses.replace(call2, dissimilarity_matrix=_decode_rounded(call2))
    s = _native.Session(520, ndim, precision="f64")
    s.load_dense(call2.dissimilarity_matrix, call2.threshold_matrix, call2.degrees)
    s.set_edges(call2.edge_i, call2.edge_j, call2.edge_dist, call2.edge_thresh)
    s.set_positions(call2.initial_positions)
    s.begin(6, call2.k0, call2.cooling_rate, call2.c_repulsion, 1e-12, 1000, 3, 42, 4)
    s.run()
    got = s.get_positions()
    s.close()
    want, _k = _model_run(call_r, 42, 4, 6, "f64")
    assert got.shape == (520, ndim)
    assert np.abs(got - want).max() <= 1e-9 * max(np.abs(want - call2.initial_positions).max(), 1.0)
    f32 = _native.optimize_layout_exact_arrays(*layout_call_args(call2), seed=1, schedule="slab")
    sm, cnt = orc.edge_error(f32.positions, call2.edge_i, call2.edge_j, call2.edge_dist, call2.edge_thresh)
    assert f32.final_mae == pytest.approx(sm / cnt, rel=2e-5)
    assert np.allclose(_native.est_distances(f32.positions), numpy_pdist(f32.positions), rtol=1e-14, atol=1e-14)
    with pytest.raises(_native.NativeError) as ei:
        _native.optimize_layout_exact_arrays(np.zeros((4, 65)), np.full((4, 4), np.inf), np.zeros((4, 4), np.int32),
                                             [0] * 4, [], [], [], [], 5, 1.0, 0.1, 0.1, 1e-4, 5, 3, seed=1,
                                             schedule="slab")
    assert ei.value.code == _native.ERR_UNSUPPORTED


@pytest.mark.parametrize("ndim", [17, 24, 40, 64])
def test_more_than_sixteen_dimensions_run_on_the_plain_stage_kernel(ndim):
    """The reference accepts any ndim (src/optimization.cpp:129).  Beyond 16 coordinates the slab schedule runs on a
    plain form of the stage kernel (csrc/relax_kernels.h: slab_stage_wide_kernel, instantiated for 32 and 64
    coordinates, other counts zero-padded): stage for stage against the CPU model in f64 (1e-9), fp32 close to it, the
    one-shot entry (AUTO takes the slab schedule there whatever n is) with the reported MAE equal to the oracle's edge
    error of the returned positions, thresholds included, row blocks as well; est_distances; the exact GS schedule and
    ndim > 64 are refused."""
    import dataclasses
    n = 310
    call, _ = _random_problem(n, ndim, 0.6, seed=ndim, thresholds=0.15, n_iter=6, k0=4.0)
    call_r = dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call))
    want, _k = _model_run(call_r, 11, 4, 6, "f64")
    scale = np.abs(want - call.initial_positions).max()
    for precision, tol in (("f64", 1e-9), ("f32", 5e-4)):
        s = _native.Session(n, ndim, precision=precision)
        s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
        s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        s.set_positions(call.initial_positions)
        s.begin(6, call.k0, call.cooling_rate, call.c_repulsion, 1e-12, 1000, 3, 11, 4)
        s.run()
        got = s.get_positions()
        assert got.shape == (n, ndim) and np.abs(got - want).max() <= tol * max(scale, 1.0), precision
        res = s.finish()
        sm, cnt = orc.edge_error(res.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        if res.iterations == 6:
            assert res.final_mae == pytest.approx(sm / cnt, rel=1e-12 if precision == "f64" else 2e-5)
        s.close()
    long_call = dataclasses.replace(call, n_iter=300)
    one = _native.optimize_layout_exact_arrays(*layout_call_args(long_call), seed=3)            # AUTO
    assert one.info["schedule"] == "slab" and one.positions.shape == (n, ndim) and np.isfinite(one.positions).all()
    sm, cnt = orc.edge_error(one.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    assert one.final_mae == pytest.approx(sm / cnt, rel=2e-5)
    start = orc.edge_error(call.initial_positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    assert one.final_mae < 0.5 * start[0] / start[1]
    two = _native.optimize_layout_exact_arrays(*layout_call_args(long_call), seed=3, devices=[0, 0])
    assert np.array_equal(two.positions, one.positions) and two.iterations == one.iterations
    est = _native.est_distances(one.positions)
    assert np.abs(est - numpy_pdist(one.positions)).max() <= 1e-12
    with pytest.raises(_native.NativeError, match="schedule gs") as ei:
        _native.optimize_layout_exact_arrays(*layout_call_args(long_call), seed=3, schedule="gs")
    assert ei.value.code == _native.ERR_UNSUPPORTED
    if ndim == 64:
        with pytest.raises(_native.NativeError) as ei:
            _native.Session(n, 65, precision="f32")
        assert ei.value.code == _native.ERR_UNSUPPORTED


def test_slab_no_measurements_only_repulsion():
    """All pairs unmeasured: MAE is 0 by definition (reference :296), points only repel."""
    n = 300
    rng = np.random.default_rng(0)
    pos0 = rng.normal(size=(n, 2))
    D = np.full((n, n), np.inf); T = np.zeros((n, n), np.int32)
    got = _native.optimize_layout_exact_arrays(pos0, D, T, np.zeros(n, np.int32), [], [], [], [], 9, 1.0, 0.1, 0.05,
                                               1e-4, 2, 3, seed=3, schedule="slab")
    ref = orc.optimize_layout_exact(pos0, D, T, np.zeros(n, np.int32), [], [], [], [], 9, 1.0, 0.1, 0.05, 1e-4, 2, 3,
                                    seed=3)
    assert got.final_mae == ref.final_mae == 0.0 and got.converged == ref.converged
    assert got.iterations == ref.iterations == 3       # first check improves (0 < DBL_MAX), next two plateau


# ----------------------------------------------------------------------------------------
# exact Gauss-Seidel across workgroups (tile schedule): replayed pair for pair by the oracle
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,dim,missing,thr", [(64, 2, 0.3, 0.0), (65, 3, 0.5, 0.2), (130, 5, 0.6, 0.1),
                                               (200, 2, 0.8, 0.0), (777, 5, 0.7, 0.15), (1500, 3, 0.9, 0.0)])
def test_tile_gs_f64_matches_oracle_same_order(n, dim, missing, thr):
    import dataclasses
    call, _ = _random_problem(n, dim, missing, seed=300 + n, thresholds=thr, n_iter=7, check_freq=2)
    rounded = dataclasses.replace(call, dissimilarity_matrix=_decode_rounded(call),
                                  edge_dist=np.array([_native.decode_target(_native.encode_target(v, 0))[0]
                                                      for v in call.edge_dist]))
    seed = 77
    s = _native.Session(n, dim, precision="f64")
    s.set_schedule("gs")
    s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
    s.set_edges(rounded.edge_i, rounded.edge_j, rounded.edge_dist, rounded.edge_thresh)
    s.set_positions(call.initial_positions)
    s.begin(7, call.k0, call.cooling_rate, call.c_repulsion, 1e-4, 5, 2, seed, 0)
    s.run()
    got = s.finish()
    s.close()

    def order_fn(it, arr):
        arr[:] = _native.tilegs_pair_order(n, seed, it)
    ref = orc.optimize_layout_exact(*layout_call_args(rounded), order_mode=orc.ORDER_SUPPLIED, order_fn=order_fn)
    assert np.abs(got.positions - ref.positions).max() <= 1e-11
    assert got.iterations == ref.iterations and got.converged == ref.converged
    assert got.final_mae == pytest.approx(ref.final_mae, rel=1e-11)


def test_tile_gs_through_the_call_boundary_and_statistics():
    """schedule="gs" beyond the one-workgroup limit takes the tile schedule; its error level is the
    reference schedule's."""
    n, dim = 2500, 5
    call, _ = _random_problem(n, dim, 0.7, seed=2500, n_iter=400, k0=14.76, cool=0.0364, c_rep=0.00294)
    got = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=1, schedule="gs")
    assert got.info["schedule"] == "gs" and got.info["precision"] == "f64" and got.converged
    slab = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=1, schedule="slab")
    assert abs(got.final_mae - slab.final_mae) <= 0.08 * got.final_mae
    sm, cnt = orc.edge_error(got.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    assert got.final_mae == pytest.approx(sm / cnt, rel=1e-10)


def test_interrupt_callback_and_verbose(capfd):
    """reference :364 (interrupt poll every 50 iterations) and :183-188/:298-301 (verbose progress)."""
    call, _ = _random_problem(1200, 3, 0.7, seed=12, n_iter=400)
    calls = []

    def stop_after_two():
        calls.append(1)
        return len(calls) >= 2
    with pytest.raises(_native.NativeError) as ei:
        _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=1, schedule="slab",
                                             interrupt=stop_after_two)
    assert ei.value.code == _native.ERR_INTERRUPTED and len(calls) == 2
    got = _native.optimize_layout_exact_arrays(*layout_call_args(call), True, seed=1, schedule="slab")
    out = capfd.readouterr().out
    # the reference's lines (:183-188, :298-301, :334-336 / :351-353), from the device's check trace
    assert "Points: 1200, Pairs per iteration: 719400" in out and "Parameters: k0=3, cooling=0.05, c_rep=0.02" in out
    assert "Iter 30/400, MAE=" in out and ", k=" in out and got.iterations > 0
    assert ("Converged (plateau) at iter %d, MAE=" % got.iterations in out or
            "Converged (MAE worsening, best restored) at iter %d, MAE=" % got.iterations in out)
    lines = []      # a caller-supplied sink (what the R shim maps to Rprintf) takes the lines instead of stdout
    again = _native.optimize_layout_exact_arrays(*layout_call_args(call), True, seed=1, schedule="slab",
                                                 **{"print": lines.append})
    assert "".join(lines) == out and capfd.readouterr().out == "" and again.iterations == got.iterations
    # the one-workgroup kernel is a single launch: its abort word reaches it through pinned memory
    small, _ = _random_problem(300, 3, 0.5, seed=4, n_iter=100000, eps=1e-12, window=10 ** 6)
    polls = []
    with pytest.raises(_native.NativeError) as ei:
        _native.optimize_layout_exact_arrays(*layout_call_args(small), seed=1, schedule="gs",
                                             interrupt=lambda: polls.append(1) or len(polls) >= 3)
    assert ei.value.code == _native.ERR_INTERRUPTED and len(polls) == 3


# ----------------------------------------------------------------------------------------
# BASELINE config 4 size (N = 50 000, ndim 3, 90 % missing) on one GPU: the row-sharded driver
# (caller's stream, one stage per call, replicated controller) against the session's own loop
# ----------------------------------------------------------------------------------------
def test_config4_size_sharded_driver_equals_session_loop():
    torch = pytest.importorskip("torch")
    from topolow_amd import sharded
    n, dim, iters = 50000, 3, 6
    backend = sharded.HipBackend(n, dim, 0, n, 0)
    n_edges, scale = sharded.load_synthetic_block(backend, n, 3, 0.9, 12345, 0, 1)
    assert 0.09 * n * (n - 1) / 2 < n_edges < 0.11 * n * (n - 1) / 2
    rng = np.random.Generator(np.random.PCG64(999))
    init = np.zeros((n, dim))
    init[1:] = np.cumsum(rng.uniform(0.0, 2.0 * scale / n, size=(n - 1, dim)), axis=0)
    coll = sharded.Collectives(1)
    a = sharded.relax_sharded(backend, coll, 0, 1, n, init, iters, 5.0, 0.01, 0.01, 1e-4, 10 ** 9, 3, 7, 0)
    b = sharded.relax_sharded(backend, coll, 0, 1, n, init, iters, 5.0, 0.01, 0.01, 1e-4, 10 ** 9, 3, 7, 0)
    assert np.array_equal(a.positions, b.positions) and a.final_mae == b.final_mae     # deterministic
    assert np.isfinite(a.positions).all() and a.iterations == iters
    # the same block through the session's own loop (own stream, checks beside the next iteration)
    s = backend.session
    torch.cuda.synchronize()
    s.set_stream(None, external=False)
    s.set_positions(init)
    s.begin(iters, 5.0, 0.01, 0.01, 1e-4, 10 ** 9, 3, 7, 0)
    s.run()
    r = s.finish()
    assert r.iterations == a.iterations
    assert np.array_equal(r.positions, a.positions)
    assert r.final_mae == pytest.approx(a.final_mae, rel=1e-6)
    # the error has fallen well below that of the start positions
    start = backend.to_device(init, n)
    sm, cnt = s.edge_error(start.data_ptr())
    assert cnt == n_edges and a.final_mae < 0.5 * sm / cnt
    s.close()


def test_full_size_with_thresholds_config_3b():
    """BASELINE config 3b: config 3 with 10 % of the measured pairs censored (">" at the 90th
    percentile).  The threshold-carrying kernel instance at full size: deterministic, its reported
    MAE is the oracle's edge error of the returned positions (threshold rule included), and one stage
    agrees with the slab model."""
    n, dim = 10000, 5
    prob = synthetic.make_problem(n, latent_dim=dim, missing=0.7, seed=12345)
    D = prob.dissimilarity
    iu, ju = np.triu_indices(n, 1)
    vals = D[iu, ju]
    rng = np.random.default_rng(1)
    sel = ~np.isnan(vals) & (rng.random(iu.size) < 0.1)
    q90 = np.nanquantile(vals, 0.9)
    m = core.CodedMatrix(D.copy(), np.zeros((n, n), dtype=np.int32), None, True)
    m.values[iu[sel], ju[sel]] = np.minimum(vals[sel], q90)
    m.values[ju[sel], iu[sel]] = m.values[iu[sel], ju[sel]]
    m.codes[iu[sel], ju[sel]] = 1
    m.codes[ju[sel], iu[sel]] = 1
    init = synthetic.initial_positions(D, dim, 12345)
    call = core.prepare_layout_call(m, dim, 9, 5.0, 0.01, 0.01, 1e-4, 5, init, False, 3, True)
    assert int((call.edge_thresh == 1).sum()) == int(sel.sum()) > 1_000_000
    s = _native.Session(n, dim, precision="f32")
    s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
    s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    assert s.uses_dense_mae

    def run(iters, stages):
        s.set_positions(call.initial_positions)
        s.begin(iters, 5.0, 0.01, 0.01, 1e-4, 5, 3, 21, stages)
        s.run()
        return s.finish()

    a, b = run(9, 0), run(9, 0)
    assert np.array_equal(a.positions, b.positions) and a.final_mae == b.final_mae
    sm, cnt = orc.edge_error(a.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    assert a.final_mae == pytest.approx(sm / cnt, rel=2e-5) and np.isfinite(a.positions).all()
    # one stage against the model on the thresholded rows
    s.set_positions(call.initial_positions)
    s.begin(1, 5.0, 0.01, 0.01, 1e-4, 5, 1, 11, 4)
    s.run()
    got = s.get_positions()
    pos = call.initial_positions
    for rg in _native.slab_plan(n, 4, 11, 0):
        pos = slab_model.stage(pos, call.dissimilarity_matrix, call.threshold_matrix, call.degrees,
                               [r for r in rg.reshape(2, 2) if r[1] > r[0]], 5.0, 0.01, "f64")
    scale = np.abs(pos - call.initial_positions).max()
    assert np.abs(got - pos).max() <= 3e-4 * scale
    s.close()


def test_repeated_calls_do_not_leak_device_memory():
    """A drop-in is called thousands of times from one R session (Euclidify): every call creates
    and destroys its device state (buffers, two streams, events).  Device memory must come back."""
    torch = pytest.importorskip("torch")
    call, _ = _random_problem(300, 3, 0.6, seed=5, thresholds=0.1, n_iter=6)
    args = layout_call_args(call)
    _native.optimize_layout_exact_arrays(*args, seed=1, schedule="slab")      # warm-up: library, context
    _native.optimize_layout_exact_arrays(*args, seed=1, schedule="gs")
    torch.cuda.synchronize()
    free0, _total = torch.cuda.mem_get_info()
    for q in range(150):
        _native.optimize_layout_exact_arrays(*args, seed=q, schedule="slab")
        _native.optimize_layout_exact_arrays(*args, seed=q, schedule="gs")
    _native.optimize_layout_exact_batch([call] * 40, seeds=list(range(40)))
    torch.cuda.synchronize()
    free1, _total = torch.cuda.mem_get_info()
    assert free0 - free1 < 32 * 1024 * 1024, (free0, free1)
