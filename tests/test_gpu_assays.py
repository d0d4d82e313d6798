"""BASELINE configs 2 and 5 on the GPU (run with -m gpu):
  config 2 -- Smith-2004 H3N2 HI panel (285 points, 911 censored titers), published parameters
              (inst/examples/methods-comparison-h3n2-hiv-denv.Rmd:312-316 of the reference), ndim 5
              (BASELINE) and 4 (published): device GS kernel == oracle replay bit for bit, and the
              MAE level agrees with the reference-order oracle.
  config 5 -- HIV neutralisation panel (335 points, 1249 censored), k-fold CV evaluator
              (`likelihood_function`) as ONE batched launch; the pooled hold-out MAE is compared with
              the numbers the reference publishes for exactly this pipeline
              (inst/examples/comparison_results/error_summary.csv: H3N2 0.799, HIV 1.315).
The matrices come from tests/golden/*.csv (built from the reference's data files by
tests/golden/make_assay_fixtures.py)."""
import csv
import os

import numpy as np
import pytest

from oracle import topolow_oracle as orc
from tests.conftest import layout_call_args
from topolow_amd import _native, antigenic, core, cv

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
H3N2 = dict(k0=14.76214, cooling_rate=0.03641074, c_repulsion=0.002943064)
HIV = dict(N=2, k0=3.550036, cooling_rate=0.04130713, c_repulsion=0.0007038619)


def h3n2_matrix():
    rows = list(csv.DictReader(open(os.path.join(GOLD, "h3n2_distances.csv"))))
    return antigenic.titers_list_to_matrix(rows, "virusStrain", "virusYear", "serumStrain", "serumYear",
                                           "distance", sort=True)


def hiv_matrix():
    rows = list(csv.DictReader(open(os.path.join(GOLD, "hiv_distances.csv"))))
    return antigenic.titers_list_to_matrix(rows, "Virus", "virusYear", "Antibody", None, "distance", sort=True)


@pytest.mark.parametrize("ndim", [5, 4])
def test_config2_h3n2_gs_equals_oracle_replay(ndim):
    m = h3n2_matrix()
    call = core.prepare_layout_call(m, ndim, 1000, H3N2["k0"], H3N2["cooling_rate"], H3N2["c_repulsion"],
                                    1e-4, 5, None, False, 3, False, np.random.default_rng(7))
    n = call.initial_positions.shape[0]
    assert n == 285
    seed = 2004
    got = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed)   # AUTO -> gs, f64
    assert got.info["schedule"] == "gs" and got.info["precision"] == "f64"

    def order_fn(it, arr):
        arr[:] = _native.gs_pair_order(n, seed, it)
    ref = orc.optimize_layout_exact(*layout_call_args(call), order_mode=orc.ORDER_SUPPLIED, order_fn=order_fn)
    assert got.converged == ref.converged and got.iterations == ref.iterations
    assert np.abs(got.positions - ref.positions).max() <= 1e-10
    assert got.final_mae == pytest.approx(ref.final_mae, rel=1e-10)
    # (the statistics against the reference-order oracle -- 20 seeds, contract band -- are in
    #  tests/test_gpu_contract.py::test_exact_gauss_seidel_in_tournament_order_meets_the_contract)


def test_config2_slab_schedule_on_h3n2_when_forced():
    """The slab schedule on a small, 91 %-sparse, heavily censored panel: same error level."""
    m = h3n2_matrix()
    call = core.prepare_layout_call(m, 5, 1000, H3N2["k0"], H3N2["cooling_rate"], H3N2["c_repulsion"],
                                    1e-4, 5, None, False, 3, False, np.random.default_rng(7))
    from tests import parity_problems as pp
    ref = pp.oracle_distribution("h3n2_ndim5")      # same problem definition (tests/parity_problems.py: h3n2_call)
    slab = [_native.optimize_layout_exact_arrays(*layout_call_args(call), seed=s, schedule="slab").final_mae
            for s in range(20)]
    # 285 points in slabs of 72 columns, 9 % measured: not the regime the slab schedule is built for (AUTO takes
    # the exact kernel here); stated band 5 % of the oracle mean
    assert abs(np.mean(slab) - ref["mean_final_mae"]) <= 0.05 * ref["mean_final_mae"], (np.mean(slab), ref["mean_final_mae"])


def test_config5_batched_cv_matches_published_holdout_mae():
    """20-fold CV at the parameters the reference publishes, one batched launch per panel, against the numbers the
    reference holds: its 20 per-fold errors (comparison_results/fold_stats.csv: H3N2 0.828 +- 0.064, HIV 1.329 +-
    0.096) -- |difference of the fold means| <= 3 standard errors of the difference -- and the pooled figures of
    error_summary.csv / BASELINE.md (0.799 / 1.315, computed there after dropping error outliers beyond 3.5 MAD,
    which lowers them): within 7 %.  Measured: H3N2 0.838, HIV 1.296."""
    from tests import parity_problems as pp
    rng = np.random.default_rng(11)
    for m, params, ds, published in ((h3n2_matrix(), dict(N=4, **H3N2), "H3N2", 0.799), (hiv_matrix(), HIV, "HIV", 1.315)):
        res, secs, n_emb = cv.likelihood_sweep(m, [params], 500, 1e-4, folds=20, rng=rng)
        r = res[0]
        assert n_emb == 20 and r["pct_converged"] >= 50
        f, ref = np.array(r["fold_mae"]), pp.ref_fold_stats(ds)
        assert f.size == 20 and ref.size == 20
        se = np.hypot(f.std(ddof=1) / np.sqrt(20), ref.std(ddof=1) / np.sqrt(20))
        assert abs(f.mean() - ref.mean()) <= 3 * se, (ds, f.mean(), ref.mean(), se)
        assert abs(r["Holdout_MAE"] / published - 1) <= 0.07, (ds, r["Holdout_MAE"], published)
        assert np.isfinite(r["NLL"]) and 0 < r["mean_iter"] <= 500


def test_config5_sweep_many_parameter_sets_in_one_launch():
    """A slice of an Euclidify-style sweep: 12 parameter sets x 5 folds = 60 embeddings, mixed
    ndim, one call.  Every set must come back finite and the batch must agree with running one of
    its members alone (same seed)."""
    rng = np.random.default_rng(3)
    hv = hiv_matrix()
    sets = [dict(N=int(rng.integers(2, 7)), k0=float(rng.uniform(1, 12)),
                 cooling_rate=float(rng.uniform(0.01, 0.06)), c_repulsion=float(rng.uniform(1e-4, 1e-2)))
            for _ in range(12)]
    res, secs, n_emb = cv.likelihood_sweep(hv, sets, 300, 1e-4, folds=5, rng=rng)
    assert n_emb == 60 and len(res) == 12 and secs > 0
    assert all(np.isfinite(r["Holdout_MAE"]) for r in res)
    # batch member == the same problem run alone
    call = core.prepare_layout_call(hv, 3, 200, 4.0, 0.03, 0.001, 1e-4, 5, None, False, 3, False,
                                    np.random.default_rng(5))
    alone = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=99, schedule="gs")
    batch, _ = _native.optimize_layout_exact_batch([call, call], seeds=[99, 100])
    assert np.array_equal(batch[0].positions, alone.positions) and batch[0].iterations == alone.iterations
    assert not np.array_equal(batch[1].positions, alone.positions)


def test_sparse_cv_path_equals_the_dense_sequence(monkeypatch):
    """The sweep that never builds an n x n array (cell-list folds, edge-list-defined problems,
    holdout scored on the device) must give what the reference's own sequence gives on the same
    folds: masked matrix -> prepare_layout_call -> est_distances -> error_calculator_comparison."""
    import dataclasses
    hv = hiv_matrix()
    sets = [HIV, dict(N=3, k0=5.0, cooling_rate=0.02, c_repulsion=0.005),
            dict(N=2, k0=-1.0, cooling_rate=0.02, c_repulsion=0.005)]       # invalid k0 -> NA row
    a, _, na = cv.likelihood_sweep(hv, sets, 300, 1e-4, folds=5, rng=np.random.default_rng(9), path="dense")
    b, _, nb = cv.likelihood_sweep(hv, sets, 300, 1e-4, folds=5, rng=np.random.default_rng(9), path="sparse")
    assert na == nb == 10
    for x, y in zip(a[:2], b[:2]):
        assert y["Holdout_MAE"] == pytest.approx(x["Holdout_MAE"], rel=1e-12)
        assert y["NLL"] == pytest.approx(x["NLL"], rel=1e-12)
        assert y["mean_iter"] == x["mean_iter"] and y["pct_converged"] == x["pct_converged"]
    assert np.isnan(a[2]["Holdout_MAE"]) and np.isnan(b[2]["Holdout_MAE"])
    # ... and the sweep as ONE library call (topolow_cv_sweep, the default) equals the sweep that builds a call object per
    # fold here: same draws from the stream, same seeds, the same numbers to the last bit, the stream left in the same place
    r1, r2 = np.random.default_rng(9), np.random.default_rng(9)
    c, _, nc = cv.likelihood_sweep(hv, sets, 300, 1e-4, folds=5, rng=r1, path="sparse-calls")
    d, _, nd = cv.likelihood_sweep(hv, sets, 300, 1e-4, folds=5, rng=r2, path="sparse")
    assert nc == nd == 10 and r1.uniform() == r2.uniform()
    for x, y in zip(c[:2], d[:2]):
        assert x == y
    # the edge list standing for the matrix, on the dense-matrix kernel too (matrix rebuilt from it)
    call = core.prepare_layout_call(hv, 3, 120, 4.0, 0.03, 0.001, 1e-4, 5, None, False, 3, False,
                                    np.random.default_rng(5))
    bare = dataclasses.replace(call, dissimilarity_matrix=None, threshold_matrix=None)
    hold = (np.array([0, 5, 7], dtype=np.int32), np.array([3, 1, 7], dtype=np.int32), np.array([1.0, 2.5, 0.0]))
    for dense_kernel in ("", "1"):
        if dense_kernel:
            monkeypatch.setenv("TOPOLOW_GS_DENSE", "1")
        full, _ = _native.optimize_layout_exact_batch([call], seeds=[3])
        lean, _ = _native.optimize_layout_exact_batch([bare], seeds=[3], holdouts=[hold])
        assert np.array_equal(full[0].positions, lean[0].positions) and full[0].final_mae == lean[0].final_mae
        p = lean[0].positions
        want = sum(abs(t - np.linalg.norm(p[i] - p[j])) for i, j, t in zip(*hold))
        assert lean[0].info["holdout_count"] == 3
        assert lean[0].info["holdout_sum_abs"] == pytest.approx(want, rel=1e-12)
    # a malformed stand-in list is refused
    bad = dataclasses.replace(bare, edge_i=bare.edge_j, edge_j=bare.edge_i)
    with pytest.raises(_native.NativeError):
        _native.optimize_layout_exact_batch([bad], seeds=[3])
