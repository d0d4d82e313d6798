"""The HIP path against the RESULTS the reference itself ships (run with -m gpu) -- GPU twins of
tests/test_reference_results.py, same fixtures (tests/golden/ref_results/), same bands:

  * BASELINE config 2 / 5 panels: the device (exact GS in f64, and the slab schedule when forced) relaxes the call that
    wrote the reference's coordinate file; its mean final MAE against the MAE of the reference's own coordinates
    (0.59241 H3N2, 1.22454 HIV) and against the oracle's 64-run distribution of the same call;
  * warm start from the reference's coordinates: the device has nothing left to do there either;
  * the CV evaluator (one batched launch per data set) at 48 parameter sets for which the reference's chains hold the
    reference's own Holdout_MAE / NLL (H3N2, HIV and the DENV panel), and against fold_stats.csv's 20 per-fold errors;
  * the DENV panel (10-D): the device against the oracle's distribution of the notebook's call.
"""
import json
import os

import numpy as np
import pytest

from oracle import topolow_oracle as orc
from tests import parity_problems as pp
from tests.conftest import layout_call_args
from tests.test_reference_results import BEST, oracle_cv_entries, ref_edge_mae
from topolow_amd import _native, core, cv

pytestmark = pytest.mark.gpu
SEEDS = 20


def best_params(ds):
    p = pp.ref_chain_optimum(ds) if BEST[ds] == "chain" else dict(pp.HIV_LISTED)
    return {k: p[k] for k in ("N", "k0", "cooling_rate", "c_repulsion")}


@pytest.mark.parametrize("ds", ["H3N2", "HIV"])
@pytest.mark.parametrize("schedule,precision", [("gs", "f64"), ("slab", "f32")])
def test_device_reaches_the_references_own_error(ds, schedule, precision):
    """20 device runs (fresh start positions and seed each) of the reference's call.  Bands, on the mean final MAE:
    against the reference-held MAE of the reference's coordinates max(3 sd_oracle, 1 %) + the oracle's own offset
    from it (tests/test_reference_results.py: +0.7 % H3N2, +1.1 % HIV), i.e. 2 % flat; against the oracle's
    distribution the contract band max(3 sd, 1 %) for exact GS, 5 % for the slab schedule forced onto a 285-point
    91 %-missing panel (AUTO never takes it there: tests/test_gpu_assays.py)."""
    name = f"{ds.lower()}_refrun_{BEST[ds]}"
    dist = pp.oracle_distribution(name)
    runs = []
    for s in range(SEEDS):
        call, _ = pp.PROBLEMS[name]["fn"](100 + s)
        runs.append(_native.optimize_layout_exact_arrays(*layout_call_args(call), seed=500 + s, schedule=schedule,
                                                         precision=precision))
    call, _ = pp.PROBLEMS[name]["fn"](7)
    ref_mae, _ = ref_edge_mae(ds, call)
    got = np.array([r.final_mae for r in runs])
    mean, sd = dist["mean_final_mae"], dist["sd_final_mae"]
    assert abs(got.mean() / ref_mae - 1) <= 0.02, (got.mean(), ref_mae)
    band = max(3 * sd, 0.01 * mean) if schedule == "gs" else 0.05 * mean
    assert abs(got.mean() - mean) <= band, (got.mean(), mean, sd)
    if schedule == "gs":        # spread: the device's exact GS scatters like the oracle's
        assert got.std(ddof=1) <= 2.0 * sd + 1e-3 * mean, (got.std(ddof=1), sd)


def test_device_matches_the_oracle_on_the_denv_panel():
    """The third panel the reference ships results for (83 points, 10-D, c_repulsion 0.038: the error rises again after
    the best iteration, so the best-snapshot restore decides the result): 20 device runs of the notebook's call against
    the oracle's 64 (0.25250 +- 0.0019): mean inside max(3 sd, 1 %), same scatter, same stop.  The reference's DENV
    coordinate file itself is a later state of that trajectory (tests/test_reference_results.py) and is not a target."""
    dist = pp.oracle_distribution("denv_refrun_chain")
    runs = []
    for s in range(SEEDS):
        call, _ = pp.PROBLEMS["denv_refrun_chain"]["fn"](100 + s)
        r = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=500 + s)
        assert r.info["schedule"] == "gs" and r.converged
        sm, cnt = orc.edge_error(r.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        assert r.final_mae == pytest.approx(sm / cnt, rel=1e-10)
        runs.append(r)
    got = np.array([r.final_mae for r in runs])
    mean, sd = dist["mean_final_mae"], dist["sd_final_mae"]
    assert abs(got.mean() - mean) <= max(3 * sd, 0.01 * mean), (got.mean(), mean, sd)
    assert got.std(ddof=1) <= 2.0 * sd + 1e-3 * mean, (got.std(ddof=1), sd)
    its = np.array([r.iterations for r in runs])
    assert abs(its.mean() - dist["mean_iterations"]) <= max(3 * dist["sd_iterations"] / np.sqrt(SEEDS) * 3, 0.1 * dist["mean_iterations"])


@pytest.mark.parametrize("ds", ["H3N2", "HIV"])
@pytest.mark.parametrize("schedule,precision", [("gs", "f64"), ("slab", "f32")])
def test_reference_embedding_is_a_rest_point_of_the_device(ds, schedule, precision):
    """Same statement and bands as test_reference_embedding_is_a_rest_point_of_the_oracle; the device's reported
    MAE is checked with the oracle's edge error of the device's positions."""
    names, P = pp.ref_coordinates(ds)
    call = pp.refrun_call(ds, best_params(ds), init=core.RMatrix(P, names), n_iter=60, k0=0.2)
    mae0, _ = ref_edge_mae(ds, call)
    ei, ej = np.asarray(call.edge_i), np.asarray(call.edge_j)
    d0 = np.linalg.norm(P[ei] - P[ej], axis=1)
    for seed in range(3):
        r = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, schedule=schedule,
                                                 precision=precision)
        s, c = orc.edge_error(r.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        assert r.final_mae == pytest.approx(s / c, rel=1e-10 if precision == "f64" else 2e-5)
        d1 = np.linalg.norm(r.positions[ei] - r.positions[ej], axis=1)
        move = float(np.mean(np.abs(d1 - d0)) / d0.mean())
        tol_mae, tol_move = (2e-3, 4e-3) if ds == "H3N2" else (1.5e-2, 6e-2)
        assert abs(r.final_mae / mae0 - 1) <= tol_mae and move <= tol_move, (r.final_mae, mae0, move)


@pytest.mark.parametrize("ds", ["H3N2", "HIV", "DENV"])
def test_cv_evaluator_reproduces_the_references_likelihood_calls(ds):
    """The 48 reference likelihood_function() calls per data set, all folds of all sets in ONE launch (960
    embeddings).  Bands as for the oracle (mean ratio 2 %, every call 6 %, implied held-out count 0.5 % / 2.5 %), and
    device against oracle on the same parameter sets (independent fold draws): mean ratio within 1 %."""
    m = {"H3N2": pp.h3n2_matrix, "HIV": pp.hiv_matrix, "DENV": pp.denv_matrix}[ds]()
    ent = oracle_cv_entries(ds, "chain")
    sets = [e["params"] for e in ent]
    res, secs, n_emb = cv.likelihood_sweep(m, sets, 500, 1e-4, folds=20, rng=np.random.default_rng(2024))
    assert n_emb == 960 and len(res) == 48
    ours = np.array([r["Holdout_MAE"] for r in res])
    ref = np.array([e["ref_Holdout_MAE"] for e in ent])
    orc_mae = np.array([e["Holdout_MAE"] for e in ent])
    rel = ours / ref - 1
    if ds == "DENV":      # ndim 6..13; the chain's optimum (row 0) is the minimum of 22 260 noisy calls: 12 %
        assert 0 < rel[0] <= 0.12 and np.abs(rel[1:]).max() <= 0.08 and abs(rel.mean()) <= 0.025, (rel[:3], rel.mean())
    else:
        assert abs(rel.mean()) <= 0.02 and np.abs(rel).max() <= 0.06, (rel.mean(), np.abs(rel).max())
    assert abs((ours / orc_mae - 1).mean()) <= 0.01, (ours / orc_mae - 1).mean()
    n_ref = np.array([e["ref_NLL"] for e in ent]) / (1 + np.log(2 * ref))
    n_ours = np.array([r["NLL"] for r in res]) / (1 + np.log(2 * ours))
    assert abs(n_ours.mean() / n_ref.mean() - 1) <= 0.005 and np.abs(n_ours / n_ref - 1).max() <= 0.025
    # mean_iter / pct_converged against the oracle's on the same sets (the reference's chains do not hold them)
    it_dev = np.array([r["mean_iter"] for r in res])
    it_orc = np.array([e["mean_iter"] for e in ent])
    assert abs(it_dev.mean() / it_orc.mean() - 1) <= 0.10, (it_dev.mean(), it_orc.mean())


@pytest.mark.parametrize("ds", ["H3N2", "HIV"])
def test_cv_evaluator_reproduces_the_references_fold_errors(ds):
    """fold_stats.csv (20 per-fold out-of-sample MAEs of the reference: H3N2 0.828 +- 0.064, HIV 1.329 +- 0.096)
    against the device in the notebook's procedure (500 iterations, eps 1e-10, window 3), five fold draws for either
    candidate parameter set: |difference of means| <= 3 standard errors of the difference per draw, 3 % pooled."""
    m = pp.h3n2_matrix() if ds == "H3N2" else pp.hiv_matrix()
    ref = pp.ref_fold_stats(ds)
    se_ref = ref.std(ddof=1) / np.sqrt(ref.size)
    listed = dict(pp.H3N2_LISTED if ds == "H3N2" else pp.HIV_LISTED)
    opt = {k: pp.ref_chain_optimum(ds)[k] for k in ("N", "k0", "cooling_rate", "c_repulsion")}
    res, _, n_emb = cv.likelihood_sweep(m, [listed, opt] * 5, 500, 1e-10, folds=20, rng=np.random.default_rng(5),
                                        convergence_counter=3)
    assert n_emb == 200
    pooled = []
    for r in res:
        f = np.array(r["fold_mae"])
        assert f.size == 20
        se = np.hypot(se_ref, f.std(ddof=1) / np.sqrt(f.size))
        assert abs(f.mean() - ref.mean()) <= 3 * se, (f.mean(), ref.mean(), se)
        pooled.append(f)
    pooled = np.concatenate(pooled)
    assert abs(pooled.mean() / ref.mean() - 1) <= 0.03, (pooled.mean(), ref.mean())


@pytest.mark.parametrize("ds", ["H3N2", "HIV"])
def test_device_cv_reproduces_the_references_signed_error_distribution(ds):
    """error_distribution_HIV_H3N2.csv (mean, sd, quartiles of the reference's pooled SIGNED out-of-sample errors, 20
    folds) against the product's own pieces in the notebook's procedure: folds cut by cv.make_folds, every fold relaxed
    on the device through the `.Call` payload (500 iterations, eps 1e-10, window 3), distances by the device's
    est_distances, errors by cv.error_calculator_comparison (true - predicted, R/error_metrics.R:113).  Same bands as
    the oracle's test (tests/test_reference_results.py: SIGNED_BANDS, sd 4 %)."""
    from tests.test_reference_results import check_signed_distribution
    m = core.coded_matrix(pp.h3n2_matrix() if ds == "H3N2" else pp.hiv_matrix())
    params = best_params(ds)
    n = m.values.shape[0]
    rng = np.random.default_rng(8)
    signed = []
    for f, h in enumerate(cv.make_folds(m.values, 20, rng)):
        masked = m.masked(h % n, h // n)
        call = core.prepare_layout_call(masked, int(params["N"]), 500, params["k0"], params["cooling_rate"],
                                        params["c_repulsion"], 1e-10, 3, None, False, 3, False, rng)
        r = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=900 + f)
        err = cv.error_calculator_comparison(_native.est_distances(r.positions), m, masked, pred_names=call.names,
                                             true_names=m.names)
        oe = err["OutSampleError"]
        signed.append(oe[~np.isnan(oe)])
    e = np.concatenate(signed)
    assert e.size > 4000
    check_signed_distribution(ds, dict(Mean=float(e.mean()), SD=float(e.std(ddof=1)), Median=float(np.median(e)),
                                       Q1=float(np.quantile(e, 0.25)), Q3=float(np.quantile(e, 0.75))))
