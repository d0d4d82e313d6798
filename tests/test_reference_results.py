"""The oracle against the RESULTS the reference itself ships for this path (data copied by
tests/golden/make_reference_results.py into tests/golden/ref_results/):

  topolow_H3N2_coords.csv / topolow_HIV_coords.csv   the reference's own embeddings of BASELINE config 2's and 5's panels
  fold_stats.csv                                     its 20-fold out-of-sample errors per fold
  chain_sample_<DS>.csv                              48 of its likelihood_function() calls per data set: parameters -> Holdout_MAE, NLL

The reference's pair order and start positions are unseedable (src/optimization.cpp:153-154, R/core.R:412), so the
comparisons are distributional; every band is stated where it is asserted.  What is NOT recorded upstream is which of
two parameter sets wrote the coordinate files -- the ones the notebook lists (inst/examples/methods-comparison-h3n2-hiv-denv.Rmd:312-323)
or the ones its rule picks from the shipped chains (:622-658; that set has N = 5 for H3N2, the width of the file) --
both are tested.  GPU twins of these tests: tests/test_gpu_reference_results.py."""
import json
import os

import numpy as np
import pytest

from oracle import topolow_oracle as orc
from tests import parity_problems as pp
from tests.conftest import layout_call_args
from tests.helpers import oracle_cv
from topolow_amd import core

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CANDIDATES = [("H3N2", "chain"), ("H3N2", "listed"), ("HIV", "chain"), ("HIV", "listed")]
# the candidate that explains each file best (tests below): H3N2 -- the chain optimum (the only one with N = 5);
# HIV -- the listed set (started from the reference's coordinates, the chain optimum's 28x stronger repulsion pushes
# the error UP at once, the listed set leaves them where they are)
BEST = {"H3N2": "chain", "HIV": "listed"}


def ref_edge_mae(ds, call):
    names, P = pp.ref_coordinates(ds)
    assert list(call.names) == names
    s, c = orc.edge_error(P, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    return s / c, P


def oracle_cv_entries(ds, kind):
    with open(os.path.join(GOLD, f"oracle_cv_{ds}.json")) as fh:
        return [e for e in json.load(fh)["entries"] if e["kind"] == kind]


def test_panels_are_the_ones_the_reference_embedded():
    """Input construction pinned by the row names of the reference's coordinate files: same 285 / 335
    points with the V/ and S/ prefixes; for H3N2 also the same row ORDER as titers_list_to_matrix(sort = TRUE) builds
    (R/data_preprocessing.R:743-844)."""
    h3, hv = pp.h3n2_matrix(), pp.hiv_matrix()
    n3, p3 = pp.ref_coordinates("H3N2")
    nv, pv = pp.ref_coordinates("HIV")
    assert p3.shape == (285, 5) and pv.shape == (335, 2)
    assert list(h3.names) == n3
    assert sorted(hv.names) == sorted(nv) and len(set(nv)) == 335
    assert list(pp.ref_matrix("HIV").names) == nv


@pytest.mark.parametrize("ds,which", CANDIDATES)
def test_reference_embedding_error_lies_in_the_oracle_distribution(ds, which):
    """Edge MAE (src/optimization.cpp:54-81) of the reference's shipped coordinates against 64 oracle runs of the
    call that wrote them (500 iterations, eps 1e-10, window 3; fresh start positions and pair order per run).
    Band: max(3 sd, 1 %) of the oracle mean -- BASELINE.md section 3's contract band -- for the better candidate of
    each data set, 2 % for the other.  Measured: H3N2 0.59241 vs 0.58806 +- 0.0037 (chain, +0.7 %, 1.2 sd) and
    0.58330 +- 0.0029 (listed, +1.6 %); HIV 1.22454 vs 1.21109 +- 0.0121 (listed, +1.1 %, 1.1 sd) and
    1.20589 +- 0.0081 (chain, +1.5 %)."""
    name = f"{ds.lower()}_refrun_{which}"
    dist = pp.oracle_distribution(name)
    call, _ = pp.PROBLEMS[name]["fn"](7)
    mae, _ = ref_edge_mae(ds, call)
    assert mae == pytest.approx({"H3N2": 0.5924083102, "HIV": 1.2245447514}[ds], rel=1e-9)
    mean, sd = dist["mean_final_mae"], dist["sd_final_mae"]
    assert dist["n_seeds"] >= 64
    band = max(3 * sd, 0.01 * mean) if BEST[ds] == which else 0.02 * mean
    assert abs(mae - mean) <= band, (mae, mean, sd)


@pytest.mark.parametrize("ds,which", CANDIDATES)
def test_reference_embedding_distances_agree_with_the_oracles(ds, which):
    """Rotation-free: the distances of the measured pairs in the reference's map against the oracle's seed-mean
    distances of the same pairs.  Gap = mean |d - d_mean| / mean d_mean.  The reference's map must lie no further
    from the oracle's mean map than the oracle's own runs do (H3N2: 0.026 against 0.020 mean, 0.031 max over 64
    runs; HIV, 2-D and less rigid: 0.10 against 0.09 mean, 0.18-0.20 max), and the mean distance within 2.5 %."""
    name = f"{ds.lower()}_refrun_{which}"
    dist = pp.oracle_distribution(name)
    call, _ = pp.PROBLEMS[name]["fn"](7)
    _, P = ref_edge_mae(ds, call)
    ei, ej = np.asarray(call.edge_i), np.asarray(call.edge_j)
    d_ref = np.linalg.norm(P[ei] - P[ej], axis=1)
    d_mean = np.array(dist["edge_dist_mean"])
    gap = float(np.mean(np.abs(d_ref - d_mean)) / d_mean.mean())
    assert gap <= max(dist["edge_gap"]), (gap, max(dist["edge_gap"]))
    assert abs(d_ref.mean() / d_mean.mean() - 1) <= 0.025


@pytest.mark.parametrize("ds", ["H3N2", "HIV"])
def test_reference_embedding_is_a_rest_point_of_the_oracle(ds):
    """Warm start (initial_positions, R/core.R:247-257) from the reference's coordinates late in the schedule
    (k = 0.2: what k0 has cooled to after ~200 / ~70 iterations): if the oracle's forces are the reference's, its
    relaxation has nothing left to do there.  H3N2: the error moves 0.5924 -> 0.5921 (band 0.2 %) and the measured
    pairs' distances by 0.15 % (band 0.4 %).  HIV (2-D, 93 % missing, soft): error within 1.5 %, distances within 6 %."""
    which = BEST[ds]
    params = pp.ref_chain_optimum(ds) if which == "chain" else dict(pp.HIV_LISTED)
    names, P = pp.ref_coordinates(ds)
    call = pp.refrun_call(ds, params, init=core.RMatrix(P, names), n_iter=60, k0=0.2)
    assert np.array_equal(np.asarray(call.initial_positions), P)
    mae0, _ = ref_edge_mae(ds, call)
    ei, ej = np.asarray(call.edge_i), np.asarray(call.edge_j)
    d0 = np.linalg.norm(P[ei] - P[ej], axis=1)
    for seed in range(3):
        r = orc.optimize_layout_exact(*layout_call_args(call), seed=seed)
        d1 = np.linalg.norm(r.positions[ei] - r.positions[ej], axis=1)
        move = float(np.mean(np.abs(d1 - d0)) / d0.mean())
        if ds == "H3N2":
            assert abs(r.final_mae / mae0 - 1) <= 2e-3 and move <= 4e-3, (r.final_mae, mae0, move)
        else:
            assert abs(r.final_mae / mae0 - 1) <= 1.5e-2 and move <= 6e-2, (r.final_mae, mae0, move)


def test_chain_optimum_repulsion_does_not_explain_the_hiv_file():
    """Why BEST['HIV'] is the listed set: from the reference's HIV coordinates the chain optimum's repulsion
    (c_rep 0.0194 against 0.0007) RAISES the error monotonically over the first checks."""
    names, P = pp.ref_coordinates("HIV")
    call = pp.refrun_call("HIV", pp.ref_chain_optimum("HIV"), init=core.RMatrix(P, names), n_iter=12, k0=0.2)
    r = orc.optimize_layout_exact(*layout_call_args(call), seed=0)
    tr = r.mae_trace
    assert tr[0] > 1.2245 * 1.005 and np.all(np.diff(tr[:4]) > 0), tr


def test_oracle_cv_fixture_is_what_the_oracle_computes():
    """One entry of tests/golden/oracle_cv_HIV.json recomputed live (same fold draws, same seeds)."""
    e = oracle_cv_entries("HIV", "chain")[3]
    r = oracle_cv(pp.hiv_matrix(), e["params"], 20, np.random.default_rng([11, e["row"]]), 500, 1e-4, 5,
                  seed0=1000 * e["row"])
    assert r["Holdout_MAE"] == pytest.approx(e["Holdout_MAE"], rel=1e-9)
    assert r["mean_iter"] == e["mean_iter"] and r["pct_converged"] == e["pct_converged"]


@pytest.mark.parametrize("ds", ["H3N2", "HIV"])
def test_oracle_cv_reproduces_the_references_likelihood_calls(ds):
    """48 likelihood_function() calls of the reference per data set (its adaptive-sampling chains: parameters ->
    Holdout_MAE, NLL; 20 folds, 500 iterations, eps 1e-4) re-evaluated with the oracle in the fold evaluator of
    R/adaptive_sampling.R:2552-2726.  Fold draws, start positions and pair order are random on both sides: one call's
    Holdout_MAE scatters by ~1.5 %.  Measured ratio oracle / reference - 1: H3N2 mean +1.3 % (sd 1.7 %, max 4.9 %),
    HIV mean -0.2 % (sd 1.2 %, max 4.6 %).  Bands: mean 2 %, every call 6 %.  The pooled number of held-out numeric
    cells implied by the reference's own (Holdout_MAE, NLL) pair, n = NLL / (1 + log(2 MAE)), pins the fold-size rule
    and the exclusion of threshold cells (R/error_metrics.R:90-91): it varies with the draw (how many threshold
    cells a fold happens to hold), so the means over the 48 calls are compared (0.5 %) and every call to 2.5 %."""
    ent = oracle_cv_entries(ds, "chain")
    assert len(ent) == 48
    ours = np.array([e["Holdout_MAE"] for e in ent])
    ref = np.array([e["ref_Holdout_MAE"] for e in ent])
    rel = ours / ref - 1
    assert abs(rel.mean()) <= 0.02 and np.abs(rel).max() <= 0.06, (rel.mean(), np.abs(rel).max())
    n_ref = np.array([e["ref_NLL"] for e in ent]) / (1 + np.log(2 * ref))
    n_ours = np.array([e["NLL"] for e in ent]) / (1 + np.log(2 * ours))
    assert np.allclose(n_ref, np.round(n_ref), atol=1e-3)        # the reference's n is an integer count
    assert abs(n_ours.mean() / n_ref.mean() - 1) <= 0.005, (n_ours.mean(), n_ref.mean())
    assert np.abs(n_ours / n_ref - 1).max() <= 0.025, (n_ours[:4], n_ref[:4])


@pytest.mark.parametrize("ds", ["H3N2", "HIV"])
def test_oracle_cv_reproduces_the_references_fold_errors(ds):
    """fold_stats.csv: the reference's 20 per-fold out-of-sample MAEs (H3N2 0.828 +- 0.064, HIV 1.329 +- 0.096)
    against the oracle in the same procedure (500 iterations, eps 1e-10, window 3), three independent fold draws
    for either candidate parameter set.  Band: |difference of the means| <= 3 standard errors of the difference
    (both sides are 20-fold means), and the fold-to-fold spread within a factor 2."""
    ref = pp.ref_fold_stats(ds)
    assert ref.size == 20
    se_ref = ref.std(ddof=1) / np.sqrt(ref.size)
    for e in oracle_cv_entries(ds, "notebook"):
        f = np.array(e["fold_mae"])
        se = np.hypot(se_ref, f.std(ddof=1) / np.sqrt(f.size))
        assert abs(f.mean() - ref.mean()) <= 3 * se, (e["params_from"], f.mean(), ref.mean(), se)
        assert 0.5 <= f.std(ddof=1) / ref.std(ddof=1) <= 2.0
    pooled = np.concatenate([e["fold_mae"] for e in oracle_cv_entries(ds, "notebook")])
    assert abs(pooled.mean() / ref.mean() - 1) <= 0.03, (pooled.mean(), ref.mean())
