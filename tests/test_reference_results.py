"""The oracle against the RESULTS the reference itself ships for this path (data copied by
tests/golden/make_reference_results.py into tests/golden/ref_results/):

  topolow_H3N2_coords.csv / topolow_HIV_coords.csv   the reference's own embeddings of BASELINE config 2's and 5's panels
  fold_stats.csv                                     its 20-fold out-of-sample errors per fold
  chain_sample_<DS>.csv                              48 of its likelihood_function() calls per data set: parameters -> Holdout_MAE, NLL

The reference's pair order and start positions are unseedable (src/optimization.cpp:153-154, R/core.R:412), so the
comparisons are distributional; every band is stated where it is asserted.  What is NOT recorded upstream is which of
two parameter sets wrote the coordinate files -- the ones the notebook lists (inst/examples/methods-comparison-h3n2-hiv-denv.Rmd:312-323)
or the ones its rule picks from the shipped chains (:622-658; that set has N = 5 for H3N2, the width of the file) --
both are tested.  A third panel, DENV (83 points, 10-D, repeated titrations averaged), pins the input construction,
the parameter-selection rule (digit for digit) and 48 more likelihood_function() calls; its coordinate file is NOT
what the reference's current code returns at the listed parameters (test_denv_coordinate_file_is_a_late_state...).
GPU twins of these tests: tests/test_gpu_reference_results.py."""
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
    # DENV: 1 838 titrations of 752 (virus, serum) pairs -- process_antigenic_data averages the repeats
    # (R/data_preprocessing.R:631-646) -- 83 points, the coordinate file's rows in titers_list_to_matrix's order
    nd, pd_ = pp.ref_coordinates("DENV")
    assert pd_.shape == (83, 10) and list(pp.denv_matrix().names) == nd
    call, _ = pp.PROBLEMS["denv_refrun_chain"]["fn"](7)
    assert len(call.edge_i) == 752 and not np.any(np.asarray(call.edge_thresh))


def test_selection_rule_reproduces_the_notebooks_listed_denv_parameters():
    """get_optimal_topolow_params() as restated in tests/golden/make_reference_results.py (finite rows with
    log_N >= log 2, clean_data(k = 3.5) per column, argmin Holdout_MAE) applied to the five shipped DENV chains
    (22 260 rows after cleaning) returns the parameter set the notebook prints for DENV
    (methods-comparison-h3n2-hiv-denv.Rmd:326-331: N 10, k0 7.1, cooling_rate 0.01232407, c_repulsion 0.03830152) to
    every printed digit.  (For H3N2 and HIV the printed sets predate the shipped chains and differ from the rule's pick.)"""
    opt = pp.ref_chain_optimum("DENV")
    assert opt["N"] == pp.DENV_LISTED["N"] and opt["rows_after_cleaning"] == 22260
    assert round(opt["k0"], 1) == 7.1 and round(opt["cooling_rate"], 8) == 0.01232407
    assert round(opt["c_repulsion"], 8) == 0.03830152


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


def test_denv_coordinate_file_is_a_late_state_of_the_oracles_trajectory_not_its_result():
    """topolow_DENV_coords.csv is examined, not claimed: its edge MAE on the DENV panel is 0.28363, the oracle's result
    for the call that is said to have written it (listed = chain-optimum parameters, 500 iterations, eps 1e-10,
    window 3) is 0.25250 +- 0.0019 over 64 runs (best iteration ~150, k ~ 1.1), and none of the other shipped DENV
    chains' optima gives more than 0.265.  What the file matches is a LATER state of the same trajectory: with
    c_repulsion 0.038 the error rises again once the spring has cooled (0.2525 at iteration ~170, 0.284 at ~375 where
    k ~ 0.07, 0.39 at 500) and the current code restores the best snapshot (src/optimization.cpp:368-374) -- the file
    was evidently written without that restore (an earlier package version; the chunk is eval = FALSE).  Asserted:
    (1) the MAE itself; (2) it lies between the oracle's best and its no-restore end state; (3) warm-started at
    k = 0.2 the file is within 1.5 % of a rest point of the oracle's forces and the error then RISES, as it does
    after the oracle's own best iteration; (4) the measured pairs' distances are the oracle's mean map to 3.5 % (the
    oracle's own runs: 1 %) -- the same map, inflated by the late repulsion.  The H3N2 / HIV files carry small
    c_repulsion (0.012 / 0.0007): there the late state and the best state coincide and the files ARE pins."""
    import dataclasses
    names, P = pp.ref_coordinates("DENV")
    call, _ = pp.PROBLEMS["denv_refrun_chain"]["fn"](7)
    mae, _ = ref_edge_mae("DENV", call)
    assert mae == pytest.approx(0.2836262225, rel=1e-9)
    dist = pp.oracle_distribution("denv_refrun_chain")
    assert dist["n_seeds"] >= 64 and mae > dist["mean_final_mae"] + 10 * dist["sd_final_mae"]
    free = dataclasses.replace(call, convergence_window=10 ** 6)
    tr = np.asarray(orc.optimize_layout_exact(*layout_call_args(free), seed=3).mae_trace)
    assert tr.min() < 0.26 < mae < tr[-1] and tr[-1] > 0.35
    late = int(np.argmax((tr > mae) & (np.arange(tr.size) > tr.argmin()))) * 3 + 3
    assert 330 <= late <= 420, late                                  # the iteration whose state has the file's error
    warm = pp.refrun_call("DENV", pp.ref_chain_optimum("DENV"), init=core.RMatrix(P, names), n_iter=60, k0=0.2)
    r = orc.optimize_layout_exact(*layout_call_args(warm), seed=0)
    assert abs(r.final_mae / mae - 1) <= 0.015 and r.mae_trace[5] > r.mae_trace[2]
    ei, ej = np.asarray(call.edge_i), np.asarray(call.edge_j)
    d_ref = np.linalg.norm(P[ei] - P[ej], axis=1)
    d_mean = np.array(dist["edge_dist_mean"])
    assert float(np.mean(np.abs(d_ref - d_mean)) / d_mean.mean()) <= 0.035


@pytest.mark.parametrize("ds", ["H3N2", "HIV", "DENV"])
def test_oracle_cv_reproduces_the_references_likelihood_calls(ds):
    """48 likelihood_function() calls of the reference per data set (its adaptive-sampling chains: parameters ->
    Holdout_MAE, NLL; 20 folds, 500 iterations, eps 1e-4) re-evaluated with the oracle in the fold evaluator of
    R/adaptive_sampling.R:2552-2726.  Fold draws, start positions and pair order are random on both sides: one call's
    Holdout_MAE scatters by ~1.5 %.  Measured ratio oracle / reference - 1: H3N2 mean +1.3 % (sd 1.7 %, max 4.9 %),
    HIV mean -0.2 % (sd 1.2 %, max 4.6 %), DENV (83 points, 38 held-out pairs per fold: one call scatters by 2.4 %)
    mean +1.7 %, max 6.2 % -- and 9.9 % for the chain's optimum itself, the minimum of 22 260 noisy calls, which
    cannot be reproduced by an independent draw (its band: 12 %).  Bands: mean 2 %, every call 6 % (DENV 7 %).  The pooled number of held-out numeric
    cells implied by the reference's own (Holdout_MAE, NLL) pair, n = NLL / (1 + log(2 MAE)), pins the fold-size rule
    and the exclusion of threshold cells (R/error_metrics.R:90-91): it varies with the draw (how many threshold
    cells a fold happens to hold), so the means over the 48 calls are compared (0.5 %) and every call to 2.5 %."""
    ent = oracle_cv_entries(ds, "chain")
    assert len(ent) == 48
    ours = np.array([e["Holdout_MAE"] for e in ent])
    ref = np.array([e["ref_Holdout_MAE"] for e in ent])
    rel = ours / ref - 1
    if ds == "DENV":
        assert ref[0] == ref.min() and 0 < rel[0] <= 0.12 and np.abs(rel[1:]).max() <= 0.07, rel[:3]
        assert abs(rel.mean()) <= 0.02, rel.mean()
    else:
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


SIGNED_BANDS = dict(Mean=0.06, Median=0.08, Q1=0.12, Q3=0.14)      # absolute, in the panels' log2 titer units (sd 1.1-1.3)


def check_signed_distribution(ds, got):
    """One 20-fold draw's pooled signed out-of-sample errors against the reference's error_distribution_HIV_H3N2.csv."""
    ref = pp.ref_error_distribution(ds)
    assert ref["Mean"] < -0.3 and got["Mean"] < 0            # same sign: true - predicted < 0, the maps OVER-predict held-out distances
    for k, band in SIGNED_BANDS.items():
        assert abs(got[k] - ref[k]) <= band, (ds, k, got[k], ref[k])
    assert abs(got["SD"] / ref["SD"] - 1) <= 0.04, (ds, got["SD"], ref["SD"])


@pytest.mark.parametrize("ds", ["H3N2", "HIV"])
def test_oracle_cv_reproduces_the_references_signed_error_distribution(ds):
    """error_distribution_HIV_H3N2.csv: mean, sd and quartiles of the SIGNED out-of-sample errors of the notebook's 20
    folds pooled (~4 950 / ~4 430 cells; OutSampleError = true - predicted, R/error_metrics.R:113) -- reference H3N2
    -0.326 / 1.067 / -0.263 / -0.928 / 0.304, HIV -0.989 / 1.340 / -0.999 / -1.837 / -0.108 -- against the oracle in the
    same procedure, three fold draws for either candidate parameter set.  Measured: H3N2 mean -0.305 ... -0.362, sd
    +0.7 ... +3.4 %, quartiles within 0.05; HIV mean -0.970 ... -1.024, sd -0.1 ... -2.9 %, Q1 within 0.11, Q3 within
    0.13.  The bias (sign and size) is a property of the update rule -- repulsion and the '>' / '<' rule -- that the
    absolute error alone does not pin.  Bands: SIGNED_BANDS, sd 4 %."""
    ent = oracle_cv_entries(ds, "notebook")
    assert len(ent) == 6
    for e in ent:
        check_signed_distribution(ds, e["signed"])
