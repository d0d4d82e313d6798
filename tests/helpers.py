"""Shared test helpers: run the host driver with the CPU oracle standing in for the `.Call`
(tests only -- the product has no such path)."""
import numpy as np

import oracle
from oracle import topolow_oracle as orc
from topolow_amd import core


def oracle_native(seed=0, order_mode=orc.ORDER_SEEDED):
    def fn(call):
        return orc.optimize_layout_exact(
            call.initial_positions, call.dissimilarity_matrix, call.threshold_matrix, call.degrees,
            call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.n_iter, call.k0,
            call.cooling_rate, call.c_repulsion, call.relative_epsilon, call.convergence_window,
            call.convergence_check_freq, call.verbose, seed=seed, order_mode=order_mode)
    return fn


def numpy_pdist(p):
    p = np.asarray(p, float)
    d = p[:, None, :] - p[None, :, :]
    return np.sqrt((d * d).sum(-1))


def embed_with_oracle(D, ndim, mapping_max_iter=1000, k0=core._MISSING, cooling_rate=core._MISSING,
                      c_repulsion=core._MISSING, relative_epsilon=1e-4, convergence_counter=5,
                      initial_positions=None, write_positions_to_csv=False, output_dir=core._MISSING,
                      verbose=False, convergence_check_freq=3, preserve_order=False, seed=0,
                      rng=None):
    return core._embed_with(oracle_native(seed), numpy_pdist, D, ndim, mapping_max_iter, k0,
                            cooling_rate, c_repulsion, relative_epsilon, convergence_counter,
                            initial_positions, write_positions_to_csv, output_dir, verbose,
                            convergence_check_freq, preserve_order,
                            rng if rng is not None else np.random.default_rng(seed))


def quickstart_matrix():
    """Reference README.md:54-69: S1,S2,S3,V1,V2 with V1-V2 missing."""
    pts = np.array([[0, 0], [3, 0], [4, 4], [2, 2], [0, 4]], dtype=float)
    D = numpy_pdist(pts)
    D[3, 4] = D[4, 3] = np.nan
    return core.RMatrix(D, ["S1", "S2", "S3", "V1", "V2"])


def oracle_cv(matrix, params, folds, rng, mapping_max_iter=500, relative_epsilon=1e-4, convergence_counter=5,
              seed0=0):
    """The reference's k-fold evaluator (R/adaptive_sampling.R:2552-2726: folds of floor(#non-NA / 2 folds) cells,
    masked symmetrically, one embedding per fold, out-of-sample errors of the numeric held-out cells) with the CPU
    oracle standing in for the `.Call` -- the CPU twin of topolow_amd.cv.likelihood_sweep(path="dense").
    Returns the pooled dict of likelihood_function plus `fold_mae` (mean |error| per fold: what the reference's
    notebook aggregates into fold_stats.csv, inst/examples/methods-comparison-h3n2-hiv-denv.Rmd:2024-2028)."""
    from tests.conftest import layout_call_args
    from topolow_amd import cv
    m = core.coded_matrix(matrix)
    n = m.values.shape[0]
    rows, signed = [], []
    for f, h in enumerate(cv.make_folds(m.values, folds, rng)):
        masked = m.masked(h % n, h // n)
        call = core.prepare_layout_call(masked, int(params["N"]), mapping_max_iter, params["k0"],
                                        params["cooling_rate"], params["c_repulsion"], relative_epsilon,
                                        convergence_counter, None, False, 3, False, rng)
        r = orc.optimize_layout_exact(*layout_call_args(call), seed=seed0 + f)
        err = cv.error_calculator_comparison(numpy_pdist(r.positions), m, masked, pred_names=call.names,
                                             true_names=m.names)
        oe = err["OutSampleError"]
        oe = oe[~np.isnan(oe)]
        rows.append(dict(n_samples=int(oe.size), sum_abs_errors=float(np.abs(oe).sum()), iter=int(r.iterations),
                         converged=int(r.converged)))
        signed.append(oe)
    out = cv._pooled([rows])[0]
    out["fold_mae"] = [r["sum_abs_errors"] / r["n_samples"] for r in rows if r["n_samples"] > 0]
    # the SIGNED out-of-sample errors of all folds pooled, summarised as the reference's notebook does
    # (inst/examples/methods-comparison-h3n2-hiv-denv.Rmd:2339-2349 -> comparison_results/error_distribution_HIV_H3N2.csv;
    # quantile() type 7 = numpy's default)
    e = np.concatenate(signed) if signed else np.zeros(0)
    if e.size > 1:
        out["signed"] = dict(Mean=float(e.mean()), SD=float(e.std(ddof=1)), Median=float(np.median(e)),
                             Q1=float(np.quantile(e, 0.25)), Q3=float(np.quantile(e, 0.75)), n=int(e.size))
    return out
