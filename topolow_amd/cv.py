"""k-fold cross-validation evaluator on the batched relaxation kernel.

Mirrors the reference's `likelihood_function()` (R/adaptive_sampling.R:2552-2726) -- the consumer
that calls `euclidean_embedding()` folds x samples times during `Euclidify()` -- and the part of
`error_calculator_comparison()` (R/error_metrics.R:55-144) it uses.  Where the reference runs
the folds one after another (or one per forked process), here every fold of every parameter
set becomes one workgroup of a single `topolow_optimize_layout_exact_batch` launch.
"""
from __future__ import annotations

import math
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _native, core


def _numeric(x) -> np.ndarray:
    """as.numeric(<matrix>): threshold strings -> NA."""
    return core.coded_matrix(x).as_numeric()


def error_calculator_comparison(predicted, true, input_=None, pred_names=None, true_names=None):
    """OutSampleError / InSampleError vectors and Completeness (R/error_metrics.R:55-144).
    Threshold strings become NA under as.numeric (:90-91) and therefore drop out."""
    pred = np.asarray(predicted, dtype=np.float64)
    truth_m = _numeric(true)
    input_m = truth_m if input_ is None else _numeric(input_)
    if pred.shape != truth_m.shape or pred.shape != input_m.shape:
        raise ValueError("All matrices must have the same dimensions")
    if true_names is None and isinstance(true, (core.RMatrix, core.CodedMatrix)):
        true_names = true.names
    if pred_names is not None and true_names is not None:
        lookup = {nm: q for q, nm in enumerate(pred_names)}
        try:
            order = [lookup[nm] for nm in true_names]
        except KeyError:
            raise ValueError("Row and column names must match between matrices after ordering") from None
        pred = pred[np.ix_(order, order)]
    # R flattens column-major; the statistics below do not depend on the order
    truth = truth_m.ravel(order="F")
    inputv = input_m.ravel(order="F")
    predv = pred.ravel(order="F")
    missing = np.isnan(inputv)
    in_err = np.where(~missing, truth - predv, np.nan)
    out_err = np.where(missing, truth - predv, np.nan)
    validation = int(np.sum(~np.isnan(truth[missing])))
    got = int(np.sum(~np.isnan(out_err)))
    if validation > 0:
        completeness = got / validation
    else:
        total = int(np.sum(~np.isnan(truth)))
        completeness = (int(np.sum(~np.isnan(predv))) / total) if total > 0 else 0.0
    return dict(InSampleError=in_err, OutSampleError=out_err, Completeness=completeness)


def make_folds(values: np.ndarray, folds: int, rng: np.random.Generator):
    """Holdout index sets of R/adaptive_sampling.R:2570-2598: `folds` disjoint draws of
    floor(#non-NA / (2 folds)) linear (column-major) indices; a drawn cell and its mirror leave
    the pool.  `values`: stripped numeric matrix, NaN = NA."""
    n = values.shape[0]
    pool = ~np.isnan(np.asarray(values, dtype=np.float64))
    num_elements = int(pool.sum())
    holdout_size = num_elements // (folds * 2)
    out = []
    for _ in range(folds):
        avail = np.flatnonzero(pool.ravel(order="F"))
        if avail.size < holdout_size:
            warnings.warn("Could not create all folds due to data sparsity. Using fewer folds.")
            break
        pick = rng.choice(avail, size=holdout_size, replace=False)
        out.append(pick)
        r, c = pick % n, pick // n
        pool[r, c] = False
        pool[c, r] = False
    return out


def likelihood_sweep(dissimilarity_matrix, param_sets: Sequence[Dict[str, float]], mapping_max_iter: int,
                     relative_epsilon: float, folds: int = 20, preserve_order: bool = False,
                     rng: Optional[np.random.Generator] = None, precision: str = "f64"):
    """`likelihood_function` for MANY parameter sets at once: all folds of all sets are relaxed
    in ONE batched launch.  param_sets: dicts with N (ndim), k0, cooling_rate, c_repulsion.
    Returns (list of result dicts, device_seconds, embeddings)."""
    rng = rng if rng is not None else _native.host_rng()
    if not hasattr(rng, "choice"):   # R-stream generator: fold sampling uses a NumPy stream seeded from it
        rng = np.random.default_rng(rng.integers(0, 2 ** 53))
    m = core.coded_matrix(dissimilarity_matrix)   # strings are parsed once, not once per fold
    if m is None:
        raise ValueError("dissimilarity_matrix must be a matrix")
    calls, owners, inputs = [], [], []
    for s_idx, ps in enumerate(param_sets):
        fold_sets = make_folds(m.values, folds, rng)
        n_pts = m.values.shape[0]
        for h in fold_sets:
            masked = m.masked(h % n_pts, h // n_pts)
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    call = core.prepare_layout_call(masked, int(ps["N"]), mapping_max_iter, ps["k0"],
                                                    ps["cooling_rate"], ps["c_repulsion"], relative_epsilon,
                                                    5, None, False, 3, preserve_order, rng)
            except ValueError:
                call = None  # the reference's tryCatch turns a failed fold into an NA row
            calls.append(call)
            owners.append(s_idx)
            inputs.append(masked)
    live = [c for c in calls if c is not None]
    seeds = [int(rng.integers(0, 2 ** 63 - 1)) for _ in live]
    results, secs = _native.optimize_layout_exact_batch(live, seeds=seeds, precision=precision) if live else ([], 0.0)
    it = iter(results)
    per_set: List[List[dict]] = [[] for _ in param_sets]
    for call, owner, masked in zip(calls, owners, inputs):
        if call is None:
            continue
        res = next(it)
        if isinstance(res, Exception):
            continue
        p = res.positions
        diff = p[:, None, :] - p[None, :, :]
        est = np.sqrt((diff * diff).sum(-1))
        err = error_calculator_comparison(est, m, masked, pred_names=call.names, true_names=m.names)
        oe = err["OutSampleError"]
        oe = oe[~np.isnan(oe)]
        per_set[owner].append(dict(n_samples=int(oe.size), sum_abs_errors=float(np.abs(oe).sum()),
                                   iter=res.iterations, converged=int(res.converged)))
    out = []
    for rows in per_set:
        rows = [r for r in rows if r["n_samples"] > 0]
        if not rows:
            out.append(dict(Holdout_MAE=math.nan, NLL=math.nan, mean_iter=math.nan, pct_converged=math.nan))
            continue
        total = sum(r["n_samples"] for r in rows)
        tot_err = sum(r["sum_abs_errors"] for r in rows)
        mae = tot_err / total if total > 0 else math.nan
        nll = total * (1 + math.log(2 * mae)) if not math.isnan(mae) and mae > 0 else math.nan
        out.append(dict(Holdout_MAE=mae, NLL=nll, mean_iter=float(np.mean([r["iter"] for r in rows])),
                        pct_converged=100.0 * float(np.mean([r["converged"] for r in rows]))))
    return out, secs, len(live)


def likelihood_sweep_distributed(dissimilarity_matrix, param_sets: Sequence[Dict[str, float]],
                                 mapping_max_iter: int, relative_epsilon: float, folds: int = 20,
                                 preserve_order: bool = False, seed: int = 0, precision: str = "f64",
                                 batch_fn=None):
    """BASELINE config 5 across GPUs: the parameter sets of a sweep are dealt round-robin to the
    ranks of the current torch.distributed group (one process per GPU); every rank relaxes its share
    as one batched launch on its own device; the per-set results are then all-gathered (a few
    floats per set -- there is no data-path collective, the embeddings are independent, exactly as
    in the reference's mclapply fan-out, R/adaptive_sampling.R:666,1301).  Without an initialised
    process group this is `likelihood_sweep`.  Returns the results in the order of `param_sets`."""
    rank, world = 0, 1
    dist = None
    try:
        import torch.distributed as dist_mod
        if dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
            rank, world = dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    mine = list(range(rank, len(param_sets), world))
    rng = np.random.default_rng([int(seed), rank])
    fn = batch_fn if batch_fn is not None else likelihood_sweep
    res, secs, n_emb = fn(dissimilarity_matrix, [param_sets[q] for q in mine], mapping_max_iter,
                          relative_epsilon, folds, preserve_order, rng, precision) if mine else ([], 0.0, 0)
    if dist is None or world == 1:
        return res, secs, n_emb
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, res, secs, n_emb))
    out = [None] * len(param_sets)
    total_emb, max_secs = 0, 0.0
    for idx, rr, sc, ne in gathered:
        for q, r in zip(idx, rr):
            out[q] = r
        total_emb += ne
        max_secs = max(max_secs, sc)
    return out, max_secs, total_emb


def likelihood_function(dissimilarity_matrix, mapping_max_iter, relative_epsilon, N, k0, cooling_rate,
                        c_repulsion, folds=20, num_cores=1, preserve_order=False):
    """Drop-in for the reference's `likelihood_function()` (R/adaptive_sampling.R:2552-2555):
    pooled Holdout_MAE, NLL = n(1 + log(2 MAE)), mean_iter, pct_converged.  `num_cores` is kept
    for signature compatibility; the folds always run as one GPU batch."""
    res, _, _ = likelihood_sweep(dissimilarity_matrix,
                                 [dict(N=N, k0=k0, cooling_rate=cooling_rate, c_repulsion=c_repulsion)],
                                 mapping_max_iter, relative_epsilon, folds, preserve_order)
    return res[0]
