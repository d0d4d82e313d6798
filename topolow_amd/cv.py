"""k-fold cross-validation evaluator on the batched relaxation kernel.

Mirrors the reference's `likelihood_function()` (R/adaptive_sampling.R:2552-2726) -- the consumer
that calls `euclidean_embedding()` folds x samples times during `Euclidify()` -- and the part of
`error_calculator_comparison()` (R/error_metrics.R:55-144) it uses.  Where the reference runs
the folds one after another (or one per forked process), here every fold of every parameter
set becomes one workgroup of a single `topolow_optimize_layout_exact_batch` launch.
"""
from __future__ import annotations

import math
import os
import warnings
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
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


@dataclass
class SparseCall:
    """A `core.LayoutCall` without the two n x n arrays: the edge list IS the matrix (the batch
    entry of the library accepts that form, include/topolow_relax.h)."""
    initial_positions: np.ndarray
    degrees: np.ndarray
    edge_i: np.ndarray
    edge_j: np.ndarray
    edge_dist: np.ndarray
    edge_thresh: np.ndarray
    n_iter: int
    k0: float
    cooling_rate: float
    c_repulsion: float
    relative_epsilon: float
    convergence_window: int
    convergence_check_freq: int
    names: Optional[List[str]] = None
    order: Optional[np.ndarray] = None
    dissimilarity_matrix: None = None
    threshold_matrix: None = None


class FoldBuilder:
    """`prepare_layout_call(masked matrix)` + the fold's out-of-sample cells, for the folds of ONE
    matrix, without re-deriving everything from an n x n matrix per fold: the non-NA cells are listed
    once; a fold drops its held-out cells from the list and rebuilds degrees, ordering, edge list and
    start positions from what is left (same arithmetic in the same order as core.prepare_layout_call,
    so the result is identical -- tests/test_host_driver.py)."""

    def __init__(self, m: core.CodedMatrix):
        self.m = m
        n = self.n = m.values.shape[0]
        non_na = ~np.isnan(m.values)
        cols, rows = np.nonzero(non_na.T)            # column-major enumeration, as R's which()
        self.rows, self.cols = rows.astype(np.int64), cols.astype(np.int64)
        self.vals = m.values[rows, cols]
        self.codes = m.codes[rows, cols].astype(np.int32)
        self.pos_of = np.full(n * n, -1, dtype=np.int64)
        self.pos_of[self.rows + self.cols * n] = np.arange(self.rows.shape[0])
        self.zeroed = np.where(non_na, m.values, 0.0)   # NaN -> 0, the form np.nanmean sums
        self.offdiag = ~np.eye(n, dtype=bool)
        self.row_cnt = (non_na & self.offdiag).sum(axis=1)
        self.col_cnt = (non_na & self.offdiag).sum(axis=0)
        self._cells = None          # the library's view of the list, built on first use

    def folds(self, folds: int, rng: np.random.Generator):
        """`make_folds` on the cell list: the same draws from the same stream (the pool handed to
        rng.choice is the same array), without an n x n pass per fold."""
        n = self.n
        lin = self.rows + self.cols * n                  # ascending: the list is in column-major order
        alive = np.ones(lin.shape[0], dtype=bool)
        holdout_size = int(lin.shape[0]) // (folds * 2)
        out = []
        for _ in range(folds):
            avail = lin[alive]
            if avail.size < holdout_size:
                warnings.warn("Could not create all folds due to data sparsity. Using fewer folds.")
                break
            pick = rng.choice(avail, size=holdout_size, replace=False)
            out.append(pick)
            r, c = pick % n, pick // n
            alive[self.pos_of[pick]] = False
            mirror = self.pos_of[c + r * n]
            alive[mirror[mirror >= 0]] = False
        return out

    def _order(self, dr, dc):
        """core.spectral_order of the matrix with cells (dr, dc) set to NA."""
        n = self.n
        z = self.zeroed.copy()
        z[dr, dc] = 0.0
        np.fill_diagonal(z, 0.0)
        off = dr != dc
        rc = self.row_cnt - np.bincount(dr[off], minlength=n)
        cc = self.col_cnt - np.bincount(dc[off], minlength=n)
        with np.errstate(invalid="ignore", divide="ignore"):
            avg = (z.sum(axis=1) / rc + z.sum(axis=0) / cc) / 2.0
        avg[np.isnan(avg)] = 0.0
        if int(np.sum(avg > 0)) > 1:
            return np.argsort(avg, kind="stable")
        return None

    def fold(self, picks: np.ndarray, ndim: int, mapping_max_iter, k0, cooling_rate, c_repulsion,
             relative_epsilon, convergence_counter, convergence_check_freq, preserve_order, rng):
        """(SparseCall, (hold_i, hold_j, hold_truth)) for the fold holding out the linear
        (column-major) cell indices `picks` and their mirrors.  The list work runs in the library's
        host code (`topolow_cv_fold`, topolow_amd/csrc/relax_fold.h); `fold_numpy` is the same thing
        in NumPy and the two are tested to agree to the last bit."""
        return self.fold_from_draw(picks, ndim, mapping_max_iter, k0, cooling_rate, c_repulsion, relative_epsilon,
                                   convergence_counter, convergence_check_freq, preserve_order, rng=rng)

    def cells(self):
        if self._cells is None:
            self._cells = _native.CellList(self.n, self.rows, self.cols, self.vals, self.codes, self.pos_of)
        return self._cells

    def fold_from_draw(self, picks: np.ndarray, ndim: int, mapping_max_iter, k0, cooling_rate, c_repulsion,
                       relative_epsilon, convergence_counter, convergence_check_freq, preserve_order, rng=None,
                       unit_draw=None):
        """`fold` with the start positions' random numbers either drawn here (rng) or handed in (unit_draw: the
        (ndim, n - 1) array rng.random would have returned at this point of the stream -- uniform(0, 2a) is
        2a * random(), bit for bit), so that many folds can be built side by side after one sequential pass
        over the random stream (likelihood_sweep)."""
        n = self.n
        order, degrees, ei, ej, ed, et, hi, hj, ht, vmax = _native.cv_fold(
            self.cells(), picks, preserve_order, self.m.names is not None)
        if ei.shape[0] == 0:
            raise ValueError("No valid off-diagonal measurements found in dissimilarity matrix")
        init_step = vmax / n
        if unit_draw is None:
            steps = rng.uniform(0.0, 2.0 * init_step, size=(int(ndim), n - 1)).T
        else:
            steps = (0.0 + (2.0 * init_step - 0.0) * unit_draw).T      # Generator.uniform's own arithmetic
        init = np.vstack([np.zeros((1, int(ndim))), np.cumsum(steps, axis=0)])
        names = self.m.names
        if names is not None and order is not None:
            names = [names[q] for q in order]
        call = SparseCall(
            initial_positions=np.ascontiguousarray(init, dtype=np.float64), degrees=degrees,
            edge_i=ei, edge_j=ej, edge_dist=ed, edge_thresh=et,
            n_iter=int(mapping_max_iter), k0=float(k0), cooling_rate=float(cooling_rate),
            c_repulsion=float(c_repulsion), relative_epsilon=float(relative_epsilon),
            convergence_window=int(convergence_counter),
            convergence_check_freq=int(convergence_check_freq), names=names, order=order)
        return call, (hi, hj, ht)

    def fold_numpy(self, picks: np.ndarray, ndim: int, mapping_max_iter, k0, cooling_rate, c_repulsion,
                   relative_epsilon, convergence_counter, convergence_check_freq, preserve_order, rng):
        """`fold` in NumPy (the cross-check of the library routine)."""
        n = self.n
        r, c = picks % n, picks // n
        lin = np.unique(np.concatenate([r + c * n, c + r * n]))     # every cell once
        at = self.pos_of[lin]
        at = at[at >= 0]
        keep = np.ones(self.rows.shape[0], dtype=bool)
        keep[at] = False
        dr, dc = self.rows[at], self.cols[at]
        order = None
        if n > 1 and not preserve_order:
            order = self._order(dr, dc)
        inv = np.arange(n)
        if order is not None:
            inv = np.empty(n, dtype=np.int64)
            inv[order] = np.arange(n)
        rows, cols = inv[self.rows[keep]], inv[self.cols[keep]]
        vals, codes = self.vals[keep], self.codes[keep]
        degrees = np.bincount(rows, minlength=n).astype(np.int32)
        up = rows < cols
        if not up.any():
            raise ValueError("No valid off-diagonal measurements found in dissimilarity matrix")
        er, ec, ev, ek = rows[up], cols[up], vals[up], codes[up]
        srt = np.lexsort((er, ec))                                   # column-major scan
        numeric = vals[codes == 0]
        init_step = (numeric.max() if numeric.size else np.nan) / n
        steps = rng.uniform(0.0, 2.0 * init_step, size=(int(ndim), n - 1)).T
        init = np.vstack([np.zeros((1, int(ndim))), np.cumsum(steps, axis=0)])
        names = self.m.names
        if names is not None and order is not None:
            names = [names[q] for q in order]
        call = SparseCall(
            initial_positions=np.ascontiguousarray(init, dtype=np.float64), degrees=degrees,
            edge_i=er[srt].astype(np.int32), edge_j=ec[srt].astype(np.int32),
            edge_dist=ev[srt].astype(np.float64), edge_thresh=ek[srt].astype(np.int32),
            n_iter=int(mapping_max_iter), k0=float(k0), cooling_rate=float(cooling_rate),
            c_repulsion=float(c_repulsion), relative_epsilon=float(relative_epsilon),
            convergence_window=int(convergence_counter),
            convergence_check_freq=int(convergence_check_freq), names=names, order=order)
        # out-of-sample cells: held out AND numeric in the truth (as.numeric drops thresholds)
        numeric_truth = self.codes[at] == 0
        # error_calculator_comparison lines the prediction up with the truth BY NAME
        # (R/error_metrics.R:100-112); an unnamed matrix is compared cell by cell as it comes back,
        # i.e. in the reordered numbering -- reproduced
        to_pred = inv if self.m.names is not None else np.arange(n)
        hold = (to_pred[dr[numeric_truth]].astype(np.int32), to_pred[dc[numeric_truth]].astype(np.int32),
                self.vals[at][numeric_truth].astype(np.float64))
        return call, hold


def _fold_workers() -> int:
    """Threads that build folds side by side: the cores this process may use, at most 16."""
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(16, os.cpu_count() or 1))


def build_fold_calls(m, builder, param_sets, folds, rng, mapping_max_iter, relative_epsilon, preserve_order,
                     parallel: bool, convergence_counter: int = 5):
    """All folds of all parameter sets: (calls, owners, masked inputs, holdouts).  One sequential pass over the
    random stream (fold picks, then each fold's start-position draws, in the order the reference's loop consumes
    them); parallel = True: the list work of the folds (topolow_cv_fold, which releases the interpreter lock) then
    runs on a thread pool.  A fold that fails draws nothing in the sequential order, so a failure inside the pool
    makes the caller redo the pass with parallel = False from the same stream state (likelihood_sweep)."""
    tiny = core.CodedMatrix(np.array([[0.0, 1.0], [1.0, 0.0]]), np.zeros((2, 2), dtype=np.int32))
    calls, owners, inputs, holds, jobs = [], [], [], [], []
    n_pts = m.values.shape[0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for s_idx, ps in enumerate(param_sets):
            fold_sets = builder.folds(folds, rng) if builder is not None else make_folds(m.values, folds, rng)
            set_ok = True
            if builder is not None:
                try:    # the parameter checks of R/core.R:202-264, once per set instead of once per fold
                    core._validate(tiny, int(ps["N"]), mapping_max_iter, ps["k0"], ps["cooling_rate"],
                                   ps["c_repulsion"], relative_epsilon, 5, 3, None)
                except ValueError:
                    set_ok = False
            for h in fold_sets:
                masked, hold, call = None, None, None
                try:
                    if builder is None:
                        masked = m.masked(h % n_pts, h // n_pts)
                        call = core.prepare_layout_call(masked, int(ps["N"]), mapping_max_iter, ps["k0"],
                                                        ps["cooling_rate"], ps["c_repulsion"], relative_epsilon,
                                                        convergence_counter, None, False, 3, preserve_order, rng)
                    elif set_ok and parallel:
                        jobs.append((len(calls), h, ps, rng.random((int(ps["N"]), n_pts - 1))))
                    elif set_ok:
                        call, hold = builder.fold(h, int(ps["N"]), mapping_max_iter, ps["k0"],
                                                  ps["cooling_rate"], ps["c_repulsion"], relative_epsilon,
                                                  convergence_counter, 3, preserve_order, rng)
                except ValueError:
                    call = None  # the reference's tryCatch turns a failed fold into an NA row
                calls.append(call)
                owners.append(s_idx)
                inputs.append(masked)
                holds.append(hold)
        if jobs:
            builder.cells()

            def one(job):
                q, h, ps, u = job
                return q, builder.fold_from_draw(h, int(ps["N"]), mapping_max_iter, ps["k0"], ps["cooling_rate"],
                                                 ps["c_repulsion"], relative_epsilon, convergence_counter, 3,
                                                 preserve_order, unit_draw=u)
            with ThreadPoolExecutor(max_workers=_fold_workers()) as pool:
                for q, (call, hold) in pool.map(one, jobs):
                    calls[q], holds[q] = call, hold
    return calls, owners, inputs, holds


def _pooled(per_set):
    out = []
    for rows in per_set:
        rows = [r for r in rows if r["n_samples"] > 0]
        if not rows:
            out.append(dict(Holdout_MAE=math.nan, NLL=math.nan, mean_iter=math.nan, pct_converged=math.nan, fold_mae=[]))
            continue
        total = sum(r["n_samples"] for r in rows)
        tot_err = sum(r["sum_abs_errors"] for r in rows)
        mae = tot_err / total if total > 0 else math.nan
        nll = total * (1 + math.log(2 * mae)) if not math.isnan(mae) and mae > 0 else math.nan
        out.append(dict(Holdout_MAE=mae, NLL=nll, mean_iter=float(np.mean([r["iter"] for r in rows])),
                        pct_converged=100.0 * float(np.mean([r["converged"] for r in rows])),
                        fold_mae=[r["sum_abs_errors"] / r["n_samples"] for r in rows]))
    return out


def _sweep_in_the_library(m, builder, param_sets, folds, rng, mapping_max_iter, relative_epsilon, preserve_order,
                          precision, convergence_counter: int = 5):
    """The sweep as ONE library call.  Draws from `rng` exactly what the fold-by-fold loop draws when no fold fails
    (per set: the fold picks, then per fold its start positions' numbers; then one seed per fold); returns None --
    with the stream spent, the caller rewinds it -- when a fold has no valid measurements, because such a fold draws
    nothing in the reference's order."""
    tiny = core.CodedMatrix(np.array([[0.0, 1.0], [1.0, 0.0]]), np.zeros((2, 2), dtype=np.int32))
    n = m.values.shape[0]
    owners, picks, draws, nd, k0, cr, cp = [], [], [], [], [], [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for s_idx, ps in enumerate(param_sets):
            fold_sets = builder.folds(folds, rng)
            try:    # the parameter checks of R/core.R:202-264, once per set
                core._validate(tiny, int(ps["N"]), mapping_max_iter, ps["k0"], ps["cooling_rate"], ps["c_repulsion"],
                               relative_epsilon, 5, 3, None)
            except ValueError:
                continue
            for h in fold_sets:
                owners.append(s_idx)
                picks.append(h)
                draws.append(rng.random((int(ps["N"]), n - 1)))
                nd.append(int(ps["N"])); k0.append(float(ps["k0"])); cr.append(float(ps["cooling_rate"]))
                cp.append(float(ps["c_repulsion"]))
    seeds = [int(rng.integers(0, 2 ** 63 - 1)) for _ in picks]
    hsum, hcnt, its, conv, ec, secs = _native.cv_sweep(builder.cells(), m.names is not None, preserve_order, nd, k0, cr, cp,
                                                       picks, draws, seeds, mapping_max_iter, relative_epsilon,
                                                       convergence_counter, 3, precision)
    if np.any(ec == _native.ERR_BAD_ARGUMENT):
        return None
    per_set: List[List[dict]] = [[] for _ in param_sets]
    for f, owner in enumerate(owners):
        if ec[f] != _native.OK:
            continue
        per_set[owner].append(dict(n_samples=int(hcnt[f]), sum_abs_errors=float(hsum[f]), iter=int(its[f]),
                                   converged=int(conv[f])))
    return _pooled(per_set), secs, len(picks)


def likelihood_sweep(dissimilarity_matrix, param_sets: Sequence[Dict[str, float]], mapping_max_iter: int,
                     relative_epsilon: float, folds: int = 20, preserve_order: bool = False,
                     rng: Optional[np.random.Generator] = None, precision: str = "f64",
                     path: str = "sparse", convergence_counter: int = 5):
    """`likelihood_function` for MANY parameter sets at once: all folds of all sets are relaxed
    in ONE batched launch, and the held-out cells are scored on the device (no est_distances, no
    n x n arrays).  param_sets: dicts with N (ndim), k0, cooling_rate, c_repulsion.
    path = "sparse" (default) hands the whole sweep to the library in one call (topolow_cv_sweep: folds built on host
    threads from the picks and the start positions' unit draws, one batch, only the scores come back);
    "sparse-calls" builds one call object per fold here and relaxes them with optimize_layout_exact_batch -- same
    draws, same seeds, same numbers (tests/test_gpu_assays.py);
    path = "dense" runs the reference's own sequence per fold instead (masked n x n matrix ->
    prepare_layout_call -> est_distances -> error_calculator_comparison): same folds, same start
    positions, same pooled numbers; kept as the cross-check.
    convergence_counter: 5 is what likelihood_function passes (R/adaptive_sampling.R:2620-2631); the reference's
    notebooks run the same folds with 3 (inst/examples/methods-comparison-h3n2-hiv-denv.Rmd:1894-1902).
    Each result dict carries likelihood_function's four fields plus `fold_mae` (mean |error| per fold).
    Returns (list of result dicts, device_seconds, embeddings)."""
    rng = rng if rng is not None else _native.host_rng()
    if not hasattr(rng, "choice"):   # R-stream generator: fold sampling uses a NumPy stream seeded from it
        rng = np.random.default_rng(rng.integers(0, 2 ** 53))
    m = core.coded_matrix(dissimilarity_matrix)   # strings are parsed once, not once per fold
    if m is None:
        raise ValueError("dissimilarity_matrix must be a matrix")
    if path not in ("sparse", "sparse-calls", "dense"):
        raise ValueError("path must be 'sparse', 'sparse-calls' or 'dense'")
    builder = FoldBuilder(m) if path != "dense" else None
    if builder is not None:      # the matrix half of R/core.R:202-264 once; per set only the parameters
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            core._validate(m, 2, 1, 1.0, 0.5, 1.0, 1.0, 1, 1, None)
    state0 = rng.bit_generator.state
    if path == "sparse":
        fused = _sweep_in_the_library(m, builder, param_sets, folds, rng, mapping_max_iter, relative_epsilon,
                                      preserve_order, precision, convergence_counter)
        if fused is not None:
            return fused
        rng.bit_generator.state = state0      # a fold failed: the fold-by-fold order of draws decides (below)
    try:
        calls, owners, inputs, holds = build_fold_calls(
            m, builder, param_sets, folds, rng, mapping_max_iter, relative_epsilon, preserve_order,
            parallel=builder is not None and len(param_sets) * folds >= 16, convergence_counter=convergence_counter)
    except Exception:
        rng.bit_generator.state = state0
        calls, owners, inputs, holds = build_fold_calls(
            m, builder, param_sets, folds, rng, mapping_max_iter, relative_epsilon, preserve_order, parallel=False,
            convergence_counter=convergence_counter)
    live = [q for q, c in enumerate(calls) if c is not None]
    seeds = [int(rng.integers(0, 2 ** 63 - 1)) for _ in live]
    results, secs = ([], 0.0)
    if live:
        results, secs = _native.optimize_layout_exact_batch(
            [calls[q] for q in live], seeds=seeds, precision=precision,
            holdouts=[holds[q] for q in live] if builder is not None else None)
    it = iter(results)
    per_set: List[List[dict]] = [[] for _ in param_sets]
    for call, owner, masked in zip(calls, owners, inputs):
        if call is None:
            continue
        res = next(it)
        if isinstance(res, Exception):
            continue
        if builder is not None:
            n_samples, sum_abs = int(res.info["holdout_count"]), float(res.info["holdout_sum_abs"])
        else:
            p = res.positions
            diff = p[:, None, :] - p[None, :, :]
            est = np.sqrt((diff * diff).sum(-1))
            err = error_calculator_comparison(est, m, masked, pred_names=call.names, true_names=m.names)
            oe = err["OutSampleError"]
            oe = oe[~np.isnan(oe)]
            n_samples, sum_abs = int(oe.size), float(np.abs(oe).sum())
        per_set[owner].append(dict(n_samples=n_samples, sum_abs_errors=sum_abs,
                                   iter=res.iterations, converged=int(res.converged)))
    out = _pooled(per_set)
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
    return {k: v for k, v in res[0].items() if k != "fold_mae"}      # the reference's four fields
