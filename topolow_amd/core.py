"""Host driver of the relaxation path: a Python mirror of the reference's R driver.

Mirrors, step by step, `euclidean_embedding()` (R/core.R:184-528 of the reference),
`create_topolow_map()` (R/core.R:616-664) and the `topolow` S3 methods (R/core.R:684-719):
same argument names, defaults, validation messages and returned fields.  The native step
-- the reference's `.Call("_topolow_optimize_layout_exact_cpp", ...)` at R/core.R:439-456 --
goes to the HIP library through :mod:`topolow_amd._native`; there is no CPU fallback.

R matrices map to Python as follows:
  * numeric matrix with NA      -> 2-D float ndarray, NaN = NA
  * character matrix (">5",...) -> 2-D object/str ndarray, None/NaN = NA
  * dimnames                    -> a pandas DataFrame's index, or the `names=` of `RMatrix`
"""
from __future__ import annotations

import math
import os
import re
import warnings
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

_MISSING = object()


# --------------------------------------------------------------------------------------
# R-matrix adaptor
# --------------------------------------------------------------------------------------
@dataclass
class RMatrix:
    """A matrix with optional row names (R's `matrix` + `rownames`)."""
    values: np.ndarray
    names: Optional[List[str]] = None


def _as_rmatrix(x: Any) -> Optional[RMatrix]:
    """Return an RMatrix if `x` is matrix-like in the R sense (`is.matrix`), else None."""
    if isinstance(x, RMatrix):
        return RMatrix(np.asarray(x.values), list(x.names) if x.names is not None else None)
    if hasattr(x, "to_numpy") and hasattr(x, "index") and hasattr(x, "columns"):
        return RMatrix(x.to_numpy(), [str(v) for v in x.index])
    if isinstance(x, np.ndarray) and x.ndim == 2:
        return RMatrix(x, None)
    return None


def _is_character(v: np.ndarray) -> bool:
    return v.dtype.kind in ("U", "S", "O")


def _is_na_cell(c: Any) -> bool:
    if c is None:
        return True
    if isinstance(c, float) and math.isnan(c):
        return True
    if isinstance(c, str) and c == "NA":
        return True
    return False


_NUM_RE = re.compile(r"^\s*[-+]?(?:(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?|Inf|inf|NaN|nan)\s*$")


def _as_numeric_scalar(c: Any) -> float:
    """R's as.numeric() on one cell: unparsable strings become NA (NaN here)."""
    if _is_na_cell(c):
        return math.nan
    if isinstance(c, (int, float, np.integer, np.floating)):
        return float(c)
    s = str(c)
    if _NUM_RE.match(s):
        return float(s)
    return math.nan


def _as_numeric(v: np.ndarray) -> np.ndarray:
    if not _is_character(v):
        return np.asarray(v, dtype=np.float64)
    out = np.empty(v.shape, dtype=np.float64)
    flat = out.reshape(-1)
    for q, c in enumerate(v.reshape(-1)):
        flat[q] = _as_numeric_scalar(c)
    return out


def _is_na(v: np.ndarray) -> np.ndarray:
    if not _is_character(v):
        return np.isnan(np.asarray(v, dtype=np.float64))
    out = np.empty(v.shape, dtype=bool)
    flat = out.reshape(-1)
    for q, c in enumerate(v.reshape(-1)):
        flat[q] = _is_na_cell(c)
    return out


@dataclass
class CodedMatrix:
    """Numeric twin of R's character dissimilarity matrix: `values` holds the number of every
    cell with any "<"/">" prefix stripped (NaN = NA), `codes` the prefix (0 none, 1 ">", -1 "<").
    All of the driver's matrix logic runs on this form; strings are parsed once, up front."""
    values: np.ndarray
    codes: np.ndarray
    names: Optional[List[str]] = None
    character: bool = False   # was the source a character matrix? (is.character(), R/core.R:278)

    def as_numeric(self) -> np.ndarray:
        """as.numeric(matrix): threshold strings become NA."""
        return np.where(self.codes == 0, self.values, np.nan)

    def reordered(self, order: np.ndarray) -> "CodedMatrix":
        ix = np.ix_(order, order)
        return CodedMatrix(self.values[ix], self.codes[ix],
                           [self.names[q] for q in order] if self.names is not None else None,
                           self.character)

    def masked(self, rows: np.ndarray, cols: np.ndarray) -> "CodedMatrix":
        v, c = self.values.copy(), self.codes.copy()
        v[rows, cols] = np.nan
        v[cols, rows] = np.nan
        c[rows, cols] = 0
        c[cols, rows] = 0
        return CodedMatrix(v, c, self.names, self.character)


def coded_matrix(x: Any) -> Optional[CodedMatrix]:
    """Any accepted matrix-like -> CodedMatrix (None if `x` is not a matrix in R's sense)."""
    if isinstance(x, CodedMatrix):
        return x
    m = _as_rmatrix(x)
    if m is None:
        return None
    v = m.values
    if not _is_character(v):
        vals = np.array(v, dtype=np.float64)
        return CodedMatrix(vals, np.zeros(vals.shape, dtype=np.int8), m.names, False)
    vals = np.full(v.shape, np.nan)
    codes = np.zeros(v.shape, dtype=np.int8)
    fv, fc = vals.reshape(-1), codes.reshape(-1)
    for q, c in enumerate(v.reshape(-1)):
        if _is_na_cell(c):
            continue
        if isinstance(c, str):
            if c.startswith(">"):
                fc[q] = 1
                fv[q] = _as_numeric_scalar(c[1:])
            elif c.startswith("<"):
                fc[q] = -1
                fv[q] = _as_numeric_scalar(c[1:])
            else:
                fv[q] = _as_numeric_scalar(c)
        else:
            fv[q] = _as_numeric_scalar(c)
    return CodedMatrix(vals, codes, m.names, True)


# --------------------------------------------------------------------------------------
# result object (R/core.R:505-527) and its S3 methods (R/core.R:684-719)
# --------------------------------------------------------------------------------------
@dataclass
class Topolow:
    positions: np.ndarray
    est_distances: np.ndarray
    mae: float
    iter: int
    parameters: Dict[str, Any]
    convergence: Dict[str, Any]
    names: Optional[List[str]] = None
    # side channel (not part of the reference object): timing / schedule of the native run
    native_info: Dict[str, Any] = field(default_factory=dict, repr=False)

    r_class = "topolow"

    def __getitem__(self, key: str):
        if key in ("positions", "est_distances", "mae", "iter", "parameters", "convergence"):
            return getattr(self, key)
        raise KeyError(key)

    def keys(self):
        return ["positions", "est_distances", "mae", "iter", "parameters", "convergence"]

    def __contains__(self, key):
        return key in self.keys()

    def format(self) -> str:
        """Text of print.topolow (R/core.R:684-692)."""
        achieved = "TRUE" if self.convergence["achieved"] else "FALSE"
        return ("topolow optimization result:\n"
                f"Dimensions: {int(self.parameters['ndim'])}\n"
                f"Iterations: {int(self.iter)}\n"
                f"MAE: {self.mae:.4f}\n"
                f"Convergence achieved: {achieved}\n"
                f"Final convergence error: {self.convergence['error']:.4f}\n")

    def __str__(self) -> str:
        return self.format()

    def summary(self) -> str:
        """Text of summary.topolow (R/core.R:713-719)."""
        return (self.format() + "\nParameters:\n"
                f"k0: {self.parameters['k0']:.4f}\n"
                f"cooling_rate: {self.parameters['cooling_rate']:.4f}\n"
                f"c_repulsion: {self.parameters['c_repulsion']:.4f}\n")


def print_topolow(x: Topolow) -> Topolow:
    print(x.format(), end="")
    return x


def summary_topolow(x: Topolow) -> None:
    print(x.summary(), end="")


# --------------------------------------------------------------------------------------
# the .Call payload
# --------------------------------------------------------------------------------------
@dataclass
class LayoutCall:
    """Arguments of `optimize_layout_exact_cpp` exactly as R/core.R:439-456 passes them."""
    initial_positions: np.ndarray      # n x ndim float64
    dissimilarity_matrix: np.ndarray   # n x n float64, Inf = unmeasured, symmetric
    threshold_matrix: np.ndarray       # n x n int32 {0, 1, -1}, symmetric
    degrees: np.ndarray                # n int32
    edge_i: np.ndarray                 # E int32, 0-based
    edge_j: np.ndarray
    edge_dist: np.ndarray              # E float64
    edge_thresh: np.ndarray            # E int32
    n_iter: int
    k0: float
    cooling_rate: float
    c_repulsion: float
    relative_epsilon: float
    convergence_window: int
    convergence_check_freq: int
    verbose: bool
    # bookkeeping for the post-processing half
    names: Optional[List[str]] = None
    order: Optional[np.ndarray] = None            # permutation applied (None = identity)
    reordered_matrix: Optional[Any] = None         # the (reordered) input as a CodedMatrix


def _stop(msg: str):
    raise ValueError(msg)


def _is_number(x: Any) -> bool:
    return isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, bool)


def _validate(m: Optional[CodedMatrix], ndim, mapping_max_iter, k0, cooling_rate, c_repulsion,
              relative_epsilon, convergence_counter, convergence_check_freq,
              initial_positions) -> None:
    """R/core.R:202-264, messages verbatim."""
    if m is None:
        _stop("dissimilarity_matrix must be a matrix")
    v = m.values
    if v.shape[0] != v.shape[1]:
        _stop("dissimilarity_matrix must be square")
    finite = m.as_numeric().copy()
    finite[np.isinf(finite)] = np.nan
    if int(np.sum(~np.isnan(finite) & (finite != 0))) == 0:
        warnings.warn("No finite non-zero dissimilarities found. Results may be unreliable.",
                      UserWarning, stacklevel=3)
    if not _is_number(ndim) or ndim < 1 or ndim != round(ndim):
        _stop("ndim must be a positive integer")
    if not _is_number(mapping_max_iter) or mapping_max_iter < 1 or \
            mapping_max_iter != round(mapping_max_iter):
        _stop("mapping_max_iter must be a positive integer")
    if not _is_number(k0) or k0 <= 0:
        _stop("k0 must be a positive number")
    if k0 > 30:
        warnings.warn("High k0 value (> 30) may lead to instability", UserWarning, stacklevel=3)
    if not _is_number(cooling_rate) or cooling_rate <= 0 or cooling_rate >= 1:
        _stop("cooling_rate must be between 0 and 1")
    if not _is_number(c_repulsion) or c_repulsion <= 0:
        _stop("c_repulsion must be a positive number")
    if not _is_number(relative_epsilon) or relative_epsilon <= 0:
        _stop("relative_epsilon must be a positive number")
    if not _is_number(convergence_counter) or convergence_counter < 1 or \
            convergence_counter != round(convergence_counter):
        _stop("convergence_counter must be a positive integer")
    if not _is_number(convergence_check_freq) or convergence_check_freq < 1:
        _stop("convergence_check_freq must be a positive integer")
    if initial_positions is not None:
        ip = _as_rmatrix(initial_positions)
        if ip is None:
            _stop("initial_positions must be a matrix")
        if ip.values.shape[0] != v.shape[0]:
            _stop("initial_positions must have same number of rows as dissimilarity_matrix")
        if ip.values.shape[1] != ndim:
            _stop("initial_positions must have ndim columns")
    if v.shape[0] < 2:
        _stop("dissimilarity_matrix must have at least 2 rows/columns")


def spectral_order(stripped: np.ndarray) -> Optional[np.ndarray]:
    """R/core.R:269-319: ascending order of each point's mean dissimilarity
    (mean of row mean and column mean over non-NA off-diagonal cells; threshold prefixes
    stripped -- `stripped` is CodedMatrix.values).  Returns None where the reference keeps
    the input order."""
    try:
        numeric = np.array(stripped, dtype=np.float64)
        np.fill_diagonal(numeric, np.nan)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            row_means = np.nanmean(numeric, axis=1)
            col_means = np.nanmean(numeric, axis=0)
        avg = (row_means + col_means) / 2.0
        avg[np.isnan(avg)] = 0.0
        if int(np.sum(avg > 0)) > 1:
            return np.argsort(avg, kind="stable")
    except Exception:  # the reference wraps this block in tryCatch and skips reordering
        return None
    return None


def prepare_layout_call(dissimilarity_matrix, ndim, mapping_max_iter, k0, cooling_rate,
                        c_repulsion, relative_epsilon, convergence_counter, initial_positions,
                        verbose, convergence_check_freq, preserve_order,
                        rng: Optional[np.random.Generator] = None) -> LayoutCall:
    """Everything `euclidean_embedding` does before the `.Call` (R/core.R:202-436)."""
    m = coded_matrix(dissimilarity_matrix)
    _validate(m, ndim, mapping_max_iter, k0, cooling_rate, c_repulsion, relative_epsilon,
              convergence_counter, convergence_check_freq, initial_positions)
    names = m.names
    n = m.values.shape[0]
    ndim = int(ndim)

    # -- reordering (R/core.R:269-322)
    order = None
    if n > 1 and not preserve_order:
        order = spectral_order(m.values)
        if order is not None:
            m = m.reordered(order)
            names = m.names
            if verbose:
                print("Matrix reordered for spectral pattern (largest values in corners)")
        elif verbose:
            print("Insufficient data for meaningful spectral ordering")
    elif preserve_order and verbose:
        print("Preserving original row/column order (preserve_order = TRUE)")
    v = m

    # -- initial positions follow the matrix only through row names (R/core.R:325-333)
    init = None
    if initial_positions is not None:
        ip = _as_rmatrix(initial_positions)
        init = np.asarray(ip.values, dtype=np.float64)
        if ip.names is not None and names is not None and list(ip.names) != list(names):
            lookup = {nm: q for q, nm in enumerate(ip.names)}
            try:
                init = init[[lookup[nm] for nm in names], :]
            except KeyError:
                raise IndexError("subscript out of bounds") from None

    # -- degrees and parsing (R/core.R:340-374)
    non_na = ~np.isnan(m.values)
    degrees = non_na.sum(axis=1).astype(np.int32)
    distances = np.where(non_na, m.values, np.inf)
    codes = np.where(non_na, m.codes, 0).astype(np.int32)

    # -- COO edge list, upper triangle, column-major scan like which(arr.ind=TRUE)
    #    (R/core.R:383-402)
    with np.errstate(invalid="ignore"):
        valid = np.triu(np.ones((n, n), dtype=bool), k=1) & (distances != np.inf)
    cols, rows = np.nonzero(valid.T)  # column-major enumeration
    if rows.shape[0] == 0:
        _stop("No valid off-diagonal measurements found in dissimilarity matrix")
    edge_i = rows.astype(np.int32)
    edge_j = cols.astype(np.int32)
    edge_dist = distances[rows, cols].astype(np.float64)
    edge_thresh = codes[rows, cols].astype(np.int32)

    # -- initial positions (R/core.R:407-415)
    if init is None:
        numeric = m.as_numeric()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            init_step = np.nanmax(numeric) / n
        gen = rng if rng is not None else np.random.default_rng()
        # runif fills the (n-1) x ndim matrix column by column
        steps = gen.uniform(0.0, 2.0 * init_step, size=(ndim, n - 1)).T
        init = np.vstack([np.zeros((1, ndim)), np.cumsum(steps, axis=0)])

    # -- symmetric dense fill: lower triangle <- transpose of upper (R/core.R:429-436)
    low = np.tril(np.ones((n, n), dtype=bool), k=-1)
    dense = distances.copy()
    dense[low] = distances.T[low]
    tdense = codes.copy()
    tdense[low] = codes.T[low]

    return LayoutCall(
        initial_positions=np.ascontiguousarray(init, dtype=np.float64),
        dissimilarity_matrix=dense, threshold_matrix=tdense, degrees=degrees,
        edge_i=edge_i, edge_j=edge_j, edge_dist=edge_dist, edge_thresh=edge_thresh,
        n_iter=int(mapping_max_iter), k0=float(k0), cooling_rate=float(cooling_rate),
        c_repulsion=float(c_repulsion), relative_epsilon=float(relative_epsilon),
        convergence_window=int(convergence_counter),
        convergence_check_freq=int(convergence_check_freq), verbose=bool(verbose),
        names=names, order=order, reordered_matrix=v)


def post_mae(reordered_matrix, est_distances: np.ndarray) -> float:
    """R/core.R:479-481: mean |as.numeric(D) - est| over every non-NA cell."""
    raw = coded_matrix(reordered_matrix).as_numeric()
    valid = ~np.isnan(raw)
    if not valid.any():
        return float("nan")
    return float(np.mean(np.abs(raw[valid] - est_distances[valid])))


def _fmt_csv_number(x: float) -> str:
    return repr(float(f"{x:.15g}")) if math.isfinite(x) else ("NA" if math.isnan(x) else
                                                               ("Inf" if x > 0 else "-Inf"))


def write_positions_csv(path: str, positions: np.ndarray, names: Optional[Sequence[str]]):
    """utils::write.csv(positions, row.names=TRUE) of an unnamed-column matrix."""
    n, d = positions.shape
    with open(path, "w") as fh:
        fh.write(",".join(['""'] + [f'"V{c + 1}"' for c in range(d)]) + "\n")
        for r in range(n):
            label = names[r] if names is not None else str(r + 1)
            fh.write(",".join([f'"{label}"'] + [_fmt_csv_number(x) for x in positions[r]]) + "\n")


# --------------------------------------------------------------------------------------
# public entry points
# --------------------------------------------------------------------------------------
def _finish(call: LayoutCall, native_result, ndim, k0, cooling_rate, c_repulsion,
            write_positions_to_csv, output_dir, verbose, pdist_fn) -> Topolow:
    positions = np.asarray(native_result.positions, dtype=np.float64)
    est = pdist_fn(positions)
    mae = post_mae(call.reordered_matrix, est)
    if write_positions_to_csv:
        if output_dir is None or output_dir is _MISSING:
            raise ValueError("An 'output_dir' must be provided when 'write_positions_to_csv' "
                             "is TRUE.")
        os.makedirs(output_dir, exist_ok=True)
        fname = "Positions_dim_%d_k0_%.4f_cooling_%.4f_c_repulsion_%.4f.csv" % (
            int(ndim), k0, cooling_rate, c_repulsion)
        full = os.path.join(output_dir, fname)
        write_positions_csv(full, positions, call.names)
        if verbose:
            print("Positions saved to:", full)
    return Topolow(
        positions=positions, est_distances=est, mae=mae, iter=int(native_result.iterations),
        parameters=dict(ndim=ndim, k0=k0, cooling_rate=cooling_rate, c_repulsion=c_repulsion,
                        method="cpp_exact_full_pairwise"),
        convergence=dict(achieved=bool(native_result.converged),
                         error=float(native_result.final_mae),
                         final_k=float(native_result.final_k)),
        names=call.names, native_info=dict(getattr(native_result, "info", {}) or {}))


def _embed_with(native_fn, pdist_fn, dissimilarity_matrix, ndim, mapping_max_iter, k0,
                cooling_rate, c_repulsion, relative_epsilon, convergence_counter,
                initial_positions, write_positions_to_csv, output_dir, verbose,
                convergence_check_freq, preserve_order, rng=None) -> Topolow:
    for nm, val in (("k0", k0), ("cooling_rate", cooling_rate), ("c_repulsion", c_repulsion)):
        if val is _MISSING:
            raise TypeError(f'argument "{nm}" is missing, with no default')
    call = prepare_layout_call(dissimilarity_matrix, ndim, mapping_max_iter, k0, cooling_rate,
                               c_repulsion, relative_epsilon, convergence_counter,
                               initial_positions, verbose, convergence_check_freq,
                               preserve_order, rng)
    if verbose:
        print("Starting C++ optimization...")
    import time
    t0 = time.time()
    res = native_fn(call)
    if verbose:
        print("Optimization finished in %.2f seconds." % (time.time() - t0))
    return _finish(call, res, ndim, k0, cooling_rate, c_repulsion, write_positions_to_csv,
                   output_dir, verbose, pdist_fn)


def euclidean_embedding(dissimilarity_matrix, ndim, mapping_max_iter=1000, k0=_MISSING,
                        cooling_rate=_MISSING, c_repulsion=_MISSING, relative_epsilon=1e-4,
                        convergence_counter=5, initial_positions=None,
                        write_positions_to_csv=False, output_dir=_MISSING, verbose=False,
                        convergence_check_freq=3, preserve_order=False) -> Topolow:
    """Drop-in for the reference's `euclidean_embedding()` (R/core.R:184-197); the native
    relaxation runs on the MI355X through libtopolow_relax.so (no CPU fallback)."""
    from . import _native
    return _embed_with(_native.optimize_layout_exact, _native.est_distances,
                       dissimilarity_matrix, ndim, mapping_max_iter, k0, cooling_rate,
                       c_repulsion, relative_epsilon, convergence_counter, initial_positions,
                       write_positions_to_csv, output_dir, verbose, convergence_check_freq,
                       preserve_order, _native.host_rng())


def create_topolow_map(distance_matrix, ndim, mapping_max_iter=1000, k0=_MISSING,
                       cooling_rate=_MISSING, c_repulsion=_MISSING, relative_epsilon=1e-4,
                       convergence_counter=3, initial_positions=None,
                       write_positions_to_csv=False, output_dir=_MISSING,
                       verbose=False) -> Topolow:
    """Deprecated alias (R/core.R:616-664): warns, then forwards with convergence_counter
    defaulting to 3 and the default convergence_check_freq / preserve_order."""
    warnings.warn("`create_topolow_map()` was deprecated in topolow 2.0.0.\n"
                  "Please use `euclidean_embedding()` instead.\n"
                  "i The new function provides the same functionality with improvements:\n"
                  "* Parameter name: 'distance_matrix' --> 'dissimilarity_matrix'\n"
                  "* Enhanced matrix reordering for better optimization",
                  DeprecationWarning, stacklevel=2)
    if write_positions_to_csv and output_dir is _MISSING:
        output_dir = os.getcwd()
        warnings.warn("output_dir not specified, using current working directory", UserWarning,
                      stacklevel=2)
    return euclidean_embedding(
        dissimilarity_matrix=distance_matrix, ndim=ndim, mapping_max_iter=mapping_max_iter,
        k0=k0, cooling_rate=cooling_rate, c_repulsion=c_repulsion,
        relative_epsilon=relative_epsilon, convergence_counter=convergence_counter,
        initial_positions=initial_positions, write_positions_to_csv=write_positions_to_csv,
        output_dir=output_dir if write_positions_to_csv else None, verbose=verbose)
