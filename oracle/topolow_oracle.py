"""ctypes front end of oracle/libtopolow_oracle.so -- TEST INFRASTRUCTURE ONLY.

The argument list of :func:`optimize_layout_exact` is the reference's `.Call` payload
(R/RcppExports.R:4-6; src/optimization.cpp:109-126) plus the oracle-only switches
documented in topolow_oracle.cpp.  Matrices follow R's layout: column-major float64
positions (n x dim) and dissimilarities (n x n), int32 threshold codes.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtopolow_oracle.so")
_lib = None

ORDER_SEEDED = 0   # std::shuffle, mt19937(seed)
ORDER_RANDOM_DEVICE = 1  # std::shuffle, mt19937(random_device()) -- the reference verbatim
ORDER_SUPPLIED = 2  # pair order supplied per iteration
ORDER_NATURAL = 3  # row-wise i<j order, never shuffled

_ORDER_FN = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_int32), C.c_int64, C.c_void_p)


class OracleError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


def build(force: bool = False) -> str:
    """Compile the oracle with the recipe in oracle/Makefile (g++ -O2 -std=gnu++17)."""
    src = os.path.join(_HERE, "topolow_oracle.cpp")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libtopolow_oracle.so"], check=True,
                       capture_output=True)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    lib = C.CDLL(_LIB_PATH)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    lib.topolow_oracle_optimize_layout_exact.restype = C.c_int
    lib.topolow_oracle_optimize_layout_exact.argtypes = [
        dp, C.c_int, C.c_int, dp, ip, ip, ip, ip, dp, ip, C.c_int64,
        C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
        C.c_int, C.c_uint64, _ORDER_FN, C.c_void_p, C.c_int,
        dp, C.POINTER(C.c_int), C.POINTER(C.c_int), dp, dp, dp, C.POINTER(C.c_int),
        C.POINTER(C.c_int), C.c_char_p, C.c_size_t,
    ]
    lib.topolow_oracle_edge_error.restype = C.c_int
    lib.topolow_oracle_edge_error.argtypes = [dp, C.c_int, C.c_int, ip, ip, dp, ip, C.c_int64,
                                              dp, C.POINTER(C.c_int64)]
    lib.topolow_oracle_controller_script.restype = C.c_int
    lib.topolow_oracle_controller_script.argtypes = [
        dp, C.POINTER(C.c_int), dp, C.c_int, C.c_double, C.c_int, C.c_double,
        C.POINTER(C.c_int), C.POINTER(C.c_int), dp, dp, C.POINTER(C.c_int)]
    _lib = lib
    return lib


def _f64(a, order="F"):
    return np.require(np.asarray(a, dtype=np.float64), requirements=["A", "O"] +
                      (["F"] if order == "F" else ["C"]))


def _i32(a, order="F"):
    return np.require(np.asarray(a, dtype=np.int32), requirements=["A", "O"] +
                      (["F"] if order == "F" else ["C"]))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


@dataclass
class OracleResult:
    positions: np.ndarray   # n x dim float64
    converged: bool
    iterations: int         # = best_iter (src/optimization.cpp:373,378)
    final_mae: float
    final_k: float
    mae_trace: np.ndarray   # MAE at every convergence check (oracle-only)
    iters_run: int          # iterations actually executed (oracle-only)


def optimize_layout_exact(initial_positions, dissimilarity_matrix, threshold_matrix, degrees,
                          edge_i, edge_j, edge_dist, edge_thresh, n_iter, k0, cooling_rate,
                          c_repulsion, relative_epsilon, convergence_window,
                          convergence_check_freq, verbose=False, *, seed: int = 0,
                          order_mode: int = ORDER_SEEDED,
                          order_fn: Optional[Callable[[int, np.ndarray], None]] = None,
                          arith: str = "f64") -> OracleResult:
    """Restated `optimize_layout_exact_cpp` (src/optimization.cpp:109-382).

    order_fn(iter, pairs) -- with order_mode=ORDER_SUPPLIED -- must fill the int32 array
    `pairs` of shape (n(n-1)/2, 2) with that iteration's visiting order.
    """
    lib = _load()
    pos0 = _f64(initial_positions)
    if pos0.ndim != 2:
        raise ValueError("initial_positions must be 2-D")
    n, dim = pos0.shape
    D = _f64(dissimilarity_matrix)
    T = _i32(threshold_matrix)
    if n >= 2 and (D.shape != (n, n) or T.shape != (n, n)):
        raise ValueError("dissimilarity/threshold matrices must be n x n")
    deg = _i32(degrees, "C")
    ei = _i32(edge_i, "C")
    ej = _i32(edge_j, "C")
    ed = _f64(edge_dist, "C")
    et = _i32(edge_thresh, "C")
    n_edges = int(ei.shape[0])
    out = np.zeros((n, dim), dtype=np.float64, order="F")
    conv = C.c_int(0)
    iters = C.c_int(0)
    fmae = C.c_double(0.0)
    fk = C.c_double(0.0)
    freq = int(convergence_check_freq)
    max_checks = int(n_iter) // max(1, freq if freq >= 1 else 10) + 2
    trace = np.zeros(max_checks, dtype=np.float64)
    n_checks = C.c_int(0)
    iters_run = C.c_int(0)
    err = C.create_string_buffer(256)

    if order_mode == ORDER_SUPPLIED:
        if order_fn is None:
            raise ValueError("ORDER_SUPPLIED needs order_fn")

        def _cb(it, ptr, npairs, _user):
            arr = np.ctypeslib.as_array(ptr, shape=(int(npairs), 2))
            order_fn(int(it), arr)
        cb = _ORDER_FN(_cb)
    else:
        cb = _ORDER_FN(0)

    rc = lib.topolow_oracle_optimize_layout_exact(
        _dp(pos0), n, dim, _dp(D), _ip(T), _ip(deg), _ip(ei), _ip(ej), _dp(ed), _ip(et), n_edges,
        int(n_iter), float(k0), float(cooling_rate), float(c_repulsion), float(relative_epsilon),
        int(convergence_window), freq, int(bool(verbose)),
        int(order_mode), int(seed) & 0xFFFFFFFFFFFFFFFF, cb, None, 1 if arith == "f32" else 0,
        _dp(out), C.byref(conv), C.byref(iters), C.byref(fmae), C.byref(fk), _dp(trace),
        C.byref(n_checks), C.byref(iters_run), err, len(err))
    if rc != 0:
        raise OracleError(rc, err.value.decode())
    return OracleResult(np.ascontiguousarray(out), bool(conv.value), int(iters.value),
                        float(fmae.value), float(fk.value), trace[: n_checks.value].copy(),
                        int(iters_run.value))


def edge_error(positions, edge_i, edge_j, edge_dist, edge_thresh):
    """(sum, count) of src/optimization.cpp:54-81 on float64 positions (n x dim)."""
    lib = _load()
    pos = _f64(positions)
    n, dim = pos.shape
    ei, ej = _i32(edge_i, "C"), _i32(edge_j, "C")
    ed, et = _f64(edge_dist, "C"), _i32(edge_thresh, "C")
    s = C.c_double(0.0)
    c = C.c_int64(0)
    lib.topolow_oracle_edge_error(_dp(pos), n, dim, _ip(ei), _ip(ej), _dp(ed), _ip(et),
                                  int(ei.shape[0]), C.byref(s), C.byref(c))
    return float(s.value), int(c.value)


def controller_script(mae_seq, iter_seq, k_seq, k0, window, eps):
    """Run the convergence controller (src/optimization.cpp:303-357) on scripted MAEs.

    Returns dict(stopped_at, snapshots, best_mae, best_k, best_iter)."""
    lib = _load()
    m = _f64(mae_seq, "C")
    it = np.require(np.asarray(iter_seq, dtype=np.intc), requirements=["C", "A"])
    ks = _f64(k_seq, "C")
    n_obs = int(m.shape[0])
    stopped = C.c_int(-1)
    snaps = np.zeros(n_obs, dtype=np.intc)
    bm, bk, bi = C.c_double(0), C.c_double(0), C.c_int(0)
    lib.topolow_oracle_controller_script(
        _dp(m), it.ctypes.data_as(C.POINTER(C.c_int)), _dp(ks), n_obs, float(k0), int(window),
        float(eps), C.byref(stopped), snaps.ctypes.data_as(C.POINTER(C.c_int)), C.byref(bm),
        C.byref(bk), C.byref(bi))
    return dict(stopped_at=int(stopped.value), snapshots=snaps.astype(bool),
                best_mae=float(bm.value), best_k=float(bk.value), best_iter=int(bi.value))


def post_metrics(positions, dissimilarity_numeric):
    """Restates R/core.R:474-481: est_distances = as.matrix(dist(positions)); mae = mean
    |D - est| over every cell of the as.numeric() matrix that is not NA (both triangles and
    the diagonal; threshold strings became NA under as.numeric and are excluded)."""
    p = np.asarray(positions, dtype=np.float64)
    diff = p[:, None, :] - p[None, :, :]
    est = np.sqrt((diff * diff).sum(-1))
    d = np.asarray(dissimilarity_numeric, dtype=np.float64)
    valid = ~np.isnan(d)
    mae = float(np.mean(np.abs(d[valid] - est[valid]))) if valid.any() else float("nan")
    return est, mae
