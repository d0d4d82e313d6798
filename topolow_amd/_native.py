"""ctypes binding of libtopolow_relax.so (include/topolow_relax.h) -- the MI355X native path.

This is the Python twin of the R `.Call` shim: it hands the `.Call` payload of
`optimize_layout_exact_cpp` (reference R/RcppExports.R:4-6) to the HIP library.  There is
NO fallback: if the library is missing or no HIP device is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtopolow_relax.so")

SCHEDULE_AUTO, SCHEDULE_SLAB, SCHEDULE_GS = 0, 1, 2
PRECISION_AUTO, PRECISION_F32, PRECISION_F64 = 0, 1, 2

FAR_F32 = 1.0e18   # phantom coordinate of padding points (relax_common.h)

OK = 0
ERR_TOO_FEW_POINTS, ERR_NONFINITE, ERR_BAD_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, \
    ERR_INTERRUPTED = 1, 2, 3, 4, 5, 6, 7


class NativeError(RuntimeError):
    """An error raised by libtopolow_relax (the R shim turns these into R errors)."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


INTERRUPT_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p)
PRINT_CB = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)


class TopolowOptions(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("schedule", C.c_int32), ("precision", C.c_int32),
                ("slab_stages", C.c_int32), ("device", C.c_int32), ("gs_max_n", C.c_int32),
                ("n_devices", C.c_int32), ("keep_labels", C.c_int32), ("reserved", C.c_int32 * 3),
                ("interrupt_cb", INTERRUPT_CB), ("interrupt_user", C.c_void_p),
                ("print_cb", PRINT_CB), ("print_user", C.c_void_p), ("devices", C.POINTER(C.c_int32))]


class TopolowRunStats(C.Structure):
    _fields_ = [("schedule_used", C.c_int32), ("precision_used", C.c_int32),
                ("iterations_run", C.c_int32), ("n_checks", C.c_int32),
                ("device_seconds", C.c_double), ("total_seconds", C.c_double),
                ("stage_launches", C.c_int64), ("setup_seconds", C.c_double), ("reserved", C.c_int64 * 3)]


class TopolowShardStats(C.Structure):
    _fields_ = [("blocks", C.c_int32), ("iterations_run", C.c_int32), ("n_checks", C.c_int32),
                ("groups", C.c_int32), ("loop_seconds", C.c_double), ("total_seconds", C.c_double),
                ("stage_kernel_seconds", C.c_double), ("check_kernel_seconds", C.c_double),
                ("stage_launches", C.c_int64), ("exchanges", C.c_int64), ("warmup_iterations", C.c_int32),
                ("symmetric_segments", C.c_int32), ("timed_seconds", C.c_double), ("reserved", C.c_int64 * 2)]


class TopolowProblem(C.Structure):
    _fields_ = [("initial_positions", C.POINTER(C.c_double)), ("dissimilarity_matrix", C.POINTER(C.c_double)),
                ("threshold_matrix", C.POINTER(C.c_int32)), ("degrees", C.POINTER(C.c_int32)),
                ("edge_i", C.POINTER(C.c_int32)), ("edge_j", C.POINTER(C.c_int32)),
                ("edge_dist", C.POINTER(C.c_double)), ("edge_thresh", C.POINTER(C.c_int32)),
                ("n_edges", C.c_int64), ("n", C.c_int32), ("ndim", C.c_int32), ("n_iter", C.c_int32),
                ("convergence_window", C.c_int32), ("convergence_check_freq", C.c_int32),
                ("reserved0", C.c_int32), ("k0", C.c_double), ("cooling_rate", C.c_double),
                ("c_repulsion", C.c_double), ("relative_epsilon", C.c_double), ("seed", C.c_uint64),
                ("holdout_i", C.POINTER(C.c_int32)), ("holdout_j", C.POINTER(C.c_int32)),
                ("holdout_truth", C.POINTER(C.c_double)), ("n_holdout", C.c_int64)]


class TopolowCellList(C.Structure):
    _fields_ = [("n", C.c_int32), ("reserved", C.c_int32), ("n_cells", C.c_int64),
                ("row", C.POINTER(C.c_int32)), ("col", C.POINTER(C.c_int32)),
                ("value", C.POINTER(C.c_double)), ("code", C.POINTER(C.c_int32)),
                ("pos_of", C.POINTER(C.c_int64)), ("by_row", C.POINTER(C.c_int64)),
                ("row_ptr", C.POINTER(C.c_int64))]


class TopolowResult(C.Structure):
    _fields_ = [("positions_out", C.POINTER(C.c_double)), ("final_mae", C.c_double),
                ("final_k", C.c_double), ("converged", C.c_int32), ("iterations", C.c_int32),
                ("iterations_run", C.c_int32), ("n_checks", C.c_int32), ("error_code", C.c_int32),
                ("error_iteration", C.c_int32), ("holdout_sum_abs", C.c_double),
                ("holdout_count", C.c_int64)]


# Additive backend options (the reference's function signatures stay untouched; this mirrors
# what an R user would set through options(topolow.*)).
options: Dict[str, Any] = dict(seed=None, schedule="auto", precision="auto", slab_stages=0,
                               device=-1, gs_max_n=0)
_host_rng = np.random.default_rng()

_SCHEDULES = {"auto": SCHEDULE_AUTO, "slab": SCHEDULE_SLAB, "gs": SCHEDULE_GS}
_PRECISIONS = {"auto": PRECISION_AUTO, "f32": PRECISION_F32, "f64": PRECISION_F64}


def set_seed(seed: Optional[int]) -> None:
    """The Python spelling of R's `set.seed(seed)`: the random-walk initial positions are then drawn
    from R's own Mersenne-Twister stream (topolow_amd/r_rng.py), i.e. they equal what the reference
    computes after `set.seed(seed)` (R/core.R:412); the native pair/slab order seed is the next draw
    of the same stream, so the whole run is reproducible (the reference's shuffle is not: it is
    seeded from std::random_device, src/optimization.cpp:153-154).  set_seed(None) returns to
    OS entropy."""
    global _host_rng
    options["seed"] = None
    if seed is None:
        _host_rng = np.random.default_rng()
    else:
        from .r_rng import RUnif
        _host_rng = RUnif(int(seed))


def host_rng() -> np.random.Generator:
    return _host_rng


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C topolow_amd/csrc`. topolow_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    lib.topolow_relax_version.restype = C.c_char_p
    lib.topolow_default_options.argtypes = [C.POINTER(TopolowOptions)]
    lib.topolow_optimize_layout_exact.restype = C.c_int
    lib.topolow_optimize_layout_exact.argtypes = [
        dp, C.c_int32, C.c_int32, dp, ip, ip, ip, ip, dp, ip, C.c_int64,
        C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32,
        C.c_int32, C.POINTER(TopolowOptions), dp, ip, ip, dp, dp, C.POINTER(TopolowRunStats),
        C.c_char_p, C.c_size_t]
    lib.topolow_optimize_layout_exact_batch.restype = C.c_int
    lib.topolow_optimize_layout_exact_batch.argtypes = [C.POINTER(TopolowProblem), C.POINTER(TopolowResult),
                                                        C.c_int32, C.c_int32, C.c_int32, dp, C.c_char_p,
                                                        C.c_size_t]
    lib.topolow_cv_fold.restype = C.c_int
    i64p = C.POINTER(C.c_int64)
    lib.topolow_cv_fold.argtypes = [C.POINTER(TopolowCellList), i64p, C.c_int64, C.c_int32, C.c_int32, ip, ip,
                                    ip, ip, dp, ip, i64p, ip, ip, dp, i64p, dp]
    u64p = C.POINTER(C.c_uint64)
    lib.topolow_cv_sweep.restype = C.c_int
    lib.topolow_cv_sweep.argtypes = [C.POINTER(TopolowCellList), C.c_int32, C.c_int32, C.c_int32, ip, dp, dp, dp, i64p, i64p,
                                     dp, i64p, u64p, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     dp, i64p, ip, ip, ip, dp, C.c_char_p, C.c_size_t]
    lib.topolow_session_profile_symmetric.restype = C.c_int
    lib.topolow_session_profile_symmetric.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                                      C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_char_p, C.c_size_t]
    lib.topolow_session_profile_fused.restype = C.c_int
    lib.topolow_session_profile_fused.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_char_p,
                                                  C.c_size_t]
    lib.topolow_session_set_relabel.restype = C.c_int
    lib.topolow_session_set_relabel.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
    lib.topolow_session_labels.restype = C.c_int
    lib.topolow_session_labels.argtypes = [C.c_void_p, ip]
    lib.topolow_session_check_partial.restype = C.c_int
    lib.topolow_session_check_partial.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    lib.topolow_session_can_fuse_checks.restype = C.c_int32
    lib.topolow_session_can_fuse_checks.argtypes = [C.c_void_p]
    lib.topolow_session_stage_fused.restype = C.c_int
    lib.topolow_session_stage_fused.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p,
                                                C.c_char_p, C.c_size_t]
    lib.topolow_session_controller_step.restype = C.c_int
    lib.topolow_session_controller_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double,
                                                    C.c_char_p, C.c_size_t]
    lib.topolow_session_first_nonfinite.restype = C.c_int
    lib.topolow_session_first_nonfinite.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.topolow_shard_rows.restype = C.c_int32
    lib.topolow_shard_rows.argtypes = [C.c_int32, C.c_int32, C.c_int32, ip, ip]
    lib.topolow_optimize_layout_exact_sharded.restype = C.c_int
    lib.topolow_optimize_layout_exact_sharded.argtypes = [
        dp, C.c_int32, C.c_int32, dp, ip, ip, ip, ip, dp, ip, C.c_int64,
        C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32,
        C.c_int32, C.POINTER(TopolowOptions), dp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, dp,
        C.POINTER(TopolowShardStats), C.c_char_p, C.c_size_t]
    lib.topolow_sessions_run_sharded.restype = C.c_int
    lib.topolow_sessions_run_sharded.argtypes = [
        C.POINTER(C.c_void_p), C.c_int32, dp, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
        C.c_int32, C.c_int32, C.c_uint64, C.c_int32, INTERRUPT_CB, C.c_void_p, C.c_int32, dp,
        C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, dp, C.POINTER(TopolowShardStats), C.c_char_p, C.c_size_t]
    lib.topolow_cell_list_index.restype = C.c_int
    lib.topolow_cell_list_index.argtypes = [C.c_int32, C.c_int64, ip, ip, i64p, i64p, i64p]
    lib.topolow_session_check_trace.restype = C.c_int
    lib.topolow_session_check_trace.argtypes = [C.c_void_p, dp, C.c_int32, C.POINTER(C.c_int32)]
    lib.topolow_est_distances.restype = C.c_int
    lib.topolow_est_distances.argtypes = [dp, C.c_int32, C.c_int32, dp, C.c_int32, C.c_char_p,
                                          C.c_size_t]
    lib.topolow_est_distances_rows.restype = C.c_int
    lib.topolow_est_distances_rows.argtypes = [dp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, dp, C.c_int32,
                                               C.c_char_p, C.c_size_t]
    vp = C.c_void_p
    lib.topolow_session_create.restype = C.c_int
    lib.topolow_session_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_size_t]
    lib.topolow_session_destroy.argtypes = [vp]
    lib.topolow_session_destroy.restype = None
    lib.topolow_session_load_dense.restype = C.c_int
    lib.topolow_session_load_dense.argtypes = [vp, dp, ip, ip, C.c_char_p, C.c_size_t]
    lib.topolow_session_load_coo.restype = C.c_int
    lib.topolow_session_load_coo.argtypes = [vp, ip, ip, dp, ip, C.c_int64, ip, C.c_char_p,
                                             C.c_size_t]
    lib.topolow_symm_stage_bounds.restype = C.c_int32
    lib.topolow_symm_stage_bounds.argtypes = [C.c_int32, C.c_int32, ip]
    lib.topolow_symm_stage_order.restype = C.c_int32
    lib.topolow_symm_stage_order.argtypes = [C.c_uint64, C.c_int32, C.c_int32, ip]
    lib.topolow_symm_segment_rows.restype = C.c_int32
    lib.topolow_symm_segment_rows.argtypes = [C.c_int32, C.c_int32, C.c_int32, ip, ip]
    lib.topolow_session_has_thresholds.restype = C.c_int32
    lib.topolow_session_has_thresholds.argtypes = [vp]
    lib.topolow_session_symm_segment_eligible.restype = C.c_int32
    lib.topolow_session_symm_segment_eligible.argtypes = [vp, C.c_int32]
    lib.topolow_session_wait.restype = C.c_int
    lib.topolow_session_wait.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.topolow_session_degree_terms.restype = vp
    lib.topolow_session_degree_terms.argtypes = [vp]
    lib.topolow_session_symm_moves.restype = vp
    lib.topolow_session_symm_moves.argtypes = [vp]
    lib.topolow_session_symm_segment_build.restype = C.c_int
    lib.topolow_session_symm_segment_build.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_int32, C.c_int32, C.c_int32,
                                                       C.c_char_p, C.c_size_t]
    lib.topolow_session_symm_segment_sweep.restype = C.c_int
    lib.topolow_session_symm_segment_sweep.argtypes = [vp, vp, C.c_int32, C.c_double, vp, C.c_char_p, C.c_size_t]
    lib.topolow_session_symm_segment_apply.restype = C.c_int
    lib.topolow_session_symm_segment_apply.argtypes = [vp, vp, vp, C.c_int32, C.c_char_p, C.c_size_t]
    lib.topolow_session_encoded_ptr.restype = vp
    lib.topolow_session_encoded_ptr.argtypes = [vp]
    lib.topolow_session_encoded_ld.restype = C.c_int32
    lib.topolow_session_encoded_ld.argtypes = [vp]
    lib.topolow_session_commit_encoded.restype = C.c_int
    lib.topolow_session_commit_encoded.argtypes = [vp, ip, C.c_char_p, C.c_size_t]
    lib.topolow_session_set_edges.restype = C.c_int
    lib.topolow_session_set_edges.argtypes = [vp, ip, ip, dp, ip, C.c_int64, C.c_char_p, C.c_size_t]
    lib.topolow_session_set_positions.restype = C.c_int
    lib.topolow_session_set_positions.argtypes = [vp, dp, C.c_char_p, C.c_size_t]
    lib.topolow_session_get_positions.restype = C.c_int
    lib.topolow_session_get_positions.argtypes = [vp, dp, C.c_char_p, C.c_size_t]
    lib.topolow_session_begin.restype = C.c_int
    lib.topolow_session_begin.argtypes = [vp, C.c_int32, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_int32, C.c_int32, C.c_uint64, C.c_int32,
                                          C.c_char_p, C.c_size_t]
    lib.topolow_session_enqueue.restype = C.c_int
    lib.topolow_session_enqueue.argtypes = [vp, C.c_int32, ip, C.c_char_p, C.c_size_t]
    lib.topolow_session_sync.restype = C.c_int
    lib.topolow_session_sync.argtypes = [vp, ip, ip, dp, C.c_char_p, C.c_size_t]
    lib.topolow_session_finish.restype = C.c_int
    lib.topolow_session_finish.argtypes = [vp, dp, ip, ip, dp, dp, C.c_char_p, C.c_size_t]
    lib.topolow_session_set_profiling.restype = C.c_int
    lib.topolow_session_set_profiling.argtypes = [vp, C.c_int32]
    lib.topolow_session_profile.restype = C.c_int
    lib.topolow_session_profile.argtypes = [vp, dp, C.POINTER(C.c_int64), dp, C.POINTER(C.c_int64),
                                            C.c_char_p, C.c_size_t]
    lib.topolow_session_set_stream.restype = C.c_int
    lib.topolow_session_set_stream.argtypes = [vp, vp, C.c_int32]
    lib.topolow_session_stream.restype = vp
    lib.topolow_session_stream.argtypes = [vp]
    lib.topolow_session_set_schedule.restype = C.c_int
    lib.topolow_session_set_schedule.argtypes = [vp, C.c_int32]
    lib.topolow_tilegs_pair_order.restype = C.c_int64
    lib.topolow_tilegs_pair_order.argtypes = [C.c_int32, C.c_uint64, C.c_int32, ip]
    lib.topolow_session_position_rows.restype = C.c_int32
    lib.topolow_session_position_rows.argtypes = [vp]
    lib.topolow_session_uses_dense_mae.restype = C.c_int32
    lib.topolow_session_uses_dense_mae.argtypes = [vp]
    lib.topolow_session_stage_launches.restype = C.c_int64
    lib.topolow_session_stage_launches.argtypes = [vp]
    lib.topolow_session_bytes_per_iteration.restype = C.c_int64
    lib.topolow_session_bytes_per_iteration.argtypes = [vp]
    lib.topolow_session_stage.restype = C.c_int
    lib.topolow_session_stage.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                          C.c_char_p, C.c_size_t]
    lib.topolow_session_edge_error.restype = C.c_int
    lib.topolow_session_edge_error.argtypes = [vp, vp, dp, C.POINTER(C.c_int64), C.c_char_p,
                                               C.c_size_t]
    lib.topolow_slab_plan.restype = C.c_int32
    lib.topolow_slab_plan.argtypes = [C.c_int32, C.c_int32, C.c_uint64, C.c_int32, ip, C.c_int32]
    lib.topolow_slab_stages_for_k.restype = C.c_int32
    lib.topolow_slab_stages_for_k.argtypes = [C.c_double, C.c_int32]
    lib.topolow_slab_stages_at.restype = C.c_int32
    lib.topolow_slab_stages_at.argtypes = [C.c_int32, C.c_double, C.c_int32]
    lib.topolow_gs_pair_order.restype = C.c_int64
    lib.topolow_gs_pair_order.argtypes = [C.c_int32, C.c_uint64, C.c_int32, ip]
    lib.topolow_encode_target.restype = C.c_uint32
    lib.topolow_encode_target.argtypes = [C.c_double, C.c_int32]
    lib.topolow_decode_target.restype = C.c_double
    lib.topolow_decode_target.argtypes = [C.c_uint32, ip]
    lib.topolow_controller_script.restype = C.c_int
    lib.topolow_controller_script.argtypes = [dp, ip, dp, C.c_int32, C.c_double, C.c_int32,
                                              C.c_double, ip, ip, dp, dp, ip]
    _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64F(a):
    return np.require(np.asarray(a, dtype=np.float64), requirements=["F", "A"])


def _i32F(a):
    return np.require(np.asarray(a, dtype=np.int32), requirements=["F", "A"])


def _check(rc: int, err) -> None:
    if rc != OK:
        raise NativeError(rc, err.value.decode(errors="replace") or f"libtopolow_relax error {rc}")


def make_options(**kw) -> TopolowOptions:
    lib = load()
    o = TopolowOptions()
    lib.topolow_default_options(C.byref(o))
    cfg = dict(options)
    cfg.update({k: v for k, v in kw.items() if v is not None})
    seed = cfg.get("seed")
    if seed is None:
        seed = int(_host_rng.integers(0, 2 ** 63 - 1))
    o.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    sch = cfg.get("schedule", "auto")
    o.schedule = _SCHEDULES[sch] if isinstance(sch, str) else int(sch)
    pr = cfg.get("precision", "auto")
    o.precision = _PRECISIONS[pr] if isinstance(pr, str) else int(pr)
    o.slab_stages = int(cfg.get("slab_stages", 0) or 0)
    o.device = int(cfg.get("device", -1))
    o.gs_max_n = int(cfg.get("gs_max_n", 0) or 0)
    o.keep_labels = int(bool(cfg.get("keep_labels", False)))
    cb = cfg.get("interrupt")
    if cb is not None:   # Python callable() -> truthy to stop; keep a reference alive on the struct
        o._cb_keepalive = INTERRUPT_CB(lambda _user: 1 if cb() else 0)
        o.interrupt_cb = o._cb_keepalive
    pr = cfg.get("print")
    if pr is not None:   # Python callable(str): receives the verbose lines instead of stdout
        o._pr_keepalive = PRINT_CB(lambda line, _user: pr(line.decode(errors="replace")))
        o.print_cb = o._pr_keepalive
    devs = cfg.get("devices")
    if devs is not None:   # row-block sharded over these HIP ordinals (repeats allowed)
        o._dev_keepalive = (C.c_int32 * len(devs))(*[int(d) for d in devs])
        o.devices = C.cast(o._dev_keepalive, C.POINTER(C.c_int32))
        o.n_devices = len(devs)
    return o


@dataclass
class NativeResult:
    positions: np.ndarray
    converged: bool
    iterations: int
    final_mae: float
    final_k: float
    info: Dict[str, Any] = field(default_factory=dict)


def optimize_layout_exact_arrays(initial_positions, dissimilarity_matrix, threshold_matrix,
                                 degrees, edge_i, edge_j, edge_dist, edge_thresh, n_iter, k0,
                                 cooling_rate, c_repulsion, relative_epsilon, convergence_window,
                                 convergence_check_freq, verbose=False, **opt_kw) -> NativeResult:
    """`optimize_layout_exact_cpp(...)` (reference src/optimization.cpp:109-126) on the GPU."""
    lib = load()
    pos0 = _f64F(initial_positions)
    n, dim = pos0.shape
    D = _f64F(dissimilarity_matrix)
    T = _i32F(threshold_matrix)
    deg = np.ascontiguousarray(degrees, dtype=np.int32)
    ei = np.ascontiguousarray(edge_i, dtype=np.int32)
    ej = np.ascontiguousarray(edge_j, dtype=np.int32)
    ed = np.ascontiguousarray(edge_dist, dtype=np.float64)
    et = np.ascontiguousarray(edge_thresh, dtype=np.int32)
    out = np.zeros((n, dim), dtype=np.float64, order="F")
    conv, iters = C.c_int32(0), C.c_int32(0)
    fmae, fk = C.c_double(0.0), C.c_double(0.0)
    stats = TopolowRunStats()
    err = C.create_string_buffer(512)
    opt = make_options(**opt_kw)
    rc = lib.topolow_optimize_layout_exact(
        _dp(pos0), n, dim, _dp(D), _ip(T), _ip(deg), _ip(ei), _ip(ej), _dp(ed), _ip(et),
        int(ei.shape[0]), int(n_iter), float(k0), float(cooling_rate), float(c_repulsion),
        float(relative_epsilon), int(convergence_window), int(convergence_check_freq),
        int(bool(verbose)), C.byref(opt), _dp(out), C.byref(conv), C.byref(iters), C.byref(fmae),
        C.byref(fk), C.byref(stats), err, len(err))
    _check(rc, err)
    info = dict(schedule={SCHEDULE_SLAB: "slab", SCHEDULE_GS: "gs"}.get(stats.schedule_used),
                precision={PRECISION_F32: "f32", PRECISION_F64: "f64"}.get(stats.precision_used),
                iterations_run=stats.iterations_run, n_checks=stats.n_checks,
                device_seconds=stats.device_seconds, total_seconds=stats.total_seconds,
                setup_seconds=stats.setup_seconds, stage_launches=stats.stage_launches, seed=int(opt.seed))
    return NativeResult(np.ascontiguousarray(out), bool(conv.value), int(iters.value),
                        float(fmae.value), float(fk.value), info)


def optimize_layout_exact(call) -> NativeResult:
    """Takes a core.LayoutCall."""
    return optimize_layout_exact_arrays(
        call.initial_positions, call.dissimilarity_matrix, call.threshold_matrix, call.degrees,
        call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.n_iter, call.k0,
        call.cooling_rate, call.c_repulsion, call.relative_epsilon, call.convergence_window,
        call.convergence_check_freq, call.verbose)


def optimize_layout_exact_batch(calls, seeds=None, precision="f64", device=-1, holdouts=None):
    """Relaxes a list of calls as ONE grid of the exact-GS kernel (one workgroup per embedding).
    A call is a core.LayoutCall, or any object with the same fields whose `dissimilarity_matrix` is
    None: its edge list then IS the matrix (every unlisted pair unmeasured) and no n x n array is
    built or uploaded.  `holdouts[b]` = (i, j, truth) arrays of held-out pairs to score on the
    returned positions (info["holdout_sum_abs"], info["holdout_count"]).
    Returns (list of NativeResult-or-NativeError, device_seconds)."""
    lib = load()
    count = len(calls)
    probs = (TopolowProblem * count)()
    ress = (TopolowResult * count)()
    keep = []
    outs = []
    for b, c in enumerate(calls):
        pos0 = _f64F(c.initial_positions)
        n, dim = pos0.shape
        deg = np.ascontiguousarray(c.degrees, dtype=np.int32)
        ei = np.ascontiguousarray(c.edge_i, dtype=np.int32)
        ej = np.ascontiguousarray(c.edge_j, dtype=np.int32)
        ed = np.ascontiguousarray(c.edge_dist, dtype=np.float64)
        et = np.ascontiguousarray(c.edge_thresh, dtype=np.int32)
        out = np.zeros((n, dim), dtype=np.float64, order="F")
        p = probs[b]
        if c.dissimilarity_matrix is not None:
            D, T = _f64F(c.dissimilarity_matrix), _i32F(c.threshold_matrix)
            p.dissimilarity_matrix, p.threshold_matrix = _dp(D), _ip(T)
            keep.append((D, T))
        keep.append((pos0, deg, ei, ej, ed, et))
        outs.append(out)
        p.initial_positions = _dp(pos0)
        p.degrees, p.edge_i, p.edge_j, p.edge_dist, p.edge_thresh = _ip(deg), _ip(ei), _ip(ej), _dp(ed), _ip(et)
        p.n_edges, p.n, p.ndim, p.n_iter = int(ei.shape[0]), n, dim, int(c.n_iter)
        p.convergence_window, p.convergence_check_freq = int(c.convergence_window), int(c.convergence_check_freq)
        p.k0, p.cooling_rate, p.c_repulsion = float(c.k0), float(c.cooling_rate), float(c.c_repulsion)
        p.relative_epsilon = float(c.relative_epsilon)
        p.seed = int(seeds[b] if seeds is not None else _host_rng.integers(0, 2 ** 63 - 1)) & 0xFFFFFFFFFFFFFFFF
        if holdouts is not None and holdouts[b] is not None and len(holdouts[b][0]) > 0:
            hi = np.ascontiguousarray(holdouts[b][0], dtype=np.int32)
            hj = np.ascontiguousarray(holdouts[b][1], dtype=np.int32)
            ht = np.ascontiguousarray(holdouts[b][2], dtype=np.float64)
            keep.append((hi, hj, ht))
            p.holdout_i, p.holdout_j, p.holdout_truth, p.n_holdout = _ip(hi), _ip(hj), _dp(ht), int(hi.shape[0])
        ress[b].positions_out = _dp(out)
    secs = C.c_double(0.0)
    err = C.create_string_buffer(512)
    rc = lib.topolow_optimize_layout_exact_batch(probs, ress, count, _PRECISIONS[precision], int(device),
                                                 C.byref(secs), err, len(err))
    _check(rc, err)
    results = []
    for b in range(count):
        r = ress[b]
        if r.error_code != OK:
            results.append(NativeError(r.error_code, "Numerical instability at iteration %d. Reduce k0 or "
                                                     "c_repulsion." % r.error_iteration))
        else:
            results.append(NativeResult(np.ascontiguousarray(outs[b]), bool(r.converged), int(r.iterations),
                                        float(r.final_mae), float(r.final_k),
                                        dict(schedule="gs", precision=precision, iterations_run=r.iterations_run,
                                             n_checks=r.n_checks, seed=int(probs[b].seed),
                                             holdout_sum_abs=float(r.holdout_sum_abs),
                                             holdout_count=int(r.holdout_count))))
    return results, float(secs.value)


class CellList:
    """The non-NA cells of a matrix in the form `topolow_cv_fold` reads (include/topolow_relax.h)."""

    def __init__(self, n, rows, cols, values, codes, pos_of):
        self.n = int(n)
        self.rows = np.ascontiguousarray(rows, dtype=np.int32)
        self.cols = np.ascontiguousarray(cols, dtype=np.int32)
        self.values = np.ascontiguousarray(values, dtype=np.float64)
        self.codes = np.ascontiguousarray(codes, dtype=np.int32)
        self.pos_of = np.ascontiguousarray(pos_of, dtype=np.int64)
        self.by_row = np.ascontiguousarray(np.lexsort((self.cols, self.rows)), dtype=np.int64)
        counts = np.bincount(self.rows, minlength=self.n)
        self.row_ptr = np.ascontiguousarray(np.concatenate([[0], np.cumsum(counts)]), dtype=np.int64)
        c = self.c = TopolowCellList()
        c.n, c.n_cells = self.n, int(self.rows.shape[0])
        c.row, c.col, c.value, c.code = _ip(self.rows), _ip(self.cols), _dp(self.values), _ip(self.codes)
        i64 = C.POINTER(C.c_int64)
        c.pos_of, c.by_row, c.row_ptr = (self.pos_of.ctypes.data_as(i64), self.by_row.ctypes.data_as(i64),
                                         self.row_ptr.ctypes.data_as(i64))


def cv_fold(cells: CellList, picks, preserve_order: bool, named: bool):
    """One fold's problem from the cell list (host-side helper of the CV evaluator).  Returns
    (order or None, degrees, edge_i, edge_j, edge_dist, edge_thresh, hold_i, hold_j, hold_truth,
    numeric_max)."""
    lib = load()
    picks = np.ascontiguousarray(picks, dtype=np.int64)
    n, m = cells.n, int(cells.rows.shape[0])
    order, deg = np.empty(n, np.int32), np.empty(n, np.int32)
    ei, ej, ed, et = np.empty(m, np.int32), np.empty(m, np.int32), np.empty(m, np.float64), np.empty(m, np.int32)
    hi, hj, ht = np.empty(m, np.int32), np.empty(m, np.int32), np.empty(m, np.float64)
    ne, nh, vmax = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
    rc = lib.topolow_cv_fold(C.byref(cells.c), picks.ctypes.data_as(C.POINTER(C.c_int64)), int(picks.shape[0]),
                             int(bool(preserve_order)), int(bool(named)), _ip(order), _ip(deg), _ip(ei), _ip(ej),
                             _dp(ed), _ip(et), C.byref(ne), _ip(hi), _ip(hj), _dp(ht), C.byref(nh), C.byref(vmax))
    if rc != OK:
        raise NativeError(rc, "topolow_cv_fold failed")
    e, h = int(ne.value), int(nh.value)
    return (None if order[0] < 0 else order.astype(np.int64), deg, ei[:e].copy(), ej[:e].copy(), ed[:e].copy(),
            et[:e].copy(), hi[:h].copy(), hj[:h].copy(), ht[:h].copy(), float(vmax.value))


def cv_sweep(cells: CellList, named: bool, preserve_order: bool, ndims, k0s, cooling_rates, c_repulsions, picks, unit_draws,
             seeds, n_iter: int, relative_epsilon: float, convergence_window: int = 5, convergence_check_freq: int = 3,
             precision: str = "f64", device: int = -1):
    """All folds of a CV sweep in one library call (topolow_cv_sweep): fold f holds out the cells picks[f] and starts
    from the random walk built from unit_draws[f] ((ndim, n - 1) uniform(0, 1) numbers).  Returns
    (holdout_sum_abs, holdout_count, iterations, converged, error_code) arrays and the device seconds."""
    lib = load()
    nf = len(picks)
    nd = np.ascontiguousarray(ndims, dtype=np.int32)
    k0 = np.ascontiguousarray(k0s, dtype=np.float64)
    cr = np.ascontiguousarray(cooling_rates, dtype=np.float64)
    cp = np.ascontiguousarray(c_repulsions, dtype=np.float64)
    p_off = np.zeros(nf + 1, dtype=np.int64)
    d_off = np.zeros(nf + 1, dtype=np.int64)
    if nf:
        np.cumsum([len(p) for p in picks], out=p_off[1:])
        np.cumsum([u.size for u in unit_draws], out=d_off[1:])
    p_all = np.ascontiguousarray(np.concatenate(picks) if nf else np.zeros(0), dtype=np.int64)
    d_all = np.ascontiguousarray(np.concatenate([np.ravel(u) for u in unit_draws]) if nf else np.zeros(0), dtype=np.float64)
    sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
    hsum, hcnt = np.zeros(nf, np.float64), np.zeros(nf, np.int64)
    its, conv, ec = np.zeros(nf, np.int32), np.zeros(nf, np.int32), np.zeros(nf, np.int32)
    secs = C.c_double(0.0)
    err = C.create_string_buffer(512)
    i64 = C.POINTER(C.c_int64)
    rc = lib.topolow_cv_sweep(C.byref(cells.c), int(bool(named)), int(bool(preserve_order)), nf, _ip(nd), _dp(k0), _dp(cr),
                              _dp(cp), p_all.ctypes.data_as(i64), p_off.ctypes.data_as(i64), _dp(d_all),
                              d_off.ctypes.data_as(i64), sd.ctypes.data_as(C.POINTER(C.c_uint64)), int(n_iter),
                              float(relative_epsilon), int(convergence_window), int(convergence_check_freq),
                              _PRECISIONS[precision], int(device), _dp(hsum), hcnt.ctypes.data_as(i64), _ip(its), _ip(conv),
                              _ip(ec), C.byref(secs), err, len(err))
    _check(rc, err)
    return hsum, hcnt, its, conv, ec, float(secs.value)


def symm_stage_bounds(n: int, stages: int):
    """Slab boundaries (labels) of the symmetric form of an S-stage iteration, or None (topolow_symm_stage_bounds)."""
    out = np.zeros(stages + 1, dtype=np.int32)
    return out.tolist() if load().topolow_symm_stage_bounds(int(n), int(stages), _ip(out)) else None


def symm_stage_order(seed: int, it: int, stages: int):
    """Order of the stages of iteration `it` (topolow_symm_stage_order)."""
    out = np.zeros(stages, dtype=np.int32)
    assert load().topolow_symm_stage_order(C.c_uint64(seed), int(it), int(stages), _ip(out))
    return out.tolist()


def symm_segment_rows(n: int, segment: int, n_segments: int):
    """Rows (first, end) of the matrix that hold segment `segment` of `n_segments` of the symmetric sweep's tile list;
    None when a problem of n points has too few tiles to cut (topolow_symm_segment_rows; host only)."""
    lib = load()
    a, b = C.c_int32(0), C.c_int32(0)
    if not lib.topolow_symm_segment_rows(int(n), int(segment), int(n_segments), C.byref(a), C.byref(b)):
        return None
    return int(a.value), int(b.value)


def shard_rows(n: int, blocks: int):
    """Row blocks of the row-sharded path: list of (row_begin, row_end), empty trailing blocks dropped."""
    lib = load()
    used = lib.topolow_shard_rows(int(n), int(blocks), -1, None, None)
    out = []
    for b in range(used):
        rb, re_ = C.c_int32(0), C.c_int32(0)
        lib.topolow_shard_rows(int(n), int(blocks), b, C.byref(rb), C.byref(re_))
        out.append((int(rb.value), int(re_.value)))
    return out


def optimize_layout_exact_sharded(initial_positions, degrees, edge_i, edge_j, edge_dist, edge_thresh, n_iter, k0,
                                  cooling_rate, c_repulsion, relative_epsilon=1e-4, convergence_window=5,
                                  convergence_check_freq=3, verbose=False, dissimilarity_matrix=None,
                                  threshold_matrix=None, devices=(0,), **opt_kw) -> NativeResult:
    """The `.Call` payload as ONE embedding row-sharded over `devices` (topolow_optimize_layout_exact_sharded).
    Without the dense matrices the edge list IS the matrix (large problems)."""
    lib = load()
    pos0 = _f64F(initial_positions)
    n, dim = pos0.shape
    D = _f64F(dissimilarity_matrix) if dissimilarity_matrix is not None else None
    T = _i32F(threshold_matrix) if threshold_matrix is not None else None
    deg = np.ascontiguousarray(degrees, dtype=np.int32)
    ei = np.ascontiguousarray(edge_i, dtype=np.int32)
    ej = np.ascontiguousarray(edge_j, dtype=np.int32)
    ed = np.ascontiguousarray(edge_dist, dtype=np.float64)
    et = np.ascontiguousarray(edge_thresh, dtype=np.int32)
    out = np.zeros((n, dim), dtype=np.float64, order="F")
    conv, iters = C.c_int32(0), C.c_int32(0)
    fmae, fk = C.c_double(0.0), C.c_double(0.0)
    stats = TopolowShardStats()
    err = C.create_string_buffer(512)
    opt = make_options(devices=list(devices), **opt_kw)
    rc = lib.topolow_optimize_layout_exact_sharded(
        _dp(pos0), n, dim, _dp(D) if D is not None else None, _ip(T) if T is not None else None, _ip(deg),
        _ip(ei), _ip(ej), _dp(ed), _ip(et), int(ei.shape[0]), int(n_iter), float(k0), float(cooling_rate),
        float(c_repulsion), float(relative_epsilon), int(convergence_window), int(convergence_check_freq),
        int(bool(verbose)), C.byref(opt), _dp(out), C.byref(conv), C.byref(iters), C.byref(fmae), C.byref(fk),
        C.byref(stats), err, len(err))
    _check(rc, err)
    info = dict(schedule="slab", blocks=stats.blocks, groups=stats.groups, iterations_run=stats.iterations_run,
                n_checks=stats.n_checks, loop_seconds=stats.loop_seconds, total_seconds=stats.total_seconds,
                stage_kernel_seconds=stats.stage_kernel_seconds, check_kernel_seconds=stats.check_kernel_seconds,
                stage_launches=stats.stage_launches, exchanges=stats.exchanges, seed=int(opt.seed))
    return NativeResult(np.ascontiguousarray(out), bool(conv.value), int(iters.value), float(fmae.value),
                        float(fk.value), info)


def run_sharded(sessions, initial_positions, n_iter, k0, cooling_rate, c_repulsion, relative_epsilon=1e-4,
                convergence_window=5, convergence_check_freq=3, seed=0, slab_stages=0, interrupt=None,
                profile=False, warmup_iterations=0) -> NativeResult:
    """ONE embedding over the given row-block sessions (topolow_sessions_run_sharded): one process, one
    host thread per block, peer-stored position slices, replicated controller."""
    lib = load()
    pos0 = _f64F(initial_positions)
    n, dim = pos0.shape
    hs = (C.c_void_p * len(sessions))(*[s._h for s in sessions])
    out = np.zeros((n, dim), dtype=np.float64, order="F")
    conv, iters = C.c_int32(0), C.c_int32(0)
    fmae, fk = C.c_double(0.0), C.c_double(0.0)
    stats = TopolowShardStats()
    stats.warmup_iterations = int(warmup_iterations)   # measurement aid: timed_seconds starts after these
    err = C.create_string_buffer(512)
    cb = INTERRUPT_CB(lambda _u: 1 if interrupt() else 0) if interrupt is not None else INTERRUPT_CB()
    rc = lib.topolow_sessions_run_sharded(hs, len(sessions), _dp(pos0), int(n_iter), float(k0), float(cooling_rate),
                                          float(c_repulsion), float(relative_epsilon), int(convergence_window),
                                          int(convergence_check_freq), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                          int(slab_stages), cb, None, int(bool(profile)), _dp(out), C.byref(conv),
                                          C.byref(iters), C.byref(fmae), C.byref(fk), C.byref(stats), err, len(err))
    _check(rc, err)
    info = dict(schedule="slab", blocks=stats.blocks, groups=stats.groups, iterations_run=stats.iterations_run,
                n_checks=stats.n_checks, loop_seconds=stats.loop_seconds, stage_kernel_seconds=stats.stage_kernel_seconds,
                check_kernel_seconds=stats.check_kernel_seconds, stage_launches=stats.stage_launches,
                exchanges=stats.exchanges, timed_seconds=stats.timed_seconds,
                symmetric_segments=stats.symmetric_segments)
    return NativeResult(np.ascontiguousarray(out), bool(conv.value), int(iters.value), float(fmae.value),
                        float(fk.value), info)


def est_distances(positions) -> np.ndarray:
    """as.matrix(dist(positions)) (reference R/core.R:474) on the GPU."""
    lib = load()
    pos = _f64F(positions)
    n, dim = pos.shape
    out = np.empty((n, n), dtype=np.float64)
    err = C.create_string_buffer(512)
    rc = lib.topolow_est_distances(_dp(pos), n, dim, _dp(out), int(options.get("device", -1)),
                                   err, len(err))
    _check(rc, err)
    return out


def est_distances_rows(positions, row_begin: int, row_end: int) -> np.ndarray:
    """Rows [row_begin, row_end) of as.matrix(dist(positions)): shape (row_end - row_begin, n)."""
    lib = load()
    pos = _f64F(positions)
    n, dim = pos.shape
    out = np.empty((int(row_end) - int(row_begin), n), dtype=np.float64)
    err = C.create_string_buffer(512)
    rc = lib.topolow_est_distances_rows(_dp(pos), n, dim, int(row_begin), int(row_end), _dp(out),
                                        int(options.get("device", -1)), err, len(err))
    _check(rc, err)
    return out


# ---- host-side helpers (no GPU needed) ---------------------------------------------------
def slab_plan(n: int, slab_stages: int, seed: int, it: int) -> np.ndarray:
    lib = load()
    buf = np.zeros((64, 4), dtype=np.int32)
    ns = lib.topolow_slab_plan(int(n), int(slab_stages), int(seed), int(it), _ip(buf), 64)
    return buf[:ns].copy()


def slab_stages_for_k(k: float, ndim: int = 5) -> int:
    return int(load().topolow_slab_stages_for_k(float(k), int(ndim)))


def slab_stages_at(it: int, k: float, ndim: int = 5) -> int:
    return int(load().topolow_slab_stages_at(int(it), float(k), int(ndim)))


def gs_pair_order(n: int, seed: int, it: int) -> np.ndarray:
    lib = load()
    buf = np.zeros((n * (n - 1) // 2, 2), dtype=np.int32)
    cnt = lib.topolow_gs_pair_order(int(n), int(seed), int(it), _ip(buf))
    assert cnt == buf.shape[0]
    return buf


def tilegs_pair_order(n: int, seed: int, it: int) -> np.ndarray:
    lib = load()
    buf = np.zeros((n * (n - 1) // 2, 2), dtype=np.int32)
    cnt = lib.topolow_tilegs_pair_order(int(n), int(seed), int(it), _ip(buf))
    assert cnt == buf.shape[0], (cnt, buf.shape)
    return buf


def encode_target(d: float, code: int) -> int:
    return int(load().topolow_encode_target(float(d), int(code)))


def decode_target(bits: int):
    c = C.c_int32(0)
    v = load().topolow_decode_target(int(bits), C.byref(c))
    return float(v), int(c.value)


def controller_script(mae_seq, iter_seq, k_seq, k0, window, eps):
    lib = load()
    m = np.ascontiguousarray(mae_seq, dtype=np.float64)
    it = np.ascontiguousarray(iter_seq, dtype=np.int32)
    ks = np.ascontiguousarray(k_seq, dtype=np.float64)
    n = int(m.shape[0])
    stopped = C.c_int32(-1)
    snaps = np.zeros(n, dtype=np.int32)
    bm, bk, bi = C.c_double(0), C.c_double(0), C.c_int32(0)
    lib.topolow_controller_script(_dp(m), _ip(it), _dp(ks), n, float(k0), int(window), float(eps),
                                  C.byref(stopped), _ip(snaps), C.byref(bm), C.byref(bk),
                                  C.byref(bi))
    return dict(stopped_at=int(stopped.value), snapshots=snaps.astype(bool),
                best_mae=float(bm.value), best_k=float(bk.value), best_iter=int(bi.value))


class Session:
    """Device-resident slab session (include/topolow_relax.h, `topolow_session_*`)."""

    def __init__(self, n, ndim, row_begin=0, row_end=None, precision="f32", device=-1):
        self.lib = load()
        self.n, self.ndim = int(n), int(ndim)
        self.row_begin = int(row_begin)
        self.row_end = int(n if row_end is None else row_end)
        self.precision = precision
        self._h = C.c_void_p(None)
        self._err = C.create_string_buffer(512)
        rc = self.lib.topolow_session_create(C.byref(self._h), self.n, self.ndim, self.row_begin,
                                             self.row_end, _PRECISIONS[precision], int(device),
                                             self._err, len(self._err))
        _check(rc, self._err)

    def close(self):
        if self._h:
            self.lib.topolow_session_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_dense(self, D, T, degrees):
        D, T = _f64F(D), _i32F(T)
        deg = np.ascontiguousarray(degrees, dtype=np.int32)
        _check(self.lib.topolow_session_load_dense(self._h, _dp(D), _ip(T), _ip(deg), self._err,
                                                   len(self._err)), self._err)

    def load_coo(self, ei, ej, ed, et, degrees):
        ei = np.ascontiguousarray(ei, dtype=np.int32)
        ej = np.ascontiguousarray(ej, dtype=np.int32)
        ed = np.ascontiguousarray(ed, dtype=np.float64)
        et = np.ascontiguousarray(et, dtype=np.int32)
        deg = np.ascontiguousarray(degrees, dtype=np.int32)
        _check(self.lib.topolow_session_load_coo(self._h, _ip(ei), _ip(ej), _dp(ed), _ip(et),
                                                 int(ei.shape[0]), _ip(deg), self._err,
                                                 len(self._err)), self._err)

    @property
    def encoded_ptr(self) -> int:
        return int(self.lib.topolow_session_encoded_ptr(self._h) or 0)

    @property
    def encoded_ld(self) -> int:
        return int(self.lib.topolow_session_encoded_ld(self._h))

    def commit_encoded(self, degrees):
        deg = np.ascontiguousarray(degrees, dtype=np.int32)
        assert deg.shape[0] == self.n
        _check(self.lib.topolow_session_commit_encoded(self._h, _ip(deg), self._err,
                                                       len(self._err)), self._err)

    def set_edges(self, ei, ej, ed, et):
        ei = np.ascontiguousarray(ei, dtype=np.int32)
        ej = np.ascontiguousarray(ej, dtype=np.int32)
        ed = np.ascontiguousarray(ed, dtype=np.float64)
        et = np.ascontiguousarray(et, dtype=np.int32)
        _check(self.lib.topolow_session_set_edges(self._h, _ip(ei), _ip(ej), _dp(ed), _ip(et),
                                                  int(ei.shape[0]), self._err, len(self._err)),
               self._err)

    def set_positions(self, pos):
        pos = _f64F(pos)
        assert pos.shape == (self.n, self.ndim)
        _check(self.lib.topolow_session_set_positions(self._h, _dp(pos), self._err,
                                                      len(self._err)), self._err)

    def get_positions(self):
        out = np.zeros((self.n, self.ndim), dtype=np.float64, order="F")
        _check(self.lib.topolow_session_get_positions(self._h, _dp(out), self._err,
                                                      len(self._err)), self._err)
        return np.ascontiguousarray(out)

    def begin(self, n_iter, k0, cooling_rate, c_repulsion, relative_epsilon=1e-4,
              convergence_window=5, convergence_check_freq=3, seed=0, slab_stages=0):
        _check(self.lib.topolow_session_begin(
            self._h, int(n_iter), float(k0), float(cooling_rate), float(c_repulsion),
            float(relative_epsilon), int(convergence_window), int(convergence_check_freq),
            int(seed) & 0xFFFFFFFFFFFFFFFF, int(slab_stages), self._err, len(self._err)), self._err)

    def enqueue(self, max_iters) -> int:
        enq = C.c_int32(0)
        _check(self.lib.topolow_session_enqueue(self._h, int(max_iters), C.byref(enq), self._err,
                                                len(self._err)), self._err)
        return int(enq.value)

    def wait(self):
        """Waits for the launches enqueued so far; a check waiting to ride on the next sweep stays pending."""
        _check(self.lib.topolow_session_wait(self._h, self._err, len(self._err)), self._err)

    def sync(self):
        it, st, mae = C.c_int32(0), C.c_int32(0), C.c_double(0.0)
        _check(self.lib.topolow_session_sync(self._h, C.byref(it), C.byref(st), C.byref(mae),
                                             self._err, len(self._err)), self._err)
        return int(it.value), bool(st.value), float(mae.value)

    def finish(self) -> NativeResult:
        out = np.zeros((self.n, self.ndim), dtype=np.float64, order="F")
        conv, iters = C.c_int32(0), C.c_int32(0)
        fmae, fk = C.c_double(0.0), C.c_double(0.0)
        _check(self.lib.topolow_session_finish(self._h, _dp(out), C.byref(conv), C.byref(iters),
                                               C.byref(fmae), C.byref(fk), self._err,
                                               len(self._err)), self._err)
        return NativeResult(np.ascontiguousarray(out), bool(conv.value), int(iters.value),
                            float(fmae.value), float(fk.value))

    def run(self, chunk=64):
        while self.enqueue(chunk) > 0:
            pass
        return self.sync()

    def check_trace(self) -> np.ndarray:
        """(iteration, MAE, k) of every convergence check of the current run, shape (checks, 3)."""
        n = C.c_int32(0)
        err = C.create_string_buffer(8)
        _check(self.lib.topolow_session_check_trace(self._h, None, 0, C.byref(n)), err)
        out = np.zeros((max(int(n.value), 1), 3), dtype=np.float64)
        _check(self.lib.topolow_session_check_trace(self._h, _dp(out), int(n.value), C.byref(n)), err)
        return out[: int(n.value)]

    def set_relabel(self, seed: int):
        """Store the points in a random order drawn from `seed` (0 = the caller's order); before loading."""
        _check(self.lib.topolow_session_set_relabel(self._h, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                                                    self._err, len(self._err)), self._err)

    def labels(self) -> np.ndarray:
        """session label -> caller's label."""
        out = np.empty(self.n, dtype=np.int32)
        self.lib.topolow_session_labels(self._h, _ip(out))
        return out

    def set_schedule(self, schedule: str):
        rc = self.lib.topolow_session_set_schedule(self._h, _SCHEDULES[schedule])
        if rc != OK:
            raise NativeError(rc, "schedule not available for this session")

    def set_profiling(self, enable: bool):
        self.lib.topolow_session_set_profiling(self._h, int(bool(enable)))

    def profile(self):
        """(stage_ms, stage_launches, check_ms, checks) since profiling was enabled."""
        sm, cm = C.c_double(0.0), C.c_double(0.0)
        sl, cl = C.c_int64(0), C.c_int64(0)
        _check(self.lib.topolow_session_profile(self._h, C.byref(sm), C.byref(sl), C.byref(cm),
                                                C.byref(cl), self._err, len(self._err)), self._err)
        return float(sm.value), int(sl.value), float(cm.value), int(cl.value)

    def profile_symmetric(self):
        """(plain_ms, plain_iterations, fused_ms, fused_iterations) of the iterations that ran as a symmetric sweep +
        apply since profiling was enabled (not part of profile() / profile_fused())."""
        a, b = C.c_double(0.0), C.c_double(0.0)
        na, nb = C.c_int64(0), C.c_int64(0)
        _check(self.lib.topolow_session_profile_symmetric(self._h, C.byref(a), C.byref(na), C.byref(b), C.byref(nb),
                                                          self._err, len(self._err)), self._err)
        return float(a.value), int(na.value), float(b.value), int(nb.value)

    def profile_fused(self):
        """(ms, launches) of the stage launches that also reduced a check's MAE; ask before profile()."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        _check(self.lib.topolow_session_profile_fused(self._h, C.byref(ms), C.byref(n), self._err, len(self._err)),
               self._err)
        return float(ms.value), int(n.value)

    def set_stream(self, hip_stream, external: bool = True):
        """hip_stream: integer hipStream_t (0/None = the device's default stream)."""
        self.lib.topolow_session_set_stream(self._h, C.c_void_p(hip_stream or None),
                                            int(bool(external)))

    @property
    def stream(self) -> int:
        return int(self.lib.topolow_session_stream(self._h) or 0)

    @property
    def position_rows(self) -> int:
        return int(self.lib.topolow_session_position_rows(self._h))

    @property
    def uses_dense_mae(self) -> bool:
        return bool(self.lib.topolow_session_uses_dense_mae(self._h))

    @property
    def stage_launches(self) -> int:
        return int(self.lib.topolow_session_stage_launches(self._h))

    @property
    def bytes_per_iteration(self) -> int:
        return int(self.lib.topolow_session_bytes_per_iteration(self._h))

    def stage(self, d_pos_in: int, d_pos_out: int, it: int, stage: int, n_stages: int, k: float):
        _check(self.lib.topolow_session_stage(self._h, C.c_void_p(d_pos_in), C.c_void_p(d_pos_out),
                                              int(it), int(stage), int(n_stages), float(k),
                                              self._err, len(self._err)), self._err)

    def check_partial(self, d_pos: int, d_out2: int):
        """Enqueue this block's share of the convergence MAE on the positions at d_pos: two doubles
        (sum, count) at the device address d_out2."""
        _check(self.lib.topolow_session_check_partial(self._h, C.c_void_p(d_pos), C.c_void_p(d_out2),
                                                      self._err, len(self._err)), self._err)

    def controller_step(self, d_total2: int, d_pos: int, iter1: int, k_after: float):
        """Enqueue the controller (reference :303-357) on the all-reduced (sum, count) at d_total2."""
        _check(self.lib.topolow_session_controller_step(self._h, C.c_void_p(d_total2), C.c_void_p(d_pos),
                                                        int(iter1), float(k_after), self._err,
                                                        len(self._err)), self._err)

    @property
    def can_fuse_checks(self) -> bool:
        return bool(self.lib.topolow_session_can_fuse_checks(self._h))

    def stage_fused(self, d_pos_in: int, d_pos_out: int, it: int, k: float, d_out2: int):
        """The single stage of iteration `it`, also reducing this block's share of the MAE of d_pos_in."""
        _check(self.lib.topolow_session_stage_fused(self._h, C.c_void_p(d_pos_in), C.c_void_p(d_pos_out), int(it),
                                                    float(k), C.c_void_p(d_out2), self._err, len(self._err)),
               self._err)

    # ---- one-stage iterations as the symmetric sweep sharded over the processes (include/topolow_relax.h) ----
    @property
    def degree_terms_ptr(self) -> int:
        """Device float[n]: degree + 1 per point; a row-block session knows its own rows' (the caller completes it)."""
        return int(self.lib.topolow_session_degree_terms(self._h) or 0)

    @property
    def has_thresholds(self) -> bool:
        return bool(self.lib.topolow_session_has_thresholds(self._h))

    def symm_segment_eligible(self, n_segments: int) -> bool:
        return bool(self.lib.topolow_session_symm_segment_eligible(self._h, int(n_segments)))

    def symm_segment_build(self, segment: int, n_segments: int, d_rows: int, row_first: int, n_rows: int,
                           any_threshold: bool):
        _check(self.lib.topolow_session_symm_segment_build(self._h, int(segment), int(n_segments), C.c_void_p(d_rows),
                                                           int(row_first), int(n_rows), int(bool(any_threshold)),
                                                           self._err, len(self._err)), self._err)

    @property
    def symm_moves_ptr(self) -> int:
        """Device float[n][ndim]: the segment's share of every point's move after symm_segment_sweep."""
        return int(self.lib.topolow_session_symm_moves(self._h) or 0)

    def symm_segment_sweep(self, d_pos_in: int, it: int, k: float, d_out2: int = 0):
        _check(self.lib.topolow_session_symm_segment_sweep(self._h, C.c_void_p(d_pos_in), int(it), float(k),
                                                           C.c_void_p(d_out2) if d_out2 else None, self._err,
                                                           len(self._err)), self._err)

    def symm_segment_apply(self, d_pos_in: int, d_pos_out: int, it: int):
        _check(self.lib.topolow_session_symm_segment_apply(self._h, C.c_void_p(d_pos_in), C.c_void_p(d_pos_out),
                                                           int(it), self._err, len(self._err)), self._err)

    def first_nonfinite(self) -> int:
        it = C.c_int32(0)
        _check(self.lib.topolow_session_first_nonfinite(self._h, C.byref(it)), self._err)
        return int(it.value)

    def edge_error(self, d_pos: int):
        s, c = C.c_double(0.0), C.c_int64(0)
        _check(self.lib.topolow_session_edge_error(self._h, C.c_void_p(d_pos), C.byref(s),
                                                   C.byref(c), self._err, len(self._err)), self._err)
        return float(s.value), int(c.value)
