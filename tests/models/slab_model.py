"""ctypes front end of tests/models/slab_model.cpp (TEST INFRASTRUCTURE: a CPU model of the
slab schedule of the HIP large-N path; see the .cpp header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libslab_model.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "slab_model.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-fopenmp", "-std=gnu++17", "-shared", "-fPIC", "-o", _SO,
                        src], check=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        _lib.slab_model_stage.restype = C.c_int
        _lib.slab_model_stage.argtypes = [dp, dp, C.c_int, C.c_int, dp, ip, ip, ip, C.c_int,
                                          C.c_double, C.c_double, C.c_int]
        _lib.slab_model_run.restype = C.c_int
        _lib.slab_model_run.argtypes = [dp, dp, C.c_int, C.c_int, dp, ip, ip, ip, ip, C.c_int,
                                        C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, dp]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def stage(pos, D, T, degrees, ranges, k, c_rep, arith="f64"):
    lib = _load()
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    n, dim = pos.shape
    Df = np.asfortranarray(D, dtype=np.float64)
    Tf = np.asfortranarray(T, dtype=np.int32)
    deg = np.ascontiguousarray(degrees, dtype=np.int32)
    rg = np.ascontiguousarray(ranges, dtype=np.int32).reshape(-1, 2)
    out = np.empty_like(pos)
    rc = lib.slab_model_stage(_dp(pos), _dp(out), n, dim, _dp(Df), _ip(Tf), _ip(deg), _ip(rg),
                              rg.shape[0], float(k), float(c_rep), 1 if arith == "f32" else 0)
    assert rc == 0
    return out


def run(pos, D, T, degrees, plan, n_stages, k0, cooling, c_rep, arith="f64"):
    """plan: (n_iter, max_stages, 4) int32; n_stages: (n_iter,) int32.  Returns (pos, k)."""
    lib = _load()
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    n, dim = pos.shape
    Df = np.asfortranarray(D, dtype=np.float64)
    Tf = np.asfortranarray(T, dtype=np.int32)
    deg = np.ascontiguousarray(degrees, dtype=np.int32)
    plan = np.ascontiguousarray(plan, dtype=np.int32)
    ns = np.ascontiguousarray(n_stages, dtype=np.int32)
    n_iter, max_stages = plan.shape[0], plan.shape[1]
    out = np.empty_like(pos)
    k_out = C.c_double(0.0)
    rc = lib.slab_model_run(_dp(pos), _dp(out), n, dim, _dp(Df), _ip(Tf), _ip(deg), _ip(plan),
                            _ip(ns), max_stages, n_iter, float(k0), float(cooling), float(c_rep),
                            1 if arith == "f32" else 0, C.byref(k_out))
    assert rc == 0
    return out, float(k_out.value)
