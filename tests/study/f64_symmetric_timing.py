"""GPU study: duration of one-stage iterations of an f64 session at config 3 -- the symmetric sweep (relax_symm64.h)
against the row-owner f64 stage kernel (TOPOLOW_SYMMETRIC=0).  usage: python tests/study/f64_symmetric_timing.py [n] [ndim]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from topolow_amd import _native, core, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 5
prob = synthetic.make_problem(n, latent_dim=dim, missing=0.7, seed=12345)
init = synthetic.initial_positions(prob.dissimilarity, dim, 12345)
call = core.prepare_layout_call(prob.dissimilarity, dim, 1, 2.0, 0.01, 0.01, 1e-4, 5, init, False, 3, True)
for sym in ("1", "0"):
    os.environ["TOPOLOW_SYMMETRIC"] = sym
    s = _native.Session(n, dim, precision="f64")
    s.set_relabel(3)
    s.load_coo(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.degrees)
    s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    for profile in (False, True):
        s.set_positions(call.initial_positions)
        s.set_profiling(profile)
        s.begin(60, 2.0, 0.01, 0.01, 1e-12, 10 ** 9, 3, 5, 1)
        t0 = time.perf_counter()
        s.run()
        s.sync()
        wall = time.perf_counter() - t0
        if profile:
            sym_ms, sym_it, _e, _ei = s.profile_symmetric()
            if _ei:
                print(f"TOPOLOW_SYMMETRIC={sym}: symmetric iterations that also reduce the check's MAE: {_ei}, {1e3 * _e / _ei:.1f} us each")
            st_ms, st_n, ck_ms, ck_n = s.profile()
            print(f"TOPOLOW_SYMMETRIC={sym}: symmetric {sym_it} iterations {1e3 * sym_ms / max(sym_it, 1):.1f} us each; "
                  f"stage launches {st_n} {1e3 * st_ms / max(st_n, 1):.1f} us each; checks {ck_n} {1e3 * ck_ms / max(ck_n, 1):.1f} us each")
        else:
            print(f"TOPOLOW_SYMMETRIC={sym}: 60 one-stage iterations, 20 checks: {1e6 * wall / 60:.1f} us per iteration (wall)")
    s.close()
