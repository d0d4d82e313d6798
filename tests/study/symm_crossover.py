"""One-stage iterations: symmetric sweep (relax_symm.h) against the row-owner stage kernel by problem size and ndim
(not a test).  300 iterations from the reference's start at k = 2, check every 3 iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from tests import parity_problems as pp
from topolow_amd import _native, core, synthetic

os.environ["TOPOLOW_SYMMETRIC_MIN_N"] = "0"

def problem(n, dim, missing=0.7):
    prob = synthetic.make_problem(n, latent_dim=dim, missing=missing, seed=12345)
    init = synthetic.initial_positions(prob.dissimilarity, dim, 12345)
    return core.prepare_layout_call(prob.dissimilarity, dim, 1000, 5.0, 0.01, 0.01, 1e-4, 5, init, False, 3, True)

def run(call, n, dim, sym, iters=300):
    os.environ["TOPOLOW_SYMMETRIC"] = "1" if sym else "0"
    s = _native.Session(n, dim, precision="f32")
    s.set_relabel(2024)
    s.load_coo(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.degrees)
    s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    best = 1e9
    for _ in range(3):
        s.set_positions(call.initial_positions)
        s.begin(iters, min(2.0, float(dim)), 0.01, 0.01, 1e-4, 10 ** 9, 3, 2024, 1)
        torch.cuda.synchronize(); t = time.perf_counter()
        s.run(); s.sync(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / iters * 1e6)
    r = s.finish(); s.close()
    return best, r.final_mae

sizes = [(3072, 5), (4096, 5), (6144, 5), (8192, 5), (10000, 5), (14000, 5), (10000, 2), (10000, 3), (10000, 4), (10000, 6), (20000, 3)]
if len(sys.argv) > 1:
    sizes = [(int(a.split("x")[0]), int(a.split("x")[1])) for a in sys.argv[1:]]
for n, dim in sizes:
    call = problem(n, dim)
    a, ma = run(call, n, dim, False)
    b, mb = run(call, n, dim, True)
    print("n %6d ndim %d: row-owner %7.1f us/it  symmetric %7.1f us/it  (x%.2f)   MAE %.6f %.6f" % (n, dim, a, b, a / b, ma, mb), flush=True)
