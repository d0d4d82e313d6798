"""Where a one-stage iteration's time goes beyond its kernels (not a test): config 3, fixed 1 stage per iteration from a
relaxed start, 300 iterations per measurement; check every 3 iterations (beside the next sweep, or TOPOLOW_SERIAL_CHECKS=1
on the main stream) against a check every 10 000 (none)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from topolow_amd import _native, core, synthetic

n = 10000
prob = synthetic.make_problem(n, latent_dim=5, missing=0.7, seed=12345)
init = synthetic.initial_positions(prob.dissimilarity, 5, 12345)
call = core.prepare_layout_call(prob.dissimilarity, 5, 1000, 5.0, 0.01, 0.01, 1e-4, 5, init, False, 3, True)
s = _native.Session(n, 5, precision="f32")
s.set_relabel(2024)
s.load_coo(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.degrees)
s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
# a relaxed start: 120 iterations of the production schedule
s.set_positions(call.initial_positions)
s.begin(120, 5.0, 0.01, 0.01, 1e-4, 10 ** 9, 3, 2024, 0)
s.run()
start = s.finish().positions


def measure(check_freq, iters=300, reps=5):
    out = []
    for _ in range(reps):
        s.set_positions(start)
        s.begin(iters, 1.5, 0.01, 0.01, 1e-4, 10 ** 9, check_freq, 2024, 1)   # k = 1.5, one stage per iteration
        torch.cuda.synchronize()
        t = time.perf_counter()
        done = 0
        while done < iters:
            done += s.enqueue(iters - done)
        s.sync()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / iters * 1e6)
    return min(out), float(np.median(out))


print("serial checks env:", os.environ.get("TOPOLOW_SERIAL_CHECKS"), " fuse env:", os.environ.get("TOPOLOW_FUSE_CHECKS"))
for cf in (3, 10000):
    print("check every %5d iterations: %.1f us / iteration (min), %.1f (median)" % ((cf,) + measure(cf)))
