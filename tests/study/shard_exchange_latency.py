"""Study (not a test): wall time per cross-block exchange of the native row-sharded engine, from problems
small enough that the kernels are short: loop_seconds / exchanges for 1, 2, 4, 8 row blocks on one GPU."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import parity_problems as pp
from topolow_amd import _native

out = {}
mode = os.environ.get("TOPOLOW_SHARD_THREAD_PER_BLOCK", "0")
print("thread per block:", mode)
for n in (2048, 8192):
    call, _ = pp.random_problem(n, 3, 0.9, seed=5, n_iter=200, k0=5.0, cool=0.01, c_rep=0.01)
    for blocks in (1, 2, 4, 8):
        best = None
        for rep in range(3):
            r = _native.optimize_layout_exact_sharded(
                call.initial_positions, call.degrees, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, 200,
                5.0, 0.01, 0.01, 1e-12, 10 ** 9, 3, devices=[0] * blocks, seed=3, slab_stages=4)
            if best is None or r.info["loop_seconds"] < best.info["loop_seconds"]:
                best = r
        i = best.info
        out[f"n{n}_b{blocks}"] = i
        print(f"n={n} blocks={i['blocks']}: loop {1e3 * i['loop_seconds']:.2f} ms, {i['exchanges']} exchanges -> "
              f"{1e6 * i['loop_seconds'] / i['exchanges']:.1f} us per stage+exchange; block-0 stage kernels "
              f"{1e3 * i['stage_kernel_seconds']:.2f} ms, checks {1e3 * i['check_kernel_seconds']:.2f} ms", flush=True)
json.dump(out, open(sys.argv[1] + "." + mode, "w"))
