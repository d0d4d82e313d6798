"""Study (not a test): final-MAE statistics of the device schedules on the parity problems, for
comparison with the committed oracle distributions (tests/golden/oracle_dist_*.json).

  python tests/study/gpu_schedule_bias.py <out.json> <problem|cfg3> <seeds> <variant> [<variant> ...]
      variant: slab:<S>  (S = 0 adaptive)   |   gs  (exact tile/one-workgroup GS, f32 above the LDS limit)
               trace:<S>  slab with the MAE of every check recorded
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from tests import parity_problems as pp
from tests.conftest import layout_call_args
from topolow_amd import _native


def main():
    out_path, name, n_seeds = sys.argv[1], sys.argv[2], int(sys.argv[3])
    variants = sys.argv[4:]
    if name == "cfg3":
        call, _ = pp.cfg3_generator(10000)
    else:
        call, _ = pp.build(name)
    n, dim = call.initial_positions.shape
    res = {"problem": name, "n": n, "damp": os.environ.get("TOPOLOW_SLAB_EARLY", "-"), "variants": {}}
    for v in variants:
        kind, _, arg = v.partition(":")
        rows = []
        t0 = time.time()
        if kind in ("slab", "trace"):
            s = _native.Session(n, dim, precision="f32")
            s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
            s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
            for seed in range(n_seeds):
                s.set_positions(call.initial_positions)
                s.begin(call.n_iter, call.k0, call.cooling_rate, call.c_repulsion, call.relative_epsilon,
                        call.convergence_window, call.convergence_check_freq, seed, int(arg or 0))
                trace = []
                if kind == "trace":
                    while s.enqueue(call.convergence_check_freq) > 0:
                        it, stopped, mae = s.sync()
                        trace.append(mae)
                        if stopped:
                            break
                else:
                    s.run()
                r = s.finish()
                rows.append(dict(seed=seed, final_mae=r.final_mae, iterations=r.iterations, converged=bool(r.converged),
                                 trace=trace))
            s.close()
        else:
            for seed in range(n_seeds):
                r = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, schedule="gs",
                                                         precision="f32" if n > 1024 else "f64")
                rows.append(dict(seed=seed, final_mae=r.final_mae, iterations=r.iterations, converged=bool(r.converged)))
        fm = np.array([r["final_mae"] for r in rows])
        res["variants"][v] = dict(mean=float(fm.mean()), sd=float(fm.std(ddof=1)) if len(fm) > 1 else 0.0,
                                  iters=float(np.mean([r["iterations"] for r in rows])), seconds=time.time() - t0,
                                  runs=rows)
        if kind == "trace":
            for r in rows:
                t = r["trace"]
                print("   seed", r["seed"], "MAE@6,9,12,33,63:", " ".join("%.4f" % t[q] for q in (1, 2, 3, 10, 20) if q < len(t)),
                      "final %.5f" % r["final_mae"], flush=True)
        print(name, "early", os.environ.get("TOPOLOW_SLAB_EARLY", "-"), v, "mean %.5f sd %.5f iters %.0f (%.0f s)" %
              (fm.mean(), res["variants"][v]["sd"], res["variants"][v]["iters"], time.time() - t0), flush=True)
    json.dump(res, open(out_path, "w"))


if __name__ == "__main__":
    main()
