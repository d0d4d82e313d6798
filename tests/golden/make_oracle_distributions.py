"""Oracle distributions for the statistical parity contract (SURVEY.md
section 7.3-2): the reference's pair shuffle is seeded from std::random_device
(src/optimization.cpp:153-154 of the reference), so a device schedule is accepted when its mean
final MAE lies within  mean_ref +- max(3 sd_ref, 1 %)  over >= 20 oracle seeds.

This script runs the CPU oracle (reference shuffled Gauss-Seidel order, f64) with 20+ seeds on the
problems the GPU tests use and writes  tests/golden/oracle_dist_<problem>.json :

    final_mae / iterations / iters_run / converged / post_mae  per seed,
    head_dist_mean : mean over seeds of the pairwise distances among the first 48 points,
    head_gap       : per seed, mean relative |d_seed - head_dist_mean| / head_dist_mean.

Problems (definitions shared with the tests through tests/parity_problems.py):
    syn1500_h3n2params   tests/test_gpu_parity.py::_random_problem(1500, 5, .7, seed 777), published
                         H3N2 parameters (k0 14.76, cooling 0.0364, c_rep 0.00294)
    cfg3gen_1500, cfg3gen_2048   BASELINE config 3's generator and parameters at N = 1500 / 2048
    cfg3b_1500           the same with 10 % of the measured pairs censored (">" at the 90th percentile)
    h3n2_ndim4, h3n2_ndim5   BASELINE config 2 (Smith-2004 panel, published parameters)

Run:  python tests/golden/make_oracle_distributions.py [problem ...] [--seeds 20] [--jobs 2]
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HEAD = 48


def head_dist(p):
    p = np.asarray(p)[:HEAD]
    iu = np.triu_indices(p.shape[0], 1)
    return np.sqrt(((p[:, None, :] - p[None, :, :]) ** 2).sum(-1))[iu]


def _one(args):
    name, seed = args
    import oracle
    from oracle import topolow_oracle as orc
    from tests.conftest import layout_call_args
    from tests import parity_problems
    spec = parity_problems.PROBLEMS[name]
    call, truth = spec["fn"](seed) if spec.get("vary_init") else parity_problems.build(name)
    t0 = time.time()
    r = orc.optimize_layout_exact(*layout_call_args(call), seed=seed)
    post = None
    if truth is not None:
        _, post = oracle.post_metrics(r.positions, truth)
    edges = None
    if parity_problems.PROBLEMS[name].get("edges"):     # distances of the measured pairs (rotation-free)
        ei, ej = np.asarray(call.edge_i), np.asarray(call.edge_j)
        edges = np.linalg.norm(r.positions[ei] - r.positions[ej], axis=1).tolist()
    return dict(seed=seed, edges=edges, final_mae=r.final_mae, iterations=int(r.iterations), iters_run=int(r.iters_run),
                converged=bool(r.converged), final_k=r.final_k, post_mae=post, seconds=time.time() - t0,
                head=head_dist(r.positions).tolist())


def main():
    from tests import parity_problems
    ap = argparse.ArgumentParser()
    ap.add_argument("problems", nargs="*", default=list(parity_problems.PROBLEMS))
    ap.add_argument("--seeds", type=int, default=20)
    ap.add_argument("--jobs", type=int, default=2)
    a = ap.parse_args()
    for name in a.problems:
        t0 = time.time()
        with ProcessPoolExecutor(a.jobs) as ex:
            recs = list(ex.map(_one, [(name, 1000 + s) for s in range(a.seeds)]))
        edges = [r.pop("edges") for r in recs]
        heads = np.array([r.pop("head") for r in recs])
        mean_head = heads.mean(0)
        gaps = [float(np.mean(np.abs(h - mean_head) / mean_head)) for h in heads]
        fm = np.array([r["final_mae"] for r in recs])
        out = dict(problem=name, definition=parity_problems.PROBLEMS[name]["doc"], oracle="shuffled Gauss-Seidel "
                   "(reference order, std::shuffle with mt19937(seed)), f64, g++ -O2", n_seeds=a.seeds,
                   mean_final_mae=float(fm.mean()), sd_final_mae=float(fm.std(ddof=1)),
                   mean_iterations=float(np.mean([r["iterations"] for r in recs])),
                   sd_iterations=float(np.std([r["iterations"] for r in recs], ddof=1)),
                   runs=recs, head_points=HEAD, head_dist_mean=[round(float(v), 6) for v in mean_head],
                   head_gap=gaps)
        if edges[0] is not None:
            ed = np.array(edges)
            em = ed.mean(0)
            out["edge_dist_mean"] = [round(float(v), 6) for v in em]
            out["edge_gap"] = [float(np.mean(np.abs(e - em)) / em.mean()) for e in ed]
        path = os.path.join(HERE, f"oracle_dist_{name}.json")
        with open(path, "w") as fh:
            json.dump(out, fh)
        print(f"{name}: mean {fm.mean():.5f} sd {fm.std(ddof=1):.5f} ({100 * fm.std(ddof=1) / fm.mean():.2f} %) "
              f"iters {out['mean_iterations']:.0f} +- {out['sd_iterations']:.0f}  [{time.time() - t0:.0f} s]",
              flush=True)


if __name__ == "__main__":
    main()
