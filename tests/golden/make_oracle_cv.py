"""CV scores of the CPU oracle at the parameter sets for which the reference ships ITS scores
(tests/golden/ref_results/, copied by make_reference_results.py):

  * the rows of chain_sample_<DS>.csv (H3N2, HIV, DENV) -- each a likelihood_function() call of the reference (20
    folds, mapping_max_iter 500, relative_epsilon 1e-4; inst/examples/parameter-fitting-h3n2.Rmd:186-207,
    parameter-fitting-denv.Rmd:200-222) with its Holdout_MAE and NLL;
  * the notebook's 20-fold comparison (inst/examples/methods-comparison-h3n2-hiv-denv.Rmd:1872-1912: 500 iterations,
    relative_epsilon 1e-10, convergence_counter 3) whose per-fold errors are fold_stats.csv, at the parameters the
    notebook lists and at the ones its rule picks from the chains (the run's own choice is not recorded).

The evaluator is tests/helpers.py::oracle_cv (reference fold rule, oracle embedding per fold, f64).
Writes tests/golden/oracle_cv_<DS>.json.   Run: python tests/golden/make_oracle_cv.py [--jobs 8]
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402


def _one(job):
    ds, idx, params, eps, counter, seed = job
    from tests import parity_problems as pp
    from tests.helpers import oracle_cv
    m = {"H3N2": pp.h3n2_matrix, "HIV": pp.hiv_matrix, "DENV": pp.denv_matrix}[ds]()
    t0 = time.time()
    r = oracle_cv(m, params, 20, np.random.default_rng([seed, idx]), 500, eps, counter, seed0=1000 * idx)
    r["seconds"] = time.time() - t0
    return r


def main():
    from tests import parity_problems as pp
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--only", default="", help="comma-separated data sets (default: all)")
    a = ap.parse_args()
    for ds in ("H3N2", "HIV", "DENV"):
        if a.only and ds not in a.only.split(","):
            continue
        listed = dict({"H3N2": pp.H3N2_LISTED, "HIV": pp.HIV_LISTED, "DENV": pp.DENV_LISTED}[ds])
        opt = {k: pp.ref_chain_optimum(ds)[k] for k in ("N", "k0", "cooling_rate", "c_repulsion")}
        chain = pp.ref_chain_sample(ds)
        jobs, meta = [], []
        for q, row in enumerate(chain):
            jobs.append((ds, q, {k: row[k] for k in ("N", "k0", "cooling_rate", "c_repulsion")}, 1e-4, 5, 11))
            meta.append(dict(kind="chain", row=q, ref_Holdout_MAE=row["Holdout_MAE"], ref_NLL=row["NLL"]))
        # (DENV: the notebook's fold_stats.csv has no DENV rows, and its listed set IS the chain optimum)
        for label, ps in (() if ds == "DENV" else (("listed", listed), ("chain_optimum", opt))):
            for rep in range(3):      # three independent fold draws per parameter set
                jobs.append((ds, 100 + len(jobs), ps, 1e-10, 3, 12))
                meta.append(dict(kind="notebook", params_from=label, rep=rep))
        t0 = time.time()
        with ProcessPoolExecutor(a.jobs) as ex:
            res = list(ex.map(_one, jobs))
        out = []
        for (d, idx, ps, eps, counter, seed), mt, r in zip(jobs, meta, res):
            out.append(dict(mt, params=ps, relative_epsilon=eps, convergence_counter=counter, folds=20,
                            mapping_max_iter=500, **r))
        with open(os.path.join(HERE, f"oracle_cv_{ds}.json"), "w") as fh:
            json.dump(dict(dataset=ds, evaluator="tests/helpers.py::oracle_cv", entries=out), fh)
        ch = [e for e in out if e["kind"] == "chain"]
        rel = np.array([e["Holdout_MAE"] / e["ref_Holdout_MAE"] - 1 for e in ch])
        print(ds, f"{time.time() - t0:.0f} s; chain rows {len(ch)}: oracle/ref - 1: mean {rel.mean():+.4f} "
              f"sd {rel.std(ddof=1):.4f} max |.| {np.abs(rel).max():.4f}", flush=True)
        fs = pp.ref_fold_stats(ds)
        for e in out if len(fs) else ():
            if e["kind"] == "notebook":
                f = np.array(e["fold_mae"])
                print("   notebook CV", e["params_from"], e["rep"], f"mean fold MAE {f.mean():.4f} (sd {f.std(ddof=1):.4f}) "
                      f"reference {fs.mean():.4f} (sd {fs.std(ddof=1):.4f})  pooled {e['Holdout_MAE']:.4f}")


if __name__ == "__main__":
    main()
