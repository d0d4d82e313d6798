"""BASELINE config 5 (not a test): Euclidify-style 5-fold CV sweep on the HIV panel (335 points),
many parameter sets batched into one launch of the exact-GS kernel.  Prints embeddings/s, the CV
MAE at the published parameters (reference: 1.315 for HIV, 0.799 for H3N2 at 20 folds), and the CPU
oracle's time for one such embedding."""
import csv, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.conftest import layout_call_args
from tests.test_gpu_assays import h3n2_matrix, hiv_matrix, H3N2, HIV
from topolow_amd import _native, core, cv
from oracle import topolow_oracle as orc

def main():
    n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(2025)
    hv = hiv_matrix()
    # LHS-like sample of the default search ranges (R/core.R:891-894 of the reference)
    sets = [dict(N=int(rng.integers(2, 11)), k0=float(rng.uniform(0.1, 20)),
                 cooling_rate=float(rng.uniform(1e-4, 0.05)), c_repulsion=float(rng.uniform(1e-4, 0.05)))
            for _ in range(n_sets)]
    t0 = time.time()
    cv.likelihood_sweep(hv, sets[:2], 50, 1e-4, folds=5, rng=np.random.default_rng(1))   # the process's first call:
    first_call = time.time() - t0                                                          # library load, HIP start-up
    t0 = time.time()
    res, secs, n_emb = cv.likelihood_sweep(hv, sets, 500, 1e-4, folds=5, rng=rng)
    wall = time.time() - t0
    ok = [r for r in res if np.isfinite(r["Holdout_MAE"])]
    best = min(ok, key=lambda r: r["Holdout_MAE"])
    out = dict(embeddings=n_emb, device_seconds=secs, wall_seconds=wall, first_small_call_seconds=first_call, embeddings_per_s_device=n_emb / secs,
               embeddings_per_s_wall=n_emb / wall, finite_sets=len(ok), best_holdout_mae=best["Holdout_MAE"],
               mean_iter=float(np.mean([r["mean_iter"] for r in ok])))
    pub, _, _ = cv.likelihood_sweep(hv, [HIV], 500, 1e-4, folds=20, rng=rng)
    out["hiv_published_params_20fold"] = pub[0]
    pub, _, _ = cv.likelihood_sweep(h3n2_matrix(), [dict(N=4, **H3N2)], 500, 1e-4, folds=20, rng=rng)
    out["h3n2_published_params_20fold"] = pub[0]
    # CPU oracle: one N=335 embedding at the published HIV parameters
    call = core.prepare_layout_call(hv, 2, 500, HIV["k0"], HIV["cooling_rate"], HIV["c_repulsion"], 1e-4, 5,
                                    None, False, 3, False, rng)
    t0 = time.time(); r = orc.optimize_layout_exact(*layout_call_args(call), seed=1); cpu = time.time() - t0
    out["cpu_oracle_one_embedding_s"] = cpu
    out["cpu_oracle_iters_run"] = r.iters_run
    print(json.dumps(out))

main()
