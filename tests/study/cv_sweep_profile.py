"""Where the wall time of a CV sweep goes (not a test): cProfile of cv.likelihood_sweep on the HIV panel."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.test_gpu_assays import hiv_matrix
from topolow_amd import cv

n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(2025)
hv = hiv_matrix()
sets = [dict(N=int(rng.integers(2, 11)), k0=float(rng.uniform(0.1, 20)), cooling_rate=float(rng.uniform(1e-4, 0.05)),
             c_repulsion=float(rng.uniform(1e-4, 0.05))) for _ in range(n_sets)]
cv.likelihood_sweep(hv, sets[:2], 50, 1e-4, folds=5, rng=np.random.default_rng(1))      # warm-up (library, device)
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
res, secs, n_emb = cv.likelihood_sweep(hv, sets, 500, 1e-4, folds=5, rng=rng)
pr.disable()
print("wall", round(time.time() - t0, 3), "device", round(secs, 3), "embeddings", n_emb)
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
