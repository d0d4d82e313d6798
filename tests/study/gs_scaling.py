"""GS kernel scaling study (not a test): device time of B identical HIV-sized embeddings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.test_gpu_assays import hiv_matrix, HIV
from topolow_amd import _native, core
hv = hiv_matrix()
for ndim in (2, 5):
    call = core.prepare_layout_call(hv, ndim, 60, HIV["k0"], HIV["cooling_rate"], HIV["c_repulsion"], 1e-12, 1000,
                                    None, False, 3, False, np.random.default_rng(1))
    for prec in ("f64", "f32"):
        for B in (1, 64, 256, 1024, 2048):
            res, secs = _native.optimize_layout_exact_batch([call] * B, seeds=list(range(B)), precision=prec)
            it = res[0].info["iterations_run"]
            rounds = it * 334
            print(f"ndim={ndim} {prec} B={B:5d} device {secs*1e3:8.2f} ms  iters {it}  per-round/WG {secs/rounds*1e6:6.2f} us  "
                  f"emb/s {B/secs:9.1f}", flush=True)
