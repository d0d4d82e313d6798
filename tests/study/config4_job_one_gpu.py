"""GPU study: BASELINE config 4 (N = 50 000, 90 % missing, ndim 3) as ONE block on ONE MI355X through the native engine,
the whole job to the controller's own stop -- with the multi-stage iterations on the symmetric sweep (default) and on
the row-owner stages (TOPOLOW_SYMMETRIC_TWO_STAGE=0).  usage: python tests/study/config4_job_one_gpu.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401  (first: it ships its own HIP runtime)
from topolow_amd import _native, sharded  # noqa: E402

n, dim = 50000, 3
for two_stage in ("1", "0"):
    os.environ["TOPOLOW_SYMMETRIC_TWO_STAGE"] = two_stage
    bk = sharded.HipBackend(n, dim, 0, n, 0)
    _ne, scale = sharded.load_synthetic_block(bk, n, 3, 0.9, 12345, 0, 1, rows=(0, n))
    torch.cuda.synchronize()
    rng = np.random.Generator(np.random.PCG64(999))
    init = np.zeros((n, dim))
    init[1:] = np.cumsum(rng.uniform(0.0, 2.0 * scale / n, size=(n - 1, dim)), axis=0)
    ss = [bk.session]
    _native.run_sharded(ss, init, 3, 5.0, 0.01, 0.01, 1e-4, 5, 3, 7)            # code objects, buffers
    for rep in range(2):
        t0 = time.perf_counter()
        r = _native.run_sharded(ss, init, 1000, 5.0, 0.01, 0.01, 1e-4, 5, 3, 7)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        it = r.info["iterations_run"]
        print(f"TOPOLOW_SYMMETRIC_TWO_STAGE={two_stage}: {it} iterations in {r.info['loop_seconds']:.3f} s of loop "
              f"({wall:.3f} s wall) = {it / r.info['loop_seconds']:.1f} iterations/s; converged {r.converged} at "
              f"{r.iterations}, final MAE {r.final_mae:.6f}", flush=True)
    bk.session.close()
    del bk, ss
    torch.cuda.empty_cache()
