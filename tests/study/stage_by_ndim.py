"""Stage-kernel time by ndim and by threshold content (not a test): N points, 70 % missing, with and
without 10 % of the measured pairs turned into ">" targets (BASELINE config 3 / 3b shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from topolow_amd import _native, core, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
dims = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 3, 5, 7, 8, 10]
for dim in dims:
    prob = synthetic.make_problem(n, latent_dim=min(dim, 5), missing=0.7, seed=12345)
    init = synthetic.initial_positions(prob.dissimilarity, dim, 12345)
    for thr in (False, True):
        D = prob.dissimilarity
        if thr:
            rng = np.random.default_rng(1)
            Dm = np.array(D, dtype=object)
            iu, ju = np.triu_indices(n, 1)
            meas = ~np.isnan(D[iu, ju])
            pick = rng.random(iu.size) < 0.1
            q90 = np.nanquantile(D[iu, ju], 0.9)
            sel = meas & pick
            vals = np.minimum(D[iu, ju][sel], q90)
            m = core.CodedMatrix(D.copy(), np.zeros((n, n), dtype=np.int32), None, True)
            m.values[iu[sel], ju[sel]] = vals; m.values[ju[sel], iu[sel]] = vals
            m.codes[iu[sel], ju[sel]] = 1; m.codes[ju[sel], iu[sel]] = 1
            src = m
        else:
            src = D
        call = core.prepare_layout_call(src, dim, 30, 5.0, 0.01, 0.01, 1e-4, 1000, init, False, 3, True)
        s = _native.Session(n, dim, precision="f32")
        s.load_dense(call.dissimilarity_matrix, call.threshold_matrix, call.degrees)
        s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        s.set_positions(call.initial_positions)
        s.set_profiling(True)
        s.begin(30, 5.0, 0.01, 0.01, 1e-4, 1000, 3, 7, 4)
        s.run()
        stage_ms, launches, check_ms, checks = s.profile()
        r = s.finish()
        bytes_launch = 4.0 * n * n / 4
        print(f"n={n} ndim={dim:2d} thresholds={int(thr)}  stage {stage_ms / launches * 1e3:7.1f} us  "
              f"{bytes_launch / (stage_ms / launches * 1e-3) / 1e9:7.0f} GB/s  check {check_ms / max(checks, 1) * 1e3:6.1f} us  mae {r.final_mae:.4f}",
              flush=True)
        s.close()
