"""GPU study (not a test): distribution of final MAE / iterations for the slab schedule (f32, f64)
and the on-device exact GS schedule against the CPU oracle, same problem as
tests/test_gpu_parity.py::test_slab_statistical_parity_with_oracle."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.conftest import layout_call_args
from topolow_amd import _native, core, synthetic
from oracle import topolow_oracle as orc

def main():
    n, dim = int(sys.argv[1]) if len(sys.argv) > 1 else 1500, 5
    k0, cool, c_rep = (14.76, 0.0364, 0.00294) if len(sys.argv) < 3 else (5.0, 0.01, 0.01)
    prob = synthetic.make_problem(n, latent_dim=dim, missing=0.7, seed=777)
    init = synthetic.initial_positions(prob.dissimilarity, dim, 777)
    call = core.prepare_layout_call(prob.dissimilarity, dim, 1000, k0, cool, c_rep, 1e-4, 5, init, False, 3, True)
    out = {}
    for name, kw in (("slab_f32", dict(schedule="slab")), ("slab_f64", dict(schedule="slab", precision="f64")),
                     ("slab_f32_S16", dict(schedule="slab", slab_stages=16)),
                     ("gs_f64", dict(schedule="gs"))):
        rows = []
        for seed in range(8):
            t0 = time.time()
            r = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, **kw)
            rows.append((r.final_mae, r.iterations, r.converged, round(time.time() - t0, 3)))
        out[name] = rows
        print(name, "mae mean %.4f sd %.4f" % (np.mean([x[0] for x in rows]), np.std([x[0] for x in rows])),
              "iters", [x[1] for x in rows], "t", [x[3] for x in rows], flush=True)
    rows = []
    for seed in range(6):
        r = orc.optimize_layout_exact(*layout_call_args(call), seed=seed)
        rows.append((r.final_mae, r.iterations, r.converged))
    out["oracle"] = rows
    print("oracle mae mean %.4f sd %.4f" % (np.mean([x[0] for x in rows]), np.std([x[0] for x in rows])),
          "iters", [x[1] for x in rows], flush=True)
    print(json.dumps(out))

main()
