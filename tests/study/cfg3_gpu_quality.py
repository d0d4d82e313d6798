"""Full-size quality study on the GPU (not a test): BASELINE config 3 run to its own stop with the
slab schedule (f32) and with the exact tile Gauss-Seidel schedule (f32 and f64), several seeds.
Compared with tests/golden/cfg3_oracle_seed*.json (the CPU oracle's record) when present."""
import glob, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.conftest import layout_call_args
from topolow_amd import _native, core, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
prob = synthetic.make_problem(n, latent_dim=5, missing=0.7, seed=12345)
init = synthetic.initial_positions(prob.dissimilarity, 5, 12345)
call = core.prepare_layout_call(prob.dissimilarity, 5, 1000, 5.0, 0.01, 0.01, 1e-4, 5, init, False, 3, True)
out = {}
for name, kw, seeds in (("slab_f32", dict(schedule="slab"), range(4)),
                        ("tilegs_f32", dict(schedule="gs", precision="f32"), range(2)),
                        ("tilegs_f64", dict(schedule="gs", precision="f64"), range(1))):
    rows = []
    for seed in seeds:
        t0 = time.time()
        r = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, **kw)
        rows.append(dict(seed=seed, final_mae=r.final_mae, iterations=r.iterations, converged=r.converged,
                         iterations_run=r.info["iterations_run"], device_seconds=r.info["device_seconds"],
                         total_seconds=r.info["total_seconds"]))
        print(name, rows[-1], flush=True)
    out[name] = rows
for f in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "golden", "cfg3_oracle_seed*.json"))):
    d = json.load(open(f))
    if d.get("n") == n:
        out.setdefault("oracle", []).append({k: d[k] for k in ("seed", "final_mae", "iterations", "converged", "iters_run", "seconds")})
print(json.dumps(out))
