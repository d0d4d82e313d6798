"""Full-size quality reference (not a test): the CPU oracle (reference schedule, f64) on BASELINE
config 3 -- N=10 000, 70 % missing, ndim=5, k0=5, cooling=0.01, c_repulsion=0.01, default
controller (eps 1e-4, window 5, check every 3) -- run to its own stop.  ~13 s per iteration on
one core, so this takes 1-2 hours; the result is committed as tests/golden/cfg3_oracle_seed<k>.json
and compared with the device run by tests/test_gpu_parity.py::test_cfg3_full_size_vs_oracle_record."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.conftest import layout_call_args
from topolow_amd import core, synthetic
from oracle import topolow_oracle as orc

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
prob = synthetic.make_problem(n, latent_dim=5, missing=0.7, seed=12345)
init = synthetic.initial_positions(prob.dissimilarity, 5, 12345)
call = core.prepare_layout_call(prob.dissimilarity, 5, 1000, 5.0, 0.01, 0.01, 1e-4, 5, init, False, 3, True)
t0 = time.time()
r = orc.optimize_layout_exact(*layout_call_args(call), seed=seed)
out = dict(n=n, seed=seed, seconds=time.time() - t0, converged=r.converged, iterations=r.iterations,
           iters_run=r.iters_run, final_mae=r.final_mae, final_k=r.final_k, mae_trace=r.mae_trace.tolist(),
           positions_head=r.positions[:64].tolist())
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden",
                    f"cfg3_oracle_seed{seed}" + ("" if n == 10000 else f"_n{n}") + ".json")
json.dump(out, open(path, "w"))
print("wrote", path, out["seconds"], out["iterations"], out["final_mae"])
