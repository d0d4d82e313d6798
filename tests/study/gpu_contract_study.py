"""Study (not a test): final-MAE statistics of the production path (slab schedule with random labels through the
one-shot entry) against the oracle distributions of tests/golden -- the numbers behind DESIGN.md section 2b and the
bands of tests/test_gpu_contract.py.
  python tests/study/gpu_contract_study.py <out.json> <problem|cfg3> <seeds>
Round 2 ran it (and variants of it, with knobs that are no longer in the library) to decide: index-contiguous against
random labels, the stage floor of the unfolding phase (4 / 16 / 32 / 64 stages for 12-100 iterations), the stage
floor afterwards (4 / 2 / 1)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import parity_problems as pp
from tests.conftest import layout_call_args
from topolow_amd import _native

out_path, name, n_seeds = sys.argv[1], sys.argv[2], int(sys.argv[3])
call = pp.cfg3_generator(10000)[0] if name == "cfg3" else pp.build(name)[0]
t0 = time.time()
runs = [_native.optimize_layout_exact_arrays(*layout_call_args(call), seed=1 + s, schedule="slab") for s in range(n_seeds)]
fm = np.array([r.final_mae for r in runs])
if name == "cfg3":
    recs = pp.cfg3_oracle_records()
    ref = np.array([r["final_mae"] for r in recs]); m, sd = ref.mean(), ref.std(ddof=1)
else:
    try:
        d = pp.oracle_distribution(name); m, sd = d["mean_final_mae"], d["sd_final_mae"]
    except FileNotFoundError:      # device side first; the oracle's distribution is compared when it exists
        m, sd = float(fm.mean()), float("nan")
band = max(3 * sd, 0.01 * m)
print(f"{name}: mean {fm.mean():.5f} sd {fm.std(ddof=1):.5f} "
      f"[{fm.min():.4f}, {fm.max():.4f}] | oracle {m:.5f} sd {sd:.5f} band {band:.5f} -> diff {fm.mean() - m:+.5f} "
      f"({100 * (fm.mean() / m - 1):+.2f} %) {'OK' if abs(fm.mean() - m) <= band else 'OUT'}; iters "
      f"{np.mean([r.iterations for r in runs]):.0f}; device s/run {np.mean([r.info['device_seconds'] for r in runs]):.4f} "
      f"({time.time() - t0:.0f} s)", flush=True)
json.dump(dict(problem=name, final_mae=fm.tolist(), iterations=[r.iterations for r in runs]), open(out_path, "w"))
