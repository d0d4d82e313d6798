"""GPU study: multi-stage iterations as symmetric sweeps (TOPOLOW_SYMMETRIC_TWO_STAGE) against the row-owner stages, seed for seed, on problems
without an oracle distribution -- config 3b at full size (10 % of the measured pairs censored), a stiffer spring
(k0 = 6: the longest two-stage phase the policy allows) and f64.  usage: python tests/study/two_stage_ab.py [seeds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import parity_problems as pp  # noqa: E402
from tests.conftest import layout_call_args  # noqa: E402
from topolow_amd import _native  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cases = [("config 3b (10 % censored), fp32", pp.cfg3_generator(10000, censored=0.1)[0], "f32"),
         ("config 3, k0 = 6, fp32", pp.cfg3_generator(10000, k0=6.0)[0], "f32"),
         ("config 3, f64", pp.cfg3_generator(10000)[0], "f64"),
         ("config 3, k0 = 12 (4-, 2-, 1-stage phases), fp32", pp.cfg3_generator(10000, k0=12.0)[0], "f32"),
         ("config 3, k0 = 20 (8-, 4-, 2-, 1-stage phases), fp32", pp.cfg3_generator(10000, k0=20.0)[0], "f32")]
if len(sys.argv) > 2:
    cases = [c for c in cases if sys.argv[2] in c[0]]
for name, call, prec in cases:
    out = {}
    for v in ("1", "0"):
        os.environ["TOPOLOW_SYMMETRIC_TWO_STAGE"] = v
        n_seeds = seeds if prec == "f32" else max(8, seeds // 4)
        runs = [_native.optimize_layout_exact_arrays(*layout_call_args(call), seed=1 + s, schedule="slab", precision=prec)
                for s in range(n_seeds)]
        fm = np.array([r.final_mae for r in runs])
        out[v] = fm
        print(f"{name}: TWO_STAGE={v}: mean {fm.mean():.5f} sd {fm.std(ddof=1):.5f} [{fm.min():.4f}, {fm.max():.4f}] "
              f"iterations {np.mean([r.iterations for r in runs]):.0f} device s/run "
              f"{np.mean([r.info['device_seconds'] for r in runs]):.4f} ({n_seeds} seeds)", flush=True)
    d = out["1"] - out["0"]
    print(f"   paired difference: mean {d.mean():+.5f} ({100 * d.mean() / out['0'].mean():+.3f} %), sd {d.std(ddof=1):.5f}", flush=True)
