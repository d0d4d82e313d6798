"""Study (not a test): how often does a production-path run of a pinned problem end far from the rest, and what does such
a run look like?   python tests/study/outlier_probe.py <problem> <seeds>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import parity_problems as pp
from tests.conftest import layout_call_args
from topolow_amd import _native

name, n_seeds = sys.argv[1], int(sys.argv[2])
call = pp.build(name)[0]
runs = [_native.optimize_layout_exact_arrays(*layout_call_args(call), seed=1 + s, schedule="slab") for s in range(n_seeds)]
fm = np.array([r.final_mae for r in runs])
med = float(np.median(fm))
bad = [s for s in range(n_seeds) if fm[s] > 1.1 * med]
print(f"{name}: {n_seeds} seeds, median {med:.5f}, outliers (> 1.1 median): {len(bad)}: "
      + ", ".join(f"seed {1 + s}: {fm[s]:.4f} it {runs[s].iterations} conv {runs[s].converged}" for s in bad), flush=True)
