"""Study (not a test): does a random relabelling of the points (slabs = random subsets instead of
index-contiguous runs, which are spatially contiguous on the reference's random-walk start,
R/core.R:407-415) remove the run-to-run spread / bias of the slab schedule at config 3?
  python tests/study/gpu_relabel_study.py <out.json> <problem|cfg3> <seeds> <S> [<S> ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import parity_problems as pp
from topolow_amd import _native

out_path, name, n_seeds = sys.argv[1], sys.argv[2], int(sys.argv[3])
call = pp.cfg3_generator(10000)[0] if name == "cfg3" else pp.build(name)[0]
n, dim = call.initial_positions.shape
res = {}
for relabel in (1,):
    for S in [int(v) if v != "gs" else "gs" for v in sys.argv[4:]]:
        rows = []
        t0 = time.time()
        s = None
        for seed in range(n_seeds):
            if relabel:
                perm = np.random.default_rng(1000 + seed).permutation(n)      # new label -> old label
            else:
                perm = np.arange(n)
            if s is None or relabel:
                if s is not None:
                    s.close()
                inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
                D = call.dissimilarity_matrix[np.ix_(perm, perm)]
                T = call.threshold_matrix[np.ix_(perm, perm)]
                ei, ej = inv[call.edge_i], inv[call.edge_j]
                lo, hi = np.minimum(ei, ej), np.maximum(ei, ej)
                s = _native.Session(n, dim, precision="f32")
                s.load_dense(D, T, call.degrees[perm])
                s.set_edges(lo.astype(np.int32), hi.astype(np.int32), call.edge_dist, call.edge_thresh)
            if S == "gs":
                r = _native.optimize_layout_exact_arrays(
                    call.initial_positions[perm], D, T, call.degrees[perm], lo.astype(np.int32), hi.astype(np.int32),
                    call.edge_dist, call.edge_thresh, call.n_iter, call.k0, call.cooling_rate, call.c_repulsion,
                    call.relative_epsilon, call.convergence_window, call.convergence_check_freq, seed=seed,
                    schedule="gs", precision="f32" if n > 1024 else "f64")
                rows.append(dict(seed=seed, final_mae=r.final_mae, iterations=r.iterations, mae6=0.0, mae33=0.0))
                continue
            s.set_positions(call.initial_positions[perm])
            s.begin(call.n_iter, call.k0, call.cooling_rate, call.c_repulsion, call.relative_epsilon,
                    call.convergence_window, call.convergence_check_freq, seed, S)
            s.run()
            r = s.finish()
            tr = s.check_trace()
            rows.append(dict(seed=seed, final_mae=r.final_mae, iterations=r.iterations, mae6=float(tr[1, 1]),
                             mae33=float(tr[10, 1])))
        s.close(); s = None
        fm = np.array([r["final_mae"] for r in rows])
        m6 = np.array([r["mae6"] for r in rows])
        res[f"relabel{relabel}_S{S}"] = dict(mean=float(fm.mean()), sd=float(fm.std(ddof=1)), runs=rows)
        print(f"{name} early={os.environ.get('TOPOLOW_SLAB_EARLY', '-')} relabel={relabel} S={S}: final mean {fm.mean():.5f} sd {fm.std(ddof=1):.5f} "
              f"[{fm.min():.4f}, {fm.max():.4f}]  MAE@6 mean {m6.mean():.3f} sd {m6.std(ddof=1):.3f}  "
              f"iters {np.mean([r['iterations'] for r in rows]):.0f}  ({time.time() - t0:.0f} s)", flush=True)
json.dump(res, open(out_path, "w"))
