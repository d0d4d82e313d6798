"""Schedule study (CPU, not a test): does a slab-parallel schedule reproduce what the
reference's shuffled Gauss-Seidel sweep produces?

Compares, on the same inputs,
  GS    -- the oracle (reference semantics, shuffled all-pairs Gauss-Seidel, f64)
  SLAB  -- a NumPy model of the schedule the HIP large-N path uses: every iteration is cut
           into S stages; in a stage every point accumulates its own half of each pair
           update over one contiguous column slab from positions frozen at the stage start,
           then all points move at once.
Run:  python tests/study/schedule_study.py [n] [ndim] [missing] [n_iter]
"""
import sys
import os
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402
from oracle import topolow_oracle as orc  # noqa: E402
from topolow_amd import core, synthetic  # noqa: E402


def slab_stage(pos, D, T, g, cols, k, c_rep):
    """Row-owner half-updates of all points against the column slab `cols`."""
    delta = pos[cols][None, :, :] - pos[:, None, :]            # (n, w, d)
    r = np.sqrt((delta * delta).sum(-1))
    rs = r + 0.01
    t = D[:, cols]
    code = T[:, cols]
    measured = np.isfinite(t)
    with np.errstate(invalid="ignore"):
        spring = measured & ((code == 0) | ((code == 1) & (r < t)) | ((code == -1) & (r > t)))
    tt = np.where(measured, t, 0.0)
    f_spring = 2.0 * k * (tt - r) / rs / (4.0 * g[:, None] + k)
    f_rep = c_rep / (2.0 * rs ** 3) / g[:, None]
    coef = np.where(spring, f_spring, f_rep)
    coef[np.arange(pos.shape[0])[:, None] == cols[None, :]] = 0.0
    return pos - (delta * coef[:, :, None]).sum(1)


def run_slab(call, S, seed, stage_policy=None, trace=False):
    n = call.initial_positions.shape[0]
    pos = call.initial_positions.copy()
    D, T = call.dissimilarity_matrix, call.threshold_matrix
    g = call.degrees.astype(np.float64) + 1.0
    rng = np.random.default_rng(seed)
    k = call.k0
    best = dict(mae=np.finfo(float).max, k=call.k0, it=0, pos=pos.copy())
    plateau = worsen = 0
    converged = False
    eps, W = call.relative_epsilon, call.convergence_window
    maes = []
    for it in range(call.n_iter):
        s_now = S if stage_policy is None else stage_policy(k, S)
        w = -(-n // s_now)
        off = int(rng.integers(0, n))
        order = rng.permutation(s_now)
        for s in order:
            cols = (off + s * w + np.arange(min(w, n - s * w))) % n
            if cols.size:
                pos = slab_stage(pos, D, T, g, cols, k, call.c_repulsion)
        k *= 1.0 - call.cooling_rate
        if (it + 1) % call.convergence_check_freq == 0 or it == call.n_iter - 1:
            s_, c_ = orc.edge_error(pos, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
            err = s_ / c_ if c_ else 0.0
            maes.append(err)
            if err < best["mae"] * (1 - eps):
                best.update(mae=err, k=k, it=it + 1, pos=pos.copy()); plateau = worsen = 0
            elif err <= best["mae"] * (1 + eps):
                if err < best["mae"]:
                    best.update(mae=err, k=k, it=it + 1, pos=pos.copy())
                worsen = 0; plateau += 1
                if plateau >= W:
                    converged = True; break
            else:
                plateau = 0; worsen += 1
                if worsen >= W:
                    converged = True; break
    return best["pos"], converged, best["it"], best["mae"], np.array(maes)


def gs(call, seed):
    return orc.optimize_layout_exact(
        call.initial_positions, call.dissimilarity_matrix, call.threshold_matrix, call.degrees,
        call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.n_iter, call.k0,
        call.cooling_rate, call.c_repulsion, call.relative_epsilon, call.convergence_window,
        call.convergence_check_freq, seed=seed)


def rel_diff(a, b):
    iu = np.triu_indices(a.shape[0], 1)
    return float(np.mean(np.abs(a[iu] - b[iu])) / np.mean(b[iu]))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    ndim = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    missing = float(sys.argv[3]) if len(sys.argv) > 3 else 0.7
    n_iter = int(sys.argv[4]) if len(sys.argv) > 4 else 300
    k0 = float(sys.argv[5]) if len(sys.argv) > 5 else 5.0
    cool = float(sys.argv[6]) if len(sys.argv) > 6 else 0.01
    c_rep = float(sys.argv[7]) if len(sys.argv) > 7 else 0.01
    prob = synthetic.make_problem(n, latent_dim=ndim, missing=missing, seed=12345)
    init = synthetic.initial_positions(prob.dissimilarity, ndim, 12345)
    call = core.prepare_layout_call(prob.dissimilarity, ndim, n_iter, k0, cool, c_rep, 1e-4, 5,
                                    init, False, 3, True)
    print(f"n={n} ndim={ndim} missing={missing} iters={n_iter} k0={k0} cool={cool} c_rep={c_rep} "
          f"E={call.edge_i.size}")
    truth = prob.dissimilarity
    gs_runs = []
    for seed in range(4):
        t0 = time.time()
        r = gs(call, seed)
        est, mae = oracle.post_metrics(r.positions, truth)
        gs_runs.append((r, est, mae))
        print(f"GS   seed={seed}: conv={r.converged} best_iter={r.iterations} ran={r.iters_run} "
              f"final_mae={r.final_mae:.5f} mae={mae:.5f}  ({time.time() - t0:.1f}s)")
    base = gs_runs[0][1]
    for q in range(1, len(gs_runs)):
        print(f"  est_distances GS seed {q} vs seed 0: mean rel diff {rel_diff(gs_runs[q][1], base):.4f}")
    for S in (1, 2, 4, 8, 16, 32):
        for seed in range(2):
            t0 = time.time()
            try:
                pos, conv, it, fm, maes = run_slab(call, S, seed)
            except FloatingPointError:
                print(f"SLAB S={S}: fp error"); continue
            est, mae = oracle.post_metrics(pos, truth)
            print(f"SLAB S={S:2d} seed={seed}: conv={conv} best_iter={it} final_mae={fm:.5f} "
                  f"mae={mae:.5f} relD_vs_GS0={rel_diff(est, base):.4f} ({time.time() - t0:.1f}s)")


if __name__ == "__main__":
    main()
