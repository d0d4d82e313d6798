import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.models import slab_model as sm
from topolow_amd import core, synthetic
import oracle
from oracle import topolow_oracle as orc

def make_plan(n, S, n_iter, rng, align=4, random_offset=True):
    w = -(-n // S); w = -(-w // align) * align
    plan = np.zeros((n_iter, S + 1, 4), np.int32); ns = np.zeros(n_iter, np.int32)
    for it in range(n_iter):
        off = (int(rng.integers(0, w)) // align) * align if random_offset else 0
        cuts = [0] + [c for c in range(off, n, w) if c > 0] + [n]
        slabs = [(cuts[q], cuts[q+1]) for q in range(len(cuts)-1)]
        # merge first partial and last partial into one wrap-around stage
        if off > 0 and len(slabs) > S:
            first = slabs[0]; last = slabs[-1]; mid = slabs[1:-1]
            stages = [(last[0], last[1], first[0], first[1])] + [(a, b, 0, 0) for a, b in mid]
        else:
            stages = [(a, b, 0, 0) for a, b in slabs]
        order = rng.permutation(len(stages))
        for q, s in enumerate(order): plan[it, q] = stages[s]
        ns[it] = len(stages)
    return plan, ns

def run_slab(call, S_policy, seed, arith="f64"):
    n = call.initial_positions.shape[0]
    rng = np.random.default_rng(seed)
    pos = call.initial_positions.copy(); k = call.k0
    best = dict(mae=np.finfo(float).max, k=k, it=0, pos=pos.copy()); plateau = worsen = 0; conv = False
    it = 0; F = call.convergence_check_freq; eps = call.relative_epsilon; W = call.convergence_window
    while it < call.n_iter:
        chunk = min(F, call.n_iter - it)
        S = S_policy(k)
        plan, ns = make_plan(n, S, chunk, rng)
        pos, k = sm.run(pos, call.dissimilarity_matrix, call.threshold_matrix, call.degrees, plan, ns, k, call.cooling_rate, call.c_repulsion, arith)
        it += chunk
        s_, c_ = orc.edge_error(pos, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        err = s_ / c_ if c_ else 0.0
        if err < best["mae"] * (1 - eps): best.update(mae=err, k=k, it=it, pos=pos.copy()); plateau = worsen = 0
        elif err <= best["mae"] * (1 + eps):
            if err < best["mae"]: best.update(mae=err, k=k, it=it, pos=pos.copy())
            worsen = 0; plateau += 1
            if plateau >= W: conv = True; break
        else:
            plateau = 0; worsen += 1
            if worsen >= W: conv = True; break
    return best["pos"], conv, best["it"], best["mae"], it

def rel_diff(a, b):
    iu = np.triu_indices(a.shape[0], 1); return float(np.mean(np.abs(a[iu] - b[iu])) / np.mean(b[iu]))

def case(n, ndim, missing, n_iter, k0, cool, c_rep, policies, nseed_gs=3, nseed_slab=2, spectral=False):
    prob = synthetic.make_problem(n, latent_dim=ndim, missing=missing, seed=777)
    init = synthetic.initial_positions(prob.dissimilarity, ndim, 777)
    call = core.prepare_layout_call(prob.dissimilarity, ndim, n_iter, k0, cool, c_rep, 1e-4, 5, init, False, 3, not spectral)
    truth = call.reordered_matrix
    print(f"--- n={n} d={ndim} miss={missing} iters={n_iter} k0={k0} cool={cool} c_rep={c_rep} spectral={spectral}", flush=True)
    for seed in range(nseed_gs):
        t0 = time.time()
        r = orc.optimize_layout_exact(call.initial_positions, call.dissimilarity_matrix, call.threshold_matrix, call.degrees, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.n_iter, call.k0, call.cooling_rate, call.c_repulsion, call.relative_epsilon, call.convergence_window, call.convergence_check_freq, seed=seed)
        est, mae = oracle.post_metrics(r.positions, truth)
        if seed == 0: base = est
        print(f"GS seed={seed} conv={r.converged} it={r.iterations} ran={r.iters_run} fm={r.final_mae:.4f} mae={mae:.4f} relD={rel_diff(est, base):.4f} ({time.time()-t0:.0f}s)", flush=True)
    for name, pol in policies:
        for seed in range(nseed_slab):
            t0 = time.time()
            pos, conv, it, fm, ran = run_slab(call, pol, seed)
            est, mae = oracle.post_metrics(pos, truth)
            print(f"SLAB {name} seed={seed} conv={conv} it={it} ran={ran} fm={fm:.4f} mae={mae:.4f} relD={rel_diff(est, base):.4f} ({time.time()-t0:.0f}s)", flush=True)

if __name__ == "__main__":
    import math
    def adaptive(k): return max(4, 1 << max(0, math.ceil(math.log2(max(k / 2.5, 1e-9)))))
    pols = [("S4", lambda k: 4), ("S16", lambda k: 16), ("S64", lambda k: 64), ("adapt", adaptive)]
    which = sys.argv[1]
    if which == "a": case(1500, 5, 0.7, 1000, 5, 0.01, 0.01, pols)
    if which == "b": case(1500, 5, 0.7, 1000, 14.76, 0.0364, 0.00294, [("S8", lambda k: 8), ("S16", lambda k: 16), ("S64", lambda k: 64), ("adapt", adaptive)])
    if which == "c": case(3000, 5, 0.7, 300, 5, 0.01, 0.01, [("S4", lambda k: 4), ("S32", lambda k: 32)], nseed_gs=2, nseed_slab=1)
    if which == "d": case(1500, 5, 0.7, 1000, 14.76, 0.0364, 0.00294, [("S8", lambda k: 8), ("adapt", adaptive)], spectral=True)
