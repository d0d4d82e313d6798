"""Generates tests/golden/relax_golden.json.

The reference (R + Rcpp) cannot run in this image and ships no numeric vectors for its
relaxation kernel, so these known-answer vectors come from a SECOND, independent restatement
written here in plain Python floats (IEEE double, same operation order as the reference's
src/optimization.cpp:203-289 and :54-81, :303-357).  They pin the C++ oracle against
transcription slips; they are not outputs of the reference itself.

Run: python tests/golden/make_golden.py
"""
import json
import math
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))


def pair_step(pos, i, j, target, code, gi, gj, k, c_rep):
    dim = len(pos[0])
    dist_sq = 0.0
    for d in range(dim):
        diff = pos[j][d] - pos[i][d]
        dist_sq += diff * diff
    dist = math.sqrt(dist_sq)
    dist_stable = dist + 0.01
    spring = False
    if math.isfinite(target):
        if code == 0:
            spring = True
        elif code == 1:
            spring = dist < target
        else:
            spring = dist > target
    if spring:
        factor = 2.0 * k * (target - dist) / dist_stable
        ni = 4.0 * gi + k
        nj = 4.0 * gj + k
        for d in range(dim):
            delta = pos[j][d] - pos[i][d]
            f = delta * factor
            pos[i][d] -= f / ni
            pos[j][d] += f / nj
    else:
        mag = c_rep / (2.0 * dist_stable * dist_stable * dist_stable)
        for d in range(dim):
            delta = pos[j][d] - pos[i][d]
            f = delta * mag
            pos[i][d] -= f / gi
            pos[j][d] += f / gj


def edge_error(pos, edges):
    total, cnt = 0.0, 0
    for (a, b, t, c) in edges:
        q = 0.0
        for d in range(len(pos[0])):
            diff = pos[b][d] - pos[a][d]
            q += diff * diff
        r = math.sqrt(q)
        if c == 0 or (c == 1 and r < t) or (c == -1 and r > t):
            total += abs(t - r)
            cnt += 1
    return total, cnt


def controller(maes, iters, ks, k0, window, eps):
    best, best_k, best_it = 1.7976931348623157e308, k0, 0
    plateau = worsen = 0
    snaps, stopped = [], -1
    for o, (e, it, k) in enumerate(zip(maes, iters, ks)):
        snap = False
        if e < best * (1.0 - eps):
            best, best_k, best_it, snap = e, k, it, True
            plateau = worsen = 0
        elif e <= best * (1.0 + eps):
            if e < best:
                best, best_k, best_it, snap = e, k, it, True
            worsen = 0
            plateau += 1
            if plateau >= window:
                snaps.append(snap); stopped = o; break
        else:
            plateau = 0
            worsen += 1
            if worsen >= window:
                snaps.append(snap); stopped = o; break
        snaps.append(snap)
    return dict(stopped_at=stopped, snapshots=snaps, best_mae=best, best_k=best_k, best_iter=best_it)


def main():
    rnd = random.Random(20240607)
    out = {}

    # G1: one pair, each branch
    g1 = []
    cases = [
        ("spring_exact", 1.7, 0), ("gt_violated_spring", 3.0, 1), ("gt_satisfied_repulse", 0.5, 1),
        ("lt_violated_spring", 0.5, -1), ("lt_satisfied_repulse", 3.0, -1),
        ("unmeasured_repulse", float("inf"), 0), ("coincident_points", 1.0, 0),
        ("coincident_unmeasured", float("inf"), 0),
    ]
    for name, target, code in cases:
        p = [[0.3, -0.2, 1.1], [1.0, 0.9, 0.4]]
        if name.startswith("coincident"):
            p = [[0.5, 0.5, 0.5], [0.5, 0.5, 0.5]]
        before = [row[:] for row in p]
        pair_step(p, 0, 1, target, code, 3.0, 6.0, 2.5, 0.07)
        g1.append(dict(name=name, before=before, target=("inf" if math.isinf(target) else target),
                       code=code, gi=3.0, gj=6.0, k=2.5, c_rep=0.07, after=p))
    out["G1_single_pair"] = g1

    # G2: full sweeps in a supplied pair order, with cooling and the MAE afterwards
    g2 = []
    for n, dim, iters in ((5, 2, 3), (8, 3, 4), (33, 5, 2)):
        pos = [[rnd.uniform(-2, 2) for _ in range(dim)] for _ in range(n)]
        D = [[float("inf")] * n for _ in range(n)]
        T = [[0] * n for _ in range(n)]
        edges = []
        for a in range(n):
            for b in range(a + 1, n):
                u = rnd.random()
                if u < 0.55:
                    t = round(rnd.uniform(0.2, 4.0), 3)
                    c = 0 if u < 0.4 else (1 if u < 0.48 else -1)
                    D[a][b] = D[b][a] = t
                    T[a][b] = T[b][a] = c
                    edges.append((a, b, t, c))
        deg = [sum(1 for b in range(n) if math.isfinite(D[a][b])) + 1 for a in range(n)]  # diag counts
        k, cool, c_rep = 3.0, 0.05, 0.02
        start = [row[:] for row in pos]
        orders = []
        for it in range(iters):
            pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
            rnd.shuffle(pairs)
            orders.append(pairs)
            for (a, b) in pairs:
                pair_step(pos, a, b, D[a][b], T[a][b], deg[a] + 1.0, deg[b] + 1.0, k, c_rep)
            k *= (1.0 - cool)
        s, c = edge_error(pos, edges)
        g2.append(dict(n=n, dim=dim, iters=iters, k0=3.0, cooling=cool, c_rep=c_rep, start=start,
                       D=[["inf" if math.isinf(x) else x for x in row] for row in D], T=T,
                       degrees=deg, orders=orders, final=pos, final_k=k, err_sum=s, err_cnt=c,
                       edges=[list(e) for e in edges]))
    out["G2_supplied_order_sweeps"] = g2

    # G3: edge error on mixed thresholds incl. zero contributing edges
    pos = [[0.0, 0.0], [3.0, 4.0], [6.0, 8.0], [1.0, 1.0]]
    e1 = [(0, 1, 4.0, 0), (0, 2, 12.0, 1), (1, 2, 4.0, 1), (0, 3, 1.0, -1), (1, 3, 9.0, -1)]
    s1, c1 = edge_error(pos, e1)
    e2 = [(0, 1, 4.0, 1), (1, 2, 9.0, -1)]  # both satisfied -> nothing contributes
    s2, c2 = edge_error(pos, e2)
    out["G3_edge_error"] = [dict(pos=pos, edges=[list(e) for e in e1], sum=s1, count=c1),
                            dict(pos=pos, edges=[list(e) for e in e2], sum=s2, count=c2)]

    # G4: controller scripts
    def ks(n, k0=5.0, cool=0.1, freq=3):
        return [k0 * (1 - cool) ** (freq * (o + 1)) for o in range(n)]
    scripts = {
        "improve_then_plateau_stop": [1.0, 0.8, 0.7, 0.69999, 0.69998, 0.70001, 0.69997, 0.7],
        "worsening_stop": [1.0, 0.5, 0.6, 0.7, 0.8, 0.9],
        "never_converge": [1.0, 0.9, 0.8, 0.7, 0.6, 0.5],
        "nan_errors": [1.0, float("nan"), float("nan"), float("nan")],
        "plateau_interrupted": [1.0, 1.00005, 0.99996, 1.2, 1.00001, 0.99999, 1.0, 1.0],
        "zero_error": [0.0, 0.0, 0.0, 0.0],
    }
    g4 = []
    for name, maes in scripts.items():
        iters = [3 * (o + 1) for o in range(len(maes))]
        kk = ks(len(maes))
        for window in (1, 3):
            r = controller(maes, iters, kk, 5.0, window, 1e-4)
            g4.append(dict(name=name, window=window, eps=1e-4, k0=5.0,
                           maes=["nan" if (isinstance(m, float) and math.isnan(m)) else m for m in maes],
                           iters=iters, ks=kk, expect=r))
    out["G4_controller"] = g4

    with open(os.path.join(HERE, "relax_golden.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", os.path.join(HERE, "relax_golden.json"))


if __name__ == "__main__":
    main()
