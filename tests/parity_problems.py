"""Named problems of the statistical parity contract: the SAME definitions feed the oracle
distributions committed under tests/golden/oracle_dist_*.json (tests/golden/make_oracle_distributions.py)
and the GPU tests that compare the device schedules with them.  Test infrastructure."""
import csv
import functools
import json
import os

import numpy as np

from topolow_amd import antigenic, core, synthetic

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H3N2 = dict(k0=14.76214, cooling_rate=0.03641074, c_repulsion=0.002943064)   # reference
# inst/examples/methods-comparison-h3n2-hiv-denv.Rmd:312-316


def random_problem(n, dim, missing, seed, thresholds=0.0, n_iter=20, k0=3.0, cool=0.05, c_rep=0.02,
                   check_freq=3, window=5, eps=1e-4):
    """Synthetic mixture problem (topolow_amd.synthetic), optionally with a fraction of the measured
    pairs turned into ">" / "<" thresholds that the true distance satisfies."""
    prob = synthetic.make_problem(n, latent_dim=dim, missing=missing, seed=seed)
    D = prob.dissimilarity
    if thresholds > 0:
        rng = np.random.default_rng(seed + 1)
        M = D.astype(object)
        iu, ju = np.triu_indices(n, 1)
        for a, b in zip(iu, ju):
            if not np.isnan(D[a, b]):
                u = rng.random()
                if u < thresholds / 2:
                    M[a, b] = M[b, a] = ">" + repr(float(D[a, b]) * 0.9)
                elif u < thresholds:
                    M[a, b] = M[b, a] = "<" + repr(float(D[a, b]) * 1.1)
            else:
                M[a, b] = M[b, a] = None
        for a in range(n):
            M[a, a] = "0"
        D = M
    init = synthetic.initial_positions(prob.dissimilarity, dim, seed)
    return core.prepare_layout_call(D, dim, n_iter, k0, cool, c_rep, eps, window, init, False, check_freq,
                                    True), prob


def cfg3_generator(n, censored=0.0, n_iter=1000, eps=1e-4, window=5, check_freq=3, k0=5.0, cool=0.01, c_rep=0.01):
    """BASELINE config 3's generator and parameters at `n` points; `censored`:
    fraction of the measured pairs turned into ">" censoring at the 90th percentile (config 3b)."""
    dim = 5
    prob = synthetic.make_problem(n, latent_dim=dim, missing=0.7, seed=12345)
    D = prob.dissimilarity
    init = synthetic.initial_positions(D, dim, 12345)
    m = D
    if censored > 0:
        iu, ju = np.triu_indices(n, 1)
        vals = D[iu, ju]
        rng = np.random.default_rng(1)
        sel = ~np.isnan(vals) & (rng.random(iu.size) < censored)
        q90 = np.nanquantile(vals, 0.9)
        m = core.CodedMatrix(D.copy(), np.zeros((n, n), dtype=np.int32), None, True)
        m.values[iu[sel], ju[sel]] = np.minimum(vals[sel], q90)
        m.values[ju[sel], iu[sel]] = m.values[iu[sel], ju[sel]]
        m.codes[iu[sel], ju[sel]] = 1
        m.codes[ju[sel], iu[sel]] = 1
    call = core.prepare_layout_call(m, dim, n_iter, k0, cool, c_rep, eps, window, init, False, check_freq, True)
    return call, prob


def h3n2_matrix():
    rows = list(csv.DictReader(open(os.path.join(GOLD, "h3n2_distances.csv"))))
    return antigenic.titers_list_to_matrix(rows, "virusStrain", "virusYear", "serumStrain", "serumYear",
                                           "distance", sort=True)


def h3n2_call(ndim):
    return core.prepare_layout_call(h3n2_matrix(), ndim, 1000, H3N2["k0"], H3N2["cooling_rate"],
                                    H3N2["c_repulsion"], 1e-4, 5, None, False, 3, False, np.random.default_rng(7))


def hiv_matrix():
    rows = list(csv.DictReader(open(os.path.join(GOLD, "hiv_distances.csv"))))
    return antigenic.titers_list_to_matrix(rows, "Virus", "virusYear", "Antibody", None, "distance", sort=True)


def denv_matrix():
    rows = list(csv.DictReader(open(os.path.join(GOLD, "denv_distances.csv"))))
    return antigenic.titers_list_to_matrix(rows, "virus_strain", "virusYear", "serum_strain", "serumYear",
                                           "distance", sort=True)


# ---- the results the reference itself ships (tests/golden/ref_results/, copied by make_reference_results.py) ----
REF_RESULTS = os.path.join(GOLD, "ref_results")
HIV_LISTED = dict(N=2, k0=3.550036, cooling_rate=0.04130713, c_repulsion=0.0007038619)   # ...h3n2-hiv-denv.Rmd:319-323
H3N2_LISTED = dict(N=4, **H3N2)
DENV_LISTED = dict(N=10, k0=7.1, cooling_rate=0.01232407, c_repulsion=0.03830152)                # ...h3n2-hiv-denv.Rmd:326-331


def ref_coordinates(ds):
    """(names, positions) of the embedding the reference ships for data set `ds` ("H3N2" | "HIV" | "DENV")."""
    rows = list(csv.reader(open(os.path.join(REF_RESULTS, f"topolow_{ds}_coords.csv"))))
    return [r[0] for r in rows[1:]], np.array([[float(x) for x in r[1:]] for r in rows[1:]])


def ref_chain_optimum(ds):
    with open(os.path.join(REF_RESULTS, "chain_optimum.json")) as fh:
        return json.load(fh)[ds]


def ref_chain_sample(ds):
    """Rows of the reference's adaptive-sampling chains: dicts N, k0, cooling_rate, c_repulsion, Holdout_MAE, NLL."""
    out = []
    for r in csv.DictReader(open(os.path.join(REF_RESULTS, f"chain_sample_{ds}.csv"))):
        out.append(dict(N=int(round(np.exp(float(r["log_N"])))), k0=float(np.exp(float(r["log_k0"]))),
                        cooling_rate=float(np.exp(float(r["log_cooling_rate"]))),
                        c_repulsion=float(np.exp(float(r["log_c_repulsion"]))),
                        Holdout_MAE=float(r["Holdout_MAE"]), NLL=float(r["NLL"])))
    return out


def ref_fold_stats(ds, algorithm="Topolow"):
    rows = csv.DictReader(open(os.path.join(REF_RESULTS, "fold_stats.csv")))
    return np.array([float(r["OutSampleError"]) for r in rows if r["Dataset"] == ds and r["Algorithm"] == algorithm])


def ref_error_distribution(ds, algorithm="Topolow"):
    """Mean, SD, Median, Q1, Q3 of the reference's pooled signed out-of-sample errors (error_distribution_HIV_H3N2.csv)."""
    for r in csv.DictReader(open(os.path.join(REF_RESULTS, "error_distribution_HIV_H3N2.csv"))):
        if r["Dataset"] == ds and r["Algorithm"] == algorithm:
            return {k: float(r[k]) for k in ("Mean", "SD", "Median", "Q1", "Q3")}
    raise KeyError(ds)


@functools.lru_cache(maxsize=4)
def ref_matrix(ds):
    """The panel of data set `ds` with its rows in the order of the reference's coordinate file (H3N2: that IS the
    order titers_list_to_matrix builds; HIV: the file lists the viruses in another order), so that a run with
    preserve_order relaxes exactly the problem whose solution the reference holds."""
    m = core.coded_matrix({"H3N2": h3n2_matrix, "HIV": hiv_matrix, "DENV": denv_matrix}[ds]())
    names, _ = ref_coordinates(ds)
    at = {nm: q for q, nm in enumerate(m.names)}
    perm = np.array([at[nm] for nm in names])
    return m if np.array_equal(perm, np.arange(len(names))) else m.reordered(perm)


def refrun_call(ds, params, init_seed=7, init=None, n_iter=500, k0=None):
    """The call that wrote the reference's coordinate file (methods-comparison-h3n2-hiv-denv.Rmd:902-912):
    mapping_max_iter 500, relative_epsilon 1e-10, convergence_counter 3, default check frequency."""
    return core.prepare_layout_call(ref_matrix(ds), params["N"], n_iter, params["k0"] if k0 is None else k0,
                                    params["cooling_rate"], params["c_repulsion"], 1e-10, 3, init, False, 3, True,
                                    np.random.default_rng(init_seed))


def _refrun(ds, which, init_seed=7):
    params = ref_chain_optimum(ds) if which == "chain" else dict({"HIV": HIV_LISTED, "H3N2": H3N2_LISTED,
                                                                    "DENV": DENV_LISTED}[ds])
    if ds == "H3N2":
        params["N"] = 5          # the width of the shipped coordinate file
    return refrun_call(ds, params, init_seed), None


def _syn1500():
    call, prob = random_problem(1500, 5, 0.7, seed=777, n_iter=1000, k0=14.76, cool=0.0364, c_rep=0.00294)
    return call, prob.dissimilarity


def _syn_lowdim(n, dim, missing, seed):
    call, prob = random_problem(n, dim, missing, seed=seed, n_iter=1000, k0=5.0, cool=0.01, c_rep=0.01)
    return call, prob.dissimilarity


def _cfg3gen(n, censored=0.0, eps=1e-4, **kw):
    call, prob = cfg3_generator(n, censored, eps=eps, **kw)
    return call, (prob.dissimilarity if censored == 0 else None)


PROBLEMS = {
    "syn1500_h3n2params": dict(fn=_syn1500, doc="random_problem(1500, 5, 0.7, seed=777, n_iter=1000, k0=14.76, "
                               "cool=0.0364, c_rep=0.00294); eps 1e-4, window 5, check every 3"),
    "cfg3gen_1500": dict(fn=functools.partial(_cfg3gen, 1500), doc="cfg3_generator(1500): config 3's generator "
                         "(seed 12345, 70 % missing, ndim 5, k0 5, cooling 0.01, c_rep 0.01), default controller"),
    "cfg3gen_2048": dict(fn=functools.partial(_cfg3gen, 2048), doc="cfg3_generator(2048)"),
    "cfg3b_1500": dict(fn=functools.partial(_cfg3gen, 1500, 0.1), doc="cfg3_generator(1500, censored=0.1): 10 % "
                       "of the measured pairs are '>' thresholds at the 90th percentile"),
    # the relative_epsilon values the reference's own callers pass: Euclidify's final embedding 1e-6
    # (R/core.R:1263), the notebooks 1e-10 (inst/examples/methods-comparison-h3n2-hiv-denv.Rmd:905)
    "cfg3gen_1500_eps1e-6": dict(fn=functools.partial(_cfg3gen, 1500, 0.0, 1e-6),
                                 doc="cfg3_generator(1500, eps=1e-6)"),
    "cfg3gen_1500_eps1e-10": dict(fn=functools.partial(_cfg3gen, 1500, 0.0, 1e-10),
                                  doc="cfg3_generator(1500, eps=1e-10)"),
    # a soft, slowly cooling spring: after the unfolding phase every iteration is ONE Jacobi sweep (k <= 3)
    "cfg3gen_1500_lowk": dict(fn=functools.partial(_cfg3gen, 1500, 0.0, 1e-4, k0=2.0, cool=0.004, c_rep=0.01),
                              doc="cfg3_generator(1500, k0=2.0, cool=0.004)"),
    # low dimensions: the stage policy is k / S <= min(3, ndim) (a Jacobi stage is stable for k / S < 2 ndim)
    "syn1500_ndim2": dict(fn=lambda: _syn_lowdim(1500, 2, 0.7, 11), doc="random_problem(1500, 2, 0.7, seed=11, "
                          "n_iter=1000, k0=5, cool=0.01, c_rep=0.01)"),
    "syn2000_ndim3_sparse": dict(fn=lambda: _syn_lowdim(2000, 3, 0.9, 12), doc="random_problem(2000, 3, 0.9, seed=12, "
                                 "n_iter=1000, k0=5, cool=0.01, c_rep=0.01): BASELINE config 4's shape (ndim 3, 90 % "
                                 "missing) at a size the oracle finishes"),
    # the same two shapes at a size that takes the symmetric sweep (>= 7 168 points): pinned before any change of the
    # schedule's constants (unfolding iterations, k / S bound) is adopted
    "syn7168_ndim2": dict(fn=lambda: _syn_lowdim(7168, 2, 0.7, 21), doc="random_problem(7168, 2, 0.7, seed=21, n_iter=1000, "
                          "k0=5, cool=0.01, c_rep=0.01)"),
    "syn7168_ndim3_sparse": dict(fn=lambda: _syn_lowdim(7168, 3, 0.9, 22), doc="random_problem(7168, 3, 0.9, seed=22, "
                                 "n_iter=1000, k0=5, cool=0.01, c_rep=0.01): BASELINE config 4's shape above the "
                                 "symmetric sweep's size gate"),
    "h3n2_ndim4": dict(fn=lambda: (h3n2_call(4), None), doc="Smith-2004 H3N2 panel (tests/golden/"
                       "h3n2_distances.csv), ndim 4, published parameters, start positions default_rng(7)"),
    "h3n2_ndim5": dict(fn=lambda: (h3n2_call(5), None), doc="the same, ndim 5 (BASELINE config 2)"),
    # (vary_init: the oracle distribution draws new start positions per seed -- fn(init_seed) -- as the reference's
    #  runs do; the other problems fix the start and vary the pair order only)
    # the runs whose results the reference ships (tests/golden/ref_results/): same panel in the coordinate file's row
    # order, preserve_order, 500 iterations, eps 1e-10, window 3; "chain" = the parameters the notebook's rule picks
    # from the shipped chains (chain_optimum.json), "listed" = the ones its text prints (...Rmd:312-323)
    "h3n2_refrun_chain": dict(fn=functools.partial(_refrun, "H3N2", "chain"), edges=True, vary_init=True,
                              doc="refrun_call('H3N2', chain optimum): ndim 5, k0 4.84, cooling 0.0161, c_rep 0.0118"),
    "h3n2_refrun_listed": dict(fn=functools.partial(_refrun, "H3N2", "listed"), edges=True, vary_init=True,
                               doc="refrun_call('H3N2', listed parameters at ndim 5)"),
    "hiv_refrun_chain": dict(fn=functools.partial(_refrun, "HIV", "chain"), edges=True, vary_init=True,
                             doc="refrun_call('HIV', chain optimum): ndim 2, k0 8.20, cooling 0.0310, c_rep 0.0194"),
    "hiv_refrun_listed": dict(fn=functools.partial(_refrun, "HIV", "listed"), edges=True, vary_init=True,
                              doc="refrun_call('HIV', listed parameters): ndim 2, k0 3.55, cooling 0.0413, c_rep 0.000704"),
    # DENV: the chain optimum IS the listed set (make_reference_results.py), so one problem
    "denv_refrun_chain": dict(fn=functools.partial(_refrun, "DENV", "chain"), edges=True, vary_init=True,
                              doc="refrun_call('DENV', chain optimum = listed): 83 points, ndim 10, k0 7.10, cooling "
                                  "0.0123, c_rep 0.0383"),
}


@functools.lru_cache(maxsize=2)
def build(name):
    """(layout call, truth matrix or None) of a named problem."""
    return PROBLEMS[name]["fn"]()


def oracle_distribution(name):
    with open(os.path.join(GOLD, f"oracle_dist_{name}.json")) as fh:
        return json.load(fh)


def cfg3_oracle_records():
    import glob
    return [json.load(open(f)) for f in sorted(glob.glob(os.path.join(GOLD, "cfg3_oracle_seed*.json")))
            if "_n" not in os.path.basename(f)]
