"""Copies, AS DATA, the results the reference itself ships for this path into tests/golden/ref_results/
(nothing of the reference is executed; CSV in, CSV/JSON out):

  inst/examples/comparison_results/coordinates/topolow_H3N2_coords.csv   285 x 5 embedding of BASELINE config 2's panel
  inst/examples/comparison_results/coordinates/topolow_HIV_coords.csv    335 x 2 embedding of config 5's panel
  inst/examples/comparison_results/coordinates/topolow_DENV_coords.csv   83 x 10 embedding of the DENV panel
      (written by inst/examples/methods-comparison-h3n2-hiv-denv.Rmd:902-923: euclidean_embedding(..., ndim = N,
       mapping_max_iter = 500, relative_epsilon = 1e-10, convergence_counter = 3) at the parameters
       get_optimal_topolow_params() picks from the shipped chains, :622-658)
  inst/examples/comparison_results/fold_stats.csv        per-fold out-of-sample MAE, 20 folds x {H3N2, HIV} (:1872-1912, :2024-2028)
  inst/examples/comparison_results/error_summary.csv     the pooled numbers BASELINE.md quotes (0.799 / 1.315)
  inst/examples/comparison_results/error_distribution_HIV_H3N2.csv   mean / sd / quartiles of the SIGNED out-of-sample
                                                         errors of the same 20 folds pooled (:2339-2349)
  inst/examples/model_parameters/{H3N2_2003_data_AMC20[1235],HIV_BC_AMC20[2456],denv_data_AMC10[1-5]}_model_parameters.csv
      the adaptive-sampling chains: one row per likelihood_function() call of the reference
      (log parameters -> Holdout_MAE, NLL; 20 folds, mapping_max_iter 500, relative_epsilon 1e-4:
      inst/examples/parameter-fitting-h3n2.Rmd:186-207).  From them:
        chain_optimum.json       the parameter set the notebook's rule selects (rows with finite values and
                                 log_N >= log 2, every column cleaned with clean_data(k = 3.5) = median +- 3.5 MAD,
                                 then argmin Holdout_MAE) -- it has N = 5 for H3N2, N = 2 for HIV and N = 10 for
                                 DENV, the widths of the three coordinate files; for DENV it IS the parameter set
                                 the notebook lists (:326-331: 7.1, 0.01232407, 0.03830152), digit for digit --
                                 which pins this restatement of the selection rule;
        chain_sample_<DS>.csv    48 rows of the cleaned chain, evenly spaced in the order of Holdout_MAE (the
                                 optimum first): reference-held (parameters -> CV score) pairs.

Run: python tests/golden/make_reference_results.py
"""
import csv
import json
import math
import os
import shutil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ref_results")
REF = "/root/reference/inst/examples"
CHAINS = {
    "H3N2": ["H3N2_2003_data_AMC201", "H3N2_2003_data_AMC202", "H3N2_2003_data_AMC203", "H3N2_2003_data_AMC205"],
    "HIV": ["HIV_BC_AMC202", "HIV_BC_AMC204", "HIV_BC_AMC205", "HIV_BC_AMC206"],
    "DENV": ["denv_data_AMC101", "denv_data_AMC102", "denv_data_AMC103", "denv_data_AMC104", "denv_data_AMC105"],
}   # methods-comparison-h3n2-hiv-denv.Rmd:577-599
COLS = ("Holdout_MAE", "NLL", "log_N", "log_k0", "log_cooling_rate", "log_c_repulsion")
N_SAMPLE = 48


def read_chain(files):
    rows = []
    for f in files:
        with open(os.path.join(REF, "model_parameters", f + "_model_parameters.csv")) as fh:
            for r in csv.DictReader(fh):
                try:
                    rows.append([float(str(r[c]).replace('"', "")) for c in COLS])
                except (ValueError, KeyError):
                    pass
    a = np.array(rows)
    a = a[np.isfinite(a).all(1)]
    return a[a[:, 2] >= math.log(2)]


def mad_clean(a, k=3.5):
    """clean_data(k) of the reference per column (R/data_preprocessing.R:864-879, :956-996):
    |x - median| > k * 1.4826 * median|x - median| -> NA; rows with any NA are dropped (na.omit)."""
    keep = np.ones(len(a), dtype=bool)
    for c in range(a.shape[1]):
        x = a[:, c]
        med = np.median(x)
        mad = 1.4826 * np.median(np.abs(x - med))
        keep &= ~(np.abs(x - med) > k * mad)
    return a[keep]


def main():
    os.makedirs(OUT, exist_ok=True)
    for rel in ("comparison_results/coordinates/topolow_H3N2_coords.csv",
                "comparison_results/coordinates/topolow_HIV_coords.csv",
                "comparison_results/coordinates/topolow_DENV_coords.csv",
                "comparison_results/fold_stats.csv", "comparison_results/error_summary.csv",
                "comparison_results/error_distribution_HIV_H3N2.csv"):
        dst = os.path.join(OUT, os.path.basename(rel))
        shutil.copyfile(os.path.join(REF, rel), dst)
        os.chmod(dst, 0o644)
    optimum = {}
    for ds, files in CHAINS.items():
        a = mad_clean(read_chain(files))
        order = np.argsort(a[:, 0], kind="stable")
        a = a[order]
        best = a[0]
        optimum[ds] = dict(N=int(round(math.exp(best[2]))), k0=math.exp(best[3]), cooling_rate=math.exp(best[4]),
                           c_repulsion=math.exp(best[5]), Holdout_MAE=best[0], NLL=best[1], rows_after_cleaning=len(a))
        pick = np.unique(np.round(np.linspace(0, len(a) - 1, N_SAMPLE)).astype(int))
        with open(os.path.join(OUT, f"chain_sample_{ds}.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(COLS)
            for q in pick:
                w.writerow([repr(float(v)) for v in a[q]])
        print(ds, "rows", len(a), "optimum", optimum[ds])
    with open(os.path.join(OUT, "chain_optimum.json"), "w") as fh:
        json.dump(optimum, fh, indent=1)


if __name__ == "__main__":
    main()
