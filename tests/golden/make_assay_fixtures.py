"""Builds the assay fixtures of BASELINE configs 2 and 5 from the reference's DATA files
(read as data only; nothing of the reference is executed):
  /root/reference/data-raw/Smith2004-data.csv        -> tests/golden/h3n2_distances.csv
  /root/reference/data-raw/hiv_filtered_long_data.csv -> tests/golden/hiv_distances.csv
  /root/reference/data-raw/DENV_titers.csv            -> tests/golden/denv_distances.csv  (the third panel the
                                                         reference ships an embedding of)
H3N2 goes through topolow_amd.antigenic.process_antigenic_data with the arguments the reference's
own notebook uses (inst/examples/methods-comparison-h3n2-hiv-denv.Rmd:497-506: is_similarity,
base 2, scale_factor 10); the HIV table already carries the reference's `distance` column.
Run: python tests/golden/make_assay_fixtures.py
"""
import csv
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from topolow_amd import antigenic  # noqa: E402

REF = "/root/reference/data-raw"


def main():
    rows = list(csv.DictReader(open(os.path.join(REF, "Smith2004-data.csv"), encoding="utf-8-sig")))
    long_rows, m = antigenic.process_antigenic_data(rows, "virusStrain", "serumStrain", "titer",
                                                    is_similarity=True, base=2, scale_factor=10)
    with open(os.path.join(HERE, "h3n2_distances.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["virusStrain", "serumStrain", "virusYear", "serumYear", "distance"])
        for r in long_rows:
            w.writerow([r["virusStrain"], r["serumStrain"], r["virusYear"], r["serumYear"], r["distance"]])
    n = len(m.names)
    gt = sum(1 for r in long_rows if r["distance"].startswith(">"))
    print("h3n2: points", n, "pairs", len(long_rows), "'>' distances", gt)

    # DENV (Katzelnick et al. 2015): repeated titrations of a (virus, serum) pair are averaged by
    # process_antigenic_data; same arguments as H3N2 (methods-comparison-h3n2-hiv-denv.Rmd:546-556)
    rows = list(csv.DictReader(open(os.path.join(REF, "DENV_titers.csv"))))
    long_rows, m = antigenic.process_antigenic_data(rows, "virus_strain", "serum_strain", "titer",
                                                    is_similarity=True, base=2, scale_factor=10)
    with open(os.path.join(HERE, "denv_distances.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["virus_strain", "serum_strain", "virusYear", "serumYear", "distance"])
        for r in long_rows:
            w.writerow([r["virus_strain"], r["serum_strain"], r["virusYear"], r["serumYear"], r["distance"]])
    print("denv: points", len(m.names), "pairs", len(long_rows), "of", len(rows), "titrations")

    rows = list(csv.DictReader(open(os.path.join(REF, "hiv_filtered_long_data.csv"))))
    with open(os.path.join(HERE, "hiv_distances.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Virus", "Antibody", "virusYear", "distance"])
        for r in rows:
            w.writerow([r["Virus"], r["Antibody"], r["virusYear"], r["distance"]])
    thr = sum(1 for r in rows if r["distance"][0] in "<>")
    print("hiv: rows", len(rows), "thresholded", thr,
          "points", len({"V/" + r["Virus"] for r in rows} | {"S/" + r["Antibody"] for r in rows}))


if __name__ == "__main__":
    main()
