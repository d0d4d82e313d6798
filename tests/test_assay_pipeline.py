"""Input construction (topolow_amd/antigenic.py) and the CV evaluator's host logic
(topolow_amd/cv.py) against the reference's semantics (R/data_preprocessing.R:488-844,
R/error_metrics.R:55-144, R/adaptive_sampling.R:2570-2598) -- CPU only."""
import csv
import math
import os

import numpy as np
import pytest

from topolow_amd import antigenic, core, cv

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_similarity_titers_to_distances_and_threshold_inversion():
    rows = [dict(v="A", s="X", titer="320", vy=2000, sy=2001),
            dict(v="B", s="X", titer="<10", vy=2001, sy=2001),
            dict(v="B", s="Y", titer="80", vy=2001, sy=2002),
            dict(v="A", s="Y", titer=">1280", vy=2000, sy=2002),
            dict(v="A", s="Y", titer="bad", vy=2000, sy=2002),
            dict(v="C", s="Y", titer="", vy=2003, sy=2002)]
    long_rows, m = antigenic.process_antigenic_data(rows, "v", "s", "titer", is_similarity=True, base=2,
                                                    scale_factor=10, antigen_year_col="vy", serum_year_col="sy")
    d = {(r["v"], r["s"]): r["distance"] for r in long_rows}
    # serum X: log2(32)=5, log2(1)=0 -> max 5 ; serum Y: log2(8)=3, log2(128)=7 -> max 7
    assert d[("A", "X")] == "0" and d[("B", "X")] == ">5"       # "<" titer -> ">" distance
    assert d[("B", "Y")] == "4" and d[("A", "Y")] == "<0"       # ">" titer -> "<" distance
    assert m.names == ["V/A", "S/X", "V/B", "S/Y"]    # by year (stable over the sorted names), V/ S/ prefixes
    v = m.values
    ix = {nm: q for q, nm in enumerate(m.names)}
    assert v[ix["V/A"], ix["S/X"]] == "0" and v[ix["S/X"], ix["V/A"]] == "0"
    assert v[ix["V/B"], ix["S/X"]] == ">5" and v[0, 0] == "0"
    assert v[ix["V/A"], ix["V/B"]] is None and v[ix["S/X"], ix["S/Y"]] is None   # no antigen-antigen cells


def test_repeated_measurements_average_and_keep_sign():
    rows = [dict(v="A", s="X", titer="4"), dict(v="A", s="X", titer="<16"), dict(v="B", s="X", titer="64")]
    long_rows, _ = antigenic.process_antigenic_data(rows, "v", "s", "titer", is_similarity=True, base=2,
                                                    antigen_year_col=None, serum_year_col=None)
    d = {(r["v"], r["s"]): r["distance"] for r in long_rows}
    # distances 6-2=4 and ">"(6-4)=">2" -> mean 3, any "<"/">" -> sign: "<" if any "<" else ">"
    assert d[("A", "X")] == ">3" and d[("B", "X")] == "0"


def test_dissimilarity_mode_log1p():
    rows = [dict(v="A", s="X", ic50="1.718281828459045"), dict(v="B", s="X", ic50=">50")]
    long_rows, _ = antigenic.process_antigenic_data(rows, "v", "s", "ic50", antigen_year_col=None,
                                                    serum_year_col=None)
    d = {(r["v"], r["s"]): r["distance"] for r in long_rows}
    assert float(d[("A", "X")]) == pytest.approx(1.0) and d[("B", "X")].startswith(">")
    assert float(d[("B", "X")][1:]) == pytest.approx(math.log(51.0))


def test_h3n2_and_hiv_fixtures_have_the_surveyed_shape():
    """H3N2 -> 285 points; HIV -> 335 points, 1249 thresholded rows."""
    rows = list(csv.DictReader(open(os.path.join(GOLD, "h3n2_distances.csv"))))
    m = antigenic.titers_list_to_matrix(rows, "virusStrain", "virusYear", "serumStrain", "serumYear",
                                        "distance", sort=True)
    assert len(m.names) == 285 and sum(r["distance"].startswith(">") for r in rows) == 911
    call = core.prepare_layout_call(m, 5, 10, 14.76214, 0.03641074, 0.002943064, 1e-4, 5, None, False, 3,
                                    False, np.random.default_rng(0))
    assert call.edge_i.size == len(rows) and int((call.edge_thresh == 1).sum()) == 911
    rows = list(csv.DictReader(open(os.path.join(GOLD, "hiv_distances.csv"))))
    m = antigenic.titers_list_to_matrix(rows, "Virus", "virusYear", "Antibody", None, "distance", sort=True)
    assert len(m.names) == 335 and sum(r["distance"][0] in "<>" for r in rows) == 1249


def test_error_calculator_comparison_reference_cases():
    """tests/testthat/test-edge-cases.R:84-109 and the documented behaviour."""
    t = np.full((3, 3), 5.0); np.fill_diagonal(t, 0)
    assert cv.error_calculator_comparison(t.copy(), t)["Completeness"] == 1
    truth = np.array([[0, 1, 2], [1, 0, 3], [2, 3, 0]], float)
    assert cv.error_calculator_comparison(np.full((3, 3), np.nan), truth)["Completeness"] == 0
    pred = truth + 0.5
    inp = truth.copy(); inp[0, 2] = inp[2, 0] = np.nan
    e = cv.error_calculator_comparison(pred, truth, inp)
    out = e["OutSampleError"]
    assert np.sum(~np.isnan(out)) == 2 and np.allclose(out[~np.isnan(out)], -0.5)
    assert np.sum(~np.isnan(e["InSampleError"])) == 7 and e["Completeness"] == 1
    # threshold truths drop out (as.numeric -> NA, R/error_metrics.R:90-91)
    tm = np.array([["0", ">2", "3"], [">2", "0", "4"], ["3", "4", "0"]], dtype=object)
    im = tm.copy(); im[0, 1] = im[1, 0] = None; im[0, 2] = im[2, 0] = None
    e = cv.error_calculator_comparison(np.ones((3, 3)), tm, im)
    assert np.sum(~np.isnan(e["OutSampleError"])) == 2      # only the numeric "3" pair counts


def test_fold_construction():
    rng = np.random.default_rng(1)
    d = rng.uniform(1, 5, (12, 12)); d = np.triu(d, 1); d = d + d.T
    d[0, 5] = d[5, 0] = np.nan
    folds = cv.make_folds(d.copy(), 4, rng)
    size = int((~np.isnan(d)).sum()) // 8
    assert len(folds) == 4 and all(f.size == size for f in folds)
    # a draw may hold a cell AND its mirror (both are separate linear indices in the reference's
    # pool, R/adaptive_sampling.R:2588); across folds the unordered pairs are disjoint
    seen = set()
    for f in folds:
        assert len(set(f.tolist())) == f.size
        mine = set()
        for idx in f:
            r, c = int(idx % 12), int(idx // 12)
            assert not np.isnan(d[r, c])
            mine.add((min(r, c), max(r, c)))
        assert not (mine & seen)
        seen |= mine


def _coded_random(n, seed, named):
    """Symmetric matrix with NA, ">"/"<" cells and a few cells whose mirror is NA."""
    rng = np.random.default_rng(seed)
    d = rng.uniform(0.5, 6.0, (n, n)); d = np.triu(d, 1); d = d + d.T
    codes = np.zeros((n, n), dtype=np.int32)
    iu = np.triu_indices(n, 1)
    for q in range(iu[0].size):
        i, j = iu[0][q], iu[1][q]
        u = rng.uniform()
        if u < 0.35:
            d[i, j] = d[j, i] = np.nan
        elif u < 0.45:
            codes[i, j] = codes[j, i] = 1 if rng.uniform() < 0.5 else -1
        elif u < 0.48:
            d[j, i] = np.nan                       # lower-triangle cell missing, upper present
    names = [f"p{q}" for q in range(n)] if named else None
    return core.CodedMatrix(d, codes, names, True)


@pytest.mark.parametrize("source", ["hiv", "random_named", "random_unnamed", "random_named_7", "random_named_150",
                                    "random_unnamed_290"])
def test_fold_builder_equals_the_dense_preparation(source):
    """cv.FoldBuilder (cell list, no n x n work per fold) must hand the kernel exactly what
    core.prepare_layout_call builds from the masked matrix, and its holdout list must be the
    out-of-sample cells of error_calculator_comparison (R/adaptive_sampling.R:2600-2680)."""
    if source == "hiv":
        rows = list(csv.DictReader(open(os.path.join(GOLD, "hiv_distances.csv"))))
        m = core.coded_matrix(antigenic.titers_list_to_matrix(rows, "Virus", "virusYear", "Antibody", None,
                                                              "distance", sort=True))
    else:   # sizes on both sides of NumPy's pairwise-summation block limits (8, 128)
        size = int(source.rsplit("_", 1)[1]) if source[-1].isdigit() else 41
        m = _coded_random(size, 5 + size, "unnamed" not in source)
    n = m.values.shape[0]
    fb = cv.FoldBuilder(m)
    folds = cv.make_folds(m.values, 4, np.random.default_rng(3))
    assert len(folds) == 4
    mine = fb.folds(4, np.random.default_rng(3))       # the cell-list version draws the same cells
    assert len(mine) == 4 and all(np.array_equal(a, b) for a, b in zip(folds, mine))
    for q, h in enumerate(folds):
        r1, r2 = np.random.default_rng(100 + q), np.random.default_rng(100 + q)
        masked = m.masked(h % n, h // n)
        dense = core.prepare_layout_call(masked, 3, 50, 2.0, 0.02, 0.01, 1e-4, 5, None, False, 3, False, r1)
        sparse, hold = fb.fold_numpy(h, 3, 50, 2.0, 0.02, 0.01, 1e-4, 5, 3, False, r2)
        # ... and the library routine (host code of libtopolow_relax.so) equals the NumPy one
        lib_call, lib_hold = fb.fold(h, 3, 50, 2.0, 0.02, 0.01, 1e-4, 5, 3, False, np.random.default_rng(100 + q))
        assert (lib_call.order is None) == (sparse.order is None)
        if sparse.order is not None:
            assert np.array_equal(lib_call.order, sparse.order)
        assert lib_call.names == sparse.names
        for f in ("initial_positions", "degrees", "edge_i", "edge_j", "edge_dist", "edge_thresh"):
            assert np.array_equal(getattr(lib_call, f), getattr(sparse, f)), f
        assert all(np.array_equal(a, b) for a, b in zip(lib_hold, hold))
        assert (dense.order is None) == (sparse.order is None)
        if dense.order is not None:
            assert np.array_equal(dense.order, sparse.order)
        assert dense.names == sparse.names
        for f in ("initial_positions", "degrees", "edge_i", "edge_j", "edge_dist", "edge_thresh"):
            assert np.array_equal(getattr(dense, f), getattr(sparse, f)), f
        assert r1.uniform() == r2.uniform()              # same draws taken from the stream
        # holdout list == OutSampleError cells, for an arbitrary "prediction" in the returned numbering
        est = np.random.default_rng(q).uniform(0, 5, (n, n)); est = (est + est.T) / 2
        err = cv.error_calculator_comparison(est, m, masked, pred_names=dense.names, true_names=m.names)
        oe = err["OutSampleError"]; oe = oe[~np.isnan(oe)]
        mine = np.abs(hold[2] - est[hold[0], hold[1]])
        assert mine.size == oe.size and mine.sum() == pytest.approx(np.abs(oe).sum(), rel=1e-13)
    # preserve_order = TRUE: no reordering in either
    d2 = core.prepare_layout_call(m.masked(folds[0] % n, folds[0] // n), 2, 50, 2.0, 0.02, 0.01, 1e-4, 5, None,
                                  False, 3, True, np.random.default_rng(1))
    s2, _ = fb.fold(folds[0], 2, 50, 2.0, 0.02, 0.01, 1e-4, 5, 3, True, np.random.default_rng(1))
    s3, _ = fb.fold_numpy(folds[0], 2, 50, 2.0, 0.02, 0.01, 1e-4, 5, 3, True, np.random.default_rng(1))
    assert np.array_equal(s2.edge_dist, s3.edge_dist) and np.array_equal(s2.degrees, s3.degrees)
    assert s2.order is None and np.array_equal(d2.edge_i, s2.edge_i) and np.array_equal(d2.degrees, s2.degrees)
    assert np.array_equal(d2.initial_positions, s2.initial_positions)


def test_folds_built_side_by_side_equal_the_sequential_loop():
    """cv.build_fold_calls(parallel=True) -- one sequential pass over the random stream, the list work on a thread
    pool -- hands over exactly the calls, holdouts and stream position of the fold-by-fold loop."""
    m = _coded_random(60, 11, True)
    builder = cv.FoldBuilder(m)
    rng0 = np.random.default_rng(17)
    sets = [dict(N=int(rng0.integers(2, 7)), k0=float(rng0.uniform(0.5, 10)), cooling_rate=0.02, c_repulsion=0.01)
            for _ in range(9)]
    sets[3]["k0"] = -1.0                                   # a set the parameter checks reject: NA rows, no draws
    ra, rb = np.random.default_rng(5), np.random.default_rng(5)
    seq = cv.build_fold_calls(m, builder, sets, 4, ra, 60, 1e-4, False, parallel=False)
    par = cv.build_fold_calls(m, cv.FoldBuilder(m), sets, 4, rb, 60, 1e-4, False, parallel=True)
    assert ra.uniform() == rb.uniform()
    assert seq[1] == par[1] and len(seq[0]) == len(par[0]) == 36
    for a, b, ha, hb in zip(seq[0], par[0], seq[3], par[3]):
        assert (a is None) == (b is None)
        if a is None:
            continue
        for f in ("initial_positions", "degrees", "edge_i", "edge_j", "edge_dist", "edge_thresh"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
        assert a.names == b.names and (a.k0, a.cooling_rate, a.n_iter) == (b.k0, b.cooling_rate, b.n_iter)
        assert all(np.array_equal(x, y) for x, y in zip(ha, hb))
    assert sum(c is None for c in seq[0]) == 4
