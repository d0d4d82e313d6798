"""Host driver (topolow_amd/core.py) against the reference's R driver semantics
(R/core.R:184-528, :616-719) -- pure host logic, no GPU, no oracle except as a stand-in
for the `.Call` where a full run is needed."""
import os
import re
import warnings

import numpy as np
import pytest

import topolow_amd
from tests.helpers import embed_with_oracle
from topolow_amd import core


def _mat(n=4, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.uniform(size=(n, 2))
    d = np.sqrt(((p[:, None] - p[None]) ** 2).sum(-1))
    return core.RMatrix(d, [f"Point{q + 1}" for q in range(n)])


BASE = dict(ndim=2, mapping_max_iter=10, k0=1.0, cooling_rate=0.01, c_repulsion=0.01,
            relative_epsilon=1e-4, convergence_counter=5)


def test_validation_messages_verbatim():
    """tests/testthat/test-core.R:22-65."""
    with pytest.raises(ValueError, match="dissimilarity_matrix must be a matrix"):
        embed_with_oracle("not a matrix", **BASE)
    with pytest.raises(ValueError, match="dissimilarity_matrix must be square"):
        embed_with_oracle(np.arange(6.0).reshape(2, 3), **BASE)
    for key, val, msg in [("ndim", -1, "ndim must be a positive integer"),
                          ("k0", -1, "k0 must be a positive number"),
                          ("cooling_rate", 1.5, "cooling_rate must be between 0 and 1"),
                          ("c_repulsion", 0, "c_repulsion must be a positive number"),
                          ("relative_epsilon", -1, "relative_epsilon must be a positive number"),
                          ("convergence_counter", 0.5, "convergence_counter must be a positive integer"),
                          ("mapping_max_iter", 0, "mapping_max_iter must be a positive integer"),
                          ("convergence_check_freq", 0, "convergence_check_freq must be a positive integer")]:
        args = dict(BASE); args[key] = val
        with pytest.raises(ValueError, match=re.escape(msg)):
            embed_with_oracle(_mat(), **args)
    with pytest.warns(UserWarning, match="High k0 value"):
        embed_with_oracle(_mat(), **{**BASE, "k0": 35})
    with pytest.raises(ValueError, match="at least 2 rows/columns"):
        embed_with_oracle(np.zeros((1, 1)), **BASE)
    with pytest.raises(TypeError, match='argument "k0" is missing'):
        embed_with_oracle(_mat(), 2, 10)
    allna = np.full((3, 3), np.nan)
    with pytest.warns(UserWarning):
        with pytest.raises(ValueError, match="No valid off-diagonal measurements"):
            embed_with_oracle(allna, **BASE)


def test_initial_positions_validation():
    """tests/testthat/test-core.R:67-88."""
    m = _mat()
    with pytest.raises(ValueError, match="initial_positions must have same number of rows"):
        embed_with_oracle(m, initial_positions=np.zeros((5, 2)), **BASE)
    with pytest.raises(ValueError, match="initial_positions must have ndim columns"):
        embed_with_oracle(m, initial_positions=np.zeros((4, 3)), **BASE)
    with pytest.raises(ValueError, match="initial_positions must be a matrix"):
        embed_with_oracle(m, initial_positions=[1, 2, 3], **BASE)
    init = core.RMatrix(np.random.default_rng(1).uniform(size=(4, 2)), m.names)
    r = embed_with_oracle(m, initial_positions=init, **BASE)
    assert r.positions.shape == (4, 2)


def test_reorder_is_ascending_mean_dissimilarity_and_names_follow():
    """R/core.R:292-311 (the docs say 'descending'; the code sorts ascending)."""
    D = np.array([[0, 9, 8, 7], [9, 0, 1, 2], [8, 1, 0, 3], [7, 2, 3, 0]], float)
    m = core.RMatrix(D, list("abcd"))
    call = core.prepare_layout_call(m, 2, 5, 1.0, 0.1, 0.1, 1e-4, 5, None, False, 3, False,
                                    np.random.default_rng(0))
    means = (np.array([24, 12, 12, 12]) / 3)
    assert list(call.order) == list(np.argsort(means, kind="stable"))
    assert call.names == [list("abcd")[q] for q in call.order]
    keep = core.prepare_layout_call(m, 2, 5, 1.0, 0.1, 0.1, 1e-4, 5, None, False, 3, True,
                                    np.random.default_rng(0))
    assert keep.order is None and keep.names == list("abcd")
    # initial positions follow the reordered matrix only through row names (R/core.R:325-333)
    init = core.RMatrix(np.arange(8.0).reshape(4, 2), list("abcd"))
    c2 = core.prepare_layout_call(m, 2, 5, 1.0, 0.1, 0.1, 1e-4, 5, init, False, 3, False)
    assert np.array_equal(c2.initial_positions, init.values[c2.order])
    c3 = core.prepare_layout_call(m, 2, 5, 1.0, 0.1, 0.1, 1e-4, 5, init.values, False, 3, False)
    assert np.array_equal(c3.initial_positions, init.values)  # unnamed: NOT permuted


def test_parse_degrees_coo_and_dense_fill_character_matrix():
    """G5: R/core.R:340-436 on a character matrix with thresholds, NA and an asymmetric NA."""
    m = np.array([["0", ">2", None, "1.5"],
                  [">2", "0", "<5", None],
                  ["7", "<5", "0", "3"],     # [2,0]="7" but [0,2]=NA: asymmetric
                  ["1.5", "2.5", "3", "0"]], dtype=object)
    call = core.prepare_layout_call(m, 2, 5, 1.0, 0.1, 0.1, 1e-4, 5, np.zeros((4, 2)), False, 3, True)
    assert list(call.degrees) == [3, 3, 4, 4]          # diagonal counts; rows, not pairs
    # COO: upper triangle, column-major scan, 0-based
    assert list(zip(call.edge_i, call.edge_j)) == [(0, 1), (1, 2), (0, 3), (2, 3)]
    assert list(call.edge_dist) == [2.0, 5.0, 1.5, 3.0]
    assert list(call.edge_thresh) == [1, -1, 0, 0]
    Dd, Td = call.dissimilarity_matrix, call.threshold_matrix
    assert np.array_equal(Dd, Dd.T) and np.array_equal(Td, Td.T)   # lower <- t(upper)
    assert np.isinf(Dd[0, 2]) and np.isinf(Dd[2, 0])               # upper NA wins over lower "7"
    assert np.isinf(Dd[1, 3]) and np.isinf(Dd[3, 1])               # upper NA wins over lower "2.5"
    assert Dd[0, 1] == 2.0 and Td[0, 1] == 1 and Td[1, 2] == -1 and Dd[0, 0] == 0.0


def test_numeric_matrix_payload_and_init_walk():
    D = np.array([[0, 2, np.nan], [2, 0, 4], [np.nan, 4, 0]], float)
    call = core.prepare_layout_call(D, 3, 5, 1.0, 0.1, 0.1, 1e-4, 5, None, False, 3, True,
                                    np.random.default_rng(3))
    assert call.dissimilarity_matrix[0, 2] == np.inf and call.threshold_matrix.sum() == 0
    p = call.initial_positions
    assert p.shape == (3, 3) and np.all(p[0] == 0)
    step = 4.0 / 3
    inc = np.diff(p, axis=0)
    assert np.all(inc >= 0) and np.all(inc <= 2 * step)            # cumsum of U(0, 2*max/n)


def test_post_mae_counts_diagonal_both_triangles_and_drops_threshold_strings():
    """G6: R/core.R:479-481."""
    m = np.array([["0", ">2", "3"], [">2", "0", None], ["3", None, "0"]], dtype=object)
    est = np.array([[0, 5.0, 2.0], [5.0, 0, 9.0], [2.0, 9.0, 0]])
    # valid cells: 3 diagonal zeros + the two "3" cells -> mean(|0|,|0|,|0|,1,1) = 0.4
    assert core.post_mae(m, est) == pytest.approx(0.4)
    num = np.array([[0, 2.0, np.nan], [2.0, 0, 4.0], [np.nan, 4.0, 0]])
    assert core.post_mae(num, est) == pytest.approx((3 + 3 + 5 + 5) / 7)


def test_object_structure_print_summary():
    """tests/testthat/test-core.R:129-139, test-S3-methods.R:21-48."""
    m = core.RMatrix(np.array([[0, 1, 2], [1, 0, 3], [2, 3, 0]], float), ["Point1", "Point2", "Point3"])
    r = embed_with_oracle(m, 2, 10, 1.0, 0.01, 0.01)
    for key in ("positions", "est_distances", "mae", "iter", "parameters", "convergence"):
        assert key in r
    assert r.r_class == "topolow" and isinstance(r.mae, float)
    assert isinstance(r.convergence["achieved"], bool)
    assert r.parameters["method"] == "cpp_exact_full_pairwise"
    text = str(r)
    for frag in ("topolow optimization result:", "Dimensions: 2", "Iterations:", "MAE:",
                 "Convergence achieved:", "Final convergence error:"):
        assert frag in text
    s = r.summary()
    for frag in ("Parameters:", "k0: 1.0000", "cooling_rate: 0.0100", "c_repulsion: 0.0100"):
        assert frag in s


def test_csv_and_output_dir(tmp_path):
    """R/core.R:486-500; tests/testthat/test-edge-cases.R:220-241."""
    m = np.array([[0, 1, 2], [1, 0, 3], [2, 3, 0]], float)
    out = tmp_path / "non_existent_subdir"
    r = embed_with_oracle(m, 2, 10, 1.0, 0.01, 0.01, write_positions_to_csv=True, output_dir=str(out))
    f = out / "Positions_dim_2_k0_1.0000_cooling_0.0100_c_repulsion_0.0100.csv"
    assert f.exists()
    lines = f.read_text().strip().split("\n")
    assert lines[0] == '"","V1","V2"' and len(lines) == 4
    assert float(lines[1].split(",")[1]) == pytest.approx(r.positions[0, 0], rel=1e-14)
    with pytest.raises(ValueError, match="An 'output_dir' must be provided"):
        embed_with_oracle(m, 2, 10, 1.0, 0.01, 0.01, write_positions_to_csv=True)


def test_create_topolow_map_deprecation_and_signature():
    """R/core.R:616-664; tests/testthat/test-deprecated.R:4-25."""
    import inspect
    sig = inspect.signature(topolow_amd.create_topolow_map)
    assert list(sig.parameters) == ["distance_matrix", "ndim", "mapping_max_iter", "k0", "cooling_rate",
                                    "c_repulsion", "relative_epsilon", "convergence_counter",
                                    "initial_positions", "write_positions_to_csv", "output_dir", "verbose"]
    assert sig.parameters["convergence_counter"].default == 3
    sig2 = inspect.signature(topolow_amd.euclidean_embedding)
    assert list(sig2.parameters) == ["dissimilarity_matrix", "ndim", "mapping_max_iter", "k0",
                                     "cooling_rate", "c_repulsion", "relative_epsilon",
                                     "convergence_counter", "initial_positions",
                                     "write_positions_to_csv", "output_dir", "verbose",
                                     "convergence_check_freq", "preserve_order"]
    assert sig2.parameters["convergence_counter"].default == 5
    assert sig2.parameters["convergence_check_freq"].default == 3
    assert sig2.parameters["mapping_max_iter"].default == 1000
    assert sig2.parameters["relative_epsilon"].default == 1e-4


def test_r_compatible_seed_reproduces_r_runif_and_initial_positions():
    """set.seed(s); runif() of R's default generator.  The expected
    numbers are the well-known heads of R's streams (e.g. `set.seed(123); runif(3)`)."""
    from topolow_amd.r_rng import RUnif
    assert np.allclose(RUnif(123).runif(5), [0.2875775, 0.7883051, 0.4089769, 0.8830174, 0.9404673], atol=5e-8)
    assert np.allclose(RUnif(42).runif(3), [0.9148060, 0.9370754, 0.2861395], atol=5e-8)
    assert np.allclose(RUnif(1).runif(3), [0.2655087, 0.3721239, 0.5728534], atol=5e-8)
    # initial positions of R/core.R:407-415 after set.seed(123): cumsum of runif((n-1)*ndim, 0, 2*max/n),
    # filled column by column
    D = np.array([[0, 2, 3], [2, 0, 4], [3, 4, 0]], float)
    call = core.prepare_layout_call(D, 2, 5, 1.0, 0.1, 0.1, 1e-4, 5, None, False, 3, True, RUnif(123))
    u = np.array([0.2875775201246142, 0.7883051354438066, 0.4089769218116999, 0.8830174040049314])
    step = 2 * 4.0 / 3
    want = np.array([[0, 0], [u[0] * step, u[2] * step], [(u[0] + u[1]) * step, (u[2] + u[3]) * step]])
    assert np.allclose(call.initial_positions, want, atol=1e-7)
    import topolow_amd
    topolow_amd.set_seed(123)
    from topolow_amd import _native
    assert abs(_native.host_rng().unif_rand() - 0.2875775201246142) < 1e-12
    topolow_amd.set_seed(None)
