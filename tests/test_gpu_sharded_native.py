"""The native row-sharded path (BASELINE config 4: ONE embedding over several GPUs from ONE process,
include/topolow_relax.h: topolow_optimize_layout_exact_sharded / topolow_sessions_run_sharded) on the one
GPU a test box has: several row blocks on the same device exchange their position slices exactly as
blocks on different GPUs would (peer stores from the stage kernel's epilogue, HIP-event barriers, replicated
controller) -- only the wire differs.  Run with -m gpu.

The slab stage moves every row independently of how rows are grouped into blocks, and the controller is
replicated, so a sharded run must reproduce the one-session run of the same seed: positions bit for bit,
the MAE to the rounding of its partial sums (they are grouped by block)."""
import numpy as np
import pytest

from oracle import topolow_oracle as orc
from tests import parity_problems as pp
from tests.conftest import layout_call_args
from topolow_amd import _native

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert np.array_equal(a.positions, b.positions)
    assert (a.converged, a.iterations, a.final_k) == (b.converged, b.iterations, b.final_k)
    assert a.final_mae == pytest.approx(b.final_mae, rel=1e-6)


@pytest.mark.parametrize("thr,thread_per_block", [(0.0, "0"), (0.15, "0"), (0.15, "1")])
def test_row_blocks_on_one_device_equal_the_single_session(thr, thread_per_block, monkeypatch):
    """thread_per_block = "1": every block gets its own host thread and stream and the blocks meet at the HIP-event
    barrier -- the path GPUs of one node take -- instead of sharing their GPU's stream."""
    monkeypatch.setenv("TOPOLOW_SHARD_THREAD_PER_BLOCK", thread_per_block)
    call, _ = pp.random_problem(1203, 5, 0.7, seed=31, thresholds=thr, n_iter=400, k0=8.0, cool=0.03, c_rep=0.01)
    one = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=5, schedule="slab")
    assert one.converged and one.info["schedule"] == "slab"
    for devs in ([0], [0, 0], [0, 0, 0], [0] * 8):
        got = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=5, devices=devs)
        _same(got, one)
        assert got.info["iterations_run"] == one.info["iterations_run"]
    # the edge list as the matrix (no dense n x n host arrays), through the sharded entry itself
    coo = _native.optimize_layout_exact_sharded(
        call.initial_positions, call.degrees, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.n_iter,
        call.k0, call.cooling_rate, call.c_repulsion, call.relative_epsilon, call.convergence_window,
        call.convergence_check_freq, devices=[0, 0, 0, 0], seed=5)
    _same(coo, one)
    assert coo.info["blocks"] == 4 and coo.info["exchanges"] > coo.info["iterations_run"]
    assert coo.info["groups"] == (4 if thread_per_block == "1" else 1)
    sm, cnt = orc.edge_error(coo.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    assert coo.final_mae == pytest.approx(sm / cnt, rel=2e-5)


def test_sharded_small_problem_drops_empty_blocks_and_runs_exhausted():
    """n = 20 over 8 requested blocks: 8-row blocks, 3 of them hold rows (an empty block is dropped, it never
    becomes a session); a run that exhausts its iterations restores the best snapshot like the one-session run."""
    assert _native.shard_rows(20, 8) == [(0, 8), (8, 16), (16, 20)]
    call, _ = pp.random_problem(20, 2, 0.2, seed=3, n_iter=7, check_freq=2)
    one = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=9, schedule="slab")
    got = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=9, devices=[0] * 8)
    _same(got, one)
    assert not got.converged


def test_sharded_guards_interrupt_and_verbose():
    call, _ = pp.random_problem(900, 3, 0.5, seed=8, n_iter=200)
    bad = call.initial_positions.copy()
    bad[17, 1] = np.inf
    args = list(layout_call_args(call)); args[0] = bad
    with pytest.raises(_native.NativeError, match=r"Numerical instability at iteration 10\. Reduce k0 or c_repulsion\.") as ei:
        _native.optimize_layout_exact_arrays(*args, seed=1, devices=[0, 0])
    assert ei.value.code == _native.ERR_NONFINITE
    with pytest.raises(_native.NativeError) as ei:
        _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=1, devices=[0, 0], schedule="gs")
    assert ei.value.code == _native.ERR_UNSUPPORTED
    polls = []

    def stop_at_second_poll():
        polls.append(1)
        return len(polls) >= 2
    import dataclasses
    long_call = dataclasses.replace(call, n_iter=400, relative_epsilon=1e-12, convergence_window=10 ** 6)
    with pytest.raises(_native.NativeError) as ei:
        _native.optimize_layout_exact_arrays(*layout_call_args(long_call), seed=1, devices=[0, 0, 0],
                                             interrupt=stop_at_second_poll)
    assert ei.value.code == _native.ERR_INTERRUPTED and len(polls) == 2     # polled every 50 iterations (:364)
    lines = []
    got = _native.optimize_layout_exact_arrays(*layout_call_args(call), True, seed=1, devices=[0, 0],
                                               **{"print": lines.append})
    text = "".join(lines)
    assert "Points: 900, Pairs per iteration: 404550" in text and "2 row blocks" in text
    assert "Iter " in text and ", k=" in text and got.iterations > 0


def test_config4_size_two_blocks_equal_one_block_and_host_overhead(monkeypatch):
    """BASELINE config 4 (N = 50 000, ndim 3, 90 % missing), generated block by block in HBM: two row blocks on the
    one GPU against one block, bit for bit; and the loop's wall time against block 0's own kernel time -- what
    the host adds per iteration (launches, event barriers, thread hand-offs) must stay below 10 % of the stage
    time."""
    torch = pytest.importorskip("torch")
    from topolow_amd import sharded
    # (the last four of the twelve iterations are two-stage ones: ONE block would take them as symmetric half sweeps, a
    #  schedule the row blocks of a sharded run do not have -- the comparison is about the row-owner engine)
    monkeypatch.setenv("TOPOLOW_SYMMETRIC_TWO_STAGE", "0")
    n, dim, iters = 50000, 3, 12
    results = {}
    for blocks in (1, 2):
        rows = _native.shard_rows(n, blocks)
        backs = [sharded.HipBackend(n, dim, rb, re_, 0) for rb, re_ in rows]
        scale = None
        for b, bk in enumerate(backs):
            _ne, scale = sharded.load_synthetic_block(bk, n, 3, 0.9, 12345, b, len(backs), rows=rows[b])
        torch.cuda.synchronize()
        rng = np.random.Generator(np.random.PCG64(999))
        init = np.zeros((n, dim))
        init[1:] = np.cumsum(rng.uniform(0.0, 2.0 * scale / n, size=(n - 1, dim)), axis=0)
        ss = [bk.session for bk in backs]
        _native.run_sharded(ss, init, 3, 5.0, 0.01, 0.01, 1e-4, 10 ** 9, 3, 7)               # warm-up
        res = _native.run_sharded(ss, init, iters, 5.0, 0.01, 0.01, 1e-4, 10 ** 9, 3, 7)
        prof = _native.run_sharded(ss, init, iters, 5.0, 0.01, 0.01, 1e-4, 10 ** 9, 3, 7, profile=True)
        assert np.array_equal(prof.positions, res.positions)
        results[blocks] = (res, prof)
        for s in ss:
            s.close()
    (r1, p1), (r2, p2) = results[1], results[2]
    assert np.array_equal(r1.positions, r2.positions) and r1.iterations == r2.iterations == iters
    assert r1.final_mae == pytest.approx(r2.final_mae, rel=1e-6)
    # one block: everything the loop does besides block 0's kernels is host / barrier overhead
    busy = p1.info["stage_kernel_seconds"] + p1.info["check_kernel_seconds"]
    print("config 4, one block : loop %.2f ms, block-0 kernels %.2f ms" % (1e3 * r1.info["loop_seconds"], 1e3 * busy))
    print("config 4, two blocks: loop %.2f ms, %d exchanges" % (1e3 * r2.info["loop_seconds"], r2.info["exchanges"]))
    assert p1.info["stage_kernel_seconds"] > 0 and p1.info["check_kernel_seconds"] > 0
    assert r1.info["loop_seconds"] <= 1.10 * busy, (r1.info, p1.info)
    # two blocks on the one GPU share its stream (stream order is their barrier).  The same pairs in twice as many
    # launches: each half-size grid pays its own fill and drain (3 125 workgroups = 2.4 rounds of the chip's 1 280
    # resident ones, against 4.9 rounds for the whole grid), so the GPU time grows (measured 1.2 x); what the HOST
    # adds is the loop's wall time beyond the summed durations of all kernels on that stream:
    busy2 = p2.info["stage_kernel_seconds"] + p2.info["check_kernel_seconds"]
    print("config 4, two blocks: kernels %.2f ms" % (1e3 * busy2))
    assert r2.info["groups"] == 1
    assert r2.info["loop_seconds"] <= 1.10 * busy2, (r2.info, p2.info)
    assert r2.info["loop_seconds"] <= 1.35 * r1.info["loop_seconds"], (r1.info, r2.info)


@pytest.mark.parametrize("thread_per_block", ["0", "1"])
def test_sharded_fuzz_equals_single_session(thread_per_block, monkeypatch):
    """Random problems (ragged sizes, 1..16 coordinates, thresholds, few or many iterations, every check cadence)
    over 2..5 row blocks: the same trajectory and verdicts as the one-session run, bit for bit."""
    from tests.test_gpu_fuzz import _fuzz_problem
    import dataclasses
    monkeypatch.setenv("TOPOLOW_SHARD_THREAD_PER_BLOCK", thread_per_block)
    rng = np.random.default_rng(4200)
    done = 0
    for case in range(10):
        n = int(rng.integers(9, 420))
        dim = int(rng.choice([1, 2, 3, 5, 8, 10, 12]))
        call = _fuzz_problem(rng, n, dim)
        call = dataclasses.replace(call, n_iter=int(rng.integers(2, 90)), relative_epsilon=1e-3)
        seed = int(rng.integers(1, 2 ** 62))
        blocks = int(rng.integers(2, 6))
        try:
            one = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, schedule="slab")
        except _native.NativeError as e:
            with pytest.raises(_native.NativeError) as ei:
                _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, devices=[0] * blocks)
            assert str(ei.value) == str(e)
            continue
        got = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, devices=[0] * blocks)
        assert np.array_equal(got.positions, one.positions), (n, dim, blocks)
        assert (got.converged, got.iterations, got.final_k) == (one.converged, one.iterations, one.final_k)
        assert got.final_mae == pytest.approx(one.final_mae, rel=1e-5, abs=1e-12)
        done += 1
    assert done >= 6


# ----------------------------------------------------------------------------------------
# one-stage iterations as the symmetric sweep sharded over the row-block sessions (csrc/relax_symm.h: segments of the
# tile list, folded partials peer-stored into the owners' inboxes, the owners move their points)
# ----------------------------------------------------------------------------------------
def _sessions(call, n, dim, blocks, env):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ss = []
        for rb, re_ in _native.shard_rows(n, blocks):
            s = _native.Session(n, dim, rb, re_, precision="f32")
            s.load_coo(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.degrees)
            lo, hi = np.minimum(call.edge_i, call.edge_j), np.maximum(call.edge_i, call.edge_j)
            own = np.where((lo + hi) % 2 == 0, lo, hi)          # the parity rule of the sharded MAE (include/topolow_relax.h)
            m = (own >= rb) & (own < re_)
            s.set_edges(call.edge_i[m], call.edge_j[m], call.edge_dist[m], call.edge_thresh[m])
            ss.append(s)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return ss


@pytest.mark.parametrize("dim,thr", [(3, 0.0), (5, 0.0), (5, 0.15), (2, 0.0)])
@pytest.mark.parametrize("thread_per_block", ["0", "1"])
def test_sharded_symmetric_sweep_equals_the_single_block(dim, thr, thread_per_block, monkeypatch):
    """2, 3 and 4 row blocks on one device, one-stage iterations as the sharded symmetric sweep, against ONE block (the
    whole matrix, plain symmetric sweep): the same pairs, the same arithmetic per pair; the partial sums are grouped
    by segment, so positions agree to the fp32 summation band of test_symmetric_sweep_equals_the_row_owner_sweep
    (2e-5 of the coordinate scale per iteration), every check's MAE to 2e-6, same verdicts.  2 973 points: the
    segments cut tile-rows in the middle and the blocks' row ranges (multiples of 8) do not coincide with tile-rows."""
    monkeypatch.setenv("TOPOLOW_SHARD_THREAD_PER_BLOCK", thread_per_block)
    n = 2973
    call, _ = pp.random_problem(n, dim, 0.7, seed=40 + dim, thresholds=0.0, n_iter=10, k0=1.5)
    if thr > 0:
        rng = np.random.default_rng(3)
        code = rng.choice([0, 1, -1], size=call.edge_thresh.shape[0], p=[1 - thr, thr / 2, thr / 2])
        call.edge_thresh[:] = code.astype(call.edge_thresh.dtype)
    env = {"TOPOLOW_SYMMETRIC": "1", "TOPOLOW_SYMMETRIC_MIN_N": "0"}
    scale = float(np.abs(call.initial_positions).max())
    iters = 9                                   # checks at 3 and 6 ride on the sweeps of 4 and 7, the last one is separate
    runs = {}
    for blocks in (1, 2, 3, 4):
        ss = _sessions(call, n, dim, blocks, env)
        r = _native.run_sharded(ss, call.initial_positions, iters, 1.5, 0.01, 0.01, 1e-12, 10 ** 9, 3, 5, 1)
        runs[blocks] = (r, ss[0].check_trace())
        for s in ss:
            s.close()
    one, t_one = runs[1]
    assert t_one.shape[0] == 3
    for blocks in (2, 3, 4):
        got, tr = runs[blocks]
        assert np.abs(got.positions - one.positions).max() <= 2e-5 * scale * iters, blocks
        assert tr.shape == t_one.shape and np.array_equal(tr[:, 0], t_one[:, 0])
        assert np.allclose(tr[:, 1], t_one[:, 1], rtol=2e-6, atol=0), (blocks, tr[:, 1], t_one[:, 1])
        assert got.iterations == one.iterations and got.final_mae == pytest.approx(one.final_mae, rel=2e-6)
        assert got.info["groups"] == (blocks if thread_per_block == "1" else 1)
        assert got.info["symmetric_segments"] == blocks                  # the sharded sweep really ran
    assert one.info["symmetric_segments"] == 0
    # ... and the sharded sweep against the row-owner engine (TOPOLOW_SHARD_SYMMETRIC=0): same band
    monkeypatch.setenv("TOPOLOW_SHARD_SYMMETRIC", "0")
    ss = _sessions(call, n, dim, 2, env)
    ro = _native.run_sharded(ss, call.initial_positions, iters, 1.5, 0.01, 0.01, 1e-12, 10 ** 9, 3, 5, 1)
    for s in ss:
        s.close()
    assert ro.info["symmetric_segments"] == 0
    assert np.abs(ro.positions - runs[2][0].positions).max() <= 2e-5 * scale * iters


def test_sharded_symmetric_sweep_whole_run_through_the_one_shot_entry(monkeypatch):
    """The production entry over four blocks on a problem above the size gate: multi-stage iterations on the row-owner
    kernel, one-stage iterations on the sharded symmetric sweep, fused checks, early stop -- against the one-session
    run of the same seed (same schedule; only fp32 summation order differs): same stop within two checks, same MAE."""
    call, _ = pp.cfg3_generator(7400)
    one = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=3)
    got = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=3, devices=[0, 0, 0, 0])
    monkeypatch.setenv("TOPOLOW_SHARD_SYMMETRIC", "0")
    row = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=3, devices=[0, 0, 0, 0])
    for x in (got, row):
        assert x.converged and one.converged
        assert abs(x.iterations - one.iterations) <= 6
        assert x.final_mae == pytest.approx(one.final_mae, rel=1e-3)
    sm, cnt = orc.edge_error(got.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
    assert got.final_mae == pytest.approx(sm / cnt, rel=2e-5)


@pytest.mark.parametrize("thread_per_block", ["0", "1"])
def test_sharded_symmetric_sweep_fuzz(thread_per_block, monkeypatch):
    """Ragged sizes (tile-rows cut anywhere, fewer tile-rows than blocks, n not a multiple of 8 or 64), 2..6
    coordinates, thresholds, 2..6 blocks, the size gate at zero: the sharded symmetric sweep against the one-session
    symmetric sweep of the same seed through the production entry -- same schedule, same arithmetic per pair, partial
    sums grouped by segment: positions within the fp32 summation band, same verdicts while no check falls on a
    plateau edge (compared to 1e-3)."""
    import dataclasses
    monkeypatch.setenv("TOPOLOW_SHARD_THREAD_PER_BLOCK", thread_per_block)
    monkeypatch.setenv("TOPOLOW_SYMMETRIC_MIN_N", "0")
    rng = np.random.default_rng(77)
    done = 0
    for case in range(8):
        n = int(rng.integers(130, 900))
        dim = int(rng.choice([2, 3, 4, 5, 6]))
        # a soft spring, so that most iterations are ONE stage (the sweep under test); no early stop
        call, _ = pp.random_problem(n, dim, float(rng.choice([0.3, 0.7, 0.9])), seed=1000 + case,
                                    thresholds=float(rng.choice([0.0, 0.2])), n_iter=int(rng.integers(12, 40)),
                                    k0=float(rng.uniform(0.5, 2.5)), cool=0.01, c_rep=0.01, check_freq=int(rng.integers(1, 5)),
                                    window=10 ** 6, eps=1e-12)
        seed = int(rng.integers(1, 2 ** 62))
        blocks = int(rng.integers(2, 7))
        try:
            one = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, schedule="slab", precision="f32")
        except _native.NativeError:
            continue
        got = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=seed, devices=[0] * blocks)
        scale = max(1.0, float(np.abs(one.positions).max()))
        assert np.abs(got.positions - one.positions).max() <= 2e-5 * scale * call.n_iter, (n, dim, blocks)
        assert got.final_mae == pytest.approx(one.final_mae, rel=1e-3, abs=1e-9)
        sm, cnt = orc.edge_error(got.positions, call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        if cnt > 0:
            assert got.final_mae == pytest.approx(sm / cnt, rel=5e-5, abs=1e-9)
        done += 1
    assert done >= 5
