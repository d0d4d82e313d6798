"""ONE embedding row-sharded over several PROCESSES (one per GPU in production: topolow_amd/sharded.py over
torch.distributed) with the one-stage iterations as the symmetric sweep sharded over the ranks
(include/topolow_relax.h: topolow_session_symm_segment_*): every rank gets the matrix rows that hold its segment of the
upper triangle's tile list from their owners, sweeps its segment, the ranks all-reduce the n x ndim moves and every
rank moves all points.  Rehearsed on the one GPU of a test box: the ranks share device 0 and talk over gloo (RCCL
refuses two ranks on one device) -- kernels, buffers, row exchange and the driver's schedule are the production ones,
only the wire differs.  Run with -m gpu.

Reference: src/optimization.cpp:198-283 (each pair visited once, both ends moved), :294-357 (check + controller)."""
import os
import socket

import numpy as np
import pytest

from oracle import topolow_oracle as orc
from tests import parity_problems as pp

pytestmark = pytest.mark.gpu

N_POINTS = 2973     # the segments cut tile-rows in the middle; row blocks (multiples of 8) do not coincide with tile-rows
ITERS = 9           # checks at 3 and 6 ride on the sweeps of iterations 4 and 7, the last one is a separate pass


def _problem(dim, thr):
    call, _ = pp.random_problem(N_POINTS, dim, 0.7, seed=40 + dim, thresholds=0.0, n_iter=ITERS, k0=1.5)
    if thr > 0:
        rng = np.random.default_rng(3)
        code = rng.choice([0, 1, -1], size=call.edge_thresh.shape[0], p=[1 - thr, thr / 2, thr / 2])
        call.edge_thresh[:] = code.astype(call.edge_thresh.dtype)
    return call


def _rank_main(rank, world, port, q, dim, thr, symmetric):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      TOPOLOW_SYMMETRIC="1", TOPOLOW_SYMMETRIC_MIN_N="0",
                      TOPOLOW_SEGMENT_PIECE_ROWS="100" if dim == 5 else "0")     # the row slices travel in several messages
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)     # a rank that hangs says where, and the parent sees it die
    import torch
    import torch.distributed as dist
    from topolow_amd import sharded
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    call = _problem(dim, thr)
    n = N_POINTS
    b, e, _per = sharded.row_block(n, world, rank)
    backend = sharded.HipBackend(n, dim, b, e, 0)
    s = backend.session
    s.load_coo(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.degrees)
    lo, hi = np.minimum(call.edge_i, call.edge_j), np.maximum(call.edge_i, call.edge_j)
    own = np.where((lo + hi) % 2 == 0, lo, hi)          # the parity rule of the sharded MAE (include/topolow_relax.h)
    m = (own >= b) & (own < e)
    s.set_edges(call.edge_i[m], call.edge_j[m], call.edge_dist[m], call.edge_thresh[m])
    coll = sharded.Collectives(world)
    took = backend.symm_prepare(coll, rank, world) if symmetric else False
    res = sharded.relax_sharded(backend, coll, rank, world, n, call.initial_positions, ITERS, 1.5, 0.01, 0.01, 1e-12,
                                10 ** 9, 3, seed=5, slab_stages=1)
    q.put((rank, took, res.positions, res.iterations, res.final_mae, s.check_trace()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch(world, *extra):
    import torch.multiprocessing as mp
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    outs, t0 = [], time.time()
    while len(outs) < world:
        try:
            outs.append(q.get(timeout=5))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead and time.time() - t0 < 600, f"rank processes ended with {dead}"
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(outs, key=lambda o: o[0])


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("dim,thr", [(3, 0.0), (5, 0.15)])
def test_ranks_with_the_symmetric_sweep_equal_the_row_owner_run(dim, thr):
    """2 and 3 ranks, every iteration one stage: the sharded symmetric sweep against the row-owner sweep of ONE rank
    (the same update per point, every pair's factor computed once instead of twice; fp32 sums grouped differently):
    positions to the summation band of tests/test_gpu_symmetric.py (2e-5 of the coordinate scale per iteration), every
    check's MAE to 2e-6, every rank the same embedding bit for bit, and the reported MAE the oracle's edge error of
    the returned positions.  The ranks really took the path (symm_prepare -> True); with it switched off two ranks
    reproduce the one-rank run bit for bit (row-owner stage + all-gather)."""
    call = _problem(dim, thr)
    scale = float(np.abs(call.initial_positions).max())
    one = _launch(1, dim, thr, False)[0]
    assert not one[1] and one[5].shape[0] == 3
    for world in (2, 3):
        outs = _launch(world, dim, thr, True)
        for rank_out in outs:
            assert rank_out[1], "symm_prepare refused the path"
            assert np.array_equal(rank_out[2], outs[0][2])
            assert np.abs(rank_out[2] - one[2]).max() <= 2e-5 * scale * ITERS, world
            assert rank_out[3] == one[3] and rank_out[4] == pytest.approx(one[4], rel=2e-6)
            assert rank_out[5].shape == one[5].shape and np.array_equal(rank_out[5][:, 0], one[5][:, 0])
            assert np.allclose(rank_out[5][:, 1], one[5][:, 1], rtol=2e-6, atol=0)
        sm, cnt = orc.edge_error(outs[0][2], call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)
        assert outs[0][4] == pytest.approx(sm / cnt, rel=2e-5)
    plain = _launch(2, dim, thr, False)
    for rank_out in plain:
        assert not rank_out[1] and np.array_equal(rank_out[2], one[2]) and rank_out[3] == one[3]
