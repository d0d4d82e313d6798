"""Row-sharded driver (topolow_amd/sharded.py) on CPU: two gloo ranks must reproduce the
single-rank run exactly.  The HIP stage is replaced by the CPU slab model (tests only), so this
covers the partitioning, the in-place all-gather of position slices, the all-reduce of the
MAE scalars and the replicated controller -- everything except the kernels themselves."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import topolow_oracle as orc
from tests.models import slab_model
from topolow_amd import _native, core, sharded, synthetic


class ModelBackend:
    """CPU stand-in for sharded.HipBackend: same interface; the slab model moves the points, the oracle
    scores them, and the controller is the library's own (host helper, no GPU), fed one check at a time."""

    def __init__(self, call, row_begin, row_end, edge_slice):
        self.call = call
        self.n, self.ndim = call.initial_positions.shape
        self.rb, self.re = row_begin, row_end
        self.edges = edge_slice
        self.seed = 0

    def new_positions(self, rows_total):
        return torch.zeros((rows_total, self.ndim), dtype=torch.float64)

    def to_device(self, pos_np, rows_total):
        t = self.new_positions(rows_total)
        t[: self.n] = torch.from_numpy(np.asarray(pos_np, dtype=np.float64))
        return t

    def begin(self, pos0, n_iter, k0, cool, c_rep, eps, window, freq, seed):
        self.seed, self.c_rep = seed, c_rep
        self.k0, self.eps, self.window = k0, eps, window
        self.maes, self.iters, self.ks = [], [], []
        self.best = np.array(pos0, dtype=np.float64)
        self.state = dict(best_mae=np.finfo(np.float64).max, best_k=k0, best_iter=0)
        self.stopped, self.iters_run, self.bad = False, 0, 0

    def stage(self, pos_in, pos_out, it, slot, stages, k):
        if self.stopped:      # like the kernels: no-ops once the controller said stop
            return
        c = self.call
        rg = _native.slab_plan(self.n, stages, self.seed, it)[slot].reshape(2, 2)
        rg = [r for r in rg if r[1] > r[0]]
        new = slab_model.stage(pos_in[: self.n].numpy(), c.dissimilarity_matrix, c.threshold_matrix,
                               c.degrees, rg, k, self.c_rep, "f64")
        pos_out[self.rb:self.re] = torch.from_numpy(new[self.rb:self.re])
        if not np.isfinite(new[self.rb:self.re]).all() and not self.bad:
            self.bad = it + 1
        self.iters_run = max(self.iters_run, it + 1)

    can_fuse_checks = True

    def stage_fused(self, pos_in, pos_out, it, k):
        total = self.check_partial(pos_in)       # the MAE of the positions the stage READS
        self.stage(pos_in, pos_out, it, 0, 1, k)
        return total

    def check_partial(self, pos):
        c = self.call
        sl = self.edges
        s_, c_ = orc.edge_error(pos[: self.n].numpy(), c.edge_i[sl], c.edge_j[sl], c.edge_dist[sl],
                                c.edge_thresh[sl])
        return torch.tensor([0.0, 0.0] if self.stopped else [s_, float(c_)], dtype=torch.float64)

    def controller_step(self, total2, pos, iter1, k_after):
        if self.stopped:
            return
        s_, c_ = float(total2[0]), float(total2[1])
        self.maes.append(s_ / c_ if c_ > 0 else 0.0); self.iters.append(iter1); self.ks.append(k_after)
        r = _native.controller_script(self.maes, self.iters, self.ks, self.k0, self.window, self.eps)
        if r["snapshots"][len(self.maes) - 1]:
            self.best = pos[: self.n].numpy().copy()
        self.state = r
        if r["stopped_at"] >= 0:
            self.stopped = True
            self.iters_run = iter1

    def poll(self):
        return self.stopped, self.iters_run

    def first_nonfinite(self):
        return self.bad

    def finish(self):
        return _native.NativeResult(self.best, self.stopped, self.state["best_iter"], self.state["best_mae"],
                                    self.state["best_k"], dict(n_checks=len(self.maes)))

    def synchronize(self):
        pass


def _problem():
    prob = synthetic.make_problem(151, latent_dim=3, missing=0.6, seed=5)
    init = synthetic.initial_positions(prob.dissimilarity, 3, 5)
    return core.prepare_layout_call(prob.dissimilarity, 3, 40, 9.0, 0.05, 0.02, 1e-4, 3, init, False, 3, True)


def _run(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    call = _problem()
    n = call.initial_positions.shape[0]
    b, e, _ = sharded.row_block(n, world, rank)
    E = call.edge_i.shape[0]
    sl = slice(rank * E // world, (rank + 1) * E // world)
    backend = ModelBackend(call, b, e, sl)
    coll = sharded.Collectives(world)
    res = sharded.relax_sharded(backend, coll, rank, world, n, call.initial_positions, call.n_iter,
                                call.k0, call.cooling_rate, call.c_repulsion, call.relative_epsilon,
                                call.convergence_window, call.convergence_check_freq, seed=31)
    q.put((rank, res.positions, res.converged, res.iterations, res.final_mae, res.final_k,
           res.iterations_run))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(outs, key=lambda o: o[0])


@pytest.mark.timeout(600)
def test_two_ranks_equal_one_rank():
    one = _launch(1)[0]
    two = _launch(2)
    for rank_out in two:
        assert np.array_equal(rank_out[1], one[1])          # identical positions on every rank
        assert rank_out[2:] == pytest.approx(one[2:], rel=1e-12)
    assert one[6] >= one[3] > 0


def test_row_blocks_cover_everything():
    for n, world in ((50000, 8), (10, 4), (7, 8), (151, 2)):
        blocks = [sharded.row_block(n, world, r) for r in range(world)]
        cover = np.zeros(n, int)
        for b, e, per in blocks:
            cover[b:e] += 1
            assert e - b <= per
        assert np.all(cover == 1)


def test_torch_encoder_matches_c_encoder():
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.uniform(0, 40, 500), [0.0, 0.1, 1e-30, 3.3e38, np.inf, np.nan]])
    w = sharded.encode_words_torch(torch, torch.from_numpy(vals)).numpy()
    for v, got in zip(vals, w):
        assert int(got) == _native.encode_target(float(v), 0)


# ---- config 5 fan-out: parameter sets dealt over ranks, results gathered (no data-path collective) ----
def _fake_sweep(matrix, sets, max_iter, eps, folds, preserve, rng, precision):
    """CPU stand-in for cv.likelihood_sweep: a deterministic function of the parameter set."""
    return ([dict(Holdout_MAE=ps["k0"] * 2 + ps["N"], NLL=1.0, mean_iter=3.0, pct_converged=100.0) for ps in sets],
            0.01 * len(sets), len(sets) * folds)


def _run_sweep(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from topolow_amd import cv
    sets = [dict(N=2 + (s % 3), k0=float(s), cooling_rate=0.01, c_repulsion=0.01) for s in range(7)]
    res, secs, n_emb = cv.likelihood_sweep_distributed(np.zeros((4, 4)), sets, 10, 1e-4, folds=5, seed=1,
                                                       batch_fn=_fake_sweep)
    q.put((rank, [r["Holdout_MAE"] for r in res], n_emb))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_parameter_sweep_fans_out_over_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_sweep, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in range(2)], key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [float(s) * 2 + 2 + (s % 3) for s in range(7)]
    for _, maes, n_emb in outs:
        assert maes == want and n_emb == 35      # every rank holds every result, in the caller's order
