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

    # ---- one-stage iterations as the symmetric sweep sharded over the ranks (sharded.HipBackend.symm_*) ----
    symm_ready = False

    def symm_prepare(self, rank, world):
        """Which cells of the upper triangle are this rank's: the tiles (64 rows x 32 columns, tile-row-major from
        each tile-row's diagonal square on: csrc/relax_symm.h) cut into `world` equal segments.  Cells of the diagonal
        squares come in both orders with weight 1/2, so the two halves of such a pair may sit in different segments."""
        n = self.n
        npad = -(-n // 64) * 64
        TR, TC = npad // 64, npad // 32
        total = TR * (TR + 1)
        i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        R, J = i // 64, j // 32
        tile = R * TC - R * (R - 1) + (J - 2 * R)
        mine = (J >= 2 * R) & (i != j) & (tile >= total * rank // world) & (tile < total * (rank + 1) // world)
        self.cell_weight = np.where(mine, np.where(J < 2 * R + 2, 0.5, 1.0), 0.0)
        self.symm_ready = True

    def _pair_halves(self, pos, k):
        """H[i, j] = the move of point i caused by the pair (i, j) at `pos` (reference src/optimization.cpp:203-281)."""
        c = self.call
        D, T = np.asarray(c.dissimilarity_matrix, float), np.asarray(c.threshold_matrix)
        g = np.asarray(c.degrees, float)[:, None] + 1.0
        delta = pos[None, :, :] - pos[:, None, :]
        r = np.sqrt((delta ** 2).sum(-1))
        rs = r + 0.01
        with np.errstate(invalid="ignore"):
            spring = np.isfinite(D) & ((T == 0) | ((T == 1) & (r < D)) | ((T == -1) & (r > D)))
            f = np.where(spring, 2.0 * k * (np.where(spring, D, 0.0) - r) / rs / (4.0 * g + k),
                         self.c_rep / (2.0 * rs ** 3) / g)
        np.fill_diagonal(f, 0.0)
        return -delta * f[:, :, None]

    def symm_sweep(self, pos_in, it, k, with_error):
        total = self.check_partial(pos_in) if with_error else None
        if self.stopped:
            return torch.zeros(self.n * self.ndim, dtype=torch.float64), total
        H = self._pair_halves(pos_in[: self.n].numpy(), k)
        w = self.cell_weight
        moves = (w[:, :, None] * H).sum(1) + (w.T[:, :, None] * H).sum(1)   # cell (i, j): i's half and j's half
        self.iters_run = max(self.iters_run, it + 1)
        return torch.from_numpy(moves.reshape(-1).copy()), total

    def symm_apply(self, pos_in, pos_out, moves, it):
        if self.stopped:
            return
        new = pos_in[: self.n] + moves.reshape(self.n, self.ndim)
        pos_out[: self.n] = new
        if not torch.isfinite(new).all() and not self.bad:
            self.bad = it + 1

    def poll(self):
        return self.stopped, self.iters_run

    def first_nonfinite(self):
        return self.bad

    def finish(self):
        return _native.NativeResult(self.best, self.stopped, self.state["best_iter"], self.state["best_mae"],
                                    self.state["best_k"], dict(n_checks=len(self.maes)))

    def synchronize(self):
        pass


def _problem(n=151):
    prob = synthetic.make_problem(n, latent_dim=3, missing=0.6, seed=5)
    init = synthetic.initial_positions(prob.dissimilarity, 3, 5)
    return core.prepare_layout_call(prob.dissimilarity, 3, 40, 9.0, 0.05, 0.02, 1e-4, 3, init, False, 3, True)


def _run(rank, world, port, q, n_points=151, symmetric=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    call = _problem(n_points)
    n = call.initial_positions.shape[0]
    b, e, _ = sharded.row_block(n, world, rank)
    E = call.edge_i.shape[0]
    sl = slice(rank * E // world, (rank + 1) * E // world)
    backend = ModelBackend(call, b, e, sl)
    if symmetric and world > 1:
        backend.symm_prepare(rank, world)
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


def _launch(world, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, q) + extra) for r in range(world)]
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


@pytest.mark.timeout(600)
def test_two_ranks_with_the_symmetric_sweep_equal_one_rank():
    """One-stage iterations as the symmetric sweep sharded over the ranks (each rank a segment of the tile list, the
    n x ndim moves all-reduced, every rank moving all points): the same embedding as the row-owner run of one rank up
    to the order of the f64 sums, the same checks, the same verdict; multi-stage iterations stay row-owner."""
    n = 300
    call = _problem(n)
    its = [it for it in range(call.n_iter)
           if _native.slab_stages_at(it, call.k0 * (1 - call.cooling_rate) ** it, 3) == 1]
    assert 5 < len(its) < call.n_iter - 8          # both kinds of iteration take part
    one = _launch(1, n, False)[0]
    two = _launch(2, n, True)
    for rank_out in two:
        np.testing.assert_allclose(rank_out[1], one[1], rtol=0, atol=1e-10)
        assert np.array_equal(rank_out[1], two[0][1])       # every rank holds the same embedding, bit for bit
        assert rank_out[2:] == pytest.approx(one[2:], rel=1e-10)
    # the segments tile the upper triangle: all weights of a pair add up to one
    w = np.zeros((n, n))
    for r in range(3):
        bk = ModelBackend(call, 0, n, slice(0, 0))
        bk.symm_prepare(r, 3)
        w += bk.cell_weight
    assert np.array_equal(w + w.T, 1.0 - np.eye(n))


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
