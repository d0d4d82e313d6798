"""Multi-GPU drivers of the relaxation path: one process per GPU (torch.distributed, RCCL).

Two modes:

* row-sharded (BASELINE config 4): ONE embedding whose N x N target matrix does not have to
  fit one GPU.  Rank r owns the row block [r*ceil(N/P), ...) of the encoded matrix and moves
  only its own points; after every slab stage the ranks all-gather their position slices
  (the only data-path collective), and at every convergence check they all-reduce the
  two scalars (error sum, count) of the edge MAE.  Every rank runs the same deterministic
  controller on identical inputs, so no further coordination is needed.
* replicas (BASELINE config 5 style): independent embeddings, one per rank, no data-path
  collective -- the reference's own parallel mode (one embedding per forked process,
  R/adaptive_sampling.R:666 of the reference).

The per-stage compute goes through a small backend interface so that the control flow can be
tested on CPU (gloo, world_size 2) with the slab model standing in for the HIP kernels; the
product backend is `HipBackend` (libtopolow_relax.so, no fallback).
"""
from __future__ import annotations

import json
import math
import os
import time
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import _native


# --------------------------------------------------------------------------------------
# helpers shared by both modes
# --------------------------------------------------------------------------------------
def dist_info():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, world, local


def row_block(n: int, world: int, rank: int):
    """Rows [b, e) of rank `rank`: contiguous blocks of `per` rows, per a multiple of 8 (whole stage-kernel
    workgroups; the same rule as the library's topolow_shard_rows).  Trailing ranks of a small problem get an
    empty block (b == e == n); relax_sharded refuses to run with one."""
    per = -(-n // world)
    per = (per + 7) & ~7
    b = min(n, rank * per)
    e = min(n, b + per)
    return b, e, per


@dataclass
class ShardedResult:
    positions: np.ndarray
    converged: bool
    iterations: int
    final_mae: float
    final_k: float
    iterations_run: int
    n_checks: int
    stage_seconds: float = 0.0
    gather_seconds: float = 0.0
    check_seconds: float = 0.0


class Collectives:
    """torch.distributed wrappers that degrade to no-ops at world_size 1 and stage through the
    host when the backend cannot take device tensors (gloo with GPU tensors)."""

    def __init__(self, world: int):
        self.world = world
        if world > 1:
            import torch.distributed as dist
            self.dist = dist
            self.backend = dist.get_backend()

    def all_gather_rows(self, full, per: int, rank: int):
        """In-place all-gather: rank r contributes rows [r*per, (r+1)*per) of `full`."""
        if self.world == 1:
            return
        # the buffer may hold extra padding rows (roundup4(n)); only the world*per leading rows take
        # part in the collective
        out = full[: per * self.world]
        mine = out[rank * per:(rank + 1) * per]
        if full.is_cuda and self.backend == "gloo":
            host = out.cpu()
            self.dist.all_gather_into_tensor(host, host[rank * per:(rank + 1) * per].clone())
            out.copy_(host)
        else:
            self.dist.all_gather_into_tensor(out, mine.clone() if self.backend == "gloo" else mine)

    def all_reduce_tensor(self, t):
        """In-place sum of a small tensor over the ranks; enqueued on the tensor's stream (RCCL), no host
        synchronisation."""
        if self.world == 1:
            return
        if t.is_cuda and self.backend == "gloo":
            host = t.cpu()
            self.dist.all_reduce(host)
            t.copy_(host)
        else:
            self.dist.all_reduce(t)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def move_rows(self, src: int, dst: int, rank: int, send=None, recv=None):
        """Rows of a device (or host) tensor from rank `src` to rank `dst`: `send` on src, `recv` on dst (both
        contiguous, same shape), nothing on the others.  Point to point over the backend (RCCL send/recv over xGMI);
        gloo takes host tensors only, so device tensors are staged."""
        if rank == src:
            self.dist.send(send.cpu() if (send.is_cuda and self.backend == "gloo") else send, dst)
        elif rank == dst:
            if recv.is_cuda and self.backend == "gloo":
                host = recv.cpu()
                self.dist.recv(host, src)
                recv.copy_(host)
            else:
                self.dist.recv(recv, src)

    def _host_reduce(self, v: float, op):
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max_float(self, v: float) -> float:
        return v if self.world == 1 else self._host_reduce(v, self.dist.ReduceOp.MAX)

    def min_float(self, v: float) -> float:
        return v if self.world == 1 else self._host_reduce(v, self.dist.ReduceOp.MIN)


# --------------------------------------------------------------------------------------
# row-sharded relaxation
# --------------------------------------------------------------------------------------
class HipBackend:
    """Row block of one embedding on this rank's GPU (libtopolow_relax.so session).  Stage kernels,
    the error pass, the controller and its snapshot all run on the device, on the stream the
    collectives use: the loop never waits for the GPU except where `poll` is called."""

    def __init__(self, n, ndim, row_begin, row_end, device):
        import torch
        self.torch = torch
        self.n, self.ndim = n, ndim
        self.session = _native.Session(n, ndim, row_begin, row_end, precision="f32", device=device)
        self.device = torch.device("cuda", device)
        # launch on torch's current stream so kernels and collectives are ordered without
        # host synchronisation
        self.session.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def new_positions(self, rows_total):
        # at least roundup4(n) rows; rows >= n are the padding columns' phantom points
        rows = max(rows_total, self.session.position_rows)
        t = self.torch.zeros((rows, self.ndim), dtype=self.torch.float32, device=self.device)
        t[self.n:, 0] = _native.FAR_F32
        return t

    def to_device(self, pos_np, rows_total):
        t = self.new_positions(rows_total)
        t[: self.n] = self.torch.from_numpy(np.ascontiguousarray(pos_np, dtype=np.float32)).to(self.device)
        return t

    def begin(self, pos0, n_iter, k0, cool, c_rep, eps, window, freq, seed):
        # the session's own buffers take the start positions too: its best snapshot starts from them
        self.session.set_stream(self.torch.cuda.current_stream(self.device).cuda_stream)
        self.session.set_positions(pos0)
        self.session.begin(n_iter, k0, cool, c_rep, eps, window, freq, seed, 0)
        self._t2 = self.torch.zeros(2, dtype=self.torch.float64, device=self.device)

    def stage(self, pos_in, pos_out, it, slot, stages, k):
        self.session.stage(pos_in.data_ptr(), pos_out.data_ptr(), it, slot, stages, k)

    def check_partial(self, pos):
        self.session.check_partial(pos.data_ptr(), self._t2.data_ptr())
        return self._t2

    @property
    def can_fuse_checks(self):
        return self.session.can_fuse_checks

    def stage_fused(self, pos_in, pos_out, it, k):
        """The single stage of iteration `it`, which also reduces the MAE of pos_in (the previous check's positions)."""
        self.session.stage_fused(pos_in.data_ptr(), pos_out.data_ptr(), it, k, self._t2.data_ptr())
        return self._t2

    def controller_step(self, total2, pos, iter1, k_after):
        self.session.controller_step(total2.data_ptr(), pos.data_ptr(), iter1, k_after)

    def poll(self):
        """(stopped, iterations_run) after waiting for everything enqueued."""
        iters_run, stopped, _mae = self.session.sync()
        return bool(stopped), int(iters_run)

    def first_nonfinite(self):
        return self.session.first_nonfinite()

    def finish(self):
        return self.session.finish()

    def edge_error(self, pos):
        return self.session.edge_error(pos.data_ptr())

    # ---- one-stage iterations as the symmetric sweep sharded over the ranks (include/topolow_relax.h:
    # topolow_session_symm_segment_*) ----
    symm_ready = False

    def symm_prepare(self, coll: Collectives, rank: int, world: int) -> bool:
        """Brings the rows that hold this rank's segment of the upper triangle's tile list together (point to point
        from their owners), completes the degree terms over the ranks and builds the segment.  Collective: every
        rank calls it, and either all ranks take the path or none (a rank whose device cannot hold the extra buffers
        vetoes it).  TOPOLOW_SHARD_SYMMETRIC=0 keeps the row-owner sweep."""
        torch, s, n = self.torch, self.session, self.n
        self.symm_ready = False
        if world < 2 or os.environ.get("TOPOLOW_SHARD_SYMMETRIC", "1") == "0":
            return False
        vote = torch.tensor([1.0 if s.symm_segment_eligible(world) else 0.0], dtype=torch.float32, device=self.device)
        coll.all_reduce_tensor(vote)
        if vote.item() < world:
            return False
        needs = [_native.symm_segment_rows(n, r, world) for r in range(world)]
        blocks = [row_block(n, world, r)[:2] for r in range(world)]
        ld = s.encoded_ld
        g = _as_tensor(torch, s.degree_terms_ptr, (n,), torch.float32, self.device)
        b, e = blocks[rank]
        full = torch.zeros(n, dtype=torch.float32, device=self.device)
        full[b:e] = g[b:e]
        coll.all_reduce_tensor(full)
        g.copy_(full)
        flag = torch.tensor([1.0 if s.has_thresholds else 0.0], dtype=torch.float32, device=self.device)
        coll.all_reduce_tensor(flag)
        any_thr = bool(flag.item() > 0)
        enc = _as_tensor(torch, s.encoded_ptr, (e - b, ld), torch.int32, self.device)
        first, end = needs[rank]
        ok = 1.0
        try:
            stage = torch.empty((end - first, ld), dtype=torch.int32, device=self.device)
        except RuntimeError:
            ok, stage = 0.0, None
        vote = torch.tensor([ok], dtype=torch.float32, device=self.device)
        coll.all_reduce_tensor(vote)
        if vote.item() < world:
            return False
        # rows per message: at most 1 GB (element counts stay far inside 32 bits); tests make the pieces small
        piece = int(os.environ.get("TOPOLOW_SEGMENT_PIECE_ROWS", 0)) or max(1, (1 << 28) // ld)
        for dst in range(world):            # the same order on every rank: a transfer at a time, no cycle to wait in
            nf, ne = needs[dst]
            for src in range(world):
                lo, hi = max(nf, blocks[src][0]), min(ne, blocks[src][1])
                if hi <= lo:
                    continue
                if src == dst:
                    if rank == dst:
                        stage[lo - nf:hi - nf] = enc[lo - b:hi - b]
                elif rank in (src, dst):
                    for r0 in range(lo, hi, piece):
                        r1 = min(hi, r0 + piece)
                        coll.move_rows(src, dst, rank, send=enc[r0 - b:r1 - b] if rank == src else None,
                                       recv=stage[r0 - nf:r1 - nf] if rank == dst else None)
        ok = 1.0
        try:
            s.symm_segment_build(rank, world, stage.data_ptr(), first, end - first, any_thr)
        except _native.NativeError:
            ok = 0.0
        del stage
        vote = torch.tensor([ok], dtype=torch.float32, device=self.device)
        coll.all_reduce_tensor(vote)
        if vote.item() < world:
            return False
        self._moves = _as_tensor(torch, s.symm_moves_ptr, (n * self.ndim,), torch.float32, self.device)
        self.symm_ready = True
        return True

    def symm_sweep(self, pos_in, it, k, with_error: bool):
        """This rank's segment of iteration `it`: returns (moves, total2) -- the session's moves buffer (the caller
        sums it over the ranks in place) and, with_error, the segment's share of the MAE of pos_in."""
        self.session.symm_segment_sweep(pos_in.data_ptr(), it, k, self._t2.data_ptr() if with_error else 0)
        return self._moves, (self._t2 if with_error else None)

    def symm_apply(self, pos_in, pos_out, moves, it):
        """pos_out = pos_in + moves for ALL points; `moves` is the session's own buffer, summed over the ranks."""
        assert moves.data_ptr() == self._moves.data_ptr()
        self.session.symm_segment_apply(pos_in.data_ptr(), pos_out.data_ptr(), it)

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)


class ShardedRelaxation:
    """The slab relaxation of one embedding over `world` processes (one per GPU).  Mirrors the
    single-GPU session loop (topolow_relax.hip: topolow_session_enqueue) and the reference's iteration
    structure (src/optimization.cpp:193-374 of the reference).

    Per stage: one kernel launch + one all-gather of the position slices; per check: the block's error
    pass, an all-reduce of two doubles and the controller kernel -- all enqueued, nothing read back.  The
    controller is replicated (identical inputs, identical decisions), so the ranks learn of a stop by
    reading their own state: every `sync_every`-th check each rank waits for its stream and looks; they all
    see the same state at the same point of the schedule, so they leave the loop together.  Work enqueued
    after the stop is a no-op on the device (kernels test the stop flag).

    `advance(m)` enqueues the next m iterations (bench.py times the job in slices of K iterations this way);
    `finish()` flushes a pending check, applies the reference's non-finite guard and returns the result."""

    def __init__(self, backend, coll: Collectives, rank: int, world: int, n: int, initial_positions,
                 n_iter: int, k0: float, cooling_rate: float, c_repulsion: float,
                 relative_epsilon: float = 1e-4, convergence_window: int = 5,
                 convergence_check_freq: int = 3, seed: int = 0, slab_stages: int = 0,
                 timers: bool = False, sync_every: int = 8):
        b, e, per = row_block(n, world, rank)
        if e <= b:
            raise ValueError(f"row-sharded run: rank {rank} of {world} would own no rows of {n} points "
                             f"(blocks are whole 8-row workgroups); use at most {-(-n // 8)} ranks")
        self.backend, self.coll, self.rank, self.world, self.n, self.per = backend, coll, rank, world, n, per
        rows_total = per * world
        self.pos = [backend.to_device(initial_positions, rows_total), backend.new_positions(rows_total)]
        self.cur = 0
        self.freq = convergence_check_freq if convergence_check_freq >= 1 else 10
        backend.begin(initial_positions, n_iter, k0, cooling_rate, c_repulsion, relative_epsilon,
                      convergence_window, self.freq, seed)
        self.n_iter, self.k, self.cooling_rate = n_iter, k0, cooling_rate
        self.seed, self.slab_stages, self.timers, self.sync_every = seed, slab_stages, timers, sync_every
        self.t_stage = self.t_gather = self.t_check = 0.0
        self.checks = 0
        self.it = 0
        self.stopped = False
        self.ndim = int(np.asarray(initial_positions).shape[1])
        # a check rides on the next sweep only when EVERY rank's block can carry it (an odd row count cannot): the ranks
        # must agree, the check's all-reduce is part of the schedule they share
        self.fusable = coll.min_float(1.0 if getattr(backend, "can_fuse_checks", False) else 0.0) > 0 and not timers
        # one-stage iterations as the symmetric sweep sharded over the ranks (every pair once; one all-reduce of the
        # n x ndim moves instead of the all-gather): when the backend built its segment (HipBackend.symm_prepare)
        self.symmetric = bool(getattr(backend, "symm_ready", False)) and world > 1
        if self.symmetric:
            self.fusable = not timers     # the sweep's ERR instance has no even-rows rule
        self.pending = None   # (iter1, k_after, buffer index): a check that rides on the next iteration's single stage

    def _n_slots_of(self, it_, k_):
        st = self.slab_stages if self.slab_stages > 0 else _native.slab_stages_at(it_, k_, self.ndim)
        return st, len(_native.slab_plan(self.n, st, self.seed, it_))

    def _separate_check(self, buf, iter1, k_after):
        total = self.backend.check_partial(self.pos[buf])
        self.coll.all_reduce_tensor(total)
        self.backend.controller_step(total, self.pos[buf], iter1, k_after)

    def advance(self, max_iters: int) -> int:
        """Enqueues up to max_iters further iterations; returns how many (0: the run is over)."""
        backend, coll, pos, per, rank, timers = self.backend, self.coll, self.pos, self.per, self.rank, self.timers
        done = 0
        while done < max_iters and self.it < self.n_iter and not self.stopped:
            it, k = self.it, self.k
            stages, n_slots = self._n_slots_of(it, k)
            fuse_now = self.pending is not None and n_slots == 1
            if self.pending is not None and not fuse_now:
                self._separate_check(self.pending[2], self.pending[0], self.pending[1])
                self.pending = None
            if self.symmetric and n_slots == 1:
                cur = self.cur
                if timers:
                    backend.synchronize()
                    t0 = time.perf_counter()
                moves, total = backend.symm_sweep(pos[cur], it, k, fuse_now)
                if timers:
                    backend.synchronize()
                    t1 = time.perf_counter()
                coll.all_reduce_tensor(moves)
                if fuse_now:
                    coll.all_reduce_tensor(total)
                    backend.controller_step(total, pos[self.pending[2]], self.pending[0], self.pending[1])
                    self.pending = None
                if timers:
                    backend.synchronize()
                    t2 = time.perf_counter()
                backend.symm_apply(pos[cur], pos[cur ^ 1], moves, it)
                if timers:
                    backend.synchronize()
                    self.t_stage += (t1 - t0) + (time.perf_counter() - t2)
                    self.t_gather += t2 - t1
                self.cur ^= 1
                n_slots = 0
            for slot in range(n_slots):
                cur = self.cur
                if timers:   # breakdown pass: host-synchronised, so slower than the timed pass
                    backend.synchronize()
                    t0 = time.perf_counter()
                    backend.stage(pos[cur], pos[cur ^ 1], it, slot, stages, k)
                    backend.synchronize()
                    t1 = time.perf_counter()
                    coll.all_gather_rows(pos[cur ^ 1], per, rank)
                    backend.synchronize()
                    self.t_stage += t1 - t0
                    self.t_gather += time.perf_counter() - t1
                elif fuse_now:
                    # one sweep: the stage, the MAE of the positions it reads (= the pending check), the gather of the
                    # new slices, the all-reduce of the two MAE scalars, the controller -- all enqueued
                    total = backend.stage_fused(pos[cur], pos[cur ^ 1], it, k)
                    coll.all_gather_rows(pos[cur ^ 1], per, rank)
                    coll.all_reduce_tensor(total)
                    backend.controller_step(total, pos[self.pending[2]], self.pending[0], self.pending[1])
                    self.pending = None
                else:
                    backend.stage(pos[cur], pos[cur ^ 1], it, slot, stages, k)
                    coll.all_gather_rows(pos[cur ^ 1], per, rank)
                self.cur ^= 1
            self.k = k = k * (1.0 - self.cooling_rate)
            self.it = it + 1
            done += 1
            if (it + 1) % self.freq == 0 or it == self.n_iter - 1:
                t0 = time.perf_counter() if timers else 0.0
                if self.fusable and it + 1 < self.n_iter and self._n_slots_of(it + 1, k)[1] == 1:
                    self.pending = (it + 1, k, self.cur)
                else:
                    self._separate_check(self.cur, it + 1, k)
                self.checks += 1
                if timers:
                    backend.synchronize()
                    self.t_check += time.perf_counter() - t0
                if self.checks % max(1, self.sync_every) == 0 and backend.poll()[0]:
                    self.stopped = True
        return done

    def flush(self):
        """A check still waiting for a sweep to ride on runs as a separate pass (the caller wants the state)."""
        if self.pending is not None:
            self._separate_check(self.pending[2], self.pending[0], self.pending[1])
            self.pending = None

    def finish(self) -> ShardedResult:
        backend = self.backend
        self.flush()
        _stopped, iters_run = backend.poll()
        bad = int(self.coll.min_float(float(backend.first_nonfinite() or 0x7FFFFFFF)))
        if bad != 0x7FFFFFFF:
            t = -(-bad // 10) * 10                      # reference :359-361: inspected every 10th iteration,
            if t <= iters_run and not (_stopped and t == iters_run):    # after that iteration's check
                raise _native.NativeError(
                    _native.ERR_NONFINITE, "Numerical instability at iteration %d. Reduce k0 or c_repulsion." % t)
        r = backend.finish()
        return ShardedResult(r.positions, bool(r.converged), int(r.iterations), float(r.final_mae), float(r.final_k),
                             iters_run, int(r.info.get("n_checks", self.checks)), self.t_stage, self.t_gather,
                             self.t_check)


def relax_sharded(backend, coll: Collectives, rank: int, world: int, n: int, initial_positions,
                  n_iter: int, k0: float, cooling_rate: float, c_repulsion: float,
                  relative_epsilon: float = 1e-4, convergence_window: int = 5,
                  convergence_check_freq: int = 3, seed: int = 0, slab_stages: int = 0,
                  timers: bool = False, sync_every: int = 8) -> ShardedResult:
    """One embedding row-sharded over the ranks, from start to the controller's stop (ShardedRelaxation)."""
    run = ShardedRelaxation(backend, coll, rank, world, n, initial_positions, n_iter, k0, cooling_rate, c_repulsion,
                            relative_epsilon, convergence_window, convergence_check_freq, seed, slab_stages, timers,
                            sync_every)
    run.advance(n_iter)
    return run.finish()


# --------------------------------------------------------------------------------------
# synthetic config 4 generated directly on the device, row block by row block
# --------------------------------------------------------------------------------------
def _hash31(torch, key, salt: int):
    """Cheap 31-bit integer hash of an int64 tensor (masking keeps every product positive)."""
    m = 0x7FFFFFFF
    h = (key ^ salt) & m
    h = (h * 1103515245 + 12345) & m
    h = ((h ^ (h >> 15)) * 1664525 + 1013904223) & m
    h = ((h ^ (h >> 13)) * 22695477 + 1) & m
    return h ^ (h >> 16)


def encode_words_torch(torch, t, code=None):
    """topolow_encode_target() on a float tensor (device side): 4-ulp-rounded fp32 bits with the
    threshold code in the two low bits; non-finite -> unmeasured word."""
    f = t.to(torch.float32)
    u = f.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    sign = u & 0x80000000
    mag = u & 0x7FFFFFFF
    mag = torch.where(mag >= 0x7F7FFFFC, torch.full_like(mag, 0x7F7FFFFC), (mag + 2) & ~3)
    w = sign | mag
    if code is not None:
        w = w | code.to(torch.int64)
    w = torch.where(torch.isfinite(f), w, torch.full_like(w, 0x7F800002))
    return (w & 0xFFFFFFFF).to(torch.int64)


def load_synthetic_block(backend: HipBackend, n: int, latent_dim: int, missing: float, seed: int,
                         rank: int, world: int, rows=None):
    """Fills this rank's encoded row block of a synthetic problem (same recipe as
    synthetic.make_problem, but the missing mask and the noise come from a symmetric hash of
    the pair so every rank can build its rows independently).  Returns the rank's share of
    the MAE edge list and the degrees."""
    import torch
    from . import synthetic
    s = backend.session
    dev = backend.device
    rng = np.random.Generator(np.random.PCG64(seed))
    x = torch.from_numpy(synthetic.latent_points(n, latent_dim, rng)).to(dev)
    b, e = rows if rows is not None else row_block(n, world, rank)[:2]
    ld = s.encoded_ld
    rows = e - b
    # view the session's HBM block as a torch tensor (row-major, include/topolow_relax.h:
    # topolow_session_encoded_ptr)
    enc = _as_tensor(torch, s.encoded_ptr, (rows, ld), torch.int32, dev)
    thresh = int(missing * 0x7FFFFFFF)
    cols = torch.arange(n, device=dev, dtype=torch.int64)
    deg = torch.zeros(n, dtype=torch.int64, device=dev)
    ei, ej, ed = [], [], []
    step = max(1, min(rows, (64 << 20) // max(1, n)))
    for r0 in range(b, e, step):
        r1 = min(e, r0 + step)
        ri = torch.arange(r0, r1, device=dev, dtype=torch.int64)[:, None]
        lo = torch.minimum(ri, cols[None, :])
        hi = torch.maximum(ri, cols[None, :])
        key = lo * n + hi
        measured = (_hash31(torch, key, 0x5bd1e995 + seed) >= thresh) & (ri != cols[None, :])
        g = sum((_hash31(torch, key, 0x1234567 * (q + 1) + seed).to(torch.float64) / 0x7FFFFFFF)
                for q in range(4))
        noise = 1.0 + 0.05 * (g - 2.0) * math.sqrt(3.0)      # Irwin-Hall(4) ~ N(0,1)
        d = torch.cdist(x[r0:r1], x) * noise
        d = torch.clamp(d, min=0.1)
        words = encode_words_torch(torch, torch.where(measured, d, torch.full_like(d, float("inf"))))
        block = torch.full((r1 - r0, ld), 0x7F800002, dtype=torch.int64, device=dev)
        block[:, :n] = words
        enc[r0 - b:r1 - b] = _to_i32(torch, block)
        deg[r0:r1] = measured.sum(1) + 1   # the reference counts the (zero) diagonal, R/core.R:341
        # this rank's share of the MAE edges: pair {i,j} goes to the owner of i when i+j is
        # even and i<j, or when i+j is odd and i>j  (balances the upper triangle over ranks)
        par = ((ri + cols[None, :]) & 1) == 0
        take = measured & ((par & (ri < cols[None, :])) | (~par & (ri > cols[None, :])))
        idx = take.nonzero(as_tuple=False)
        a = idx[:, 0] + r0
        c = idx[:, 1]
        tgt = d[idx[:, 0], idx[:, 1]]
        ei.append(torch.minimum(a, c).to(torch.int32).cpu())
        ej.append(torch.maximum(a, c).to(torch.int32).cpu())
        ed.append(tgt.cpu())
    deg_np = deg.to(torch.int32).cpu().numpy()
    # degrees of rows this rank does not own are never read by its kernels
    s.commit_encoded(np.maximum(deg_np, 1).astype(np.int32) - 0)
    ei = torch.cat(ei).numpy(); ej = torch.cat(ej).numpy(); ed = torch.cat(ed).numpy()
    s.set_edges(ei, ej, ed, np.zeros(ei.shape[0], np.int32))
    scale = float(torch.cdist(x[:2048], x[:2048]).max().item())
    return int(ei.shape[0]), scale


def _to_i32(torch, w64):
    """int64 words (0..2^32-1) -> int32 bit patterns."""
    return torch.where(w64 >= 0x80000000, w64 - 0x100000000, w64).to(torch.int32)


def _as_tensor(torch, ptr: int, shape, dtype, device):
    class _Holder:
        pass
    h = _Holder()
    itemsize = torch.empty((), dtype=dtype).element_size()
    typestr = {torch.int32: "<i4", torch.float32: "<f4"}[dtype]
    h.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False),
                                      version=2, strides=None)
    del itemsize
    return torch.as_tensor(h, device=device)


# --------------------------------------------------------------------------------------
# bench entry (called by bench.py when --gpus > 1 or WORLD_SIZE > 1)
# --------------------------------------------------------------------------------------
def bench_main(args):
    import torch
    import torch.distributed as dist
    rank, world, local = dist_info()
    local = local % max(1, torch.cuda.device_count())   # rehearsals may put several ranks on one GPU
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        backend = os.environ.get("TOPOLOW_DIST_BACKEND", "nccl")   # nccl == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    coll = Collectives(world)
    mode = getattr(args, "mode", None) or "auto"
    # auto (the driver's `bench.py --gpus N`): the path BASELINE.json's north_star names for more than one GPU -- ONE
    # config-4 embedding row-sharded over the ranks (strong scaling) -- is the line's value; the reference's own
    # parallel mode (independent embeddings, weak scaling) follows as the secondary field "replicas"
    if mode in ("auto", "sharded"):
        out = _bench_sharded(args, rank, world, local, coll)
        if mode == "auto" and world > 1 and not getattr(args, "devices", ""):
            rep = _bench_replicas(args, rank, world, local, coll)
            out["replicas"] = {k: rep[k] for k in ("value", "unit", "scaling", "ms_per_step", "config", "roofline")}
    else:
        out = _bench_replicas(args, rank, world, local, coll)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _bench_replicas(args, rank, world, local, coll):
    """Weak scaling: every rank relaxes its own copy of BASELINE config 3 (independent
    embeddings, no data-path collective -- the reference's own parallel mode); value = iterations/s summed
    over the ranks.  Same session flow as the one-GPU bench (bench.py: run_single)."""
    import torch
    from . import core, synthetic
    n = args.n or 10000
    ndim, k0, cool, c_rep = 5, 5.0, 0.01, 0.01
    K, W = args.steps, args.warmup
    prob = synthetic.make_problem(n, latent_dim=5, missing=0.7, seed=12345 + rank)
    init = synthetic.initial_positions(prob.dissimilarity, 5, 12345 + rank)
    call = core.prepare_layout_call(prob.dissimilarity, 5, 1, k0, cool, c_rep, 1e-4, 5, init, False, 3, True)
    s = _native.Session(n, ndim, precision="f32", device=local)
    s.set_relabel(2024 + rank)
    s.load_coo(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh, call.degrees)
    s.set_edges(call.edge_i, call.edge_j, call.edge_dist, call.edge_thresh)

    def fresh(total):
        s.set_positions(call.initial_positions)
        s.begin(total, k0, cool, c_rep, 1e-4, 10 ** 9, 3, 2024 + rank, args.stages)

    # the job: this rank's embedding relaxed to the controller's own stop; its length (max over ranks) sets how
    # many K-iteration slices a rotation has (bench.py: run_single has the reasoning)
    s.set_positions(call.initial_positions)
    s.begin(1000, k0, cool, c_rep, 1e-4, 5, 3, 2024 + rank, args.stages)
    s.run()
    iters_run, _st, _m = s.sync()
    s.finish()
    n_job = int(coll.max_float(float(iters_run)))
    P = max(1, -(-n_job // K))

    def rotation():
        """W untimed iterations of a throw-away run, then the job from its start in P slices of EXACTLY K
        iterations, every slice bracketed by barrier + device synchronisation; a slice's time is the MAX over the
        ranks."""
        if W > 0:
            fresh(W)
            done = 0
            while done < W:
                done += s.enqueue(W - done)
            s.sync()
        fresh(P * K)
        slices = []
        for _p in range(P):
            torch.cuda.synchronize()
            coll.barrier()
            t0 = time.perf_counter()
            done = 0
            while done < K:
                done += s.enqueue(K - done)
            if _p + 1 < P:      # as bench.py: run_single -- a check waiting to ride on the next sweep stays pending
                s.wait()
            else:
                s.sync()
            torch.cuda.synchronize()
            coll.barrier()
            slices.append(coll.max_float(time.perf_counter() - t0))
        return slices

    rotations = []
    while (sum(map(sum, rotations)) < args.min_timed or len(rotations) < 3) and len(rotations) < 100:   # same on every rank
        rotations.append(rotation())
    rates = [world * K / float(np.mean(r)) for r in rotations]
    elapsed = world * K / float(np.median(rates))
    res = s.finish()
    # roofline of the dominant kernel on this rank (profiled pass over the same iterations; rank 0 reports)
    fresh(P * K)
    s.set_profiling(True)
    done = 0
    while done < P * K:
        done += s.enqueue(P * K - done)
    sym_ms, sym_iters, sym_err_ms, sym_err_iters = s.profile_symmetric()
    fused_ms, fused_launches = s.profile_fused()
    stage_ms, launches, check_ms, checks = s.profile()
    s.set_profiling(False)
    bytes_iter = s.bytes_per_iteration
    if sym_iters > 0:      # one-stage iterations ran as symmetric sweep + apply: the dominant kind (bench.py: run_single)
        kernel = "symm_sweep_kernel<5> + symm_apply_kernel<5>"
        per_launch = float(bytes_iter)
        avg_s = sym_ms * 1e-3 / sym_iters
    else:
        kernel = "slab_stage_pipe_kernel<5,float>"
        plain = max(launches - fused_launches, 1)
        per_launch = bytes_iter * (P * K - fused_launches) / plain     # fused launches are whole-matrix sweeps
        avg_s = (stage_ms - fused_ms) * 1e-3 / plain
    launches += sym_iters + sym_err_iters
    achieved = per_launch / avg_s / 1e9
    s.close()
    return {
        "metric": "relaxation iterations/sec (NxN pairs)", "value": world * K / elapsed,
        "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config 3 replicated: {world} independent embeddings (one per GPU), each "
                               f"synthetic N={n}, 70% missing, ndim=5, k0=5, cooling=0.01, c_repulsion=0.01",
                   "n_points": n, "ndim": ndim, "schedule": "slab", "parallelism": f"replicas x{world}",
                   "stages_per_iteration": launches / (P * K)},
        "timing": {"job_iterations": n_job, "slices_per_rotation": P, "rotations": len(rotations),
                   "timed_seconds": float(sum(map(sum, rotations))),
                   "iterations_per_s": {"min": min(rates), "median": float(np.median(rates)), "max": max(rates)},
                   "note": "every rank's job (its embedding to the controller's own stop) is timed in slices of exactly "
                           "K iterations between barriers and device synchronisations, a slice's time the MAX over "
                           "ranks, after W untimed iterations of a throw-away run; value = world * K / mean slice, "
                           "median over rotations"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0, "traffic": None, "kernel": kernel,
                     "avg_launch_us": avg_s * 1e6, "algorithmic_bytes_per_launch": per_launch,
                     "note": "rank 0, per GPU; the dominant kind of sweep without a fused check (HIP events, separate pass)"},
        "final_mae": res.final_mae,
    }


def _bench_sharded_native(args, devices):
    """ONE config-4 embedding row-sharded over `devices` from THIS process -- the deployment the reference's
    host implies (an R session is one process): topolow_sessions_run_sharded, one host thread per row block,
    position slices peer-stored by the stage kernels, HIP-event barriers, replicated device controller."""
    import torch
    n = args.n or 50000
    ndim, k0, cool, c_rep = 3, 5.0, 0.01, 0.01
    K, W = args.steps, args.warmup
    rows = _native.shard_rows(n, len(devices))
    backs = [HipBackend(n, ndim, rb, re_, devices[b % len(devices)]) for b, (rb, re_) in enumerate(rows)]
    n_edges, scale = 0, 1.0
    for b, bk in enumerate(backs):
        ne, scale = load_synthetic_block(bk, n, 3, 0.9, 12345, b, len(backs), rows=rows[b])
        n_edges += ne
    rng = np.random.Generator(np.random.PCG64(999))
    init = np.zeros((n, ndim))
    init[1:] = np.cumsum(rng.uniform(0.0, 2.0 * scale / n, size=(n - 1, ndim)), axis=0)
    ss = [bk.session for bk in backs]
    sync = lambda: [torch.cuda.synchronize(d) for d in sorted(set(devices))]
    W = max(W, 16)     # the unfolding phase (8 iterations of 16 stages) belongs to the warm-up
    _native.run_sharded(ss, init, 3, k0, cool, c_rep, 1e-4, 10 ** 9, 3, 7, args.stages)
    sync()
    passes = []
    while (sum(passes) < args.min_timed or len(passes) < 3) and len(passes) < 50:
        # one call runs W + K iterations; the library drains the GPUs after W and clocks the remaining K
        # between two device-wide synchronisations of its own (topolow_shard_stats.timed_seconds)
        res = _native.run_sharded(ss, init, W + K, k0, cool, c_rep, 1e-4, 10 ** 9, 3, 7, args.stages,
                                  warmup_iterations=W)
        sync()
        passes.append(res.info["timed_seconds"])
    elapsed = float(np.median(passes))
    prof = _native.run_sharded(ss, init, W + K, k0, cool, c_rep, 1e-4, 10 ** 9, 3, 7, args.stages, profile=True)
    i = dict(prof.info)
    # scale the profiled pass (W + K iterations, early ones included) to per-iteration figures of its own
    per_iter = 1.0 / (W + K)
    bytes_iter_block0 = ss[0].bytes_per_iteration
    n_group0 = sum(1 for d in devices[:len(rows)] if d == devices[0])
    achieved = (sum(ss[b].bytes_per_iteration for b in range(len(rows)) if devices[b % len(devices)] == devices[0])
                * (W + K) / max(i["stage_kernel_seconds"], 1e-12) / 1e9)
    out = {
        "metric": "relaxation iterations/sec (NxN pairs)", "value": K / elapsed, "unit": "iterations/s",
        "n_gpus": len(set(devices)), "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config 4: ONE embedding, synthetic N={n}, 90% missing, ndim=3, {len(rows)} row "
                               f"blocks on devices {devices}, single process (native row-sharded engine)",
                   "n_points": n, "ndim": ndim, "schedule": "slab", "parallelism": f"rows/{len(rows)}",
                   "edges": n_edges},
        "timing": {"passes": len(passes), "timed_seconds": float(sum(passes)),
                   "iterations_per_s": {"min": K / max(passes), "median": K / elapsed, "max": K / min(passes)},
                   "note": "each pass = one topolow_sessions_run_sharded call of W + K iterations; the library drains "
                           "the GPUs after W and times the remaining K itself"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                     "traffic": None,
                     "kernel": ("slab_stage_pipe_kernel<3,float> (multi-stage iterations) + symm_sweep_kernel<3> / "
                                "symm_apply_kernel<3> (one-stage iterations: ONE block is the whole matrix)"
                                if len(rows) == 1 else "slab_stage_pipe_kernel<3,float>"),
                     "note": f"first GPU ({n_group0} row block(s)): algorithmic bytes of its row blocks / summed "
                             "durations of their sweeps (HIP events, profiled pass over W + K iterations)"},
        "breakdown_ms_per_iteration": {
            "note": "profiled pass, W + K iterations incl. the 16-stage unfolding phase, first GPU",
            "loop_wall": 1e3 * i["loop_seconds"] * per_iter,
            "stage_kernels": 1e3 * i["stage_kernel_seconds"] * per_iter,
            "check_kernels": 1e3 * i["check_kernel_seconds"] * per_iter,
            "barriers_and_host": 1e3 * (i["loop_seconds"] - i["stage_kernel_seconds"] - i["check_kernel_seconds"]) * per_iter,
            "host_overhead_fraction_of_stage_time": (i["loop_seconds"] - i["stage_kernel_seconds"] -
                                                     i["check_kernel_seconds"]) / max(i["stage_kernel_seconds"], 1e-12),
            "exchanges_per_iteration": i["exchanges"] * per_iter, "host_threads": i["groups"]},
        "final_mae": res.final_mae, "stage_launches_block0": int(i["stage_launches"]),
    }
    for s_ in ss:
        s_.close()
    return out


def _bench_sharded(args, rank, world, local, coll):
    """Strong scaling: ONE config-4 embedding (N=50 000, 90 % missing, ndim=3) row-sharded over the ranks (one
    process per GPU), all-gather of the position slices after every slab stage over RCCL.  Timed like the one-GPU
    line (bench.py: run_single): the JOB -- the embedding relaxed to the controller's own stop, unfolding phase
    included -- in slices of exactly K iterations after W untimed ones, every slice between barrier + device
    synchronisation on both sides, a slice's time the MAX over the ranks; value = K / mean slice."""
    import socket
    import torch
    if world == 1 and getattr(args, "devices", ""):
        return _bench_sharded_native(args, [int(d) for d in args.devices.split(",")])
    n = args.n or 50000
    ndim, k0, cool, c_rep = 3, 5.0, 0.01, 0.01
    K, W = args.steps, args.warmup
    b, e, per = row_block(n, world, rank)
    backend = HipBackend(n, ndim, b, e, local)
    n_edges, scale = load_synthetic_block(backend, n, 3, 0.9, 12345, rank, world)
    # one-stage iterations as the symmetric sweep sharded over the ranks: every rank gets the rows of its segment of the
    # upper triangle's tile list from their owners (once), then reads half the bytes per iteration and the ranks
    # exchange ONE all-reduce of n x ndim floats instead of the all-gather
    t0 = time.perf_counter()
    symmetric = backend.symm_prepare(coll, rank, world)
    torch.cuda.synchronize()
    symm_setup_s = coll.max_float(time.perf_counter() - t0)
    rng = np.random.Generator(np.random.PCG64(999))
    init = np.zeros((n, ndim))
    init[1:] = np.cumsum(rng.uniform(0.0, 2.0 * scale / n, size=(n - 1, ndim)), axis=0)

    def new_run(n_iter, window, timers=False):
        return ShardedRelaxation(backend, coll, rank, world, n, init, n_iter, k0, cool, c_rep, 1e-4, window, 3, 7,
                                 args.stages, timers=timers)

    def fence():
        torch.cuda.synchronize()
        coll.barrier()

    # ---- the job: one embedding to the controller's own stop (first run: code objects, communicator) ----
    for _warm in (True, False):
        run = new_run(1000, 5)
        fence()
        t0 = time.perf_counter()
        run.advance(1000)
        job = run.finish()
        fence()
        whole = coll.max_float(time.perf_counter() - t0)
    n_job = int(job.iterations_run)
    P = max(1, -(-n_job // K))

    def rotation():
        if W > 0:                    # W untimed iterations of a throw-away run
            r0 = new_run(W, 10 ** 9)
            r0.advance(W)
            r0.finish()
        run = new_run(P * K, 10 ** 9)
        slices = []
        for _p in range(P):
            fence()
            t0 = time.perf_counter()
            got = run.advance(K)
            backend.poll()
            fence()
            slices.append(coll.max_float(time.perf_counter() - t0))
            assert got == K, got
        run.finish()
        return slices

    rotations = []
    while (sum(map(sum, rotations)) < args.min_timed or len(rotations) < 3) and len(rotations) < 100:   # same on every rank
        rotations.append(rotation())
    rates = [K / float(np.mean(r)) for r in rotations]
    elapsed = K / float(np.median(rates))
    slice_rates = K / np.mean(np.array(rotations), axis=0)
    # ---- breakdown pass over the same iterations, host-synchronised around every stage and every gather ----
    brk_run = new_run(P * K, 10 ** 9, timers=True)
    brk_run.advance(P * K)
    brk = brk_run.finish()
    bytes_iter_rank = backend.session.bytes_per_iteration
    stages = backend.session.stage_launches
    achieved = bytes_iter_rank * (P * K) / max(brk.stage_seconds, 1e-9) / 1e9
    mine = dict(rank=rank, host=socket.gethostname(), device=int(local), rows=[int(b), int(e)], achieved=achieved,
                frac=achieved / 8000.0, stage_ms=1e3 * brk.stage_seconds / (P * K),
                all_gather_ms=1e3 * brk.gather_seconds / (P * K), check_ms=1e3 * brk.check_seconds / (P * K))
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        coll.dist.all_gather_object(per_rank, mine)
    backend.session.close()
    n_gpus = len({(r["host"], r["device"]) for r in per_rank})
    return {
        "metric": "relaxation iterations/sec (NxN pairs)", "value": K / elapsed, "unit": "iterations/s",
        "n_gpus": n_gpus, "ranks": world,
        "collective_backend": (coll.backend if world > 1 else None),
        "rccl_ranks": world if (world > 1 and coll.backend == "nccl") else 0,
        "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"config 4: ONE embedding, synthetic N={n}, 90% missing, ndim=3, row-block "
                               f"sharded over {world} rank(s) on {n_gpus} GPU(s), all-gather of position slices per "
                               "slab stage" + ("; one-stage iterations as the symmetric sweep over tile segments, one "
                                               "all-reduce of the n x ndim moves each" if symmetric else ""),
                   "n_points": n, "ndim": ndim, "schedule": "slab", "parallelism": f"rows/{world}",
                   "edges_rank0": n_edges, "symmetric_segments": bool(symmetric),
                   "symmetric_setup_seconds": symm_setup_s if symmetric else None},
        "timing": {"job_iterations": n_job, "slices_per_rotation": P, "rotations": len(rotations),
                   "timed_seconds": float(sum(map(sum, rotations))),
                   "iterations_per_s": {"min": min(rates), "median": float(np.median(rates)), "max": max(rates)},
                   "iterations_per_s_by_slice": [round(float(x), 1) for x in slice_rates],
                   "note": "the job (one embedding to the controller's own stop, unfolding phase included) timed in "
                           "slices of exactly K iterations, each between barrier + device synchronisation on both "
                           "sides, a slice's time the MAX over ranks, after W untimed iterations of a throw-away run; "
                           "value = K / mean slice, median over rotations -- the same rule as the one-GPU line"},
        "whole_run": {"iterations_run": n_job, "seconds": whole, "iterations_per_s": n_job / whole,
                      "converged": bool(job.converged), "best_iteration": int(job.iterations),
                      "final_mae": job.final_mae},
        "roofline": {"bound": "hbm", "achieved": per_rank[0]["achieved"], "peak": 8000.0, "unit": "GB/s",
                     "frac": per_rank[0]["frac"], "traffic": None,
                     "kernel": ("slab_stage_pipe_kernel<3,float> (multi-stage iterations) + symm_sweep_kernel<3> / "
                                "symm_partial_kernel<3> / symm_owner_apply_kernel<3> (one-stage iterations)"
                                if symmetric else "slab_stage_pipe_kernel<3,float>"),
                     "per_gpu_frac": [round(r["frac"], 4) for r in per_rank],
                     "note": "per rank: algorithmic bytes of its row block over the job / its host-synchronised "
                             "stage-kernel time (breakdown pass); `frac` is rank 0's"},
        "breakdown_ms_per_iteration": {"stage": per_rank[0]["stage_ms"], "all_gather": per_rank[0]["all_gather_ms"],
                                       "check": per_rank[0]["check_ms"],
                                       "note": "rank 0, breakdown pass (host-synchronised, so the parts add up to "
                                               "more than ms_per_step)", "per_rank": per_rank},
        "final_mae": job.final_mae, "stage_launches": int(stages),
    }
