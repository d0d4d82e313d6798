// topolow_amd/csrc/relax_sharded_engine.h -- row-sharded relaxation of ONE embedding over several
// row-block sessions, driven from one process (an R session is one process: reference
// src/RcppExports.cpp:16-39).  Implementation fragment of topolow_relax.hip (uses its session type and
// launch helpers; included inside its anonymous namespace).
//
// Layout: block b owns rows [row_begin_b, row_end_b) of the encoded target
// matrix and moves only its own points; every block keeps ALL n positions (they are tiny).  Per slab
// stage:
//   * block b's stage kernel writes its updated rows into its own next-position buffer AND, from the
//     same epilogue, into the next-position buffer of every other block (peer stores -- over xGMI when
//     the blocks sit on different GPUs): the all-gather of the position slices is fused into the
//     kernel, there is no separate collective and no staging copy;
//   * a barrier between the GPUs out of HIP events: every GPU's thread records an event behind its
//     stage kernels and makes its stream wait for the events of the others (blocks that share a GPU share
//     a stream: stream order is their barrier).  Peer-written data is consumed only by
//     kernels that START after that wait -- kernel-boundary visibility, the guarantee HIP gives for
//     ordinary (coarse-grained) device memory; nothing spins inside a kernel on a remote flag.
// Per convergence check each block reduces its parity share of the measured pairs, folds the
// partials into one (sum, count) and peer-stores it into a slot of every block's rank table; after
// one more barrier every block runs the SAME controller on the SAME numbers in the same order, so
// the decisions (stop / snapshot) are replicated without any host round trip.
// One host thread per GPU enqueues that GPU's work (the event waits are per pair of GPUs, a single
// thread would issue G^2 of them per stage); threads meet in a spin barrier between "record" and
// "wait" so an event is always recorded before anyone waits on it.  The calling thread is the first
// GPU's thread: only it polls the caller's interrupt callback (R's API is main-thread only).
// Measured on one MI355X (tests/study/shard_exchange_latency.py): with every block forced into its own
// thread and stream the barrier costs 27 / 62 / 198 us per stage at 2 / 4 / 8 blocks -- eight threads
// issuing 72 event calls per stage into ONE device's runtime lock; that is why same-GPU blocks share a
// stream, and it is the upper bound of what 8 GPUs (one lock each, 9 calls per thread) can cost.

struct ShardedAbortableBarrier {
  std::atomic<int> count{0};
  std::atomic<int> sense{0};
  std::atomic<bool> failed{false};
  int n;
  explicit ShardedAbortableBarrier(int n_) : n(n_) {}
  // false: some thread failed (it never arrives); the caller must unwind
  bool wait() {
    const int s = sense.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
      count.store(0, std::memory_order_relaxed);
      sense.store(s ^ 1, std::memory_order_release);
      return !failed.load(std::memory_order_acquire);
    }
    int spins = 0;
    while (sense.load(std::memory_order_acquire) == s) {
      if (failed.load(std::memory_order_acquire)) return false;
      if (++spins > 4000) std::this_thread::yield();
    }
    return !failed.load(std::memory_order_acquire);
  }
  void fail() { failed.store(true, std::memory_order_release); }
};

// Blocks that share a GPU form a GROUP: one host thread enqueues all of them on ONE stream, stage by
// stage, so stream order is their barrier (their kernels each fill the chip anyway).  Groups -- i.e.
// different GPUs, one block each in production -- meet at the event barrier.  TOPOLOW_SHARD_THREAD_PER_BLOCK=1
// makes every block its own group (tests drive the cross-group path on one GPU with it).
struct ShardedGroup {
  std::vector<int> blocks;     // indices into ShardedRun::ss, ascending
  hipStream_t stream = nullptr;
  int device = 0;
};

struct ShardedRun {
  std::vector<topolow_session*> ss;
  std::vector<ShardedGroup> groups;
  int P = 0;
  int n_iter = 0, check_freq = 3;
  double k0 = 0, cooling = 0;
  int fixed_stages = 0;
  std::vector<std::array<hipEvent_t, 2>> ev;      // per group
  ShardedAbortableBarrier* bar = nullptr;
  std::atomic<int> flag[2];      // published by group 0's thread before barrier g (slot g & 1): bit 0 stop, bit 1 interrupt
  int32_t (*interrupt_cb)(void*) = nullptr;
  void* interrupt_user = nullptr;
  std::mutex err_mu;
  HipError first_error{TOPOLOW_OK, ""};
  int warmup_iters = 0;          // > 0: the clock of `timed_seconds` starts when these iterations have drained
  double t_timed0 = 0.0;
  bool pair_sharded = false;     // one-stage iterations run as the symmetric sweep sharded over the sessions (relax_symm.h)
  // results of the loop
  int iters_enqueued = 0;
  bool interrupted = false;
  long long exchanges = 0;
};

// One group's enqueue loop (thread r).  Throws HipError; the caller marks the barrier failed.
inline void sharded_worker(ShardedRun& R, int r) {
  const ShardedGroup& G = R.groups[r];
  const int n_groups = (int)R.groups.size();
  topolow_session* lead = R.ss[G.blocks[0]];
  HIP_TRY(hipSetDevice(G.device));
  long long g = 0;          // exchange counter, identical in every thread
  int seen = 0;             // flag value read at the last exchange, identical in every thread
  auto exchange = [&]() -> bool {
    int f = 0;
    if (r == 0) {
      f = R.ss[0]->mailbox->stopped ? 1 : 0;
      if (R.interrupted) f |= 2;
    }
    if (n_groups == 1) { seen = f; ++g; return true; }   // stream order is the barrier
    HIP_TRY(hipEventRecord(R.ev[r][g & 1], G.stream));
    if (r == 0) R.flag[g & 1].store(f, std::memory_order_release);
    if (!R.bar->wait()) return false;
    for (int p = 0; p < n_groups; ++p)
      if (p != r) HIP_TRY(hipStreamWaitEvent(G.stream, R.ev[p][g & 1], 0));
    seen = R.flag[g & 1].load(std::memory_order_acquire);
    ++g;
    return true;
  };
  int cur = 0;
  double k = R.k0;
  int iter = 0;
  // Position buffers rotate through three: a stage reads `cur` and writes (own rows, and the other blocks' copies)
  // the next one, so the buffer a check measured is not written for two more stages.
  // A check whose error pass is deferred into the next iteration's single sweep (slab_stage_pipe_kernel<ERR>):
  // the sweep's barrier then also carries the blocks' (sum, count) slots, so the check costs no pass over the
  // block and no barrier of its own.  Its controller runs behind that barrier and snapshots the buffer the sweep
  // READ; other blocks may already be writing their next stage -- into the third buffer.  Decided from the
  // schedule alone, hence identically in every thread.
  bool pend = false;
  int pend_iter1 = 0, pend_buf = 0;
  double pend_k = 0.0;
  // Rank tables alternate between successive checks: with fused checks a block may already be pushing the NEXT
  // check's (sum, count) -- it rides on the sweep that follows the barrier -- while another block's controller
  // still reads this check's table behind the same barrier.  Two tables are enough: the push of check q + 2 is
  // issued behind the barrier that follows controller(q) in every stream.  (With one table and one stream per
  // block, controllers of different blocks saw different MAEs for the same check and stopped at different
  // iterations: tests/test_gpu_sharded_native.py::test_sharded_fuzz_equals_single_session.)
  long long n_check = 0;
  auto can_fuse = [&](const topolow_session* s) {
    return s->fuse_checks && s->dense_mae && s->precision == TOPOLOW_PRECISION_F32 && s->rows() % 2 == 0;
  };
  bool fusable = true;
  for (topolow_session* s : R.ss) fusable = fusable && can_fuse(s);
  // one-stage iterations as the sharded symmetric sweep: its ERR instance has no even-rows rule
  bool fusable_pair = R.pair_sharded;
  for (topolow_session* s : R.ss)
    fusable_pair = fusable_pair && s->fuse_checks && s->dense_mae && s->precision == TOPOLOW_PRECISION_F32;
  auto separate_check = [&](int buf, int iter1, double k_after) -> bool {
    const int tab = (int)(n_check & 1) * R.P;
    ++n_check;
    for (int b : G.blocks) {
      topolow_session* s = R.ss[b];
      ProfScope prof(s, &s->prof_check);
      TL_DISPATCH_DIM(s->dim, launch_edge_error, s, s->pos[buf].p, s->state.p);
      hipLaunchKernelGGL(reduce_push_kernel, dim3(1), dim3(1024), 0, s->stream, s->part_sum.p, s->part_cnt.p,
                         error_parts(s), s->rsum_tab.p, s->rcnt_tab.p, s->n_ranks, tab + s->rank, s->state.p);
      HIP_TRY(hipGetLastError());
    }
    if (!exchange()) return false;
    for (int b : G.blocks) {
      topolow_session* s = R.ss[b];
      ProfScope prof(s, &s->prof_check);
      launch_controller(s, s->pos[buf].p, iter1, k_after, s->rank_sum.p + tab, s->rank_cnt.p + tab, s->n_ranks);
    }
    return true;
  };
  for (; iter < R.n_iter; ++iter) {
    if (seen != 0) break;
    if (R.warmup_iters > 0 && iter == R.warmup_iters) {   // measurement only: drain, meet, start the clock
      HIP_TRY(hipStreamSynchronize(G.stream));
      if (n_groups > 1 && !R.bar->wait()) return;
      if (r == 0) R.t_timed0 = now_s();
      if (n_groups > 1 && !R.bar->wait()) return;
    }
    if (r == 0 && R.interrupt_cb != nullptr && iter > 0 && iter % 50 == 0 && !R.interrupted)   // reference :364
      R.interrupted = R.interrupt_cb(R.interrupt_user) != 0;   // published at the next exchange
    const int stages = R.fixed_stages > 0 ? R.fixed_stages : slab_stages_at(iter, k, lead->dim);
    const SlabGeom geo = slab_geom(lead->n, stages);
    const bool fuse_now = pend && geo.n_stages == 1;
    if (pend && !fuse_now) { if (!separate_check(pend_buf, pend_iter1, pend_k)) return; pend = false; }
    const int tab = (int)(n_check & 1) * R.P;   // the table of the check this sweep carries, if any
    if (fuse_now) ++n_check;
    if (R.pair_sharded && geo.n_stages == 1) {
      // every session sweeps its segment of the tile list and folds its partials into the owners' inboxes; barrier;
      // the owners move their points and store them into every session's next buffer; barrier
      const int nxt = (cur + 1) % 3;
      for (int b : G.blocks) {
        topolow_session* s = R.ss[b];
        TL_DISPATCH_DIM(s->dim, sym_sharded_sweep, s, s->pos[cur].p, iter, k, fuse_now);
        if (fuse_now) {
          ProfScope prof(s, &s->prof_check);
          hipLaunchKernelGGL(reduce_push_kernel, dim3(1), dim3(1024), 0, s->stream, s->part_sum.p, s->part_cnt.p,
                             s->sym.n_units, s->rsum_tab.p, s->rcnt_tab.p, s->n_ranks, tab + s->rank, s->state.p);
          HIP_TRY(hipGetLastError());
        }
      }
      if (!exchange()) return;
      if (fuse_now) {
        for (int b : G.blocks) {
          topolow_session* s = R.ss[b];
          ProfScope prof(s, &s->prof_check);
          launch_controller(s, s->pos[pend_buf].p, pend_iter1, pend_k, s->rank_sum.p + tab, s->rank_cnt.p + tab, s->n_ranks);
        }
        pend = false;
      }
      for (int b : G.blocks) {
        topolow_session* s = R.ss[b];
        TL_DISPATCH_DIM(s->dim, sym_sharded_apply, s, s->pos[cur].p, s->pos[nxt].p, s->push_tab[nxt].p, iter);
      }
      if (!exchange()) return;
      cur = nxt;
    } else if (R.P == 1 && lead->sym.two_stage && sym_rr_stages_ok(geo.n_stages) && sym_eligible(lead) && sym_available(lead) &&
               sym_rr_available(lead, geo.n_stages)) {
      // ONE block = the whole matrix: a 2-, 4- or 8-stage iteration as symmetric sweeps over the tiles of one stage each,
      // as in the session's own loop
      const int S = geo.n_stages;
      int order[8];
      sym_rr_order(lead->seed, iter, S, order);
      for (int t = 0; t < S; ++t) {
        const bool last = t == S - 1;
        TL_DISPATCH_DIM(lead->dim, sym_rr_stage, lead, lead->pos[cur].p, lead->pos[(cur + 1) % 3].p, iter, k,
                        last ? k * (1.0 - R.cooling) : k, S, order[t], last ? iter + 1 : iter);
        if (!exchange()) return;
        cur = (cur + 1) % 3;
      }
    } else
    for (int slot = 0; slot < geo.n_stages; ++slot) {
      const SlabRanges rg = slab_ranges(geo, lead->seed, iter, slot);
      for (int b : G.blocks) {
        topolow_session* s = R.ss[b];
        int stage_blocks = (s->rows() + CfgProd::ROWS - 1) / CfgProd::ROWS;
        if (R.P == 1 && geo.n_stages == 1 && sym_eligible(s) && sym_available(s)) {
          // ONE block = the whole matrix: a one-stage iteration may run as the symmetric sweep (relax_symm.h), as in
          // the session's own loop; its error partials are one per (unit)
          TL_DISPATCH_DIM(s->dim, sym_iteration, s, s->pos[cur].p, s->pos[(cur + 1) % 3].p, iter, k, fuse_now);
          stage_blocks = s->sym.n_units;
        } else {
          TL_DISPATCH_DIM(s->dim, launch_stage, s, s->pos[cur].p, s->pos[(cur + 1) % 3].p, s->state.p, rg, iter + 1, k,
                          s->push_tab[(cur + 1) % 3].p, s->n_push, fuse_now);
        }
        if (fuse_now) {   // the sweep's per-workgroup partials -> this block's slot of every rank table
          ProfScope prof(s, &s->prof_check);
          hipLaunchKernelGGL(reduce_push_kernel, dim3(1), dim3(1024), 0, s->stream, s->part_sum.p, s->part_cnt.p,
                             stage_blocks, s->rsum_tab.p, s->rcnt_tab.p, s->n_ranks, tab + s->rank, s->state.p);
          HIP_TRY(hipGetLastError());
        }
      }
      if (!exchange()) return;
      if (fuse_now) {
        for (int b : G.blocks) {
          topolow_session* s = R.ss[b];
          ProfScope prof(s, &s->prof_check);
          launch_controller(s, s->pos[pend_buf].p, pend_iter1, pend_k, s->rank_sum.p + tab, s->rank_cnt.p + tab, s->n_ranks);
        }
        pend = false;
      }
      cur = (cur + 1) % 3;
    }
    k *= (1.0 - R.cooling);   // reference :289
    if ((iter + 1) % R.check_freq == 0 || iter == R.n_iter - 1) {   // reference :294
      const bool fuse = (fusable || fusable_pair) && iter + 1 < R.n_iter &&
                        slab_geom(lead->n, R.fixed_stages > 0 ? R.fixed_stages
                                                              : slab_stages_at(iter + 1, k, lead->dim)).n_stages == 1;
      if (fuse) { pend = true; pend_iter1 = iter + 1; pend_k = k; pend_buf = cur; }
      else if (!separate_check(cur, iter + 1, k)) return;
    }
  }
  if (pend) { if (!separate_check(pend_buf, pend_iter1, pend_k)) return; pend = false; }   // the loop ended early
  for (int b : G.blocks) R.ss[b]->cur = cur;
  if (r == 0) { R.iters_enqueued = iter; R.exchanges = g; }
  HIP_TRY(hipStreamSynchronize(G.stream));
}

// Wires `count` loaded sessions (row blocks tiling [0, n) in order, same n / ndim / precision) into one
// another: peer access between their devices, push tables, rank tables.
inline void sharded_wire(std::vector<topolow_session*>& ss) {
  const int P = (int)ss.size();
  for (int a = 0; a < P; ++a)
    for (int b = 0; b < P; ++b) {
      if (ss[a]->device == ss[b]->device) continue;
      int can = 0;
      HIP_TRY(hipDeviceCanAccessPeer(&can, ss[a]->device, ss[b]->device));
      if (!can) throw HipError{TOPOLOW_ERR_UNSUPPORTED, "row-sharded run: the GPUs cannot access each other's memory"};
      HIP_TRY(hipSetDevice(ss[a]->device));
      const hipError_t e = hipDeviceEnablePeerAccess(ss[b]->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(e);
      (void)hipGetLastError();
    }
  for (int a = 0; a < P; ++a) {
    topolow_session* s = ss[a];
    HIP_TRY(hipSetDevice(s->device));
    s->n_ranks = P;
    s->rank = a;
    s->n_push = P - 1;
    for (int b2 = 0; b2 < 3; ++b2) {
      std::vector<void*> tab;
      for (int q = 0; q < P; ++q) if (q != a) tab.push_back((void*)ss[q]->pos[b2].p);
      s->push_tab[b2].alloc(tab.size());
      if (!tab.empty())
        HIP_TRY(hipMemcpy(s->push_tab[b2].p, tab.data(), tab.size() * sizeof(void*), hipMemcpyHostToDevice));
    }
    s->rank_sum.alloc(2 * P);   // two tables, alternating between successive checks (see sharded_worker)
    s->rank_cnt.alloc(2 * P);
    HIP_TRY(hipMemset(s->rank_sum.p, 0, sizeof(double) * 2 * P));
    HIP_TRY(hipMemset(s->rank_cnt.p, 0, sizeof(unsigned long long) * 2 * P));
  }
  for (int a = 0; a < P; ++a) {
    topolow_session* s = ss[a];
    HIP_TRY(hipSetDevice(s->device));
    std::vector<double*> ts;
    std::vector<unsigned long long*> tc;
    for (int q = 0; q < P; ++q) { ts.push_back(ss[q]->rank_sum.p); tc.push_back(ss[q]->rank_cnt.p); }
    s->rsum_tab.alloc(P);
    s->rcnt_tab.alloc(P);
    HIP_TRY(hipMemcpy(s->rsum_tab.p, ts.data(), P * sizeof(double*), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->rcnt_tab.p, tc.data(), P * sizeof(unsigned long long*), hipMemcpyHostToDevice));
  }
}
