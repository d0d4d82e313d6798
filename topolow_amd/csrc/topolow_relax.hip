// topolow_amd/csrc/topolow_relax.hip -- host side of libtopolow_relax.so: the C ABI of
// include/topolow_relax.h over the HIP kernels in relax_kernels.h / relax_gs.h.
//
// Replaces the native half of the reference's euclidean_embedding():
//   .Call(`_topolow_optimize_layout_exact_cpp`, ...)  (reference R/RcppExports.R:4-6)
//   -> optimize_layout_exact_cpp                       (reference src/optimization.cpp:109-382)
// There is no CPU fallback in this library: without a HIP device every entry point that
// computes returns TOPOLOW_ERR_NO_DEVICE.

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <array>
#include <atomic>
#include <deque>
#include <mutex>
#include <string>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/topolow_relax.h"
#include "relax_common.h"
#include "relax_kernels.h"
#include "relax_fold.h"
#include "relax_gs.h"
#include "relax_tilegs.h"
#include "relax_symm.h"
#include "relax_symm64.h"

using namespace topolow;

namespace {

constexpr int kMaxDim = 64;        // 1..16: the tuned kernels; 17..64: the plain stage kernel (relax_kernels.h: slab_stage_wide_kernel)
constexpr int kMaxTunedDim = 16;
// kernels are instantiated for 1..10, 12, 16, 32 and 64 coordinates; 11 runs as 12, 13..15 as 16, 17..31 as 32 and
// 33..63 as 64 with the extra coordinates held at exactly zero (a zero coordinate adds 0 to every distance and
// receives 0 of every move)
constexpr int kernel_dim(int ndim) { return ndim <= 10 ? ndim : (ndim <= 12 ? 12 : (ndim <= 16 ? 16 : (ndim <= 32 ? 32 : 64))); }
constexpr int kDefaultGsMaxN = 1024;

struct HipError {
  int code;
  std::string msg;
};

void set_err(char* errbuf, size_t errlen, const char* fmt, ...) {
  if (!errbuf || errlen == 0) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(errbuf, errlen, fmt, ap);
  va_end(ap);
}

// verbose lines go to the caller's sink (topolow_options.print_cb) or stdout
void emit(const topolow_options& opt, const char* fmt, ...) {
  char line[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(line, sizeof line, fmt, ap);
  va_end(ap);
  if (opt.print_cb) opt.print_cb(line, opt.print_user);
  else { std::fputs(line, stdout); std::fflush(stdout); }
}

// The reference's opening lines (src/optimization.cpp:183-188) plus which device path runs.
void emit_header(const topolow_options& opt, const char* path, int n, double k0, double cooling, double c_rep) {
  emit(opt, "=== Exact Algorithm (O(N^2) Full Pairwise) on HIP: %s ===\n", path);
  emit(opt, "Points: %d, Pairs per iteration: %lld\n", n, (long long)n * (n - 1) / 2);
  emit(opt, "Parameters: k0=%g, cooling=%g, c_rep=%g\n", k0, cooling, c_rep);
}

// Progress lines of the checks [from, to) of a trace (3 doubles per check), with the reference's
// cadence: checks that fall on a multiple of 10 iterations or on the last one (:298-301).
void emit_checks(const topolow_options& opt, const double* trace, int from, int to, int n_iter) {
  for (int c = from; c < to; ++c) {
    const int it = (int)trace[3 * c];
    if (it % 10 == 0 || it == n_iter)
      emit(opt, "Iter %d/%d, MAE=%g, k=%g\n", it, n_iter, trace[3 * c + 1], trace[3 * c + 2]);
  }
}

// Closing line of a converged run (:334-336, :351-353): the controller's counters tell which rule fired.
void emit_converged(const topolow_options& opt, bool plateau, int best_iter, double best_mae) {
  if (plateau) emit(opt, "Converged (plateau) at iter %d, MAE=%g\n", best_iter, best_mae);
  else emit(opt, "Converged (MAE worsening, best restored) at iter %d, MAE=%g\n", best_iter, best_mae);
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      throw HipError{e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice              \
                         ? TOPOLOW_ERR_NO_DEVICE                                        \
                         : TOPOLOW_ERR_HIP,                                             \
                     std::string(#expr) + ": " + hipGetErrorString(e_)};                \
    }                                                                                   \
  } while (0)

int select_device(int device) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    throw HipError{TOPOLOW_ERR_NO_DEVICE,
                   "no HIP device available (libtopolow_relax has no CPU fallback)"};
  if (device < 0) {
    HIP_TRY(hipGetDevice(&device));
  } else {
    if (device >= count) throw HipError{TOPOLOW_ERR_NO_DEVICE, "HIP device ordinal out of range"};
    HIP_TRY(hipSetDevice(device));
  }
  return device;
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  void alloc(size_t count) {
    release();
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc((void**)&p, count * sizeof(T)));
    n = count;
  }
  void release() {
    if (p) { (void)hipFree(p); p = nullptr; n = 0; }
  }
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

// Splits [0, n) over a few host threads (the one-shot call's host passes over 10^7..10^8 edges);
// small ranges run on the calling thread.  fn(begin, end) must be thread-safe.
template <typename Fn>
void host_parallel(size_t n, Fn fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t workers = std::max<size_t>(1, std::min<size_t>({n / (1u << 18), (size_t)(hw ? hw : 1), (size_t)16}));
  if (workers <= 1) { fn((size_t)0, n); return; }
  std::vector<std::thread> pool;
  const size_t step = (n + workers - 1) / workers;
  for (size_t w = 0; w < workers; ++w) {
    const size_t lo = w * step, hi = std::min(n, lo + step);
    if (lo >= hi) break;
    pool.emplace_back([=, &fn] { fn(lo, hi); });
  }
  for (auto& t : pool) t.join();
}

// Is the edge list exactly the measured strict-upper-triangle of the dense inputs (same pairs, same
// targets, same threshold codes)?  Host only, a few threads: one streaming pass over the upper
// triangle to count its finite cells, one gather per edge.  (A pair listed twice is not detected here;
// the caller cross-checks the count on the device.)
bool edges_are_the_matrix(const double* D, const int32_t* T, int n, const int32_t* ei, const int32_t* ej,
                          const double* ed, const int32_t* et, int64_t n_edges) {
  std::atomic<long long> finite{0};
  host_parallel((size_t)n, [&](size_t lo, size_t hi) {   // columns; work grows with j, close enough
    long long c = 0;
    for (size_t j = lo; j < hi; ++j) {
      const double* col = D + j * (size_t)n;
      for (size_t i = 0; i < j; ++i) c += std::isfinite(col[i]) ? 1 : 0;
    }
    finite.fetch_add(c);
  });
  if (finite.load() != (long long)n_edges) return false;
  std::atomic<bool> ok{true};
  host_parallel((size_t)n_edges, [&](size_t lo, size_t hi) {
    for (size_t e = lo; e < hi; ++e) {
      const int a = ei[e], b = ej[e];
      if (a < 0 || b <= a || b >= n) { ok.store(false); return; }
      const size_t cell = (size_t)a + (size_t)b * n;
      const int tc = T[cell], ec = et[e];
      const int tn = tc == 0 ? 0 : (tc == 1 ? 1 : -1), en = ec == 0 ? 0 : (ec == 1 ? 1 : -1);
      if (!(D[cell] == ed[e]) || !std::isfinite(ed[e]) || tn != en) { ok.store(false); return; }
    }
  });
  return ok.load();
}

}  // namespace

// =========================================================================================
// Session (slab path)
// =========================================================================================
struct topolow_session {
  int n = 0, dim = 0, udim = 0, row_begin = 0, row_end = 0, ld = 0;   // dim: coordinates the kernels carry, udim: the caller's ndim
  int precision = TOPOLOW_PRECISION_F32;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  // Convergence checks run beside the next iteration (slab schedule, own stream, not profiling):
  // the error pass and the controller go to `check_stream` and read the position buffer the
  // iteration ended in, which the stage kernels leave alone (`held`) until the next check.
  hipStream_t check_stream = nullptr;
  hipEvent_t ev_iter_done = nullptr, ev_check_done = nullptr;
  int held = -1;
  // A convergence check whose error pass has not been launched yet: when the NEXT iteration is a single
  // stage, that stage's kernel reduces the MAE of the positions it reads (exactly this check's positions)
  // on its way (slab_stage_pipe_kernel<..., ERR = true>) and the separate 2 N^2-byte pass is dropped.
  struct PendingCheck { bool active = false; int iter1 = 0; double k_after = 0.0; int buf = -1; bool beside = false; } pcheck;
  bool fuse_checks = true;      // TOPOLOW_FUSE_CHECKS=0: always the separate pass
  bool serial_checks = false;   // TOPOLOW_SERIAL_CHECKS=1: keep every check on the main stream

  // Session labels.  With a relabelling (topolow_session_set_relabel) the session stores point
  // perm[q] of the caller as its point q, so that a slab -- a run of consecutive session labels -- is
  // a random subset of the caller's points instead of a run of consecutive ones (which lie next to
  // each other on the reference's random-walk start, R/core.R:407-415, and often in the data's own
  // order).  Every entry point that takes or returns host arrays speaks the CALLER's labels.
  std::vector<int> perm, inv;      // session -> caller, caller -> session; empty = identity
  DevBuf<int> d_perm, d_inv;
  DevBuf<uint32_t> enc;
  DevBuf<float> gplus;
  DevBuf<unsigned char> rowflags;
  bool any_threshold = true;   // does any row of the block hold a ">" / "<" target?
  unsigned long long block_cells = 0;   // measured (ordered) cells of the block: 2 x the measured pairs of a whole problem
  int schedule = TOPOLOW_SCHEDULE_SLAB;   // SLAB, or GS = exact tile Gauss-Seidel (relax_tilegs.h)
  DevBuf<int> bperm;
  DevBuf<unsigned char> pos[3];   // stage ping-pong + the buffer a running check reads
  DevBuf<unsigned char> best;
  DevBuf<int> ei, ej;
  DevBuf<unsigned char> et;
  DevBuf<int8_t> ec;
  long long n_edges = 0;
  int n_parts = 0;
  bool dense_mae = false;   // edge list verified == measured cells of the encoded block
  bool list_is_block = false;   // ... the verification itself (dense_mae also wants fp32 and ndim <= 16)
  bool dense_parity = false;
  int dense_blocks = 0, dense_grid_x = 0, dense_grid_y = 0;
  DevBuf<double> part_sum;
  DevBuf<unsigned long long> part_cnt;
  DevBuf<RunState> state;
  RunState* mailbox = nullptr;      // pinned host memory
  RunState* mailbox_dev = nullptr;  // device alias of mailbox
  double* trace = nullptr;          // pinned: (iteration, MAE, k) of every check of the current run
  double* trace_dev = nullptr;
  int trace_cap = 0;

  // run parameters
  int n_iter = 0, check_freq = 3, window = 5, fixed_stages = 0;
  double k0 = 0, cooling = 0, c_rep = 0, eps = 1e-4;
  uint64_t seed = 0;
  // host-side progress
  int iters_enqueued = 0;
  double k_host = 0;
  int cur = 0;
  bool host_seen_stop = false;
  bool began = false;
  long long stage_launches = 0;
  std::deque<hipEvent_t> pending;
  std::vector<hipEvent_t> event_pool;
  // row-sharded engine (topolow_sessions_run_sharded): this block's view of the other blocks
  DevBuf<void*> push_tab[3];           // [b]: the other blocks' position buffer b (device pointers)
  int n_push = 0;
  DevBuf<double> rank_sum;             // one (sum, count) slot per block, written by every block
  DevBuf<unsigned long long> rank_cnt;
  DevBuf<double*> rsum_tab;            // every block's rank_sum / rank_cnt (self included)
  DevBuf<unsigned long long*> rcnt_tab;
  int n_ranks = 0, rank = 0;
  // Symmetric sweep (relax_symm.h): one-stage iterations of a whole-matrix fp32 session.
  struct SymState {
    bool allowed = false;          // TOPOLOW_SYMMETRIC=1 (session creation)
    int min_n = 0;                 // size gate (TOPOLOW_SYMMETRIC_MIN_N at session creation; default kSymMinPoints)
    bool ready = false;            // plan + tile-major copy built for the current block
    int npad = 0, tiles = 0, grid = 0, n_units = 0;   // npad = roundup(n, 64): whole 64-row tiles
    DevBuf<uint32_t> tenc;
    DevBuf<float> rec[2];
    int rec_cur = 0, rec_iter = -1;   // rec[rec_cur] holds the records of iteration rec_iter
    DevBuf<float> rowpart, colpart;
    DevBuf<double> rec64[2], rowpart64, colpart64;   // f64 sessions (relax_symm64.h)
    DevBuf<float> tdelta;          // ... exact target - decoded word per cell of tenc: the fused check's MAE is exact
    bool delta_ready = false;
    // multi-stage iterations (2, 4, 8 stages) as symmetric sweeps over the tiles of one stage each (relax_symm.h:
    // sym_rr_*; sym_rr_stage below): rr[log2 S] holds the S plans, built when an iteration first needs them
    struct StagePlan {
      DevBuf<SymUnit> units;
      DevBuf<SymRun> runs;
      DevBuf<int2> row_units;
      int n_units = 0;
    };
    std::vector<StagePlan> rr[4];
    bool whole = false;            // the buffers describe the whole triangle (not a segment): stage plans may be cut from it
    bool two_stage = true;         // TOPOLOW_SYMMETRIC_TWO_STAGE=0: multi-stage iterations stay on the row-owner kernel
    int rr_min_tiles = 5;          // tiles per resident wave a stage must have (TOPOLOW_SYMMETRIC_STAGE_MIN_TILES; tests: 0)
    DevBuf<SymUnit> units;
    DevBuf<SymRun> wave_first;     // per wave of the grid: its run of units (relax_symm.h: SymPlan::runs)
    DevBuf<int2> row_units;
    DevBuf<const uint32_t*> src_tab;   // the row blocks the tile-major copy is gathered from (one: the session's own)
    DevBuf<int> src_row0;
    // the sweep sharded over the row-block sessions of a run (relax_sharded_engine.h): this session's segment
    bool seg_ready = false;        // built for the run's current set of sessions
    bool seg_thr = false;          // some session of the run holds threshold targets: the classifying instance
    int seg_first = 0, seg_last = -1;   // tile-rows that hold a tile of the segment
    int seg_slots = 0;             // sessions of the run (slots of an inbox)
    int seg_slot = 0;              // the slot this session's folded partials go to
    bool seg_caller = false;       // built by topolow_session_symm_segment_build: one slot, one owner (this session's
                                   // own moves buffer = inbox), the caller sums it over the processes
    DevBuf<float> inbox;           // [seg_slots][npad][ndim]: every session's folded partials of this session's points
    DevBuf<float*> inbox_tab;      // every session's inbox (self included)
    DevBuf<int> own0;              // first row of every session, then n
    std::vector<const void*> seg_peers;   // the sessions the segment was built with (their encoded blocks)
  } sym;
  int fused_parts = 0;             // partial sums the last ERR launch wrote (stage kernel: workgroups; sweep: units)
  // profiling (roofline accounting)
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_stage, prof_stage_err, prof_check;   // _err: launches that also reduce the MAE
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_sym, prof_sym_err;                   // symmetric sweep + apply of one iteration

  size_t real_size() const { return precision == TOPOLOW_PRECISION_F64 ? 8 : 4; }
  int rows() const { return row_end - row_begin; }
  int pos_rows() const { return (n + 3) & ~3; }  // position buffers hold the padding points too

  ~topolow_session() {
    for (auto& pr : prof_stage) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto& pr : prof_stage_err) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto& pr : prof_check) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto& pr : prof_sym) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto& pr : prof_sym_err) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (hipEvent_t e : pending) (void)hipEventDestroy(e);
    for (hipEvent_t e : event_pool) (void)hipEventDestroy(e);
    if (mailbox) (void)hipHostFree(mailbox);
    if (trace) (void)hipHostFree(trace);
    if (ev_iter_done) (void)hipEventDestroy(ev_iter_done);
    if (ev_check_done) (void)hipEventDestroy(ev_check_done);
    if (check_stream) (void)hipStreamDestroy(check_stream);
    if (own_stream) (void)hipStreamDestroy(own_stream);
  }
};

namespace {

// ---- kernel dispatch over (DIM, real) ----------------------------------------------------
#define TL_DISPATCH_DIM(dim, FN, ...)                 \
  switch (dim) {                                      \
    case 1: FN<1>(__VA_ARGS__); break;                \
    case 2: FN<2>(__VA_ARGS__); break;                \
    case 3: FN<3>(__VA_ARGS__); break;                \
    case 4: FN<4>(__VA_ARGS__); break;                \
    case 5: FN<5>(__VA_ARGS__); break;                \
    case 6: FN<6>(__VA_ARGS__); break;                \
    case 7: FN<7>(__VA_ARGS__); break;                \
    case 8: FN<8>(__VA_ARGS__); break;                \
    case 9: FN<9>(__VA_ARGS__); break;                \
    case 10: FN<10>(__VA_ARGS__); break;              \
    case 12: FN<12>(__VA_ARGS__); break;              \
    case 16: FN<16>(__VA_ARGS__); break;              \
    case 32: FN<32>(__VA_ARGS__); break;              \
    case 64: FN<64>(__VA_ARGS__); break;              \
    default: throw HipError{TOPOLOW_ERR_UNSUPPORTED, "ndim must be between 1 and 64"}; \
  }

struct ProfScope {
  topolow_session* s;
  std::vector<std::pair<hipEvent_t, hipEvent_t>>* dst;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(topolow_session* s_, std::vector<std::pair<hipEvent_t, hipEvent_t>>* d) : s(s_), dst(d) {
    if (!s->profiling) return;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    HIP_TRY(hipEventRecord(a, s->stream));
  }
  ~ProfScope() {
    if (!a) return;
    (void)hipEventRecord(b, s->stream);
    dst->emplace_back(a, b);
  }
};

// Stage-kernel geometry.  Production: 256 threads, 2 rows per wave (8 rows per workgroup), chunk
// size picked per ndim (PipeGeom), falling issue priority, register budget for 5 waves per SIMD --
// the fastest of the variants measured on MI355X.  A build with
// -DTOPOLOW_TUNING also instantiates other chunk sizes / budgets (TOPOLOW_SLAB_VARIANT=<n>) and
// the per-workgroup time stamps (TOPOLOW_WG_STAMPS=<file>).
using CfgProd = StageCfg<256, 2, 0, 1, 5>;
#ifdef TOPOLOW_TUNING
// TOPOLOW_WG_STAMPS=<file>: the stage kernel's workgroups stamp their start/end times; the last
// launch's stamps are written to <file> as text when the session is destroyed.
struct WgStamps {
  unsigned long long* dev = nullptr;
  int blocks = 0;
  void arm(int nblocks) {
    if (dev != nullptr || getenv("TOPOLOW_WG_STAMPS") == nullptr) return;
    blocks = nblocks;
    if (hipMalloc(&dev, sizeof(unsigned long long) * 2 * nblocks) != hipSuccess) { dev = nullptr; return; }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_stamps), &dev, sizeof(dev));
  }
  void dump() {
    if (dev == nullptr) return;
    std::vector<unsigned long long> h(2 * (size_t)blocks);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (FILE* f = fopen(getenv("TOPOLOW_WG_STAMPS"), "w")) {
      for (int b = 0; b < blocks; ++b) fprintf(f, "%llu %llu\n", h[2 * b], h[2 * b + 1]);
      fclose(f);
    }
    unsigned long long* none = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_stamps), &none, sizeof(none));
    (void)hipFree(dev);
    dev = nullptr;
  }
};
WgStamps g_stamps;

int slab_variant() {
  static int v = [] {
    const char* e = getenv("TOPOLOW_SLAB_VARIANT");
    return e ? atoi(e) : -1;
  }();
  return v;
}
#endif

template <int DIM, typename real, typename CFG>
void launch_stage_pipe(topolow_session* s, const void* pin, void* pout, RunState* st,
                       SlabRanges rg, int iter1, double k, const void* push, int n_push, bool err) {
  const int blocks = (s->rows() + CFG::ROWS - 1) / CFG::ROWS;
#ifdef TOPOLOW_TUNING
  g_stamps.arm(blocks);
#endif
  auto launch = [&](auto kern, int which) {
    // falling issue priority only when every workgroup of the grid is resident at once
    static int resident[2] = {0, 0};     // workgroups of this (ndim, precision) kernel a device holds
    if (resident[which] == 0) {          // [0] threshold-free instance, [1] threshold-carrying one
      hipDeviceProp_t prop;
      HIP_TRY(hipGetDeviceProperties(&prop, s->device));
      int per_cu = 0;
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, CFG::THREADS, 0));
      resident[which] = std::max(1, per_cu) * prop.multiProcessorCount;
    }
    const int falling = blocks <= resident[which] ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(CFG::THREADS), 0, s->stream, s->enc.p, s->ld,
                       s->row_begin, s->row_end, s->n, (const real*)pin, (real*)pout, s->gplus.p,
                       s->rowflags.p, st, rg, iter1, k, s->c_rep, falling, (real* const*)push, n_push,
                       s->part_sum.p, s->part_cnt.p, s->block_cells);
  };
  if constexpr (sizeof(real) == 4) {
    if (err) {   // the launch also reduces the convergence MAE of the positions it reads (one-stage iterations)
      if (s->any_threshold) {
        constexpr int kThrWavesE = DIM >= 16 ? 1 : (DIM >= 12 ? 2 : (DIM >= 9 ? 3 : (DIM >= 5 ? 4 : 5)));   // the error sums cost registers: 4 waves from ndim 5
        using CfgThrE = StageCfg<CFG::THREADS, CFG::RPW, CFG::CHUNK, CFG::PRIO,
                                 CFG::MINWAVES < kThrWavesE ? CFG::MINWAVES : kThrWavesE>;
        launch(&slab_stage_pipe_kernel<DIM, real, CfgThrE, true, true>, 1);
      } else {
        launch(&slab_stage_pipe_kernel<DIM, real, CFG, false, true>, 0);
      }
      return;
    }
  }
  if (s->any_threshold) {
    // The instance that also carries the ">" / "<" classification needs more registers: from
    // ndim 7 on it would spill inside the pair loop at a 5-wave budget (9x slower at ndim 10), so
    // it is built for 4 waves per SIMD there and for 3 from ndim 9 (tests/test_capi.py checks that no
    // instantiation uses scratch).
    constexpr int kThrWaves = sizeof(real) == 4 ? (DIM >= 16 ? 1 : (DIM >= 12 ? 2 : (DIM >= 9 ? 3 : (DIM >= 7 ? 4 : 5)))) : 1;
    using CfgThr = StageCfg<CFG::THREADS, CFG::RPW, CFG::CHUNK, CFG::PRIO,
                            CFG::MINWAVES < kThrWaves ? CFG::MINWAVES : kThrWaves>;
    launch(&slab_stage_pipe_kernel<DIM, real, CfgThr, true>, 1);
  } else {
    launch(&slab_stage_pipe_kernel<DIM, real, CFG, false>, 0);
  }
}

template <int DIM>
void launch_stage(topolow_session* s, const void* pin, void* pout, RunState* st, SlabRanges rg,
                  int iter1, double k, const void* push = nullptr, int n_push = 0, bool err = false) {
  if (s->rows() <= 0) return;
  ProfScope prof(s, err ? &s->prof_stage_err : &s->prof_stage);
  if constexpr (DIM > kMaxTunedDim) {      // wide embeddings: the plain stage kernel (never an ERR launch: no dense MAE there)
    const int blocks = (s->rows() + kWaves - 1) / kWaves;
    if (s->precision == TOPOLOW_PRECISION_F64)
      hipLaunchKernelGGL((slab_stage_wide_kernel<DIM, double>), dim3(blocks), dim3(kThreads), 0, s->stream, s->enc.p, s->ld,
                         s->row_begin, s->row_end, s->n, (const double*)pin, (double*)pout, s->gplus.p, st, rg, iter1, k,
                         s->c_rep, (double* const*)push, n_push);
    else
      hipLaunchKernelGGL((slab_stage_wide_kernel<DIM, float>), dim3(blocks), dim3(kThreads), 0, s->stream, s->enc.p, s->ld,
                         s->row_begin, s->row_end, s->n, (const float*)pin, (float*)pout, s->gplus.p, st, rg, iter1, k,
                         s->c_rep, (float* const*)push, n_push);
    HIP_TRY(hipGetLastError());
    s->stage_launches += 1;
    return;
  } else
  if (s->precision == TOPOLOW_PRECISION_F64) {
    launch_stage_pipe<DIM, double, StageCfg<256, 2, 0, 1>>(s, pin, pout, st, rg, iter1, k, push, n_push, false);
  } else {
#ifdef TOPOLOW_TUNING
    switch (slab_variant()) {
      case 20: launch_stage_pipe<DIM, float, StageCfg<256, 2, 512, 0, 5>>(s, pin, pout, st, rg, iter1, k, push, n_push, err); break;
      case 21: launch_stage_pipe<DIM, float, StageCfg<256, 2, 768, 1, 5>>(s, pin, pout, st, rg, iter1, k, push, n_push, err); break;
      case 22: launch_stage_pipe<DIM, float, StageCfg<256, 2, 512, 1, 4>>(s, pin, pout, st, rg, iter1, k, push, n_push, err); break;
      case 23: launch_stage_pipe<DIM, float, StageCfg<256, 2, 256, 1, 5>>(s, pin, pout, st, rg, iter1, k, push, n_push, err); break;
      case 24: launch_stage_pipe<DIM, float, StageCfg<256, 2, 512, 1, 7>>(s, pin, pout, st, rg, iter1, k, push, n_push, err); break;
      case 25: launch_stage_pipe<DIM, float, StageCfg<512, 2, 768, 0, 4>>(s, pin, pout, st, rg, iter1, k, push, n_push, err); break;
      case 27: launch_stage_pipe<DIM, float, StageCfg<256, 2, 1024, 1, 4>>(s, pin, pout, st, rg, iter1, k, push, n_push, err); break;
      case 28: launch_stage_pipe<DIM, float, StageCfg<512, 2, 512, 0, 4>>(s, pin, pout, st, rg, iter1, k, push, n_push, err); break;
      default: launch_stage_pipe<DIM, float, CfgProd>(s, pin, pout, st, rg, iter1, k, push, n_push, err); break;
    }
#else
    // register budget: 5 waves per SIMD up to ndim 10 (CfgProd), 3 at 12 coordinates, 2 at 16
    using CfgDim = StageCfg<CfgProd::THREADS, CfgProd::RPW, CfgProd::CHUNK, CfgProd::PRIO,
                            DIM <= 10 ? CfgProd::MINWAVES : (DIM <= 12 ? 3 : 2)>;
    launch_stage_pipe<DIM, float, CfgDim>(s, pin, pout, st, rg, iter1, k, push, n_push, err);
#endif
  }
  HIP_TRY(hipGetLastError());
  s->stage_launches += 1;
}

using ErrCfg = StageCfg<256, 2, 1024>;

template <int DIM, typename real>
void launch_dense_error(topolow_session* s, const void* pos, const RunState* st) {
  const size_t lds = sizeof(real) * DIM * ErrCfg::CHUNK;
  const dim3 grid(s->dense_grid_x, s->dense_grid_y);
  if (s->dense_parity) {
    hipLaunchKernelGGL((dense_error_kernel<DIM, real, ErrCfg, true>), grid, dim3(ErrCfg::THREADS), lds,
                       s->stream, s->enc.p, s->ld, s->row_begin, s->row_end, s->n, (const real*)pos,
                       s->rowflags.p, s->part_sum.p, s->part_cnt.p, st);
  } else {
    hipLaunchKernelGGL((dense_error_kernel<DIM, real, ErrCfg, false>), grid, dim3(ErrCfg::THREADS), lds,
                       s->stream, s->enc.p, s->ld, s->row_begin, s->row_end, s->n, (const real*)pos,
                       s->rowflags.p, s->part_sum.p, s->part_cnt.p, st);
  }
}

// Number of partial sums the last launched error kernel produced.
int error_parts(const topolow_session* s) { return s->dense_mae ? s->dense_blocks : s->n_parts; }

template <int DIM>
void launch_edge_error(topolow_session* s, const void* pos, const RunState* st) {
  if (s->dense_mae) {
    if constexpr (DIM > kMaxTunedDim) {
      throw HipError{TOPOLOW_ERR_UNSUPPORTED, "dense MAE pass: ndim"};   // (wide sessions never set dense_mae)
    } else {
      if (s->precision == TOPOLOW_PRECISION_F64) launch_dense_error<DIM, double>(s, pos, st);
      else launch_dense_error<DIM, float>(s, pos, st);
    }
  } else if (s->precision == TOPOLOW_PRECISION_F64) {
    hipLaunchKernelGGL((edge_error_kernel<DIM, double, double>), dim3(s->n_parts),
                       dim3(kThreads), 0, s->stream, (const double*)pos, s->ei.p, s->ej.p,
                       (const double*)s->et.p, s->ec.p, s->n_edges, s->part_sum.p,
                       s->part_cnt.p, st);
  } else {
    hipLaunchKernelGGL((edge_error_kernel<DIM, float, float>), dim3(s->n_parts), dim3(kThreads),
                       0, s->stream, (const float*)pos, s->ei.p, s->ej.p, (const float*)s->et.p,
                       s->ec.p, s->n_edges, s->part_sum.p, s->part_cnt.p, st);
  }
  HIP_TRY(hipGetLastError());
}

// psum / pcnt / nparts: the partials to reduce (default: the block's own error-kernel partials)
void launch_controller(topolow_session* s, const void* pos, int iter1, double k_after,
                       const double* psum = nullptr, const unsigned long long* pcnt = nullptr, int nparts = 0,
                       const double* total2 = nullptr) {
  const long long nv = (long long)s->n * s->dim;
  if (psum == nullptr && total2 == nullptr) { psum = s->part_sum.p; pcnt = s->part_cnt.p; nparts = error_parts(s); }
  if (s->precision == TOPOLOW_PRECISION_F64) {
    hipLaunchKernelGGL((controller_kernel<double>), dim3(1), dim3(kCtlThreads), 0, s->stream,
                       s->state.p, s->mailbox_dev, psum, pcnt, nparts,
                       (const double*)pos, (double*)s->best.p, nv, iter1, k_after, s->trace_dev, s->trace_cap, total2);
  } else {
    hipLaunchKernelGGL((controller_kernel<float>), dim3(1), dim3(kCtlThreads), 0, s->stream,
                       s->state.p, s->mailbox_dev, psum, pcnt, nparts,
                       (const float*)pos, (float*)s->best.p, nv, iter1, k_after, s->trace_dev, s->trace_cap, total2);
  }
  HIP_TRY(hipGetLastError());
}

// positions host (n x dim f64 column-major) <-> device (n x dim row-major, session precision)
void upload_positions(topolow_session* s, const double* host_colmajor, void* dst) {
  // rows [n, roundup4(n)) are the phantom points of the padding columns (relax_common.h)
  const size_t nv = (size_t)s->pos_rows() * s->dim;
  if (s->precision == TOPOLOW_PRECISION_F64) {
    std::vector<double> tmp(nv, 0.0);
    for (int i = 0; i < s->n; ++i) {
      const int o = s->perm.empty() ? i : s->perm[i];
      for (int d = 0; d < s->udim; ++d) tmp[(size_t)i * s->dim + d] = host_colmajor[o + (size_t)d * s->n];
    }
    for (int i = s->n; i < s->pos_rows(); ++i) tmp[(size_t)i * s->dim] = kFarF64;
    HIP_TRY(hipMemcpyAsync(dst, tmp.data(), nv * 8, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  } else {
    std::vector<float> tmp(nv, 0.0f);
    for (int i = 0; i < s->n; ++i) {
      const int o = s->perm.empty() ? i : s->perm[i];
      for (int d = 0; d < s->udim; ++d)
        tmp[(size_t)i * s->dim + d] = (float)host_colmajor[o + (size_t)d * s->n];
    }
    for (int i = s->n; i < s->pos_rows(); ++i) tmp[(size_t)i * s->dim] = kFarF32;
    HIP_TRY(hipMemcpyAsync(dst, tmp.data(), nv * 4, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  }
}

void download_positions(topolow_session* s, const void* src, double* host_colmajor) {
  const size_t nv = (size_t)s->n * s->dim;
  if (s->precision == TOPOLOW_PRECISION_F64) {
    std::vector<double> tmp(nv);
    HIP_TRY(hipMemcpyAsync(tmp.data(), src, nv * 8, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (int i = 0; i < s->n; ++i) {
      const int o = s->perm.empty() ? i : s->perm[i];
      for (int d = 0; d < s->udim; ++d) host_colmajor[o + (size_t)d * s->n] = tmp[(size_t)i * s->dim + d];
    }
  } else {
    std::vector<float> tmp(nv);
    HIP_TRY(hipMemcpyAsync(tmp.data(), src, nv * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (int i = 0; i < s->n; ++i) {
      const int o = s->perm.empty() ? i : s->perm[i];
      for (int d = 0; d < s->udim; ++d)
        host_colmajor[o + (size_t)d * s->n] = (double)tmp[(size_t)i * s->dim + d];
    }
  }
}

void compute_row_flags(topolow_session* s) {
  s->sym.ready = false;   // the encoded block changed: the symmetric sweep's copy is rebuilt on first use
  s->sym.seg_ready = false;
  s->rowflags.alloc(s->rows());
  DevBuf<unsigned long long> measured;
  measured.alloc(1);
  HIP_TRY(hipMemsetAsync(measured.p, 0, sizeof(unsigned long long), s->stream));
  hipLaunchKernelGGL(row_flags_kernel, dim3(s->rows()), dim3(kThreads), 0, s->stream, s->enc.p,
                     s->rows(), s->ld, s->rowflags.p, measured.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpy(&s->block_cells, measured.p, sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<unsigned char> h(s->rows());
  HIP_TRY(hipMemcpy(h.data(), s->rowflags.p, h.size(), hipMemcpyDeviceToHost));
  s->any_threshold = false;
  for (unsigned char f : h) s->any_threshold = s->any_threshold || f != 0;
}

void upload_degrees(topolow_session* s, const int32_t* degrees) {
  std::vector<float> g(s->n);
  for (int i = 0; i < s->n; ++i) g[i] = (float)degrees[s->perm.empty() ? i : s->perm[i]] + 1.0f;  // reference :137-140
  s->gplus.alloc(s->n);
  HIP_TRY(hipMemcpy(s->gplus.p, g.data(), (size_t)s->n * 4, hipMemcpyHostToDevice));
}

hipEvent_t take_event(topolow_session* s) {
  if (!s->event_pool.empty()) {
    hipEvent_t e = s->event_pool.back();
    s->event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return e;
}

void poll_checks(topolow_session* s, size_t keep_in_flight) {
  while (s->pending.size() > keep_in_flight) {
    hipEvent_t e = s->pending.front();
    HIP_TRY(hipEventSynchronize(e));
    s->pending.pop_front();
    s->event_pool.push_back(e);
    if (s->mailbox->stopped) s->host_seen_stop = true;
  }
  // opportunistic: anything already finished
  while (!s->pending.empty() && hipEventQuery(s->pending.front()) == hipSuccess) {
    s->event_pool.push_back(s->pending.front());
    s->pending.pop_front();
    if (s->mailbox->stopped) s->host_seen_stop = true;
  }
}

// Launch helpers take the stream from the session: run a few of them on another one.
struct StreamScope {
  topolow_session* s;
  hipStream_t saved;
  StreamScope(topolow_session* s_, hipStream_t on) : s(s_), saved(s_->stream) { s->stream = on; }
  ~StreamScope() { s->stream = saved; }
};

// ---- exact tile Gauss-Seidel iteration (relax_tilegs.h): in place on `pos` -------------------
template <int DIM>
void launch_tilegs_iteration(topolow_session* s, void* pos, int iter, double k) {
  if constexpr (DIM > kMaxTunedDim) {
    throw HipError{TOPOLOW_ERR_UNSUPPORTED, "schedule gs: ndim must be between 1 and 16 (wider embeddings run the slab schedule)"};
  } else {
  const int nb = (s->n + kTile - 1) / kTile;
  if ((int)s->bperm.n < nb) s->bperm.alloc(nb);
  hipLaunchKernelGGL(tilegs_perm_kernel, dim3(1), dim3(256), 0, s->stream, s->seed, iter, nb, s->bperm.p,
                     s->state.p);
  const int M = nb + (nb & 1), m1 = M - 1;
  for (int r = 0; r < m1 && nb >= 2; ++r) {
    if (s->precision == TOPOLOW_PRECISION_F64)
      hipLaunchKernelGGL((tilegs_pair_kernel<DIM, double>), dim3(M / 2), dim3(kTile), 0, s->stream, s->enc.p,
                         s->ld, s->n, (double*)pos, s->gplus.p, s->bperm.p, nb, r, s->state.p, s->seed, iter, k,
                         s->c_rep);
    else
      hipLaunchKernelGGL((tilegs_pair_kernel<DIM, float>), dim3(M / 2), dim3(kTile), 0, s->stream, s->enc.p,
                         s->ld, s->n, (float*)pos, s->gplus.p, s->bperm.p, nb, r, s->state.p, s->seed, iter, k,
                         s->c_rep);
    s->stage_launches += 1;
  }
  if (s->precision == TOPOLOW_PRECISION_F64)
    hipLaunchKernelGGL((tilegs_intra_kernel<DIM, double>), dim3(nb), dim3(kTile), 0, s->stream, s->enc.p, s->ld,
                       s->n, (double*)pos, s->gplus.p, s->state.p, s->seed, iter, k, s->c_rep);
  else
    hipLaunchKernelGGL((tilegs_intra_kernel<DIM, float>), dim3(nb), dim3(kTile), 0, s->stream, s->enc.p, s->ld,
                       s->n, (float*)pos, s->gplus.p, s->state.p, s->seed, iter, k, s->c_rep);
  s->stage_launches += 1;
  HIP_TRY(hipGetLastError());
  }
}

void launch_tilegs_finite(topolow_session* s, const void* pos, int iter1) {
  const long long nv = (long long)s->n * s->dim;
  if (s->precision == TOPOLOW_PRECISION_F64)
    hipLaunchKernelGGL((tilegs_finite_kernel<double>), dim3(64), dim3(256), 0, s->stream, (const double*)pos, nv,
                       s->state.p, iter1);
  else
    hipLaunchKernelGGL((tilegs_finite_kernel<float>), dim3(64), dim3(256), 0, s->stream, (const float*)pos, nv,
                       s->state.p, iter1);
  HIP_TRY(hipGetLastError());
}

// ---- symmetric sweep (relax_symm.h) ----------------------------------------------------------
// Which sessions take it: the whole matrix on one GPU (no row block, nothing to push), fp32 slab schedule, ndim
// 2..6 (the register-tiled kernel keeps eight rows' coordinates, constants and sums in VGPRs: 20 x ndim + 16 of
// them), at least kSymMinPoints points.  Row-sharded runs shard it over their sessions (sym_sharded_*, below).
// Everything else -- and every multi-stage iteration -- stays on the row-owner stage kernel.
template <int DIM> constexpr bool kSymDim = DIM >= 2 && DIM <= 6;
constexpr int kSymMinPoints = 7168;   // below ~7000 points a resident wave gets fewer than 8 tiles and the row-owner sweep is faster (tests/study/symm_crossover.py)

bool sym_eligible(const topolow_session* s) {
  return s->sym.allowed && s->schedule == TOPOLOW_SCHEDULE_SLAB &&
         (s->precision == TOPOLOW_PRECISION_F32 || s->precision == TOPOLOW_PRECISION_F64) &&
         s->row_begin == 0 && s->row_end == s->n && s->n_push == 0 && s->dim >= 2 && s->dim <= 6 && s->n >= s->sym.min_n &&
         s->dim == s->udim;
}

// Builds the plan, the tile-major copy and the partial buffers of tiles [t0, t1) of the upper triangle (t1 < 0: all of
// it) from the row blocks `src` (device pointers of their encoded blocks, first rows in row0[0..n_src]).
template <int DIM>
void sym_build(topolow_session* s, const std::vector<const uint32_t*>& src, const std::vector<int>& row0, bool any_thr,
               long long t0, long long t1) {
  if constexpr (!kSymDim<DIM>) {
    throw HipError{TOPOLOW_ERR_UNSUPPORTED, "symmetric sweep: ndim"};
  } else {
    auto& y = s->sym;
    y.npad = (s->n + kSymRows - 1) & ~(kSymRows - 1);
    const int TR = y.npad / kSymRows, TC = y.npad / kSymCols;
    if (t1 < 0) t1 = (long long)TR * (TR + 1);
    y.tiles = (int)(t1 - t0);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, s->device));
    int occ = 1 << 30;   // the session's two instances (plain, ERR) share one plan: the smaller occupancy decides the grid
    auto probe = [&](auto kern) {
      int per_cu = 0;
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 64 * kSymWaves, 0));
      occ = std::min(occ, std::max(1, per_cu));
    };
    const bool f64 = s->precision == TOPOLOW_PRECISION_F64;
    if (f64) {
      if (any_thr) { probe(&symm64_sweep_kernel<DIM, true, false>); probe(&symm64_sweep_kernel<DIM, true, true>); }
      else { probe(&symm64_sweep_kernel<DIM, false, false>); probe(&symm64_sweep_kernel<DIM, false, true>); }
    } else if (any_thr) {
      probe(&symm_sweep_kernel<DIM, true, false>);
      probe(&symm_sweep_kernel<DIM, true, true>);
    } else {
      probe(&symm_sweep_kernel<DIM, false, false>);
      probe(&symm_sweep_kernel<DIM, false, true>);
    }
    y.grid = occ * prop.multiProcessorCount;
    const SymPlan plan = relax_symm_plan(y.npad, y.grid * kSymWaves, t0, t1, &y.seg_first, &y.seg_last);
    y.n_units = (int)plan.units.size();
    y.units.alloc(std::max<size_t>(plan.units.size(), 1));
    const std::vector<SymRun> runs = plan.runs();
    y.wave_first.alloc(runs.size());
    y.row_units.alloc(std::max<size_t>(plan.row_units.size(), 1));
    if (!plan.units.empty())
      HIP_TRY(hipMemcpy(y.units.p, plan.units.data(), plan.units.size() * sizeof(SymUnit), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(y.wave_first.p, runs.data(), runs.size() * sizeof(SymRun), hipMemcpyHostToDevice));
    if (!plan.row_units.empty())
      HIP_TRY(hipMemcpy(y.row_units.p, plan.row_units.data(), plan.row_units.size() * sizeof(int2), hipMemcpyHostToDevice));
    y.whole = t0 == 0 && (long long)y.tiles == (long long)TR * (TR + 1);
    for (auto& v : y.rr) v.clear();
    // a stage plan has at most one unit per wave and one more per tile-row and interval end
    const int max_units = std::max(y.n_units, y.grid * kSymWaves + 2 * TR + 8);
    y.src_tab.alloc(src.size());
    y.src_row0.alloc(row0.size());
    HIP_TRY(hipMemcpy(y.src_tab.p, src.data(), src.size() * sizeof(const uint32_t*), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(y.src_row0.p, row0.data(), row0.size() * sizeof(int), hipMemcpyHostToDevice));
    y.tenc.alloc((size_t)std::max(y.tiles, 1) * kSymTileWords);
    if (y.tiles > 0)
      hipLaunchKernelGGL(symm_tiles_kernel, dim3(y.tiles), dim3(256), 0, s->stream, y.src_tab.p, y.src_row0.p, (int)src.size(),
                         s->ld, y.tenc.p, TC, t0);
    HIP_TRY(hipGetLastError());
    const int seg_rows = y.seg_last >= y.seg_first ? y.seg_last - y.seg_first + 1 : 1;
    if (f64) {
      for (auto& r : y.rec64) r.alloc((size_t)y.npad * SymRec64<DIM>::W);
      y.rowpart64.alloc((size_t)std::max(max_units, 1) * kSymRows * DIM);
      y.colpart64.alloc((size_t)seg_rows * y.npad * DIM);
      HIP_TRY(hipMemsetAsync(y.colpart64.p, 0, (size_t)seg_rows * y.npad * DIM * sizeof(double), s->stream));
      // the fused check needs the edge list to BE the block's measured cells and to be on the device in f64
      y.delta_ready = false;
      if (s->list_is_block && !s->dense_mae && s->n_edges > 0 && t0 == 0 && (long long)y.tiles == (long long)TR * (TR + 1)) {
        y.tdelta.alloc((size_t)y.tiles * kSymTileWords);
        HIP_TRY(hipMemsetAsync(y.tdelta.p, 0, (size_t)y.tiles * kSymTileWords * sizeof(float), s->stream));
        hipLaunchKernelGGL(symm64_delta_kernel, dim3(2048), dim3(256), 0, s->stream, s->ei.p, s->ej.p, (const double*)s->et.p,
                           s->ec.p, (long long)s->n_edges, y.tdelta.p, TC, s->n);
        HIP_TRY(hipGetLastError());
        y.delta_ready = true;
      }
    } else {
      constexpr int W = SymRec<DIM>::W;
      for (auto& r : y.rec) r.alloc((size_t)y.npad * W);
      y.rowpart.alloc((size_t)std::max(max_units, 1) * kSymRows * DIM);
      y.colpart.alloc((size_t)seg_rows * y.npad * DIM);
      // (a segment's first and last tile-row are partial: the columns its tiles never reach must read as zero)
      HIP_TRY(hipMemsetAsync(y.colpart.p, 0, (size_t)seg_rows * y.npad * DIM * sizeof(float), s->stream));
    }
    if ((size_t)y.n_units > s->part_sum.n) {   // error partials: one per unit
      HIP_TRY(hipStreamSynchronize(s->stream));
      HIP_TRY(hipStreamSynchronize(s->check_stream));
      s->part_sum.alloc(y.n_units);
      s->part_cnt.alloc(y.n_units);
    }
    y.rec_iter = -1;
  }
}

template <int DIM>
void sym_prepare(topolow_session* s) {
  sym_build<DIM>(s, {s->enc.p}, {0, s->rows()}, s->any_threshold, 0, -1);
  s->sym.ready = true;
  s->sym.seg_ready = false;
}

// Builds the sweep's buffers on first use.  They cost device memory (half the encoded block again, plus the
// partials): if the device cannot give it, the session simply keeps the row-owner sweep.
bool sym_available(topolow_session* s) {
  if (s->sym.ready) return true;
  try {
    TL_DISPATCH_DIM(s->dim, sym_prepare, s);
  } catch (const HipError&) {
    (void)hipGetLastError();
    auto& y = s->sym;
    y.tenc.release(); y.rec[0].release(); y.rec[1].release(); y.rowpart.release(); y.colpart.release();
    y.rec64[0].release(); y.rec64[1].release(); y.rowpart64.release(); y.colpart64.release(); y.tdelta.release();
    y.delta_ready = false;
    y.units.release(); y.wave_first.release(); y.row_units.release();
    y.ready = false;
    y.allowed = false;
  }
  return s->sym.ready;
}

// One one-stage iteration: records of this iteration (built from `pin` unless the previous iteration's apply left
// them), sweep, apply into `pout` (and the next iteration's records).  err: the sweep also reduces the pending
// check's MAE (positions read = that check's positions).
template <int DIM>
void sym_iteration(topolow_session* s, const void* pin, void* pout, int iter, double k, bool err) {
  if constexpr (!kSymDim<DIM>) {
    throw HipError{TOPOLOW_ERR_UNSUPPORTED, "symmetric sweep: ndim"};
  } else {
    auto& y = s->sym;
    ProfScope prof(s, err ? &s->prof_sym_err : &s->prof_sym);
    const int TC = y.npad / kSymCols;
    if (s->precision == TOPOLOW_PRECISION_F64) {   // relax_symm64.h: same plan, tiles and partial layout, everything else in f64
      if (err && !y.delta_ready) throw HipError{TOPOLOW_ERR_UNSUPPORTED, "symmetric sweep (f64): no delta tiles for a fused check"};
      if (y.rec_iter != iter) {
        for (int b = 0; b < 2; ++b)
          hipLaunchKernelGGL(symm64_records_kernel<DIM>, dim3((y.npad + 255) / 256), dim3(256), 0, s->stream, (const double*)pin,
                             s->gplus.p, y.rec64[b].p, s->n, y.npad, k, s->c_rep);
        y.rec_cur = 0;
      }
      const double* rec = y.rec64[y.rec_cur].p;
      double* rec_next = y.rec64[y.rec_cur ^ 1].p;
      auto sweep64 = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(y.grid), dim3(64 * kSymWaves), 0, s->stream, y.tenc.p, rec, y.units.p, y.wave_first.p,
                           y.rowpart64.p, y.colpart64.p, y.npad, s->state.p, 0, y.tdelta.p, s->part_sum.p, s->part_cnt.p,
                           s->block_cells);
      };
      if (s->any_threshold) {
        if (err) sweep64(&symm64_sweep_kernel<DIM, true, true>); else sweep64(&symm64_sweep_kernel<DIM, true, false>);
      } else {
        if (err) sweep64(&symm64_sweep_kernel<DIM, false, true>); else sweep64(&symm64_sweep_kernel<DIM, false, false>);
      }
      hipLaunchKernelGGL(symm64_apply_kernel<DIM>, dim3(TC), dim3(32 * kSymApplyParts), 0, s->stream, rec, rec_next, (double*)pout,
                         s->gplus.p, y.rowpart64.p, y.colpart64.p, y.row_units.p, s->n, y.npad, k * (1.0 - s->cooling), s->c_rep,
                         iter + 1, s->state.p);
      HIP_TRY(hipGetLastError());
      y.rec_cur ^= 1;
      y.rec_iter = iter + 1;
      if (err) s->fused_parts = y.n_units;
      s->stage_launches += 1;
      return;
    }
    if (y.rec_iter != iter) {
      for (int b = 0; b < 2; ++b)   // both buffers need the phantom records; the second one's points are overwritten by the apply
        hipLaunchKernelGGL(symm_records_kernel<DIM>, dim3((y.npad + 255) / 256), dim3(256), 0, s->stream, (const float*)pin,
                           s->gplus.p, y.rec[b].p, s->n, y.npad, k, s->c_rep);
      y.rec_cur = 0;
    }
    const float* rec = y.rec[y.rec_cur].p;
    float* rec_next = y.rec[y.rec_cur ^ 1].p;
    auto sweep = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(y.grid), dim3(64 * kSymWaves), 0, s->stream, y.tenc.p, rec, y.units.p, y.wave_first.p,
                         y.rowpart.p, y.colpart.p, y.npad, s->state.p, s->part_sum.p, s->part_cnt.p,
                         s->block_cells, 0);
    };
    if (s->any_threshold) {
      if (err) sweep(&symm_sweep_kernel<DIM, true, true>); else sweep(&symm_sweep_kernel<DIM, true, false>);
    } else {
      if (err) sweep(&symm_sweep_kernel<DIM, false, true>); else sweep(&symm_sweep_kernel<DIM, false, false>);
    }
    const double k_next = k * (1.0 - s->cooling);
    hipLaunchKernelGGL(symm_apply_kernel<DIM>, dim3(TC), dim3(32 * kSymApplyParts), 0, s->stream, rec, rec_next, (float*)pout,
                       s->gplus.p, y.rowpart.p, y.colpart.p, y.row_units.p, s->n, y.npad, k_next, s->c_rep, iter + 1,
                       s->state.p);
    HIP_TRY(hipGetLastError());
    y.rec_cur ^= 1;
    y.rec_iter = iter + 1;
    if (err) s->fused_parts = y.n_units;
    s->stage_launches += 1;
  }
}

// Multi-stage iterations on the symmetric sweep (relax_symm.h: sym_rr_*).  The row-owner form of an S-stage iteration gives
// every point its halves of the pairs with one slab of the points per stage, the same slab for everybody; this form
// splits the PAIRS: in stage st every slab a meets its partner slab (st - a) mod S -- every point still meets one slab
// of partners per stage (the stability argument, k / S, is the same), both ends of a pair move in the same stage as in
// the reference (src/optimization.cpp:245-281), and every pair is evaluated once per iteration instead of twice.  The
// order of the stages is drawn per iteration.  S = 2, 4, 8; sixteen stages (the unfolding phase, k > 24) stay row-owner.
bool sym_rr_stages_ok(int S) { return S == 2 || S == 4 || S == 8; }

bool sym_rr_available(topolow_session* s, int S) {
  auto& y = s->sym;
  if (!sym_rr_stages_ok(S) || !y.whole) return false;
  const int TR = y.npad / kSymRows;
  if (TR < 2 * S) return false;
  // a stage must give a resident wave about five tiles: below that a wave's prologue and epilogue and the apply kernel
  // cost what the halved pair count saves, or more (config 3, tests/study/two_stage_ab.py: two stages, 6 tiles per wave:
  // a run with k0 = 6 26.1 against 27.6 ms on the row-owner stages; four stages, 3 tiles: +25 us per iteration; eight
  // stages, 1.5 tiles: 16.0 against 13.0 ms per run with k0 = 20)
  if ((long long)TR * (TR + 1) / S < (long long)y.rr_min_tiles * y.grid * kSymWaves) return false;
  int lg = S == 2 ? 1 : (S == 4 ? 2 : 3);
  if (!y.rr[lg].empty()) return true;
  std::vector<topolow_session::SymState::StagePlan> plans(S);
  for (int st = 0; st < S; ++st) {
    const SymPlan hp = relax_symm_plan_rows(y.npad, y.grid * kSymWaves, [&](int R, int& j0, int& j1) { sym_rr_row(TR, S, st, R, j0, j1); });
    auto& h = plans[st];
    h.n_units = (int)hp.units.size();
    if ((size_t)h.n_units * kSymRows * s->dim > std::max(y.rowpart.n, y.rowpart64.n)) return false;   // (never: sym_build sizes for it)
    const std::vector<SymRun> hr = hp.runs();
    h.units.alloc(std::max<size_t>(hp.units.size(), 1));
    h.runs.alloc(hr.size());
    h.row_units.alloc(hp.row_units.size());
    if (!hp.units.empty())
      HIP_TRY(hipMemcpy(h.units.p, hp.units.data(), hp.units.size() * sizeof(SymUnit), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h.runs.p, hr.data(), hr.size() * sizeof(SymRun), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h.row_units.p, hp.row_units.data(), hp.row_units.size() * sizeof(int2), hipMemcpyHostToDevice));
  }
  y.rr[lg] = std::move(plans);
  return true;
}

// Stage st of S of iteration iter.  k_records: the spring constant of the records the apply kernel leaves (this
// iteration's until its last stage, then the next one's).
template <int DIM>
void sym_rr_stage(topolow_session* s, const void* pin, void* pout, int iter, double k, double k_records, int S, int st,
                  int rec_iter_after) {
  if constexpr (!kSymDim<DIM>) {
    throw HipError{TOPOLOW_ERR_UNSUPPORTED, "symmetric sweep: ndim"};
  } else {
    auto& y = s->sym;
    auto& h = y.rr[S == 2 ? 1 : (S == 4 ? 2 : 3)][st];
    ProfScope prof(s, &s->prof_stage);
    const int TC = y.npad / kSymCols;
    const bool f64 = s->precision == TOPOLOW_PRECISION_F64;
    if (y.rec_iter != iter) {
      for (int b = 0; b < 2; ++b) {
        if (f64)
          hipLaunchKernelGGL(symm64_records_kernel<DIM>, dim3((y.npad + 255) / 256), dim3(256), 0, s->stream, (const double*)pin,
                             s->gplus.p, y.rec64[b].p, s->n, y.npad, k, s->c_rep);
        else
          hipLaunchKernelGGL(symm_records_kernel<DIM>, dim3((y.npad + 255) / 256), dim3(256), 0, s->stream, (const float*)pin,
                             s->gplus.p, y.rec[b].p, s->n, y.npad, k, s->c_rep);
      }
      y.rec_cur = 0;
    }
    if (f64) {
      const double* rec = y.rec64[y.rec_cur].p;
      auto sweep = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(y.grid), dim3(64 * kSymWaves), 0, s->stream, y.tenc.p, rec, h.units.p, h.runs.p,
                           y.rowpart64.p, y.colpart64.p, y.npad, s->state.p, 0, (const float*)nullptr, s->part_sum.p,
                           s->part_cnt.p, 0ull);
      };
      if (s->any_threshold) sweep(&symm64_sweep_kernel<DIM, true, false>); else sweep(&symm64_sweep_kernel<DIM, false, false>);
      hipLaunchKernelGGL(symm64_apply_kernel<DIM>, dim3(TC), dim3(32 * kSymApplyParts), 0, s->stream, rec, y.rec64[y.rec_cur ^ 1].p,
                         (double*)pout, s->gplus.p, y.rowpart64.p, y.colpart64.p, h.row_units.p, s->n, y.npad, k_records, s->c_rep,
                         iter + 1, s->state.p, S, st);
    } else {
      const float* rec = y.rec[y.rec_cur].p;
      auto sweep = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(y.grid), dim3(64 * kSymWaves), 0, s->stream, y.tenc.p, rec, h.units.p, h.runs.p,
                           y.rowpart.p, y.colpart.p, y.npad, s->state.p, s->part_sum.p, s->part_cnt.p, 0ull, 0);
      };
      if (s->any_threshold) sweep(&symm_sweep_kernel<DIM, true, false>); else sweep(&symm_sweep_kernel<DIM, false, false>);
      hipLaunchKernelGGL(symm_apply_kernel<DIM>, dim3(TC), dim3(32 * kSymApplyParts), 0, s->stream, rec, y.rec[y.rec_cur ^ 1].p,
                         (float*)pout, s->gplus.p, y.rowpart.p, y.colpart.p, h.row_units.p, s->n, y.npad, k_records, s->c_rep,
                         iter + 1, s->state.p, S, st);
    }
    HIP_TRY(hipGetLastError());
    y.rec_cur ^= 1;
    y.rec_iter = rec_iter_after;
    s->stage_launches += 1;
  }
}

// ---- the symmetric sweep sharded over the row-block sessions of one run (relax_symm.h, relax_sharded_engine.h) ----
bool sym_sharded_eligible(const std::vector<topolow_session*>& ss) {
  const char* e = getenv("TOPOLOW_SHARD_SYMMETRIC");
  if (e != nullptr && e[0] == '0') return false;       // row-owner sweeps only
  const int P = (int)ss.size();
  if (P < 2) return false;
  const topolow_session* a = ss[0];
  const long long TR = ((a->n + kSymRows - 1) & ~(kSymRows - 1)) / kSymRows;
  if (TR * (TR + 1) < 8ll * P) return false;
  for (const topolow_session* s : ss)
    if (!(s->sym.allowed && s->schedule == TOPOLOW_SCHEDULE_SLAB && s->precision == TOPOLOW_PRECISION_F32 && s->dim >= 2 &&
          s->dim <= 6 && s->dim == s->udim && s->n >= s->sym.min_n && s->rows() > 0 && s->fuse_checks == a->fuse_checks))
      return false;
  return true;
}

// Session b's segment: tiles [total b / P, total (b + 1) / P) of the tile-row-major list, gathered from every session's
// row block (peer reads where the sessions sit on different GPUs).  Kept while the same sessions run together again.
template <int DIM>
void sym_sharded_build(std::vector<topolow_session*>& ss, int b) {
  const int P = (int)ss.size();
  topolow_session* s = ss[b];
  std::vector<const uint32_t*> src;
  std::vector<const void*> peers;
  std::vector<int> row0;
  bool any_thr = false;
  for (topolow_session* q : ss) {
    src.push_back(q->enc.p);
    peers.push_back(q->enc.p);
    row0.push_back(q->row_begin);
    any_thr = any_thr || q->any_threshold;
  }
  row0.push_back(s->n);
  auto& y = s->sym;
  if (y.seg_ready && !y.seg_caller && y.seg_peers == peers && y.seg_thr == any_thr) { y.seg_slot = s->rank; return; }
  HIP_TRY(hipSetDevice(s->device));
  const long long npad = (s->n + kSymRows - 1) & ~(kSymRows - 1);
  const long long TR = npad / kSymRows, total = TR * (TR + 1);
  sym_build<DIM>(s, src, row0, any_thr, total * b / P, total * (b + 1) / P);
  if (y.seg_first < 0) { y.seg_first = 0; y.seg_last = -1; }
  y.ready = false;            // the buffers now describe a segment, not the session's own whole-matrix plan
  y.seg_thr = any_thr;
  y.seg_slots = P;
  y.seg_slot = s->rank;
  y.seg_caller = false;
  y.inbox.alloc((size_t)P * y.npad * DIM);
  HIP_TRY(hipMemsetAsync(y.inbox.p, 0, (size_t)P * y.npad * DIM * sizeof(float), s->stream));
  y.own0.alloc(row0.size());
  HIP_TRY(hipMemcpy(y.own0.p, row0.data(), row0.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipStreamSynchronize(s->stream));
  y.seg_peers = peers;
}

void sym_sharded_prepare(std::vector<topolow_session*>& ss) {
  const int P = (int)ss.size();
  for (topolow_session* s : ss) {   // every block is loaded before anybody gathers from it
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
  }
  for (int b = 0; b < P; ++b) TL_DISPATCH_DIM(ss[b]->dim, sym_sharded_build, ss, b);
  for (int b = 0; b < P; ++b) {
    topolow_session* s = ss[b];
    HIP_TRY(hipSetDevice(s->device));
    // a segment meets points of every row block: their degree terms come from the owners (a caller that fills a
    // block in place -- topolow_session_commit_encoded -- need only know its own rows' degrees)
    for (topolow_session* q : ss)
      if (q != s && q->rows() > 0)
        HIP_TRY(hipMemcpy(s->gplus.p + q->row_begin, q->gplus.p + q->row_begin, (size_t)q->rows() * sizeof(float),
                          hipMemcpyDeviceToDevice));
    std::vector<float*> tab;
    for (topolow_session* q : ss) tab.push_back(q->sym.inbox.p);
    s->sym.inbox_tab.alloc(P);
    HIP_TRY(hipMemcpy(s->sym.inbox_tab.p, tab.data(), P * sizeof(float*), hipMemcpyHostToDevice));
    s->sym.seg_ready = true;
  }
}

// First half of a one-stage iteration of session s: records of all points from `pin`, the sweep of its segment, its
// partials folded per point into slot `rank` of the owners' inboxes.  err: the sweep also reduces its pairs' share of
// the pending check's MAE (one partial per unit: s->fused_parts).
template <int DIM>
void sym_sharded_sweep(topolow_session* s, const void* pin, int iter, double k, bool err) {
  if constexpr (!kSymDim<DIM>) {
    throw HipError{TOPOLOW_ERR_UNSUPPORTED, "symmetric sweep: ndim"};
  } else {
    auto& y = s->sym;
    ProfScope prof(s, err ? &s->prof_sym_err : &s->prof_sym);
    const int TC = y.npad / kSymCols;
    hipLaunchKernelGGL(symm_records_kernel<DIM>, dim3((y.npad + 255) / 256), dim3(256), 0, s->stream, (const float*)pin,
                       s->gplus.p, y.rec[0].p, s->n, y.npad, k, s->c_rep);
    auto sweep = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(y.grid), dim3(64 * kSymWaves), 0, s->stream, y.tenc.p, y.rec[0].p, y.units.p,
                         y.wave_first.p, y.rowpart.p, y.colpart.p, y.npad, s->state.p, s->part_sum.p, s->part_cnt.p,
                         s->block_cells, y.seg_first);
    };
    if (y.tiles > 0) {
      if (y.seg_thr) {
        if (err) sweep(&symm_sweep_kernel<DIM, true, true>); else sweep(&symm_sweep_kernel<DIM, true, false>);
      } else {
        if (err) sweep(&symm_sweep_kernel<DIM, false, true>); else sweep(&symm_sweep_kernel<DIM, false, false>);
      }
    }
    hipLaunchKernelGGL(symm_partial_kernel<DIM>, dim3(TC), dim3(32 * 32), 0, s->stream, y.rowpart.p, y.colpart.p,
                       y.row_units.p, y.seg_first, y.seg_last, s->n, y.npad, y.inbox_tab.p, y.own0.p, y.seg_slots, y.seg_slot,
                       s->state.p);
    HIP_TRY(hipGetLastError());
    if (err) s->fused_parts = y.n_units;
    (void)iter;
    s->stage_launches += 1;
  }
}

// Second half, behind the run's barrier: the session's own points move by the sum of the inbox's slots; `push`: the
// other sessions' copies of the output buffer.
template <int DIM>
void sym_sharded_apply(topolow_session* s, const void* pin, void* pout, const void* push, int iter) {
  if constexpr (!kSymDim<DIM>) {
    throw HipError{TOPOLOW_ERR_UNSUPPORTED, "symmetric sweep: ndim"};
  } else {
    auto& y = s->sym;
    hipLaunchKernelGGL(symm_owner_apply_kernel<DIM>, dim3((s->rows() + 255) / 256), dim3(256), 0, s->stream, (const float*)pin,
                       (float*)pout, y.inbox.p, y.seg_slots, y.npad, s->row_begin, s->row_end, (float* const*)push, s->n_push,
                       iter + 1, s->state.p);
    HIP_TRY(hipGetLastError());
  }
}

// ---- the same sweep sharded over caller-driven sessions (one process per GPU: topolow_session_symm_segment_*) ----
// The segment of this session, built from the rows the caller brought together; ONE slot and ONE owner: the folded
// partials of all n points land in the session's own inbox (the "moves buffer"), which the caller sums over the
// processes; the apply then moves ALL points from it.
template <int DIM>
void sym_segment_build(topolow_session* s, int segment, int P, const uint32_t* d_rows, int row_first, int n_rows,
                       bool any_thr) {
  auto& y = s->sym;
  const long long npad = (s->n + kSymRows - 1) & ~(kSymRows - 1);
  const long long TR = npad / kSymRows, total = TR * (TR + 1);
  const long long t0 = total * segment / P, t1 = total * (segment + 1) / P;
  int rf = -1, rl = -1;
  (void)relax_symm_plan((int)npad, 1, t0, t1, &rf, &rl);
  if (rf >= 0 && (row_first > rf * kSymRows || row_first + n_rows < std::min<long long>(s->n, (long long)(rl + 1) * kSymRows)))
    throw HipError{TOPOLOW_ERR_BAD_ARGUMENT, "symm_segment_build: d_rows does not hold the rows of the segment's tiles"};
  sym_build<DIM>(s, {d_rows}, {row_first, row_first + n_rows}, any_thr, t0, t1);
  if (y.seg_first < 0) { y.seg_first = 0; y.seg_last = -1; }
  y.ready = false;            // the buffers now describe a segment, not the session's own whole-matrix plan
  y.seg_thr = any_thr;
  y.seg_slots = 1;
  y.seg_slot = 0;
  y.seg_caller = true;
  y.seg_peers.clear();
  y.inbox.alloc((size_t)y.npad * DIM);
  HIP_TRY(hipMemsetAsync(y.inbox.p, 0, (size_t)y.npad * DIM * sizeof(float), s->stream));
  const int own[2] = {0, s->n};
  y.own0.alloc(2);
  HIP_TRY(hipMemcpy(y.own0.p, own, sizeof own, hipMemcpyHostToDevice));
  float* tab[1] = {y.inbox.p};
  y.inbox_tab.alloc(1);
  HIP_TRY(hipMemcpy(y.inbox_tab.p, tab, sizeof tab, hipMemcpyHostToDevice));
  HIP_TRY(hipStreamSynchronize(s->stream));     // the tile-major copy is complete: the caller may free d_rows
  y.seg_ready = true;
}

template <int DIM>
void sym_segment_apply(topolow_session* s, const void* pin, void* pout, int iter) {
  if constexpr (!kSymDim<DIM>) {
    throw HipError{TOPOLOW_ERR_UNSUPPORTED, "symmetric sweep: ndim"};
  } else {
    auto& y = s->sym;
    hipLaunchKernelGGL(symm_owner_apply_kernel<DIM>, dim3((s->n + 255) / 256), dim3(256), 0, s->stream, (const float*)pin,
                       (float*)pout, y.inbox.p, 1, y.npad, 0, s->n, (float* const*)nullptr, 0, iter + 1, s->state.p);
    HIP_TRY(hipGetLastError());
  }
}

// One convergence check of the session's own loop.  error_pass = true: the separate pass over the block (or
// the edge list) + the controller; false: the partials were written by the stage kernel just launched
// (ERR launch), only the controller follows.  pc.beside: on the check stream, beside the next stages.
void launch_check(topolow_session* s, topolow_session::PendingCheck& pc, bool error_pass) {
  hipStream_t check_on = s->stream;
  if (pc.beside) {
    HIP_TRY(hipEventRecord(s->ev_iter_done, s->stream));
    HIP_TRY(hipStreamWaitEvent(s->check_stream, s->ev_iter_done, 0));
    check_on = s->check_stream;
  }
  {
    StreamScope on(s, check_on);
    ProfScope prof(s, &s->prof_check);
    if (error_pass) {
      TL_DISPATCH_DIM(s->dim, launch_edge_error, s, s->pos[pc.buf].p, s->state.p);
      launch_controller(s, s->pos[pc.buf].p, pc.iter1, pc.k_after);
    } else {
      launch_controller(s, s->pos[pc.buf].p, pc.iter1, pc.k_after, s->part_sum.p, s->part_cnt.p, s->fused_parts);
    }
  }
  if (pc.beside) HIP_TRY(hipEventRecord(s->ev_check_done, check_on));
  hipEvent_t e = take_event(s);
  HIP_TRY(hipEventRecord(e, check_on));
  s->pending.push_back(e);
  poll_checks(s, 3);
  pc.active = false;
}

// A check that was waiting for the next iteration's kernel and will not get one (the caller stopped
// enqueueing, or asks for results): run it as a separate pass now.
void flush_pending_check(topolow_session* s) {
  if (s->pcheck.active) launch_check(s, s->pcheck, /*error_pass=*/true);
}

template <typename F>
int guarded(char* errbuf, size_t errlen, F&& body) {
  try {
    body();
    return TOPOLOW_OK;
  } catch (const HipError& e) {
    set_err(errbuf, errlen, "%s", e.msg.c_str());
    return e.code;
  } catch (const std::bad_alloc&) {
    set_err(errbuf, errlen, "out of host memory");
    return TOPOLOW_ERR_HIP;
  }
}

#include "relax_sharded_engine.h"

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

const char* topolow_relax_version(void) { return "topolow_relax 0.1 (gfx950)"; }

void topolow_default_options(topolow_options* opt) {
  if (!opt) return;
  std::memset(opt, 0, sizeof(*opt));
  opt->seed = 0;
  opt->schedule = TOPOLOW_SCHEDULE_AUTO;
  opt->precision = TOPOLOW_PRECISION_AUTO;
  opt->slab_stages = 0;
  opt->device = -1;
  opt->gs_max_n = 0;
  opt->keep_labels = 0;
  opt->interrupt_cb = nullptr;
  opt->interrupt_user = nullptr;
}

int64_t topolow_encoded_index(int32_t row_in_block, int32_t column, int32_t ld) {
  return (int64_t)enc_index(row_in_block, column, ld);
}

uint32_t topolow_encode_target(double dissimilarity, int32_t threshold_code) {
  return encode_target(dissimilarity, threshold_code);
}
double topolow_decode_target(uint32_t bits, int32_t* threshold_code) {
  int c = 0;
  const double v = decode_target(bits, &c);
  if (threshold_code) *threshold_code = c;
  return v;
}

int32_t topolow_slab_stages_for_k(double k, int32_t ndim) { return slab_stages_for_k(k, ndim); }
int32_t topolow_slab_stages_at(int32_t iter, double k, int32_t ndim) { return slab_stages_at(iter, k, ndim); }

int32_t topolow_slab_plan(int32_t n, int32_t slab_stages, uint64_t seed, int32_t iter,
                          int32_t* ranges_out, int32_t max_stages) {
  const SlabGeom g = slab_geom(n, slab_stages);
  if (!ranges_out) return g.n_stages;
  for (int slot = 0; slot < g.n_stages && slot < max_stages; ++slot) {
    const SlabRanges r = slab_ranges(g, seed, iter, slot);
    ranges_out[4 * slot + 0] = r.b0;
    ranges_out[4 * slot + 1] = r.e0;
    ranges_out[4 * slot + 2] = r.b1;
    ranges_out[4 * slot + 3] = r.e1;
  }
  return g.n_stages;
}

int64_t topolow_gs_pair_order(int32_t n, uint64_t seed, int32_t iter, int32_t* pairs_out) {
  return gs_pair_order(n, seed, iter, pairs_out);
}

int topolow_controller_script(const double* mae_seq, const int32_t* iter_seq,
                              const double* k_seq, int32_t n_obs, double k0, int32_t window,
                              double eps, int32_t* stopped_at_obs, int32_t* snapshot_flags,
                              double* best_mae, double* best_k, int32_t* best_iter) {
  Controller c;
  c.init(k0, window, eps);
  *stopped_at_obs = -1;
  for (int o = 0; o < n_obs; ++o) {
    const int a = c.observe(mae_seq[o], iter_seq[o], k_seq[o]);
    if (snapshot_flags) snapshot_flags[o] = (a & 2) ? 1 : 0;
    if (a & 1) { *stopped_at_obs = o; break; }
  }
  *best_mae = c.best_mae;
  *best_k = c.best_k;
  *best_iter = c.best_iter;
  return TOPOLOW_OK;
}

// ---- session ---------------------------------------------------------------------------
int topolow_session_create(topolow_session** out, int32_t n, int32_t ndim, int32_t row_begin,
                           int32_t row_end, int32_t precision, int32_t device, char* errbuf,
                           size_t errlen) {
  if (!out) return TOPOLOW_ERR_BAD_ARGUMENT;
  *out = nullptr;
  if (n < 2) {
    set_err(errbuf, errlen, "Need at least 2 points for embedding");
    return TOPOLOW_ERR_TOO_FEW_POINTS;
  }
  if (ndim < 1 || ndim > kMaxDim) {
    set_err(errbuf, errlen, "ndim must be between 1 and %d in this build", kMaxDim);
    return TOPOLOW_ERR_UNSUPPORTED;
  }
  if (row_begin < 0 || row_end > n || row_begin >= row_end) {
    set_err(errbuf, errlen, "bad row block [%d,%d) for n=%d", row_begin, row_end, n);
    return TOPOLOW_ERR_BAD_ARGUMENT;
  }
  topolow_session* s = nullptr;
  const int rc = guarded(errbuf, errlen, [&] {
    const int dev = select_device(device);
    s = new topolow_session();
    s->device = dev;
    s->n = n;
    s->dim = kernel_dim(ndim);
    s->udim = ndim;
    s->row_begin = row_begin;
    s->row_end = row_end;
    s->ld = (n + kEncLdAlign - 1) & ~(kEncLdAlign - 1);
    s->precision = precision == TOPOLOW_PRECISION_F64 ? TOPOLOW_PRECISION_F64
                                                      : TOPOLOW_PRECISION_F32;
    HIP_TRY(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
    s->stream = s->own_stream;
    HIP_TRY(hipStreamCreateWithFlags(&s->check_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_iter_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_check_done, hipEventDisableTiming));
    const char* serial = getenv("TOPOLOW_SERIAL_CHECKS");
    s->serial_checks = serial != nullptr && serial[0] == '1';
    const char* fuse = getenv("TOPOLOW_FUSE_CHECKS");
    s->fuse_checks = !(fuse != nullptr && fuse[0] == '0');
    const char* symm = getenv("TOPOLOW_SYMMETRIC");
    s->sym.allowed = !(symm != nullptr && symm[0] == '0');   // TOPOLOW_SYMMETRIC=0: row-owner sweeps only
    const char* symm_min = getenv("TOPOLOW_SYMMETRIC_MIN_N");   // tests lower the size gate to reach the sweep on small problems
    s->sym.min_n = symm_min != nullptr ? atoi(symm_min) : kSymMinPoints;
    const char* symm2 = getenv("TOPOLOW_SYMMETRIC_TWO_STAGE");
    s->sym.two_stage = !(symm2 != nullptr && symm2[0] == '0');
    const char* rr_min = getenv("TOPOLOW_SYMMETRIC_STAGE_MIN_TILES");
    if (rr_min != nullptr) s->sym.rr_min_tiles = std::max(0, atoi(rr_min));
    s->enc.alloc((size_t)((s->rows() + kEncRowAlign - 1) / kEncRowAlign * kEncRowAlign) * s->ld);
    const size_t pos_bytes = (size_t)s->pos_rows() * s->dim * s->real_size();
    for (auto& b : s->pos) b.alloc(pos_bytes);
    s->best.alloc(pos_bytes);
    s->state.alloc(1);
    HIP_TRY(hipHostMalloc((void**)&s->mailbox, sizeof(RunState), hipHostMallocMapped));
    std::memset(s->mailbox, 0, sizeof(RunState));
    HIP_TRY(hipHostGetDevicePointer((void**)&s->mailbox_dev, s->mailbox, 0));
  });
  if (rc != TOPOLOW_OK) { delete s; return rc; }
  *out = s;
  return TOPOLOW_OK;
}

void topolow_session_destroy(topolow_session* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->check_stream) (void)hipStreamSynchronize(s->check_stream);
#ifdef TOPOLOW_TUNING
  g_stamps.dump();
#endif
  delete s;
}

int topolow_session_set_relabel(topolow_session* s, uint64_t seed, char* errbuf, size_t errlen) {
  if (!s) return TOPOLOW_ERR_BAD_ARGUMENT;
  if (s->gplus.p || s->part_sum.p) {
    set_err(errbuf, errlen, "the relabelling must be chosen before targets, edges or positions are loaded");
    return TOPOLOW_ERR_BAD_ARGUMENT;
  }
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    s->perm.clear();
    s->inv.clear();
    if (seed == 0) return;   // identity
    const int n = s->n;
    s->perm.resize(n);
    s->inv.resize(n);
    for (int q = 0; q < n; ++q) s->perm[q] = q;
    for (int q = n - 1; q > 0; --q) {   // Fisher-Yates on the schedule's counter-based stream
      const uint32_t r = rnd_below(rnd64(seed, 0x7e1abe1ull, (uint64_t)q), (uint32_t)(q + 1));
      std::swap(s->perm[q], s->perm[r]);
    }
    for (int q = 0; q < n; ++q) s->inv[s->perm[q]] = q;
    s->d_perm.alloc(n);
    s->d_inv.alloc(n);
    HIP_TRY(hipMemcpy(s->d_perm.p, s->perm.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_inv.p, s->inv.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  });
}

int topolow_session_labels(const topolow_session* s, int32_t* session_to_caller) {
  if (!s || !session_to_caller) return TOPOLOW_ERR_BAD_ARGUMENT;
  for (int q = 0; q < s->n; ++q) session_to_caller[q] = s->perm.empty() ? q : s->perm[q];
  return TOPOLOW_OK;
}

int topolow_session_load_dense(topolow_session* s, const double* D, const int32_t* T,
                               const int32_t* degrees, char* errbuf, size_t errlen) {
  if (!s || !D || !T || !degrees) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    const size_t nn = (size_t)s->n * s->n;
    DevBuf<double> dD;
    DevBuf<int> dT;
    dD.alloc(nn);
    dT.alloc(nn);
    HIP_TRY(hipMemcpy(dD.p, D, nn * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dT.p, T, nn * 4, hipMemcpyHostToDevice));
    dim3 grid(s->rows(), (s->ld + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(encode_dense_kernel, grid, dim3(kThreads), 0, s->stream, dD.p, dT.p, s->n,
                       s->row_begin, s->row_end, s->ld, s->enc.p, s->perm.empty() ? nullptr : s->d_perm.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    compute_row_flags(s);
    upload_degrees(s, degrees);
  });
}

int topolow_session_load_coo(topolow_session* s, const int32_t* edge_i, const int32_t* edge_j,
                             const double* edge_dist, const int32_t* edge_thresh,
                             int64_t n_edges, const int32_t* degrees, char* errbuf,
                             size_t errlen) {
  if (!s || !degrees || n_edges < 0) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    dim3 grid(s->rows(), (s->ld + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(fill_unmeasured_kernel, grid, dim3(kThreads), 0, s->stream, s->n,
                       s->row_begin, s->row_end, s->ld, s->enc.p);
    HIP_TRY(hipGetLastError());
    const int64_t chunk = 1 << 24;
    DevBuf<int> di, dj, dc;
    DevBuf<double> dd;
    const size_t cap = (size_t)std::min<int64_t>(chunk, std::max<int64_t>(n_edges, 1));
    di.alloc(cap); dj.alloc(cap); dc.alloc(cap); dd.alloc(cap);
    for (int64_t off = 0; off < n_edges; off += chunk) {
      const int64_t m = std::min<int64_t>(chunk, n_edges - off);
      HIP_TRY(hipMemcpyAsync(di.p, edge_i + off, m * 4, hipMemcpyHostToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(dj.p, edge_j + off, m * 4, hipMemcpyHostToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(dd.p, edge_dist + off, m * 8, hipMemcpyHostToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(dc.p, edge_thresh + off, m * 4, hipMemcpyHostToDevice, s->stream));
      hipLaunchKernelGGL(scatter_edges_kernel, dim3((unsigned)((m + kThreads - 1) / kThreads)),
                         dim3(kThreads), 0, s->stream, di.p, dj.p, dd.p, dc.p, (long long)m, s->n,
                         s->row_begin, s->row_end, s->ld, s->enc.p, s->inv.empty() ? nullptr : s->d_inv.p);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(s->stream));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    compute_row_flags(s);
    upload_degrees(s, degrees);
  });
}

void* topolow_session_encoded_ptr(topolow_session* s) { return s ? (void*)s->enc.p : nullptr; }
int32_t topolow_session_encoded_ld(const topolow_session* s) { return s ? s->ld : 0; }

int topolow_session_commit_encoded(topolow_session* s, const int32_t* degrees, char* errbuf,
                                   size_t errlen) {
  if (!s || !degrees) return TOPOLOW_ERR_BAD_ARGUMENT;
  if (!s->perm.empty()) {
    set_err(errbuf, errlen, "a block filled by the caller is in the caller's labels: not with a relabelled session");
    return TOPOLOW_ERR_BAD_ARGUMENT;
  }
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());  // the caller filled the block on another stream
    compute_row_flags(s);
    upload_degrees(s, degrees);
  });
}

int topolow_session_set_edges(topolow_session* s, const int32_t* edge_i, const int32_t* edge_j,
                              const double* edge_dist, const int32_t* edge_thresh,
                              int64_t n_edges, char* errbuf, size_t errlen) {
  if (!s || n_edges < 0) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    s->n_edges = n_edges;
    if (s->precision == TOPOLOW_PRECISION_F64) {   // the f64 sweep's delta tiles are made from this list: rebuilt on first use
      s->sym.ready = false;
      s->sym.delta_ready = false;
    }
    const size_t m = (size_t)n_edges;
    long long blocks = (n_edges + (long long)kThreads * 8 - 1) / ((long long)kThreads * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    s->n_parts = (int)blocks;
    // dense MAE pass: one workgroup per (column chunk, 64-row tile)
    s->dense_grid_x = (((s->n + 3) & ~3) + ErrCfg::CHUNK - 1) / ErrCfg::CHUNK;
    s->dense_grid_y = (s->rows() + kErrTileRows - 1) / kErrTileRows;
    s->dense_blocks = s->dense_grid_x * s->dense_grid_y;
    const int stage_blocks = (s->rows() + CfgProd::ROWS - 1) / CfgProd::ROWS;   // fused checks: one partial per workgroup
    s->part_sum.alloc(std::max({s->n_parts, s->dense_blocks, stage_blocks}));
    s->part_cnt.alloc(std::max({s->n_parts, s->dense_blocks, stage_blocks}));
    // Can the MAE be reduced from the encoded block instead of gathering the edge list?  Only if
    // the list is exactly the set of measured cells the dense pass would visit -- checked with an
    // order-independent fingerprint BEFORE anything is uploaded: when it holds (it does for
    // everything the R driver builds), the list itself never travels to the device.
    s->dense_mae = false;
    s->dense_parity = !(s->row_begin == 0 && s->row_end == s->n);
    const int* inv = s->inv.empty() ? nullptr : s->inv.data();
    const char* force = getenv("TOPOLOW_EDGE_MAE");
    s->list_is_block = false;
    if (s->gplus.p && !(force && atoi(force) != 0)) {
      std::atomic<bool> owned{true};
      std::atomic<unsigned long long> fp_total{0};
      host_parallel(m, [&](size_t lo_e, size_t hi_e) {
        unsigned long long fp = 0;
        for (size_t e = lo_e; e < hi_e; ++e) {
          int a = edge_i[e], b = edge_j[e];
          if (a < 0 || b < 0 || a >= s->n || b >= s->n || a == b) { owned.store(false); return; }
          if (inv) { a = inv[a]; b = inv[b]; }
          const int lo = a < b ? a : b, hi = a < b ? b : a;
          int owner = lo;
          if (s->dense_parity && (((lo + hi) & 1) != 0)) owner = hi;
          if (owner < s->row_begin || owner >= s->row_end) { owned.store(false); return; }
          const uint32_t w = encode_target(edge_dist[e], edge_thresh[e]);
          if (w == kInfWord) { owned.store(false); return; }
          fp += cell_fingerprint(lo, hi, w);
        }
        fp_total.fetch_add(fp);
      });
      if (owned.load()) {
        DevBuf<unsigned long long> d_fp;
        d_fp.alloc(2);
        // on the session's stream: it is non-blocking, a null-stream memset is not ordered with it
        HIP_TRY(hipMemsetAsync(d_fp.p, 0, 16, s->stream));
        hipLaunchKernelGGL(upper_fingerprint_kernel, dim3(s->rows()), dim3(kThreads), 0, s->stream,
                           s->enc.p, s->n, s->row_begin, s->row_end, s->ld, s->dense_parity ? 1 : 0,
                           d_fp.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s->stream));
        unsigned long long h[2];
        HIP_TRY(hipMemcpy(h, d_fp.p, 16, hipMemcpyDeviceToHost));
        s->list_is_block = (h[0] == fp_total.load()) && (h[1] == (unsigned long long)m);
        // (f64 sessions keep the exact edge-list pass: the block holds 4-byte targets; their symmetric sweep fuses the
        //  check through the delta tiles instead, relax_symm64.h)
        s->dense_mae = s->list_is_block && s->dim <= kMaxTunedDim && s->precision == TOPOLOW_PRECISION_F32;
      }
    }
    if (s->dense_mae) {   // the gather fallback is not needed: keep 1-element placeholders
      s->ei.alloc(1); s->ej.alloc(1); s->ec.alloc(1); s->et.alloc(8);
      return;
    }
    // edge-list MAE (parity sessions keep the exact f64 targets): upload the list in session labels
    s->ei.alloc(m); s->ej.alloc(m); s->ec.alloc(m);
    std::vector<int8_t> codes(m);
    std::vector<int> li, lj;
    if (inv) { li.resize(m); lj.resize(m); }
    host_parallel(m, [&](size_t lo_e, size_t hi_e) {
      for (size_t e = lo_e; e < hi_e; ++e) {
        const int c = edge_thresh[e];
        codes[e] = (int8_t)(c == 0 ? 0 : (c == 1 ? 1 : (c == -1 ? -1 : 2)));  // others never count
        if (inv) {
          const int a = edge_i[e], b = edge_j[e];
          li[e] = (a >= 0 && a < s->n) ? inv[a] : a;
          lj[e] = (b >= 0 && b < s->n) ? inv[b] : b;
        }
      }
    });
    if (m) {
      HIP_TRY(hipMemcpy(s->ei.p, inv ? li.data() : edge_i, m * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(s->ej.p, inv ? lj.data() : edge_j, m * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(s->ec.p, codes.data(), m, hipMemcpyHostToDevice));
    }
    if (s->precision == TOPOLOW_PRECISION_F64) {
      s->et.alloc(m * 8);
      if (m) HIP_TRY(hipMemcpy(s->et.p, edge_dist, m * 8, hipMemcpyHostToDevice));
    } else {
      std::vector<float> t(m);
      host_parallel(m, [&](size_t lo_e, size_t hi_e) {
        for (size_t e = lo_e; e < hi_e; ++e) t[e] = (float)edge_dist[e];
      });
      s->et.alloc(m * 4);
      if (m) HIP_TRY(hipMemcpy(s->et.p, t.data(), m * 4, hipMemcpyHostToDevice));
    }
  });
}

int topolow_session_set_positions(topolow_session* s, const double* positions, char* errbuf,
                                  size_t errlen) {
  if (!s || !positions) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    s->cur = 0;
    upload_positions(s, positions, s->pos[0].p);
    // the other buffers need the same padding rows
    for (int b = 1; b < 3; ++b)
      HIP_TRY(hipMemcpyAsync(s->pos[b].p, s->pos[0].p, (size_t)s->pos_rows() * s->dim * s->real_size(),
                             hipMemcpyDeviceToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->held = -1;
    s->pcheck.active = false;
  });
}

int topolow_session_get_positions(topolow_session* s, double* positions, char* errbuf,
                                  size_t errlen) {
  if (!s || !positions) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    download_positions(s, s->pos[s->cur].p, positions);
  });
}

int topolow_session_begin(topolow_session* s, int32_t n_iter, double k0, double cooling_rate,
                          double c_repulsion, double relative_epsilon,
                          int32_t convergence_window, int32_t convergence_check_freq,
                          uint64_t seed, int32_t slab_stages, char* errbuf, size_t errlen) {
  if (!s) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    if (!s->gplus.p) throw HipError{TOPOLOW_ERR_BAD_ARGUMENT, "session has no targets loaded"};
    if (!s->part_sum.p) throw HipError{TOPOLOW_ERR_BAD_ARGUMENT, "session has no edge list"};
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipStreamSynchronize(s->check_stream));
    for (hipEvent_t e : s->pending) s->event_pool.push_back(e);
    s->pending.clear();
    s->n_iter = n_iter;
    s->k0 = k0;
    s->cooling = cooling_rate;
    s->c_rep = c_repulsion;
    s->eps = relative_epsilon;
    s->window = convergence_window;
    s->check_freq = convergence_check_freq < 1 ? 10 : convergence_check_freq;  // reference :181
    {
      const int need = n_iter / s->check_freq + 2;
      if (need > s->trace_cap) {
        if (s->trace) { (void)hipHostFree(s->trace); s->trace = nullptr; s->trace_cap = 0; }
        HIP_TRY(hipHostMalloc((void**)&s->trace, sizeof(double) * 3 * (size_t)need, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&s->trace_dev, s->trace, 0));
        s->trace_cap = need;
      }
    }
    s->seed = seed;
    s->fixed_stages = slab_stages;
    s->iters_enqueued = 0;
    s->k_host = k0;
    s->host_seen_stop = false;
    s->held = -1;
    s->pcheck.active = false;
    s->sym.rec_iter = -1;
    s->began = true;
    RunState st;
    std::memset(&st, 0, sizeof st);
    st.ctl.init(k0, convergence_window, relative_epsilon);
    st.k_base = k0;
    st.cooling = cooling_rate;
    st.n_iter = n_iter;
    st.first_nonfinite = 0x7fffffff;
    *s->mailbox = st;
    HIP_TRY(hipMemcpyAsync(s->state.p, &st, sizeof st, hipMemcpyHostToDevice, s->stream));
    // best snapshot starts as the initial positions (reference :171)
    HIP_TRY(hipMemcpyAsync(s->best.p, s->pos[s->cur].p,
                           (size_t)s->pos_rows() * s->dim * s->real_size(), hipMemcpyDeviceToDevice,
                           s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
  });
}

int topolow_session_enqueue(topolow_session* s, int32_t max_iters, int32_t* enqueued,
                            char* errbuf, size_t errlen) {
  if (!s || !s->began) return TOPOLOW_ERR_BAD_ARGUMENT;
  if (enqueued) *enqueued = 0;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    int done = 0;
    while (done < max_iters && s->iters_enqueued < s->n_iter && !s->host_seen_stop) {
      const int iter = s->iters_enqueued;
      if (s->schedule == TOPOLOW_SCHEDULE_GS) {
        TL_DISPATCH_DIM(s->dim, launch_tilegs_iteration, s, s->pos[s->cur].p, iter, s->k_host);
      } else {
        const int stages = s->fixed_stages > 0 ? s->fixed_stages : slab_stages_at(iter, s->k_host, s->dim);
        const SlabGeom g = slab_geom(s->n, stages);
        const bool sym = g.n_stages == 1 && sym_eligible(s) && sym_available(s);
        // (the row-owner ERR instance pairs rows two by two: an odd block keeps the separate pass; the symmetric
        //  sweep's ERR instance has no such rule)
        const bool fuse_now = s->pcheck.active && g.n_stages == 1 &&
                              (s->precision == TOPOLOW_PRECISION_F64 ? (sym && s->sym.delta_ready) : (sym || s->rows() % 2 == 0));
        if (s->pcheck.active && !fuse_now) flush_pending_check(s);
        const bool symrr = s->sym.two_stage && sym_rr_stages_ok(g.n_stages) && sym_eligible(s) && sym_available(s) &&
                           sym_rr_available(s, g.n_stages);
        if (sym) {   // one sweep over the upper triangle moves both ends of every pair
          int out = 0;
          while (out == s->cur || out == s->held) ++out;
          TL_DISPATCH_DIM(s->dim, sym_iteration, s, s->pos[s->cur].p, s->pos[out].p, iter, s->k_host, fuse_now);
          s->cur = out;
        } else if (symrr) {   // S symmetric sweeps over the tiles of one stage each, in random order
          const int S = g.n_stages;
          int order[8];
          sym_rr_order(s->seed, iter, S, order);
          for (int t = 0; t < S; ++t) {
            int out = 0;
            while (out == s->cur || out == s->held) ++out;
            const bool last = t == S - 1;
            TL_DISPATCH_DIM(s->dim, sym_rr_stage, s, s->pos[s->cur].p, s->pos[out].p, iter, s->k_host,
                            last ? s->k_host * (1.0 - s->cooling) : s->k_host, S, order[t], last ? iter + 1 : iter);
            s->cur = out;
          }
        } else
        for (int slot = 0; slot < g.n_stages; ++slot) {
          const SlabRanges rg = slab_ranges(g, s->seed, iter, slot);
          int out = 0;   // a buffer that is neither the input nor the one a running check reads
          while (out == s->cur || out == s->held) ++out;
          TL_DISPATCH_DIM(s->dim, launch_stage, s, s->pos[s->cur].p, s->pos[out].p, s->state.p, rg,
                          iter + 1, s->k_host, nullptr, 0, fuse_now);
          s->cur = out;
          if (fuse_now) s->fused_parts = (s->rows() + CfgProd::ROWS - 1) / CfgProd::ROWS;
        }
        if (fuse_now) launch_check(s, s->pcheck, /*error_pass=*/false);
      }
      s->iters_enqueued = iter + 1;
      s->k_host *= (1.0 - s->cooling);  // reference :289
      ++done;
      if ((iter + 1) % s->check_freq == 0 || iter == s->n_iter - 1) {  // reference :294
        topolow_session::PendingCheck pc;
        pc.active = true;
        pc.iter1 = iter + 1;
        pc.k_after = s->k_host;
        pc.buf = s->cur;
        pc.beside = s->schedule == TOPOLOW_SCHEDULE_SLAB && s->stream == s->own_stream &&
                    !s->profiling && !s->serial_checks;
        // The check reads this iteration's positions while the next iteration's stages run.  Its
        // verdict (stop / snapshot) is the same as in the serial order: the buffer it reads is
        // not written until the next check has waited for it; stage kernels that start after a
        // stop are no-ops and those already running write buffers nobody returns.
        if (s->held >= 0) HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_check_done, 0));
        s->held = pc.beside ? s->cur : -1;
        // one stage next iteration: its kernel reduces this check's MAE (the positions it reads ARE this
        // check's positions) and the separate pass over the block is dropped
        // (fp32: the row-owner ERR instance or the symmetric sweep's; f64: the symmetric sweep's, exact through its delta tiles)
        const bool fuse = s->fuse_checks && s->schedule == TOPOLOW_SCHEDULE_SLAB && iter + 1 < s->n_iter &&
                          ((s->dense_mae && s->precision == TOPOLOW_PRECISION_F32 && (s->rows() % 2 == 0 || sym_eligible(s))) ||
                           (s->precision == TOPOLOW_PRECISION_F64 && sym_eligible(s) && sym_available(s) && s->sym.delta_ready)) &&
                          slab_geom(s->n, s->fixed_stages > 0 ? s->fixed_stages
                                                              : slab_stages_at(iter + 1, s->k_host, s->dim)).n_stages == 1;
        if (fuse) s->pcheck = pc;
        else launch_check(s, pc, /*error_pass=*/true);
      }
      // the slab kernels flag non-finite results themselves; the in-place schedule is inspected at
      // the reference's cadence (:359-361), after that iteration's check
      if (s->schedule == TOPOLOW_SCHEDULE_GS && (iter + 1) % 10 == 0)
        launch_tilegs_finite(s, s->pos[s->cur].p, iter + 1);
    }
    if (enqueued) *enqueued = done;
  });
}

int topolow_session_wait(topolow_session* s, char* errbuf, size_t errlen) {
  if (!s) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipStreamSynchronize(s->check_stream));
  });
}

int topolow_session_sync(topolow_session* s, int32_t* iterations_run, int32_t* stopped,
                         double* last_mae, char* errbuf, size_t errlen) {
  if (!s) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    flush_pending_check(s);
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipStreamSynchronize(s->check_stream));
    poll_checks(s, 0);
    RunState st;
    HIP_TRY(hipMemcpy(&st, s->state.p, sizeof st, hipMemcpyDeviceToHost));
    if (st.stopped) s->host_seen_stop = true;
    if (iterations_run) *iterations_run = st.stopped ? st.iter_base : s->iters_enqueued;
    if (stopped) *stopped = st.stopped;
    if (last_mae) *last_mae = st.last_mae;
  });
}

int topolow_session_finish(topolow_session* s, double* positions_out, int32_t* converged,
                           int32_t* iterations, double* final_mae, double* final_k,
                           char* errbuf, size_t errlen) {
  if (!s) return TOPOLOW_ERR_BAD_ARGUMENT;
  int rc_nonfinite = TOPOLOW_OK;
  const int rc = guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    flush_pending_check(s);
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipStreamSynchronize(s->check_stream));
    poll_checks(s, 0);
    RunState st;
    HIP_TRY(hipMemcpy(&st, s->state.p, sizeof st, hipMemcpyDeviceToHost));
    // Reference :359-361 -- positions are inspected after every 10th iteration (after that
    // iteration's convergence check, which may already have stopped the run).
    const int ran = st.stopped ? st.iter_base : s->iters_enqueued;
    if (st.first_nonfinite != 0x7fffffff) {
      const int t = ((st.first_nonfinite + 9) / 10) * 10;
      if (t <= ran && !(st.stopped && t == ran)) {
        set_err(errbuf, errlen,
                "Numerical instability at iteration %d. Reduce k0 or c_repulsion.", t);
        rc_nonfinite = TOPOLOW_ERR_NONFINITE;
        return;
      }
    }
    if (positions_out) download_positions(s, s->best.p, positions_out);
    if (converged) *converged = st.converged;
    if (iterations) *iterations = st.ctl.best_iter;
    if (final_mae) *final_mae = st.ctl.best_mae;
    if (final_k) *final_k = st.ctl.best_k;
  });
  return rc != TOPOLOW_OK ? rc : rc_nonfinite;
}

int topolow_session_check_trace(topolow_session* s, double* out, int32_t max_checks, int32_t* n_checks) {
  if (!s || (!out && max_checks > 0) || !n_checks) return TOPOLOW_ERR_BAD_ARGUMENT;
  (void)hipSetDevice(s->device);
  try { flush_pending_check(s); } catch (const HipError&) { return TOPOLOW_ERR_HIP; }
  (void)hipStreamSynchronize(s->stream);
  (void)hipStreamSynchronize(s->check_stream);
  const int have = std::min(s->mailbox ? s->mailbox->n_checks : 0, s->trace_cap);
  *n_checks = have;
  const int take = std::min(have, (int)max_checks);
  if (take > 0) std::memcpy(out, s->trace, sizeof(double) * 3 * (size_t)take);
  return TOPOLOW_OK;
}

int topolow_session_set_profiling(topolow_session* s, int32_t enable) {
  if (!s) return TOPOLOW_ERR_BAD_ARGUMENT;
  s->profiling = enable != 0;
  return TOPOLOW_OK;
}

int topolow_session_profile(topolow_session* s, double* stage_ms, int64_t* stage_launches,
                            double* check_ms, int64_t* checks, char* errbuf, size_t errlen) {
  if (!s) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    flush_pending_check(s);
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipStreamSynchronize(s->check_stream));
    auto drain = [](std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, double* ms, int64_t* cnt) {
      double total = 0.0;
      for (auto& pr : v) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) total += t;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
      }
      if (ms) *ms = total;
      if (cnt) *cnt = (int64_t)v.size();
      v.clear();
    };
    // stage launches: the plain ones plus those that also reduced a check's MAE (reported apart by
    // topolow_session_profile_fused, which must be asked first)
    double ms_a = 0.0, ms_b = 0.0;
    int64_t n_a = 0, n_b = 0;
    drain(s->prof_stage, &ms_a, &n_a);
    drain(s->prof_stage_err, &ms_b, &n_b);
    if (stage_ms) *stage_ms = ms_a + ms_b;
    if (stage_launches) *stage_launches = n_a + n_b;
    drain(s->prof_check, check_ms, checks);
  });
}

int topolow_session_profile_fused(topolow_session* s, double* fused_ms, int64_t* fused_launches, char* errbuf,
                                  size_t errlen) {
  if (!s) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    flush_pending_check(s);
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipStreamSynchronize(s->check_stream));
    double total = 0.0;
    for (auto& pr : s->prof_stage_err) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) total += t;
    }
    if (fused_ms) *fused_ms = total;
    if (fused_launches) *fused_launches = (int64_t)s->prof_stage_err.size();
  });
}

int topolow_session_profile_symmetric(topolow_session* s, double* plain_ms, int64_t* plain_iterations, double* fused_ms,
                                      int64_t* fused_iterations, char* errbuf, size_t errlen) {
  if (!s) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    flush_pending_check(s);
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipStreamSynchronize(s->check_stream));
    auto drain = [](std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, double* ms, int64_t* cnt) {
      double total = 0.0;
      for (auto& pr : v) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) total += t;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
      }
      if (ms) *ms = total;
      if (cnt) *cnt = (int64_t)v.size();
      v.clear();
    };
    drain(s->prof_sym, plain_ms, plain_iterations);
    drain(s->prof_sym_err, fused_ms, fused_iterations);
  });
}

int topolow_session_set_stream(topolow_session* s, void* hip_stream, int32_t external) {
  if (!s) return TOPOLOW_ERR_BAD_ARGUMENT;
  (void)hipSetDevice(s->device);
  (void)hipStreamSynchronize(s->stream);
  (void)hipStreamSynchronize(s->check_stream);
  s->stream = external ? (hipStream_t)hip_stream : s->own_stream;
  return TOPOLOW_OK;
}

void* topolow_session_stream(topolow_session* s) { return s ? (void*)s->stream : nullptr; }

int topolow_session_set_schedule(topolow_session* s, int32_t schedule) {
  if (!s || (schedule != TOPOLOW_SCHEDULE_SLAB && schedule != TOPOLOW_SCHEDULE_GS))
    return TOPOLOW_ERR_BAD_ARGUMENT;
  if (s->row_begin != 0 || s->row_end != s->n) return TOPOLOW_ERR_UNSUPPORTED;  // tile GS: whole problem
  s->schedule = schedule;
  return TOPOLOW_OK;
}

int64_t topolow_tilegs_pair_order(int32_t n, uint64_t seed, int32_t iter, int32_t* pairs_out) {
  return tilegs_pair_order(n, seed, iter, pairs_out);
}

int32_t topolow_session_position_rows(const topolow_session* s) { return s ? s->pos_rows() : 0; }
int32_t topolow_session_position_dim(const topolow_session* s) { return s ? s->dim : 0; }

int32_t topolow_session_uses_dense_mae(const topolow_session* s) {
  return s && s->dense_mae ? 1 : 0;
}

int64_t topolow_session_stage_launches(const topolow_session* s) {
  return s ? s->stage_launches : 0;
}

int64_t topolow_session_bytes_per_iteration(const topolow_session* s) {
  if (!s) return 0;
  return 4ll * s->rows() * s->n + 8ll * s->n * s->udim + 4ll * s->n;
}

int topolow_session_stage(topolow_session* s, const void* d_pos_in, void* d_pos_out,
                          int32_t iter, int32_t stage, int32_t n_stages, double k,
                          char* errbuf, size_t errlen) {
  if (!s || !d_pos_in || !d_pos_out) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    const SlabGeom g = slab_geom(s->n, n_stages);
    if (stage < 0 || stage >= g.n_stages)
      throw HipError{TOPOLOW_ERR_BAD_ARGUMENT, "stage index out of range"};
    const SlabRanges rg = slab_ranges(g, s->seed, iter, stage);
    // inside a run (topolow_session_begin) the launch honours the run's stop flag and reports
    // non-finite results through its state, like the session's own loop
    TL_DISPATCH_DIM(s->dim, launch_stage, s, d_pos_in, d_pos_out, s->began ? s->state.p : (RunState*)nullptr, rg,
                    iter + 1, k);
    s->iters_enqueued = std::max(s->iters_enqueued, iter + 1);
  });
}

int32_t topolow_session_can_fuse_checks(const topolow_session* s) {
  return s && s->fuse_checks && s->dense_mae && s->precision == TOPOLOW_PRECISION_F32 && s->rows() % 2 == 0 ? 1 : 0;
}

int topolow_session_stage_fused(topolow_session* s, const void* d_pos_in, void* d_pos_out, int32_t iter,
                                double k, double* d_out2, char* errbuf, size_t errlen) {
  if (!s || !d_pos_in || !d_pos_out || !d_out2 || !s->began) return TOPOLOW_ERR_BAD_ARGUMENT;
  if (!topolow_session_can_fuse_checks(s)) {
    set_err(errbuf, errlen, "this session cannot fuse checks (needs the fp32 block-based MAE and an even row count)");
    return TOPOLOW_ERR_UNSUPPORTED;
  }
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    const SlabGeom g = slab_geom(s->n, 1);
    const SlabRanges rg = slab_ranges(g, s->seed, iter, 0);
    TL_DISPATCH_DIM(s->dim, launch_stage, s, d_pos_in, d_pos_out, s->state.p, rg, iter + 1, k, nullptr, 0, true);
    const int stage_blocks = (s->rows() + CfgProd::ROWS - 1) / CfgProd::ROWS;
    hipLaunchKernelGGL(reduce_total_kernel, dim3(1), dim3(1024), 0, s->stream, s->part_sum.p, s->part_cnt.p,
                       stage_blocks, d_out2, s->state.p);
    HIP_TRY(hipGetLastError());
    s->iters_enqueued = std::max(s->iters_enqueued, iter + 1);
  });
}

int topolow_session_check_partial(topolow_session* s, const void* d_pos, double* d_out2, char* errbuf,
                                  size_t errlen) {
  if (!s || !d_pos || !d_out2 || !s->part_sum.p || !s->began) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    TL_DISPATCH_DIM(s->dim, launch_edge_error, s, d_pos, s->state.p);
    hipLaunchKernelGGL(reduce_total_kernel, dim3(1), dim3(1024), 0, s->stream, s->part_sum.p, s->part_cnt.p,
                       error_parts(s), d_out2, s->state.p);
    HIP_TRY(hipGetLastError());
  });
}

int topolow_session_controller_step(topolow_session* s, const double* d_total2, const void* d_pos,
                                    int32_t iter1, double k_after, char* errbuf, size_t errlen) {
  if (!s || !d_total2 || !d_pos || !s->began) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    launch_controller(s, d_pos, iter1, k_after, nullptr, nullptr, 0, d_total2);
    s->iters_enqueued = std::max(s->iters_enqueued, (int)iter1);
  });
}

int32_t topolow_symm_stage_bounds(int32_t n, int32_t stages, int32_t* first_label) {
  if (n < 2 || !first_label || !(stages == 2 || stages == 4 || stages == 8)) return 0;
  const int npad = (n + kSymRows - 1) & ~(kSymRows - 1), TR = npad / kSymRows;
  if (TR < 2 * stages) return 0;
  for (int q = 0; q <= stages; ++q) first_label[q] = std::min(n, sym_rr_bound(TR, stages, q) * kSymRows);
  return 1;
}
int32_t topolow_symm_stage_order(uint64_t seed, int32_t iter, int32_t stages, int32_t* order) {
  if (!order || !(stages == 2 || stages == 4 || stages == 8)) return 0;
  int perm[8];
  sym_rr_order(seed, iter, stages, perm);
  for (int q = 0; q < stages; ++q) order[q] = perm[q];
  return 1;
}

int32_t topolow_symm_segment_rows(int32_t n, int32_t segment, int32_t n_segments, int32_t* row_first,
                                  int32_t* row_end) {
  if (n < 2 || n_segments < 1 || segment < 0 || segment >= n_segments) return 0;
  const long long npad = ((long long)n + kSymRows - 1) & ~(long long)(kSymRows - 1);
  const long long TR = npad / kSymRows, total = TR * (TR + 1);
  if (total < 8ll * n_segments) return 0;
  int rf = -1, rl = -1;
  (void)relax_symm_plan((int)npad, 1, total * segment / n_segments, total * (segment + 1) / n_segments, &rf, &rl);
  if (rf < 0) return 0;
  if (row_first) *row_first = rf * kSymRows;
  if (row_end) *row_end = (int32_t)std::min<long long>(n, (long long)(rl + 1) * kSymRows);
  return 1;
}

int32_t topolow_session_symm_segment_eligible(const topolow_session* s, int32_t n_segments) {
  return s && s->sym.allowed && s->schedule == TOPOLOW_SCHEDULE_SLAB && s->precision == TOPOLOW_PRECISION_F32 &&
                 s->dim >= 2 && s->dim <= 6 && s->dim == s->udim && s->n >= s->sym.min_n && s->gplus.p != nullptr &&
                 topolow_symm_segment_rows(s->n, 0, n_segments, nullptr, nullptr)
             ? 1 : 0;
}

float* topolow_session_degree_terms(topolow_session* s) { return s ? s->gplus.p : nullptr; }
int32_t topolow_session_has_thresholds(const topolow_session* s) { return s && s->gplus.p && s->any_threshold ? 1 : 0; }
float* topolow_session_symm_moves(topolow_session* s) { return s && s->sym.seg_caller ? s->sym.inbox.p : nullptr; }

int topolow_session_symm_segment_build(topolow_session* s, int32_t segment, int32_t n_segments, const void* d_rows,
                                       int32_t row_first, int32_t n_rows, int32_t any_threshold, char* errbuf,
                                       size_t errlen) {
  if (!s || !d_rows || n_segments < 1 || segment < 0 || segment >= n_segments || n_rows < 0 || row_first < 0 ||
      row_first + n_rows > s->n)
    return TOPOLOW_ERR_BAD_ARGUMENT;
  if (!topolow_session_symm_segment_eligible(s, n_segments)) {
    set_err(errbuf, errlen, "this session cannot take the symmetric sweep (fp32 slab schedule, ndim 2..6, at least %d "
                            "points, targets loaded)", s->sym.min_n);
    return TOPOLOW_ERR_UNSUPPORTED;
  }
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    try {
      TL_DISPATCH_DIM(s->dim, sym_segment_build, s, segment, n_segments, (const uint32_t*)d_rows, row_first, n_rows,
                      any_threshold != 0);
    } catch (const HipError&) {      // e.g. the device cannot hold the extra buffers: the session keeps its row-owner sweep
      auto& y = s->sym;
      y.tenc.release(); y.rowpart.release(); y.colpart.release(); y.inbox.release();
      y.seg_ready = false;
      y.seg_caller = false;
      throw;
    }
  });
}

int topolow_session_symm_segment_sweep(topolow_session* s, const void* d_pos_in, int32_t iter, double k,
                                       double* d_out2, char* errbuf, size_t errlen) {
  if (!s || !d_pos_in || !s->began) return TOPOLOW_ERR_BAD_ARGUMENT;
  if (!(s->sym.seg_ready && s->sym.seg_caller)) {
    set_err(errbuf, errlen, "no segment built (topolow_session_symm_segment_build)");
    return TOPOLOW_ERR_BAD_ARGUMENT;
  }
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    TL_DISPATCH_DIM(s->dim, sym_sharded_sweep, s, d_pos_in, iter, k, d_out2 != nullptr);
    if (d_out2 != nullptr) {
      hipLaunchKernelGGL(reduce_total_kernel, dim3(1), dim3(1024), 0, s->stream, s->part_sum.p, s->part_cnt.p,
                         s->sym.tiles > 0 ? s->sym.n_units : 0, d_out2, s->state.p);
      HIP_TRY(hipGetLastError());
    }
    s->iters_enqueued = std::max(s->iters_enqueued, iter + 1);
  });
}

int topolow_session_symm_segment_apply(topolow_session* s, const void* d_pos_in, void* d_pos_out, int32_t iter,
                                       char* errbuf, size_t errlen) {
  if (!s || !d_pos_in || !d_pos_out || !s->began || !(s->sym.seg_ready && s->sym.seg_caller))
    return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    TL_DISPATCH_DIM(s->dim, sym_segment_apply, s, d_pos_in, d_pos_out, iter);
  });
}

int topolow_session_first_nonfinite(topolow_session* s, int32_t* iteration) {
  if (!s || !iteration) return TOPOLOW_ERR_BAD_ARGUMENT;
  (void)hipSetDevice(s->device);
  (void)hipStreamSynchronize(s->stream);
  RunState st;
  if (hipMemcpy(&st, s->state.p, sizeof st, hipMemcpyDeviceToHost) != hipSuccess) return TOPOLOW_ERR_HIP;
  *iteration = st.first_nonfinite == 0x7fffffff ? 0 : st.first_nonfinite;
  return TOPOLOW_OK;
}

int topolow_session_edge_error(topolow_session* s, const void* d_pos, double* sum,
                               int64_t* count, char* errbuf, size_t errlen) {
  if (!s || !d_pos || !s->part_sum.p) return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    HIP_TRY(hipSetDevice(s->device));
    TL_DISPATCH_DIM(s->dim, launch_edge_error, s, d_pos, (const RunState*)nullptr);
    const int np = error_parts(s);
    std::vector<double> ps(np);
    std::vector<unsigned long long> pc(np);
    HIP_TRY(hipMemcpyAsync(ps.data(), s->part_sum.p, np * 8, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(pc.data(), s->part_cnt.p, np * 8, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    double ts = 0.0;
    unsigned long long tc = 0;
    for (int p = 0; p < np; ++p) { ts += ps[p]; tc += pc[p]; }
    if (sum) *sum = ts;
    if (count) *count = (int64_t)tc;
  });
}

// ---- batch of independent embeddings (GS kernel, one workgroup each) --------------------
int topolow_optimize_layout_exact_batch(const topolow_problem* problems, topolow_result* results,
                                        int32_t count, int32_t precision, int32_t device,
                                        double* device_seconds, char* errbuf, size_t errlen) {
  if (count < 0 || (count > 0 && (!problems || !results))) return TOPOLOW_ERR_BAD_ARGUMENT;
  if (device_seconds) *device_seconds = 0.0;
  if (count == 0) return TOPOLOW_OK;
  const int prec = precision == TOPOLOW_PRECISION_F32 ? TOPOLOW_PRECISION_F32 : TOPOLOW_PRECISION_F64;
  int rc_all = TOPOLOW_OK;
  const int rc = guarded(errbuf, errlen, [&] {
    select_device(device);
    // The kernel is instantiated per ndim: one grid per distinct ndim.  The grids are independent and run
    // SIDE BY SIDE on their own streams (a sweep over ndim 2..10 would otherwise run nine under-filled grids
    // back to back, each with its own tail of slow embeddings); a grid is launched as soon as it is staged,
    // the costliest (largest ndim) first, so the host's staging of the next grid hides behind the device's
    // work on the previous ones.
    std::vector<int> dims;
    for (int b = 0; b < count; ++b) {
      if (problems[b].n < 2) throw HipError{TOPOLOW_ERR_TOO_FEW_POINTS, "Need at least 2 points for embedding"};
      if (std::find(dims.begin(), dims.end(), problems[b].ndim) == dims.end()) dims.push_back(problems[b].ndim);
    }
    std::sort(dims.begin(), dims.end(), std::greater<int>());
    struct Grid {
      std::vector<GsProblem> pbs;
      std::vector<GsResult> res;
      std::vector<int> idx;
      std::unique_ptr<GsBatchBase> batch;
      hipStream_t stream = nullptr;
      hipEvent_t done = nullptr;
    };
    std::vector<Grid> grids(dims.size());
    hipEvent_t start = nullptr;
    auto release = [&] {
      for (Grid& g : grids) {
        if (g.stream) { (void)hipStreamSynchronize(g.stream); (void)hipStreamDestroy(g.stream); }
        if (g.done) (void)hipEventDestroy(g.done);
        g.batch.reset();
      }
      if (start) (void)hipEventDestroy(start);
    };
    try {
      for (size_t q = 0; q < dims.size(); ++q) {
        Grid& g = grids[q];
        for (int b = 0; b < count; ++b) {
          const topolow_problem& p = problems[b];
          if (p.ndim != dims[q]) continue;
          GsProblem pb;
          pb.initial_positions = p.initial_positions; pb.D = p.dissimilarity_matrix; pb.T = p.threshold_matrix;
          pb.degrees = p.degrees; pb.edge_i = p.edge_i; pb.edge_j = p.edge_j; pb.edge_dist = p.edge_dist;
          pb.edge_thresh = p.edge_thresh; pb.n_edges = p.n_edges; pb.n = p.n; pb.dim = p.ndim;
          pb.n_iter = p.n_iter; pb.window = p.convergence_window; pb.check_freq = p.convergence_check_freq;
          pb.k0 = p.k0; pb.cooling = p.cooling_rate; pb.c_rep = p.c_repulsion; pb.eps = p.relative_epsilon;
          pb.seed = p.seed;
          pb.hold_i = p.holdout_i; pb.hold_j = p.holdout_j; pb.hold_truth = p.holdout_truth;
          pb.n_hold = (p.holdout_i && p.holdout_j && p.holdout_truth) ? p.n_holdout : 0;
          GsResult r;
          r.positions = results[b].positions_out;
          g.pbs.push_back(pb); g.res.push_back(r); g.idx.push_back(b);
        }
        g.batch.reset(gs_new_batch(prec));
        g.batch->stage(g.pbs.data(), (int)g.pbs.size());
        HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&g.done));
        if (q == 0) {
          HIP_TRY(hipEventCreate(&start));
          HIP_TRY(hipEventRecord(start, g.stream));
        }
        g.batch->launch(g.stream);
        HIP_TRY(hipEventRecord(g.done, g.stream));
      }
      double secs = 0.0;   // first launch -> last completion (the later grids' staging runs inside this span)
      for (Grid& g : grids) {
        HIP_TRY(hipEventSynchronize(g.done));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, start, g.done));
        secs = std::max(secs, (double)ms * 1e-3);
      }
      if (device_seconds) *device_seconds = secs;
      for (Grid& g : grids) {
        char local_err[256];
        local_err[0] = 0;
        const int rcb = g.batch->collect(g.res.data(), local_err, sizeof local_err);
        if (rcb != TOPOLOW_OK && rcb != TOPOLOW_ERR_NONFINITE) throw HipError{rcb, local_err};
        for (size_t q = 0; q < g.idx.size(); ++q) {
          topolow_result& o = results[g.idx[q]];
          const GsResult& r = g.res[q];
          o.final_mae = r.final_mae; o.final_k = r.final_k; o.converged = r.converged;
          o.iterations = r.iterations; o.iterations_run = r.iters_run; o.n_checks = r.n_checks;
          o.error_code = r.nonfinite_iter ? TOPOLOW_ERR_NONFINITE : TOPOLOW_OK;
          o.error_iteration = r.nonfinite_iter;
          o.holdout_sum_abs = r.hold_sum;
          o.holdout_count = r.hold_count;
        }
      }
    } catch (const GsHipError& e) {
      release();
      throw HipError{e.code, e.msg};
    } catch (...) {
      release();
      throw;
    }
    release();
  });
  return rc != TOPOLOW_OK ? rc : rc_all;
}

// ---- post metric -----------------------------------------------------------------------
int topolow_cell_list_index(int32_t n, int64_t n_cells, const int32_t* row, const int32_t* col,
                            int64_t* pos_of, int64_t* by_row, int64_t* row_ptr) {
  if (n < 1 || n_cells < 0 || !pos_of || !by_row || !row_ptr || (n_cells > 0 && (!row || !col)))
    return TOPOLOW_ERR_BAD_ARGUMENT;
  for (int64_t q = 0; q < (int64_t)n * n; ++q) pos_of[q] = -1;
  for (int i = 0; i <= n; ++i) row_ptr[i] = 0;
  for (int64_t c = 0; c < n_cells; ++c) {
    if (row[c] < 0 || row[c] >= n || col[c] < 0 || col[c] >= n) return TOPOLOW_ERR_BAD_ARGUMENT;
    pos_of[(int64_t)row[c] + (int64_t)col[c] * n] = c;   // linear column-major index, as R's which()
    row_ptr[row[c] + 1] += 1;
  }
  for (int i = 0; i < n; ++i) row_ptr[i + 1] += row_ptr[i];
  try {
    // counting sort by row; the listing is column-major, so columns ascend inside every row
    std::vector<int64_t> at(row_ptr, row_ptr + n);
    for (int64_t c = 0; c < n_cells; ++c) by_row[at[row[c]]++] = c;
  } catch (const std::bad_alloc&) {
    return TOPOLOW_ERR_HIP;
  }
  return TOPOLOW_OK;
}

int topolow_cv_fold(const topolow_cell_list* cells, const int64_t* picks, int64_t n_picks,
                    int32_t preserve_order, int32_t named, int32_t* order, int32_t* degrees,
                    int32_t* edge_i, int32_t* edge_j, double* edge_dist, int32_t* edge_thresh,
                    int64_t* n_edges, int32_t* holdout_i, int32_t* holdout_j, double* holdout_truth,
                    int64_t* n_holdout, double* numeric_max) {
  if (!cells || (!picks && n_picks > 0) || !order || !degrees || !edge_i || !edge_j || !edge_dist ||
      !edge_thresh || !n_edges || !holdout_i || !holdout_j || !holdout_truth || !n_holdout || !numeric_max)
    return TOPOLOW_ERR_BAD_ARGUMENT;
  try {
    return fold_problem(cells, picks, n_picks, preserve_order, named, order, degrees, edge_i, edge_j,
                        edge_dist, edge_thresh, n_edges, holdout_i, holdout_j, holdout_truth, n_holdout,
                        numeric_max);
  } catch (const std::bad_alloc&) {
    return TOPOLOW_ERR_HIP;
  }
}

// All folds of a cross-validation sweep in ONE call: the folds' problems are built side by side on host threads
// (fold_problem; start positions from the caller's unit draws with NumPy's / R's arithmetic: a random walk whose
// steps are uniform(0, 2 max / n), R/core.R:407-415) and relaxed as one batch; only the per-fold scores come back.
int topolow_cv_sweep(const topolow_cell_list* cells, int32_t named, int32_t preserve_order, int32_t n_folds,
                     const int32_t* ndim, const double* k0, const double* cooling_rate, const double* c_repulsion,
                     const int64_t* picks, const int64_t* picks_offset, const double* unit_draws,
                     const int64_t* draws_offset, const uint64_t* seeds, int32_t n_iter, double relative_epsilon,
                     int32_t convergence_window, int32_t convergence_check_freq, int32_t precision, int32_t device,
                     double* holdout_sum_abs, int64_t* holdout_count, int32_t* iterations, int32_t* converged,
                     int32_t* error_code, double* device_seconds, char* errbuf, size_t errlen) {
  if (!cells || n_folds < 0 || (n_folds > 0 && (!ndim || !k0 || !cooling_rate || !c_repulsion || !picks_offset ||
      !unit_draws || !draws_offset || !seeds || !holdout_sum_abs || !holdout_count || !iterations || !converged ||
      !error_code)))
    return TOPOLOW_ERR_BAD_ARGUMENT;
  if (device_seconds) *device_seconds = 0.0;
  if (n_folds == 0) return TOPOLOW_OK;
  const int n = cells->n;
  const size_t m = (size_t)cells->n_cells;
  struct Fold {
    std::vector<int32_t> order, deg, ei, ej, et, hi, hj;
    std::vector<double> ed, ht, pos, out;
    int64_t ne = 0, nh = 0;
    int rc = TOPOLOW_OK;
  };
  std::vector<Fold> F((size_t)n_folds);
  try {
    gs_parallel_for(n_folds, [&](int f) {
      Fold& x = F[(size_t)f];
      x.order.resize(n); x.deg.resize(n);
      x.ei.resize(m); x.ej.resize(m); x.et.resize(m); x.ed.resize(m);
      x.hi.resize(m); x.hj.resize(m); x.ht.resize(m);
      double vmax = 0.0;
      x.rc = fold_problem(cells, picks + picks_offset[f], picks_offset[f + 1] - picks_offset[f], preserve_order, named,
                          x.order.data(), x.deg.data(), x.ei.data(), x.ej.data(), x.ed.data(), x.et.data(), &x.ne,
                          x.hi.data(), x.hj.data(), x.ht.data(), &x.nh, &vmax);
      if (x.rc == TOPOLOW_OK && (x.ne == 0 || !(vmax == vmax))) x.rc = TOPOLOW_ERR_BAD_ARGUMENT;   // no valid measurements
      if (x.rc != TOPOLOW_OK) return;
      const int d_ = ndim[f];
      if (d_ < 1 || draws_offset[f + 1] - draws_offset[f] != (int64_t)d_ * (n - 1)) { x.rc = TOPOLOW_ERR_BAD_ARGUMENT; return; }
      const double* u = unit_draws + draws_offset[f];             // (ndim, n - 1), row-major
      const double step = vmax / (double)n;
      x.pos.assign((size_t)n * d_, 0.0);                          // column-major n x ndim
      x.out.assign((size_t)n * d_, 0.0);
      for (int d = 0; d < d_; ++d) {
        double acc = 0.0;
        for (int i = 1; i < n; ++i) {
          const double st_ = 0.0 + (2.0 * step - 0.0) * u[(size_t)d * (n - 1) + (i - 1)];   // Generator.uniform's arithmetic
          acc = i == 1 ? st_ : acc + st_;                         // cumsum
          x.pos[(size_t)i + (size_t)d * n] = acc;
        }
      }
    });
  } catch (const GsHipError& e) {
    set_err(errbuf, errlen, "%s", e.msg.c_str());
    return e.code;
  } catch (const std::bad_alloc&) {
    set_err(errbuf, errlen, "out of host memory");
    return TOPOLOW_ERR_HIP;
  }
  std::vector<topolow_problem> P;
  std::vector<topolow_result> R;
  std::vector<int> idx;
  for (int f = 0; f < n_folds; ++f) {
    Fold& x = F[(size_t)f];
    holdout_sum_abs[f] = 0.0; holdout_count[f] = 0; iterations[f] = 0; converged[f] = 0;
    error_code[f] = x.rc;
    if (x.rc != TOPOLOW_OK) continue;
    topolow_problem p;
    std::memset(&p, 0, sizeof p);
    p.initial_positions = x.pos.data();
    p.degrees = x.deg.data();
    p.edge_i = x.ei.data(); p.edge_j = x.ej.data(); p.edge_dist = x.ed.data(); p.edge_thresh = x.et.data();
    p.n_edges = x.ne; p.n = n; p.ndim = ndim[f]; p.n_iter = n_iter;
    p.convergence_window = convergence_window; p.convergence_check_freq = convergence_check_freq;
    p.k0 = k0[f]; p.cooling_rate = cooling_rate[f]; p.c_repulsion = c_repulsion[f]; p.relative_epsilon = relative_epsilon;
    p.seed = seeds[f];
    if (x.nh > 0) { p.holdout_i = x.hi.data(); p.holdout_j = x.hj.data(); p.holdout_truth = x.ht.data(); p.n_holdout = x.nh; }
    topolow_result r;
    std::memset(&r, 0, sizeof r);
    r.positions_out = x.out.data();
    P.push_back(p); R.push_back(r); idx.push_back(f);
  }
  if (P.empty()) return TOPOLOW_OK;
  const int rc = topolow_optimize_layout_exact_batch(P.data(), R.data(), (int32_t)P.size(), precision, device, device_seconds,
                                                     errbuf, errlen);
  if (rc != TOPOLOW_OK) return rc;
  for (size_t q = 0; q < idx.size(); ++q) {
    const int f = idx[q];
    error_code[f] = R[q].error_code;
    holdout_sum_abs[f] = R[q].holdout_sum_abs; holdout_count[f] = R[q].holdout_count;
    iterations[f] = R[q].iterations; converged[f] = R[q].converged;
  }
  return TOPOLOW_OK;
}

// rows [row_begin, row_end) of as.matrix(dist(positions)): out is (row_end - row_begin) x n, row-major.
// Device memory is bounded: the rows are produced in tiles of at most 256 MB.
int topolow_est_distances_rows(const double* positions, int32_t n, int32_t ndim, int32_t row_begin,
                               int32_t row_end, double* out, int32_t device, char* errbuf, size_t errlen) {
  if (!positions || !out || n < 1 || ndim < 1 || row_begin < 0 || row_end > n || row_begin > row_end)
    return TOPOLOW_ERR_BAD_ARGUMENT;
  return guarded(errbuf, errlen, [&] {
    select_device(device);
    std::vector<double> rowmajor((size_t)n * ndim);
    for (int i = 0; i < n; ++i)
      for (int d = 0; d < ndim; ++d) rowmajor[(size_t)i * ndim + d] = positions[i + (size_t)d * n];
    DevBuf<double> dp, dout;
    dp.alloc(rowmajor.size());
    HIP_TRY(hipMemcpy(dp.p, rowmajor.data(), rowmajor.size() * 8, hipMemcpyHostToDevice));
    const int tile = std::max(1, std::min(row_end - row_begin, (int)((256ll << 20) / (8ll * n))));
    dout.alloc((size_t)tile * n);
    for (int r0 = row_begin; r0 < row_end; r0 += tile) {
      const int rows = std::min(tile, row_end - r0);
      dim3 grid(rows, (n + kThreads - 1) / kThreads);
      hipLaunchKernelGGL(pdist_kernel, grid, dim3(kThreads), 0, 0, dp.p, n, ndim, r0, rows, dout.p);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpy(out + (size_t)(r0 - row_begin) * n, dout.p, (size_t)rows * n * 8, hipMemcpyDeviceToHost));
    }
  });
}

int topolow_est_distances(const double* positions, int32_t n, int32_t ndim,
                          double* est_distances, int32_t device, char* errbuf, size_t errlen) {
  return topolow_est_distances_rows(positions, n, ndim, 0, n, est_distances, device, errbuf, errlen);
}

// ---- ONE embedding row-sharded over several sessions (one process, one host thread per block) ----
int topolow_sessions_run_sharded(topolow_session** sessions, int32_t count, const double* initial_positions,
                                 int32_t n_iter, double k0, double cooling_rate, double c_repulsion,
                                 double relative_epsilon, int32_t convergence_window,
                                 int32_t convergence_check_freq, uint64_t seed, int32_t slab_stages,
                                 int32_t (*interrupt_cb)(void*), void* interrupt_user, int32_t profile,
                                 double* positions_out, int32_t* converged, int32_t* iterations,
                                 double* final_mae, double* final_k, topolow_shard_stats* stats, char* errbuf,
                                 size_t errlen) {
  if (!sessions || count < 1 || !initial_positions) return TOPOLOW_ERR_BAD_ARGUMENT;
  const int warmup_iters = stats != nullptr ? stats->warmup_iterations : 0;   // in: see the header
  int rc_extra = TOPOLOW_OK;
  const int rc = guarded(errbuf, errlen, [&] {
    ShardedRun R;
    R.P = count;
    R.ss.assign(sessions, sessions + count);
    const int n = R.ss[0]->n;
    int expect = 0;
    for (topolow_session* s : R.ss) {
      if (!s || s->n != n || s->dim != R.ss[0]->dim || s->precision != R.ss[0]->precision ||
          s->row_begin != expect || s->schedule != TOPOLOW_SCHEDULE_SLAB)
        throw HipError{TOPOLOW_ERR_BAD_ARGUMENT,
                       "row-sharded run: the sessions must be slab-schedule row blocks that tile [0, n) in order"};
      expect = s->row_end;
    }
    if (expect != n) throw HipError{TOPOLOW_ERR_BAD_ARGUMENT, "row-sharded run: the row blocks do not cover all n rows"};
    sharded_wire(R.ss);
    if (sym_sharded_eligible(R.ss)) {
      // one-stage iterations as the symmetric sweep sharded over the sessions; a device that cannot hold the extra
      // buffers (half a row block again per session) keeps the row-owner sweep
      try {
        sym_sharded_prepare(R.ss);
        R.pair_sharded = true;
      } catch (const HipError&) {
        (void)hipGetLastError();
        for (topolow_session* s : R.ss) {
          auto& y = s->sym;
          y.tenc.release(); y.rec[0].release(); y.rec[1].release(); y.rowpart.release(); y.colpart.release();
          y.units.release(); y.wave_first.release(); y.row_units.release(); y.inbox.release();
          y.ready = false; y.seg_ready = false;
        }
        R.pair_sharded = false;
      }
    }
    // groups: blocks that share a GPU share one stream and one host thread
    const char* per_block = getenv("TOPOLOW_SHARD_THREAD_PER_BLOCK");
    const bool thread_per_block = per_block != nullptr && per_block[0] == '1';
    for (int b = 0; b < count; ++b) {
      int gidx = -1;
      if (!thread_per_block)
        for (size_t q = 0; q < R.groups.size(); ++q) if (R.groups[q].device == R.ss[b]->device) gidx = (int)q;
      if (gidx < 0) {
        R.groups.emplace_back();
        gidx = (int)R.groups.size() - 1;
        R.groups[gidx].device = R.ss[b]->device;
        R.groups[gidx].stream = R.ss[b]->own_stream;
      }
      R.groups[gidx].blocks.push_back(b);
    }
    const int n_groups = (int)R.groups.size();
    for (int b = 0; b < count; ++b) {
      topolow_session* s = R.ss[b];
      int rcb = topolow_session_set_stream(s, nullptr, 0);
      if (rcb == TOPOLOW_OK) rcb = topolow_session_set_positions(s, initial_positions, errbuf, errlen);
      if (rcb == TOPOLOW_OK)
        rcb = topolow_session_begin(s, n_iter, k0, cooling_rate, c_repulsion, relative_epsilon, convergence_window,
                                    convergence_check_freq, seed, slab_stages, errbuf, errlen);
      if (rcb != TOPOLOW_OK) throw HipError{rcb, errbuf ? errbuf : "session setup failed"};
      s->profiling = false;
    }
    for (const ShardedGroup& G : R.groups)
      for (int b : G.blocks) R.ss[b]->stream = G.stream;   // (idle: set_positions / begin have synchronised)
    R.n_iter = n_iter;
    R.check_freq = convergence_check_freq < 1 ? 10 : convergence_check_freq;
    R.k0 = k0;
    R.cooling = cooling_rate;
    R.fixed_stages = slab_stages;
    R.interrupt_cb = interrupt_cb;
    R.interrupt_user = interrupt_user;
    R.flag[0].store(0);
    R.flag[1].store(0);
    ShardedAbortableBarrier bar(n_groups);
    R.bar = &bar;
    R.ev.resize(n_groups);
    for (auto& e : R.ev) e = {nullptr, nullptr};
    auto cleanup = [&] {
      for (int q = 0; q < n_groups; ++q) {
        (void)hipSetDevice(R.groups[q].device);
        (void)hipStreamSynchronize(R.groups[q].stream);
        for (hipEvent_t e : R.ev[q]) if (e) (void)hipEventDestroy(e);
      }
      for (topolow_session* s : R.ss) s->stream = s->own_stream;
    };
    try {
      for (int q = 0; q < n_groups; ++q) {
        HIP_TRY(hipSetDevice(R.groups[q].device));
        for (int e = 0; e < 2; ++e) HIP_TRY(hipEventCreateWithFlags(&R.ev[q][e], hipEventDisableTiming));
      }
      R.warmup_iters = warmup_iters;
      if (profile)   // the kernels of the first GPU's blocks are bracketed by timing events
        for (int b : R.groups[0].blocks) R.ss[b]->profiling = true;
      const long long launches0 = R.ss[0]->stage_launches;
      const double t0 = now_s();
      auto body = [&](int r) {
        try {
          sharded_worker(R, r);
        } catch (const HipError& e) {
          std::lock_guard<std::mutex> lock(R.err_mu);
          if (R.first_error.code == TOPOLOW_OK) R.first_error = e;
          bar.fail();
        } catch (const std::exception& e) {
          std::lock_guard<std::mutex> lock(R.err_mu);
          if (R.first_error.code == TOPOLOW_OK) R.first_error = HipError{TOPOLOW_ERR_HIP, e.what()};
          bar.fail();
        }
      };
      std::vector<std::thread> pool;
      for (int r = 1; r < n_groups; ++r) pool.emplace_back(body, r);
      body(0);   // the calling thread is the first GPU's thread (the interrupt callback runs here)
      for (auto& t : pool) t.join();
      if (R.first_error.code != TOPOLOW_OK) throw R.first_error;
      const double wall = now_s() - t0;
      // result: every block holds the same controller state and the same best snapshot
      topolow_session* s0 = R.ss[0];
      HIP_TRY(hipSetDevice(s0->device));
      RunState st;
      HIP_TRY(hipMemcpy(&st, s0->state.p, sizeof st, hipMemcpyDeviceToHost));
      int first_bad = 0x7fffffff;
      for (topolow_session* s : R.ss) {
        HIP_TRY(hipSetDevice(s->device));
        RunState sb;
        HIP_TRY(hipMemcpy(&sb, s->state.p, sizeof sb, hipMemcpyDeviceToHost));
        first_bad = std::min(first_bad, sb.first_nonfinite);
      }
      HIP_TRY(hipSetDevice(s0->device));
      const int ran = st.stopped ? st.iter_base : R.iters_enqueued;
      if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->blocks = count;
        stats->groups = n_groups;
        stats->iterations_run = ran;
        stats->n_checks = st.n_checks;
        stats->loop_seconds = wall;
        stats->stage_launches = s0->stage_launches - launches0;
        stats->exchanges = R.exchanges;
        stats->warmup_iterations = warmup_iters;
        stats->symmetric_segments = R.pair_sharded ? count : 0;
        stats->timed_seconds = (warmup_iters > 0 && R.t_timed0 > 0.0) ? (t0 + wall) - R.t_timed0 : wall;
        if (profile) {
          for (int b : R.groups[0].blocks) {
            double sm = 0, cm = 0;
            int64_t sl = 0, cl = 0;
            (void)topolow_session_profile(R.ss[b], &sm, &sl, &cm, &cl, nullptr, 0);
            double ya = 0, yb = 0;    // a single block's one-stage iterations may have run as symmetric sweeps
            int64_t na = 0, nb = 0;
            (void)topolow_session_profile_symmetric(R.ss[b], &ya, &na, &yb, &nb, nullptr, 0);
            stats->stage_kernel_seconds += (sm + ya + yb) * 1e-3;
            stats->check_kernel_seconds += cm * 1e-3;
          }
        }
      }
      for (topolow_session* s : R.ss) s->profiling = false;
      if (R.interrupted) {
        set_err(errbuf, errlen, "interrupted by the caller");
        rc_extra = TOPOLOW_ERR_INTERRUPTED;
      } else if (first_bad != 0x7fffffff && ((first_bad + 9) / 10) * 10 <= ran &&
                 !(st.stopped && ((first_bad + 9) / 10) * 10 == ran)) {   // reference :359-361
        set_err(errbuf, errlen, "Numerical instability at iteration %d. Reduce k0 or c_repulsion.",
                ((first_bad + 9) / 10) * 10);
        rc_extra = TOPOLOW_ERR_NONFINITE;
      } else {
        if (positions_out) download_positions(s0, s0->best.p, positions_out);
        if (converged) *converged = st.converged;
        if (iterations) *iterations = st.ctl.best_iter;
        if (final_mae) *final_mae = st.ctl.best_mae;
        if (final_k) *final_k = st.ctl.best_k;
      }
    } catch (...) {
      cleanup();
      throw;
    }
    cleanup();
  });
  return rc != TOPOLOW_OK ? rc : rc_extra;
}

int32_t topolow_shard_rows(int32_t n, int32_t blocks, int32_t block, int32_t* row_begin, int32_t* row_end) {
  if (n < 1 || blocks < 1) return 0;
  int per = (n + blocks - 1) / blocks;
  per = (per + 7) & ~7;                       // whole workgroups of 8 rows
  const int used = (n + per - 1) / per;       // blocks that hold at least one row
  if (block >= 0 && block < used) {
    if (row_begin) *row_begin = block * per;
    if (row_end) *row_end = std::min(n, (block + 1) * per);
  } else {
    if (row_begin) *row_begin = n;
    if (row_end) *row_end = n;
  }
  return used;
}

int topolow_optimize_layout_exact_sharded(
    const double* initial_positions, int32_t n, int32_t ndim, const double* dissimilarity_matrix,
    const int32_t* threshold_matrix, const int32_t* degrees, const int32_t* edge_i, const int32_t* edge_j,
    const double* edge_dist, const int32_t* edge_thresh, int64_t n_edges, int32_t n_iter, double k0,
    double cooling_rate, double c_repulsion, double relative_epsilon, int32_t convergence_window,
    int32_t convergence_check_freq, int32_t verbose, const topolow_options* opt_in, double* positions_out,
    int32_t* converged, int32_t* iterations, double* final_mae, double* final_k, topolow_shard_stats* stats,
    char* errbuf, size_t errlen) {
  if (n < 2) {  // reference :131
    set_err(errbuf, errlen, "Need at least 2 points for embedding");
    return TOPOLOW_ERR_TOO_FEW_POINTS;
  }
  if (!initial_positions || !degrees || !positions_out || !converged || !iterations || !final_mae || !final_k ||
      ((dissimilarity_matrix == nullptr) != (threshold_matrix == nullptr)) || n_edges < 0 ||
      (n_edges > 0 && (!edge_i || !edge_j || !edge_dist || !edge_thresh))) {
    set_err(errbuf, errlen, "null argument");
    return TOPOLOW_ERR_BAD_ARGUMENT;
  }
  topolow_options opt;
  if (opt_in) opt = *opt_in; else topolow_default_options(&opt);
  if (opt.schedule == TOPOLOW_SCHEDULE_GS) {
    set_err(errbuf, errlen, "the row-sharded path runs the slab schedule; exact Gauss-Seidel is a one-GPU schedule");
    return TOPOLOW_ERR_UNSUPPORTED;
  }
  const int want = opt.n_devices > 0 ? opt.n_devices : 1;
  const int blocks = topolow_shard_rows(n, want, -1, nullptr, nullptr);
  const int precision = opt.precision == TOPOLOW_PRECISION_F64 ? TOPOLOW_PRECISION_F64 : TOPOLOW_PRECISION_F32;
  const double t_start = now_s();
  std::vector<topolow_session*> ss(blocks, nullptr);
  int rc = TOPOLOW_OK;
  for (int b = 0; b < blocks && rc == TOPOLOW_OK; ++b) {
    int rb = 0, re = 0;
    topolow_shard_rows(n, want, b, &rb, &re);
    const int dev = opt.devices ? opt.devices[b % want] : (opt.n_devices > 1 ? b : opt.device);
    rc = topolow_session_create(&ss[b], n, ndim, rb, re, precision, dev, errbuf, errlen);
    if (rc) break;
    if (!opt.keep_labels) {   // every block draws the same permutation (same n, same seed)
      rc = topolow_session_set_relabel(ss[b], mix64(opt.seed ^ 0x1abe15eedull) | 1ull, errbuf, errlen);
      if (rc) break;
    }
    if (dissimilarity_matrix)
      rc = topolow_session_load_dense(ss[b], dissimilarity_matrix, threshold_matrix, degrees, errbuf, errlen);
    else
      rc = topolow_session_load_coo(ss[b], edge_i, edge_j, edge_dist, edge_thresh, n_edges, degrees, errbuf, errlen);
    if (rc) break;
    // this block's share of the convergence MAE: pair {lo, hi} belongs to the owner of lo when lo + hi
    // is even, of hi when it is odd (every block then reduces about half of its row block's pairs)
    try {
      std::vector<int32_t> bi, bj, bt;
      std::vector<double> bd;
      const int* inv = ss[b]->inv.empty() ? nullptr : ss[b]->inv.data();   // ownership is by session label
      for (int64_t e = 0; e < n_edges; ++e) {
        const int a = edge_i[e], c = edge_j[e];
        if (a < 0 || c < 0 || a >= n || c >= n) continue;
        const int sa = inv ? inv[a] : a, sc = inv ? inv[c] : c;
        const int lo = sa < sc ? sa : sc, hi = sa < sc ? sc : sa;
        const int owner = blocks == 1 ? lo : ((((lo + hi) & 1) == 0) ? lo : hi);
        if (owner >= rb && owner < re) { bi.push_back(a); bj.push_back(c); bd.push_back(edge_dist[e]); bt.push_back(edge_thresh[e]); }
      }
      rc = topolow_session_set_edges(ss[b], bi.data(), bj.data(), bd.data(), bt.data(), (int64_t)bi.size(), errbuf, errlen);
    } catch (const std::bad_alloc&) {
      set_err(errbuf, errlen, "out of host memory");
      rc = TOPOLOW_ERR_HIP;
    }
  }
  if (rc == TOPOLOW_OK) {
    if (verbose) {
      char what[64];
      snprintf(what, sizeof what, "row-owner slabs, %d row blocks", blocks);
      emit_header(opt, what, n, k0, cooling_rate, c_repulsion);
    }
    rc = topolow_sessions_run_sharded(ss.data(), blocks, initial_positions, n_iter, k0, cooling_rate, c_repulsion,
                                      relative_epsilon, convergence_window, convergence_check_freq, opt.seed,
                                      opt.slab_stages, opt.interrupt_cb, opt.interrupt_user, stats != nullptr,
                                      positions_out, converged, iterations, final_mae, final_k, stats, errbuf, errlen);
    if (rc == TOPOLOW_OK && verbose) {
      int nc = 0;
      if (topolow_session_check_trace(ss[0], nullptr, 0, &nc) == TOPOLOW_OK && nc > 0) {
        std::vector<double> trace(3 * (size_t)nc);
        if (topolow_session_check_trace(ss[0], trace.data(), nc, &nc) == TOPOLOW_OK)
          emit_checks(opt, trace.data(), 0, nc, n_iter);
      }
      if (*converged)
        emit_converged(opt, ss[0]->mailbox->ctl.plateau >= ss[0]->mailbox->ctl.window, *iterations, *final_mae);
    }
  }
  for (topolow_session* s : ss) topolow_session_destroy(s);
  if (rc == TOPOLOW_OK && stats) stats->total_seconds = now_s() - t_start;
  return rc;
}

// ---- the .Call payload -------------------------------------------------------------------
int topolow_optimize_layout_exact(
    const double* initial_positions, int32_t n, int32_t ndim,
    const double* dissimilarity_matrix, const int32_t* threshold_matrix,
    const int32_t* degrees, const int32_t* edge_i, const int32_t* edge_j,
    const double* edge_dist, const int32_t* edge_thresh, int64_t n_edges, int32_t n_iter,
    double k0, double cooling_rate, double c_repulsion, double relative_epsilon,
    int32_t convergence_window, int32_t convergence_check_freq, int32_t verbose,
    const topolow_options* opt_in, double* positions_out, int32_t* converged,
    int32_t* iterations, double* final_mae, double* final_k, topolow_run_stats* stats,
    char* errbuf, size_t errlen) {
  if (n < 2) {  // reference :131
    set_err(errbuf, errlen, "Need at least 2 points for embedding");
    return TOPOLOW_ERR_TOO_FEW_POINTS;
  }
  if (!initial_positions || !dissimilarity_matrix || !threshold_matrix || !degrees ||
      !positions_out || !converged || !iterations || !final_mae || !final_k ||
      (n_edges > 0 && (!edge_i || !edge_j || !edge_dist || !edge_thresh)) || n_edges < 0) {
    set_err(errbuf, errlen, "null argument");
    return TOPOLOW_ERR_BAD_ARGUMENT;
  }
  topolow_options opt;
  if (opt_in) opt = *opt_in; else topolow_default_options(&opt);
  const double t_start = now_s();
  if (opt.n_devices > 1 || opt.devices != nullptr) {   // ONE embedding over several GPUs / row blocks
    topolow_shard_stats sh;
    const int rcs = topolow_optimize_layout_exact_sharded(
        initial_positions, n, ndim, dissimilarity_matrix, threshold_matrix, degrees, edge_i, edge_j, edge_dist,
        edge_thresh, n_edges, n_iter, k0, cooling_rate, c_repulsion, relative_epsilon, convergence_window,
        convergence_check_freq, verbose, &opt, positions_out, converged, iterations, final_mae, final_k,
        stats ? &sh : nullptr, errbuf, errlen);
    if (rcs == TOPOLOW_OK && stats) {
      std::memset(stats, 0, sizeof *stats);
      stats->schedule_used = TOPOLOW_SCHEDULE_SLAB;
      stats->precision_used = opt.precision == TOPOLOW_PRECISION_F64 ? TOPOLOW_PRECISION_F64 : TOPOLOW_PRECISION_F32;
      stats->iterations_run = sh.iterations_run;
      stats->n_checks = sh.n_checks;
      stats->device_seconds = sh.loop_seconds;
      stats->total_seconds = sh.total_seconds;
      stats->stage_launches = sh.stage_launches;
    }
    return rcs;
  }

  int schedule = opt.schedule;
  const int gs_max_n = opt.gs_max_n > 0 ? opt.gs_max_n : kDefaultGsMaxN;
  if (schedule == TOPOLOW_SCHEDULE_AUTO)
    schedule = (n <= gs_max_n && ndim <= kMaxTunedDim) ? TOPOLOW_SCHEDULE_GS : TOPOLOW_SCHEDULE_SLAB;
  if (schedule == TOPOLOW_SCHEDULE_GS && ndim > kMaxTunedDim) {
    set_err(errbuf, errlen, "schedule gs: ndim must be between 1 and %d (wider embeddings run the slab schedule)", kMaxTunedDim);
    return TOPOLOW_ERR_UNSUPPORTED;
  }

  // exact GS: one workgroup while the problem fits its LDS, the tile schedule beyond that
  const bool gs_fits_lds =
      gs_lds_bytes(n, kernel_dim(ndim), (opt.precision == TOPOLOW_PRECISION_F32) ? 4 : 8) <= 150 * 1024 && n <= 2048;
  const bool tile_gs = schedule == TOPOLOW_SCHEDULE_GS && !gs_fits_lds;
  if (schedule == TOPOLOW_SCHEDULE_GS && !tile_gs) {
    int precision = opt.precision == TOPOLOW_PRECISION_AUTO ? TOPOLOW_PRECISION_F64 : opt.precision;
    GsProblem pb;
    pb.initial_positions = initial_positions; pb.n = n; pb.dim = ndim;
    pb.D = dissimilarity_matrix; pb.T = threshold_matrix; pb.degrees = degrees;
    pb.edge_i = edge_i; pb.edge_j = edge_j; pb.edge_dist = edge_dist; pb.edge_thresh = edge_thresh;
    pb.n_edges = n_edges; pb.n_iter = n_iter; pb.k0 = k0; pb.cooling = cooling_rate;
    pb.c_rep = c_repulsion; pb.eps = relative_epsilon; pb.window = convergence_window;
    pb.check_freq = convergence_check_freq; pb.seed = opt.seed;
    GsResult res;
    res.positions = positions_out;
    int rc_inner = TOPOLOW_OK;
    double dev_s = 0.0;
    std::vector<double> trace;
    if (verbose) emit_header(opt, "one-workgroup Gauss-Seidel", n, k0, cooling_rate, c_repulsion);
    const int rc = guarded(errbuf, errlen, [&] {
      select_device(opt.device);
      rc_inner = gs_run_batch(&pb, &res, 1, precision, &dev_s, errbuf, errlen, opt.interrupt_cb,
                              opt.interrupt_user, verbose ? &trace : nullptr);
    });
    if (rc != TOPOLOW_OK) return rc;
    if (rc_inner != TOPOLOW_OK) return rc_inner;
    *converged = res.converged; *iterations = res.iterations; *final_mae = res.final_mae;
    *final_k = res.final_k;
    if (stats) {
      std::memset(stats, 0, sizeof *stats);
      stats->schedule_used = TOPOLOW_SCHEDULE_GS;
      stats->precision_used = precision;
      stats->iterations_run = res.iters_run;
      stats->n_checks = res.n_checks;
      stats->device_seconds = dev_s;
      stats->total_seconds = now_s() - t_start;
    }
    if (verbose) {
      const int nc = (int)(trace.size() / 3);
      emit_checks(opt, trace.data(), 0, nc, n_iter);
      if (res.converged) {
        // the rule that stopped the run: the last `window` checks all lie inside the plateau band
        // (plateau) or above it (worsening); the last check decides
        const bool plateau = nc > 0 && trace[3 * (nc - 1) + 1] <= res.final_mae * (1.0 + relative_epsilon);
        emit_converged(opt, plateau, res.iterations, res.final_mae);
      }
    }
    return TOPOLOW_OK;
  }

  // ---- slab schedule, or exact tile Gauss-Seidel (same session, different iteration body) ----
  const int precision = opt.precision == TOPOLOW_PRECISION_AUTO
                            ? (tile_gs ? TOPOLOW_PRECISION_F64 : TOPOLOW_PRECISION_F32)
                            : opt.precision;
  topolow_session* s = nullptr;
  int rc = topolow_session_create(&s, n, ndim, 0, n, precision, opt.device, errbuf, errlen);
  if (rc != TOPOLOW_OK) return rc;
  double t_dev0 = 0.0, t_dev1 = 0.0;
  int iters_run = 0, stopped = 0;
  double t_setup = 0.0;
  do {
    if (tile_gs) {
      rc = topolow_session_set_schedule(s, TOPOLOW_SCHEDULE_GS);
      if (rc) break;
    }
    if (!opt.keep_labels) {   // slabs / tiles of random points instead of index-contiguous ones
      rc = topolow_session_set_relabel(s, mix64(opt.seed ^ 0x1abe15eedull) | 1ull, errbuf, errlen);
      if (rc) break;
    }
    // The 16 arguments carry the matrix twice: dense (800 + 400 MB at config 3) and as the list of its
    // measured upper-triangle cells (R/core.R:383-402 and :429-436 build both from one matrix).  When
    // the list is verified to BE the matrix, the encoded block is built from the list -- a quarter of
    // the bytes over PCIe -- and the dense arrays are only read on the host, once, to verify it.
    bool from_edges = false;
    if (precision == TOPOLOW_PRECISION_F32 && getenv("TOPOLOW_DENSE_UPLOAD") == nullptr &&
        edges_are_the_matrix(dissimilarity_matrix, threshold_matrix, n, edge_i, edge_j, edge_dist, edge_thresh, n_edges)) {
      rc = topolow_session_load_coo(s, edge_i, edge_j, edge_dist, edge_thresh, n_edges, degrees, errbuf, errlen);
      if (rc) break;
      rc = topolow_session_set_edges(s, edge_i, edge_j, edge_dist, edge_thresh, n_edges, errbuf, errlen);
      if (rc) break;
      // the device-side fingerprint (count and hash of the block's measured cells == the list) rules
      // out what the host pass cannot see cheaply: a pair listed twice
      from_edges = topolow_session_uses_dense_mae(s) != 0;
    }
    if (!from_edges) {
      rc = topolow_session_load_dense(s, dissimilarity_matrix, threshold_matrix, degrees, errbuf, errlen);
      if (rc) break;
      rc = topolow_session_set_edges(s, edge_i, edge_j, edge_dist, edge_thresh, n_edges, errbuf, errlen);
      if (rc) break;
    }
    rc = topolow_session_set_positions(s, initial_positions, errbuf, errlen);
    if (rc) break;
    rc = topolow_session_begin(s, n_iter, k0, cooling_rate, c_repulsion, relative_epsilon,
                               convergence_window, convergence_check_freq, opt.seed,
                               opt.slab_stages, errbuf, errlen);
    if (rc) break;
    t_dev0 = now_s();
    t_setup = t_dev0 - t_start;
    if (verbose)
      emit_header(opt, tile_gs ? "tile Gauss-Seidel" : "row-owner slabs", n, k0, cooling_rate, c_repulsion);
    int reported = 0;
    std::vector<double> trace;
    auto report = [&] {   // verbose: the checks since the last report (waits for the enqueued work)
      int nc = 0;
      if (topolow_session_check_trace(s, nullptr, 0, &nc) != TOPOLOW_OK || nc <= reported) return;
      trace.resize(3 * (size_t)nc);
      if (topolow_session_check_trace(s, trace.data(), nc, &nc) != TOPOLOW_OK) return;
      emit_checks(opt, trace.data(), reported, nc, n_iter);
      reported = nc;
    };
    for (;;) {
      int enq = 0;
      // 50 iterations at a time: the reference's interrupt cadence (src/optimization.cpp:364)
      rc = topolow_session_enqueue(s, 50, &enq, errbuf, errlen);
      if (rc || enq == 0) break;
      if (opt.interrupt_cb && opt.interrupt_cb(opt.interrupt_user)) {
        set_err(errbuf, errlen, "interrupted by the caller");
        rc = TOPOLOW_ERR_INTERRUPTED;
        break;
      }
      if (verbose) report();
    }
    if (rc == TOPOLOW_OK && verbose) report();
    if (rc) break;
    double last = 0.0;
    rc = topolow_session_sync(s, &iters_run, &stopped, &last, errbuf, errlen);
    if (rc) break;
    t_dev1 = now_s();
    rc = topolow_session_finish(s, positions_out, converged, iterations, final_mae, final_k,
                                errbuf, errlen);
  } while (0);
  if (rc == TOPOLOW_OK && stats) {
    std::memset(stats, 0, sizeof *stats);
    stats->schedule_used = tile_gs ? TOPOLOW_SCHEDULE_GS : TOPOLOW_SCHEDULE_SLAB;
    stats->precision_used = precision;
    stats->iterations_run = iters_run;
    stats->n_checks = s->mailbox->n_checks;
    stats->device_seconds = t_dev1 - t_dev0;
    stats->setup_seconds = t_setup;
    stats->stage_launches = s->stage_launches;
  }
  if (rc == TOPOLOW_OK && verbose && *converged)
    emit_converged(opt, s->mailbox->ctl.plateau >= s->mailbox->ctl.window, *iterations, *final_mae);
  topolow_session_destroy(s);
  if (rc == TOPOLOW_OK && stats) stats->total_seconds = now_s() - t_start;
  return rc;
}

}  // extern "C"
