// topolow_amd/csrc/relax_kernels.h -- HIP kernels of the slab (large-N) relaxation path.
// gfx950 only: 64-wide wavefronts, 256-thread workgroups, LDS-staged column points.
//
// What one stage does (reference arithmetic: src/optimization.cpp:203-281 of the reference,
// applied row-owner style -- point i applies its own half of every pair (i,c)):
//   for every row i of this rank's row block, for every column c of the stage's slab:
//     delta = p_c - p_i ; r = |delta| ; rs = r + 0.01
//     spring  (measured and [exact | ">" and r<t | "<" and r>t]): coef = 2k(t-r)/rs / (4g_i+k)
//     repulse (unmeasured, or threshold satisfied):                coef = c/(2 rs^3) / g_i
//     acc_i += delta * coef
//   p_i(new) = p_i - acc_i                      (positions frozen inside a stage: ping-pong)
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>
#include "relax_common.h"

namespace topolow {

constexpr int kThreads = 256;       // block size of the auxiliary kernels
constexpr int kWaves = kThreads / 64;

// Launch geometry of the slab stage kernel (and of the dense error pass, which shares its tiling).
//   THREADS : workgroup size (waves share one LDS image of the slab's column points)
//   RPW     : rows (points being moved) per wave; their coordinates sit in scalar registers
//   CHUNK   : slab columns per LDS buffer (multiple of 256); stage kernel: 0 = as many 256-column
//             groups as keep its double buffer near 20 KB (PipeGeom)
//   PRIO    : stage kernel: 1 = issue priority may fall as the workgroup advances through its slab
//             (switched per launch by the host: only when the whole grid is resident at once)
//   MINWAVES: second __launch_bounds__ argument (waves per SIMD the register budget must allow)
template <int THREADS_, int RPW_, int CHUNK_, int PRIO_ = 0, int MINWAVES_ = 1>
struct StageCfg {
  static constexpr int MINWAVES = MINWAVES_;
  static constexpr int PRIO = PRIO_;
  static constexpr int THREADS = THREADS_;
  static constexpr int WAVES = THREADS_ / 64;
  static constexpr int RPW = RPW_;
  static constexpr int ROWS = WAVES * RPW_;
  static constexpr int CHUNK = CHUNK_;
};

template <typename real> struct Math;
template <> struct Math<float> {
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static constexpr float far() { return kFarF32; }
};
template <> struct Math<double> {
  // The hardware estimates (v_rsq_f64, v_rcp_f64: ~2^-26) and two Newton steps: within 1 ulp of the correctly rounded
  // result, 15 instructions for the pair where the IEEE expansions (scaling for denormals, division fix-ups) take 25.
  // The arguments here are sums of squares of coordinate differences -- exactly 0 for a point with itself, never
  // denormal otherwise -- and r + 0.01 in [0.01, 1e150].  (-DTOPOLOW_F64_IEEE: the library calls.)  The exact GS
  // kernels (relax_gs.h, relax_tilegs.h), which are held to the CPU oracle pair by pair, do not come through here.
#ifdef TOPOLOW_F64_IEEE
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
#else
  static __device__ __forceinline__ double sqrt(double x) {
    const double y0 = __builtin_amdgcn_rsq(x);
    double g = x * y0, h = 0.5 * y0;
    const double e1 = fma(-h, g, 0.5);
    g = fma(g, e1, g);
    h = fma(h, e1, h);
    g = fma(fma(-g, g, x), h, g);
    return x > 0.0 ? g : 0.0;
  }
  static __device__ __forceinline__ double rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
  }
#endif
  static constexpr double far() { return kFarF64; }
};

// Marks a wave-uniform value so the compiler keeps it in scalar registers.
__device__ __forceinline__ float uniform(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ double uniform(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
  const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

template <typename real>
__device__ __forceinline__ real wave_sum(real v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// fp32 wave sum on the vector pipe alone (DPP lane permutations; __shfl_xor goes through LDS
// hardware, ds_bpermute, with a round trip per step): quads, half rows, rows, then the row sums are
// passed down the rows; the total is read from lane 63.  Fixed order, so deterministic; the value is
// valid in every lane.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  auto dpp = [](float x, auto ctrl, auto row_mask) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                                 decltype(row_mask)::value, 0xf, true));
  };
  using std::integral_constant;
  v += dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
  v += dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
  v += dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});   // row_half_mirror
  v += dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});   // row_mirror
  v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});   // row_bcast:15 -> rows 1, 3
  v += dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});   // row_bcast:31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// One encoded row as a buffer resource (wave-uniform, lives in 4 SGPRs): loads then need only a
// 32-bit lane offset instead of a 64-bit address per row.
typedef __amdgpu_buffer_rsrc_t row_rsrc_t;
// `block` = first word of the encoded block, row_in_block = row - row_begin; the resource spans the row's
// words (relax_common.h: enc_index), column c sits enc_col_offset_bytes(c) in.
__device__ __forceinline__ row_rsrc_t make_row_rsrc(const uint32_t* block, int row_in_block, int ld) {
  const uint32_t* row = block + enc_index(row_in_block, 0, ld);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(row), /*stride*/ 0,
                                           enc_row_span_bytes(row_in_block, ld), 0x00020000);
}
__device__ __forceinline__ uint4 load_words(row_rsrc_t rsrc, int byte_off) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Stage the points of columns [cb, cb+cw) into LDS as they lie in HBM (row-major n4 x DIM, so a
// chunk is one contiguous run): a straight 16-byte-per-lane copy, all of a lane's loads in
// flight before its first LDS write.  Position buffers hold roundup4(n) rows; the padding rows
// carry the phantom point (relax_common.h), so no bounds test is needed here.
template <int DIM, typename real, int THREADS, int CHUNK>
__device__ __forceinline__ void stage_points(const real* __restrict__ pos, int cb, int cw,
                                             real* lds_pos, int tid) {
  constexpr int kMaxVec = (CHUNK * DIM * (int)sizeof(real) / 16 + THREADS - 1) / THREADS;
  const uint4* src = reinterpret_cast<const uint4*>(pos + (size_t)cb * DIM);
  uint4* dst = reinterpret_cast<uint4*>(lds_pos);
  const int nvec = cw * DIM * (int)sizeof(real) / 16;
  uint4 tmp[kMaxVec];
#pragma unroll
  for (int v = 0; v < kMaxVec; ++v) {
    const int q = tid + v * THREADS;
    tmp[v] = src[q < nvec ? q : 0];   // clamped, unconditional: keeps tmp[] in registers
  }
#pragma unroll
  for (int v = 0; v < kMaxVec; ++v) {
    const int q = tid + v * THREADS;
    if (q < nvec) dst[q] = tmp[v];
  }
}

// The four column points a lane works on (columns c4..c4+3 of the chunk): 4*DIM consecutive
// reals in LDS, read as 16-byte pieces (conflict-free: lane stride 16*DIM bytes).
template <int DIM, typename real>
__device__ __forceinline__ void load_points(const real* lds_pos, int c4, real (&pc)[4][DIM]) {
  constexpr int kPer16 = 16 / (int)sizeof(real);
  constexpr int kVec = 4 * DIM / kPer16;
  const uint4* src = reinterpret_cast<const uint4*>(lds_pos + (size_t)c4 * DIM);
#pragma unroll
  for (int v = 0; v < kVec; ++v) {
    const uint4 u = src[v];
    if constexpr (sizeof(real) == 4) {
      pc[(4 * v + 0) / DIM][(4 * v + 0) % DIM] = __builtin_bit_cast(float, u.x);
      pc[(4 * v + 1) / DIM][(4 * v + 1) % DIM] = __builtin_bit_cast(float, u.y);
      pc[(4 * v + 2) / DIM][(4 * v + 2) % DIM] = __builtin_bit_cast(float, u.z);
      pc[(4 * v + 3) / DIM][(4 * v + 3) % DIM] = __builtin_bit_cast(float, u.w);
    } else {
      pc[(2 * v + 0) / DIM][(2 * v + 0) % DIM] =
          __builtin_bit_cast(double, ((unsigned long long)u.y << 32) | u.x);
      pc[(2 * v + 1) / DIM][(2 * v + 1) % DIM] =
          __builtin_bit_cast(double, ((unsigned long long)u.w << 32) | u.z);
    }
  }
}

// One ordered pair (row i, column c): accumulate i's half of the pair update.
//   ks = 2k / (4 g_i + k), cg = (c_rep / 2) / g_i   (row constants, scalar registers)
//   THR = false: the row holds no ">" / "<" targets (flag computed when the matrix is encoded),
//   so a pair springs exactly when it is measured (target < +Inf).
template <int DIM, typename real, bool THR, bool ERR = false, bool CNT = false>
__device__ __forceinline__ void pair_accum(const real (&pc)[DIM], const real (&pi)[DIM],
                                           uint32_t w, real ks, real cg, real (&acc)[DIM],
                                           float* err = nullptr, unsigned* cnt_wave = nullptr) {
  real dx[DIM];
  real s = 0;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    dx[d] = pc[d] - pi[d];
    s = fma(dx[d], dx[d], s);
  }
  const real r = Math<real>::sqrt(s);
  const real inv = Math<real>::rcp(r + (real)0.01);
  // THR = false: every measured word is an exact target (code 0: the word IS the float) and every
  // other word is kInfWord, so the raw word is class-tested (one instruction) and never masked; the NaN an
  // unmeasured word reads as only ever reaches the unselected side of the select below
  const real t = (real)bits_f32(THR ? (w & ~kCodeMask) : w);
  bool spring;
  if constexpr (THR) {
    // branch-free classification (bitwise ops on purpose: no short-circuit control flow);
    // unmeasured pairs are (+Inf, "<") and therefore never spring
    const uint32_t code = w & kCodeMask;
    spring = (code == 0u) | ((code == 1u) & (r < t)) | ((code == 2u) & (r > t));
  } else {
    spring = __builtin_amdgcn_classf(bits_f32(w), 0x1f8);   // finite of either sign
  }
  const real fs = (t - r) * inv * ks;
  const real fr = inv * inv * inv * cg;
  const real coef = spring ? fs : fr;
#pragma unroll
  for (int d = 0; d < DIM; ++d) acc[d] = fma(dx[d], coef, acc[d]);
  if constexpr (ERR) {
    // the convergence MAE of the positions this stage reads: a pair contributes exactly when its spring is
    // active (reference src/optimization.cpp:68-76 and :230-243 are the same three cases)
    // CNT = false: the block holds no threshold target, every measured pair contributes whatever the
    // positions are, and the host knows their number
    *err += spring ? (float)fabs(t - r) : 0.0f;
    if constexpr (CNT) *cnt_wave += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(spring));
  }
}

#ifdef TOPOLOW_TUNING
// Tuning builds: per-workgroup start/end stamps (100 MHz constant clock) of the stage kernel's last
// launch, for the dispatch-ramp / tail analysis in DESIGN.md section 6.
__device__ unsigned long long* g_wg_stamps = nullptr;
#define TL_WG_STAMP(slot) \
  do { if (g_wg_stamps != nullptr && threadIdx.x == 0) g_wg_stamps[blockIdx.x * 2 + (slot)] = wall_clock64(); } while (0)
#else
#define TL_WG_STAMP(slot) do { } while (0)
#endif

// ----------------------------------------------------------------------------------------
// The stage kernel.  Lane l of a wave takes columns 4l..4l+3 of every 256-column group, groups in
// slab order; nothing in the steady state waits for memory:
//   * the slab's column points move HBM/L2 -> LDS with direct-to-LDS loads (no registers), one
//     chunk ahead, into the other half of a double buffer: one barrier per chunk, no exposed
//     staging;
//   * the encoded target words are requested one 256-column group ahead, across chunk and
//     slab-part boundaries, so the 4 N^2-byte stream never drains at a barrier.
// A chunk is GPC groups (CFG::CHUNK / 256, or as many as keep the double buffer near 20 KB).  A 256-column group of points is DIM*sizeof(real)/4 KB, i.e. that many 1-KB wave
// transfers, dealt round-robin to the waves.
typedef __attribute__((address_space(3))) const unsigned char* lds_cptr_t;

template <int DIM, typename real, int CHUNK_REQ = 0>
struct PipeGeom {
  static constexpr int kGroupBytes = 256 * DIM * (int)sizeof(real);
  static constexpr int kGpcRaw = CHUNK_REQ > 0 ? CHUNK_REQ / 256 : 10240 / kGroupBytes;
  static constexpr int GPC = kGpcRaw < 1 ? 1 : (kGpcRaw > 4 ? 4 : kGpcRaw);
  static constexpr int CHUNK = 256 * GPC;
  static constexpr int kBufBytes = GPC * kGroupBytes;
  static constexpr int kXfersPerGroup = kGroupBytes / 1024;
};

// chunk `c` of the slab: parts [b0,e0) then [b1,e1), CHUNK columns at a time; width 0 past the end
template <int CHUNK>
__device__ __forceinline__ void pipe_chunk_at(const SlabRanges& rg, int nc0, int c, int& cb, int& cw) {
  if (c < nc0) {
    cb = rg.b0 + c * CHUNK;
    cw = min(CHUNK, rg.e0 - cb);
  } else {
    cb = rg.b1 + (c - nc0) * CHUNK;
    cw = max(0, min(CHUNK, rg.e1 - cb));
  }
}

// Request the points of columns [cb, cb+CHUNK) into `buf` (completion is awaited by the caller's
// barrier).  A ragged chunk is requested whole -- no control flow; what lies past its last column
// is never read back -- with the addresses clamped to the position buffer's last 16 bytes.
template <int DIM, typename real, int WAVES, int CHUNK_REQ>
__device__ __forceinline__ void pipe_request_points(const real* __restrict__ pos, int pos_bytes,
                                                    int cb, int cw, unsigned char* buf, int wave,
                                                    int lane) {
  using G = PipeGeom<DIM, real, CHUNK_REQ>;
  const int base = cb * DIM * (int)sizeof(real);
  const unsigned char* src = reinterpret_cast<const unsigned char*>(pos);
  (void)cw;
#pragma unroll
  for (int x = 0; x < (G::GPC * G::kXfersPerGroup + WAVES - 1) / WAVES; ++x) {
    const int t = x * WAVES + wave;               // wave-uniform transfer index
    if ((G::GPC * G::kXfersPerGroup) % WAVES == 0 || t < G::GPC * G::kXfersPerGroup) {
      const int off = min(base + t * 1024 + lane * 16, pos_bytes - 16);
      const unsigned lds_dst = (unsigned)(unsigned long)(lds_cptr_t)(buf + t * 1024);
      // Written as asm on purpose: hipcc drains vmcnt to 0 at every barrier while it knows of an
      // LDS transfer in flight, which would also drain the target-word prefetch.  The wave that
      // issues a transfer awaits it itself (pipe_await_points) before the barrier.
      // (m0 holds the LDS address of the transfer; it is put back, the backend treats it as its own)
      unsigned saved_m0;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(saved_m0) : "v"(src + off), "s"(lds_dst) : "memory");
    }
  }
}

// Wait until this wave's point transfers have landed, leaving its YOUNGER vector-memory operations
// (the kYounger target loads issued since) in flight: vmcnt counts in issue order.
template <int kYounger>
__device__ __forceinline__ void pipe_await_points() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kYounger) : "memory");
}

// The rows a wave moves: coordinates and row constants (wave-uniform), accumulators, target rows.
// Generic form: one scalar pair update per (row, column).
template <int DIM, typename real, int RPW, bool ANYTHR, bool PACKED = (sizeof(real) == 4 && RPW == 2)>
struct PipeRows {
  real pi[RPW][DIM];
  real acc[RPW][DIM];
  real ks[RPW], cg[RPW];
  row_rsrc_t rsrc[RPW];
  bool thr;

  __device__ __forceinline__ void set_row(int r, const real (&p)[DIM], real ks_r, real cg_r) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) { pi[r][d] = p[d]; acc[r][d] = 0; }
    ks[r] = ks_r;
    cg[r] = cg_r;
  }
  float err = 0.0f;          // ERR launches: this lane's |t - r| over its contributing pairs (folded by the caller)
  unsigned cnt_wave = 0;     // ... and the wave's count of them (wave-uniform)

  // four column points against every row; w[r] = the rows' target words of those columns
  template <bool ERR>
  __device__ __forceinline__ void group(const real (&pc)[4][DIM], const uint4 (&w)[RPW]) {
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if (ANYTHR && thr) {
        pair_accum<DIM, real, true, ERR, ANYTHR>(pc[0], pi[r], w[r].x, ks[r], cg[r], acc[r], &err, &cnt_wave);
        pair_accum<DIM, real, true, ERR, ANYTHR>(pc[1], pi[r], w[r].y, ks[r], cg[r], acc[r], &err, &cnt_wave);
        pair_accum<DIM, real, true, ERR, ANYTHR>(pc[2], pi[r], w[r].z, ks[r], cg[r], acc[r], &err, &cnt_wave);
        pair_accum<DIM, real, true, ERR, ANYTHR>(pc[3], pi[r], w[r].w, ks[r], cg[r], acc[r], &err, &cnt_wave);
      } else {
        pair_accum<DIM, real, false, ERR, ANYTHR>(pc[0], pi[r], w[r].x, ks[r], cg[r], acc[r], &err, &cnt_wave);
        pair_accum<DIM, real, false, ERR, ANYTHR>(pc[1], pi[r], w[r].y, ks[r], cg[r], acc[r], &err, &cnt_wave);
        pair_accum<DIM, real, false, ERR, ANYTHR>(pc[2], pi[r], w[r].z, ks[r], cg[r], acc[r], &err, &cnt_wave);
        pair_accum<DIM, real, false, ERR, ANYTHR>(pc[3], pi[r], w[r].w, ks[r], cg[r], acc[r], &err, &cnt_wave);
      }
    }
  }
  __device__ __forceinline__ double take_err() { const double e = (double)err; err = 0.0f; return e; }
  __device__ __forceinline__ real origin(int r, int d) const { return pi[r][d]; }
  __device__ __forceinline__ real lane_sum(int r, int d) const { return acc[r][d]; }
};

// fp32, two rows per wave: the two rows ride in the two halves of packed fp32 operations
// (v_pk_add/mul/fma_f32) -- the column point is broadcast, the rows' coordinates and constants are
// scalar-register pairs, the accumulators are register pairs -- so every non-transcendental
// operation of a column serves both rows.  Same operations in the same order per row as the
// generic form (bit-identical results).
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int DIM, bool THR, bool ERR = false, bool CNT = false>
__device__ __forceinline__ void pair_accum_rows2(const float (&pc)[DIM], const f32x2_t (&pi2)[DIM],
                                                 uint32_t w0, uint32_t w1, f32x2_t ks2, f32x2_t cg2,
                                                 f32x2_t (&acc2)[DIM], f32x2_t* err2 = nullptr,
                                                 unsigned* cnt_wave = nullptr) {
  f32x2_t dx[DIM];
  f32x2_t s = {0.0f, 0.0f};
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const f32x2_t pcd = {pc[d], pc[d]};
    dx[d] = pcd - pi2[d];
    s = __builtin_elementwise_fma(dx[d], dx[d], s);
  }
  const f32x2_t r = {Math<float>::sqrt(s.x), Math<float>::sqrt(s.y)};
  const f32x2_t rs = r + (f32x2_t){0.01f, 0.01f};
  const f32x2_t inv = {Math<float>::rcp(rs.x), Math<float>::rcp(rs.y)};
  const f32x2_t t = {bits_f32(THR ? (w0 & ~kCodeMask) : w0), bits_f32(THR ? (w1 & ~kCodeMask) : w1)};
  bool sp0, sp1;
  if constexpr (THR) {
    const uint32_t c0 = w0 & kCodeMask, c1 = w1 & kCodeMask;
    sp0 = (c0 == 0u) | ((c0 == 1u) & (r.x < t.x)) | ((c0 == 2u) & (r.x > t.x));
    sp1 = (c1 == 0u) | ((c1 == 1u) & (r.y < t.y)) | ((c1 == 2u) & (r.y > t.y));
  } else {
    sp0 = __builtin_amdgcn_classf(bits_f32(w0), 0x1f8);
    sp1 = __builtin_amdgcn_classf(bits_f32(w1), 0x1f8);
  }
  const f32x2_t e = t - r;
  const f32x2_t fs = e * inv * ks2;
  const f32x2_t fr = inv * inv * inv * cg2;
  const f32x2_t coef = {sp0 ? fs.x : fr.x, sp1 ? fs.y : fr.y};
#pragma unroll
  for (int d = 0; d < DIM; ++d) acc2[d] = __builtin_elementwise_fma(dx[d], coef, acc2[d]);
  if constexpr (ERR) {   // see pair_accum
    const f32x2_t a = {sp0 ? fabsf(e.x) : 0.0f, sp1 ? fabsf(e.y) : 0.0f};
    *err2 += a;
    if constexpr (CNT)
      *cnt_wave += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sp0)) +
                   (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sp1));
  }
}

template <int DIM, bool ANYTHR>
struct PipeRows<DIM, float, 2, ANYTHR, true> {
  f32x2_t pi2[DIM];    // (row 0, row 1) per coordinate
  f32x2_t acc2[DIM];
  f32x2_t ks2, cg2;
  row_rsrc_t rsrc[2];
  bool thr;
  f32x2_t err2 = {0.0f, 0.0f};   // ERR launches: see the generic form
  unsigned cnt_wave = 0;

  __device__ __forceinline__ void set_row(int r, const float (&p)[DIM], float ks_r, float cg_r) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      if (r == 0) { pi2[d].x = p[d]; acc2[d].x = 0; } else { pi2[d].y = p[d]; acc2[d].y = 0; }
    }
    if (r == 0) { ks2.x = ks_r; cg2.x = cg_r; } else { ks2.y = ks_r; cg2.y = cg_r; }
  }
  template <bool ERR>
  __device__ __forceinline__ void group(const float (&pc)[4][DIM], const uint4 (&w)[2]) {
    if (ANYTHR && thr) {
      pair_accum_rows2<DIM, true, ERR, ANYTHR>(pc[0], pi2, w[0].x, w[1].x, ks2, cg2, acc2, &err2, &cnt_wave);
      pair_accum_rows2<DIM, true, ERR, ANYTHR>(pc[1], pi2, w[0].y, w[1].y, ks2, cg2, acc2, &err2, &cnt_wave);
      pair_accum_rows2<DIM, true, ERR, ANYTHR>(pc[2], pi2, w[0].z, w[1].z, ks2, cg2, acc2, &err2, &cnt_wave);
      pair_accum_rows2<DIM, true, ERR, ANYTHR>(pc[3], pi2, w[0].w, w[1].w, ks2, cg2, acc2, &err2, &cnt_wave);
    } else {
      pair_accum_rows2<DIM, false, ERR, ANYTHR>(pc[0], pi2, w[0].x, w[1].x, ks2, cg2, acc2, &err2, &cnt_wave);
      pair_accum_rows2<DIM, false, ERR, ANYTHR>(pc[1], pi2, w[0].y, w[1].y, ks2, cg2, acc2, &err2, &cnt_wave);
      pair_accum_rows2<DIM, false, ERR, ANYTHR>(pc[2], pi2, w[0].z, w[1].z, ks2, cg2, acc2, &err2, &cnt_wave);
      pair_accum_rows2<DIM, false, ERR, ANYTHR>(pc[3], pi2, w[0].w, w[1].w, ks2, cg2, acc2, &err2, &cnt_wave);
    }
  }
  __device__ __forceinline__ double take_err() {
    const double e = (double)err2.x + (double)err2.y;
    err2 = (f32x2_t){0.0f, 0.0f};
    return e;
  }
  __device__ __forceinline__ float origin(int r, int d) const { return r == 0 ? pi2[d].x : pi2[d].y; }
  __device__ __forceinline__ float lane_sum(int r, int d) const { return r == 0 ? acc2[d].x : acc2[d].y; }
};

// One chunk: request the next chunk's points into `oth`, sweep this chunk's groups out of `cur`.
//   w : target words of this chunk's groups on entry, of the next chunk's groups on exit -- a
//       group's words are requested GPC-1 groups and one barrier before their use
template <int DIM, typename real, typename CFG, bool ANYTHR, bool ERR>
__device__ __forceinline__ void pipe_chunk(PipeRows<DIM, real, CFG::RPW, ANYTHR>& R,
                                           const real* __restrict__ pos, int pos_bytes,
                                           const unsigned char* cur, unsigned char* oth, int cw,
                                           int ncb, int ncw,
                                           uint4 (&w)[PipeGeom<DIM, real, CFG::CHUNK>::GPC][CFG::RPW], int wave,
                                           int lane) {
  using G = PipeGeom<DIM, real, CFG::CHUNK>;
  constexpr int RPW = CFG::RPW;
  if (ncw > 0) pipe_request_points<DIM, real, CFG::WAVES, CFG::CHUNK>(pos, pos_bytes, ncb, ncw, oth, wave, lane);
  const real* lds_pos = reinterpret_cast<const real*>(cur);
#pragma unroll
  for (int g = 0; g < G::GPC; ++g) {
    const int c4 = lane * 4 + g * 256;
    if (c4 < cw) {
      real pc[4][DIM];
      load_points<DIM, real>(lds_pos, c4, pc);
      R.template group<ERR>(pc, w[g]);
    }
    // The same group of the next chunk, requested as soon as this group's words are dead (so they
    // land in the same registers).  Unconditional: the count of loads per chunk is what
    // pipe_await_points relies on; past the slab's end the offset is out of range, which a buffer
    // load answers with 0 without touching memory.
    const int noff = g * 256 < ncw ? enc_col_offset_bytes(ncb + c4) : 0x7ffffff0;
#pragma unroll
    for (int r = 0; r < RPW; ++r) w[g][r] = load_words(R.rsrc[r], noff);
  }
  pipe_await_points<G::GPC * RPW>();
  __syncthreads();   // next chunk's points have landed; every wave is done reading `cur`
}

// One slab stage for rows [row_begin,row_end).
//   denc    : (row_end-row_begin) x ld encoded targets (layout: relax_common.h enc_index), ld % 64 == 0
//   pos_in  : n x DIM row-major, all points, read-only in this launch
//   pos_out : n x DIM row-major; rows [row_begin,row_end) are written
//   st      : run state (nullable): launch is a no-op once st->stopped is set; non-finite
//             results are reported through st->first_nonfinite
//   falling_priority: see StageCfg::PRIO
//   push / n_push: row-sharded runs -- the same rows are also stored into the position buffers of the
//             other row blocks (peer stores: on another GPU of the node they travel over xGMI), so the
//             all-gather of the updated slices is part of this kernel's epilogue
//   ANYTHR = false: the host has checked that NO row of the block holds a threshold target,
//   so only the cheaper classification is compiled in (fewer registers, one more wave per SIMD).
//   ERR = true (one-stage iterations only): the launch also reduces the convergence MAE of the positions it
//             READS -- with one stage per iteration every ordered pair is met from the positions the
//             previous iteration left, which is what the reference's check of that iteration measures
//             (src/optimization.cpp:294-296); each unordered pair is met twice with the same |t - r|, so
//             sum and count double and their ratio is the MAE.  Per-workgroup partials in f64.
template <int DIM, typename real, typename CFG, bool ANYTHR, bool ERR = false>
__global__ __launch_bounds__(CFG::THREADS, CFG::MINWAVES) void slab_stage_pipe_kernel(
    const uint32_t* __restrict__ denc, int ld, int row_begin, int row_end, int n,
    const real* __restrict__ pos_in, real* __restrict__ pos_out,
    const float* __restrict__ gplus, const unsigned char* __restrict__ rowflags, RunState* st,
    SlabRanges rg, int iter1, double k, double c_rep, int falling_priority,
    real* const* __restrict__ push, int n_push, double* __restrict__ part_sum,
    unsigned long long* __restrict__ part_cnt, unsigned long long fixed_cnt) {
  if (st != nullptr && st->stopped) return;
  TL_WG_STAMP(0);
  using G = PipeGeom<DIM, real, CFG::CHUNK>;
  constexpr int RPW = CFG::RPW;
  __shared__ __attribute__((aligned(16))) unsigned char bufs[2 * G::kBufBytes];   // double buffer

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = row_begin + blockIdx.x * CFG::ROWS + wave * RPW;
  const int pos_bytes = ((n + 3) & ~3) * DIM * (int)sizeof(real);
  const int nc0 = (rg.e0 - rg.b0 + G::CHUNK - 1) / G::CHUNK;
  const int nch = nc0 + (rg.e1 - rg.b1 + G::CHUNK - 1) / G::CHUNK;

  PipeRows<DIM, real, RPW, ANYTHR> R;
  double err_d = 0.0;
  int rr[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int row = row0 + r;
    rr[r] = row < row_end ? row : row_end - 1;  // clamp: result discarded below
    R.rsrc[r] = make_row_rsrc(denc, rr[r] - row_begin, ld);
  }
  // first chunk's points and target words are on their way before anything else
  int cb, cw;
  pipe_chunk_at<G::CHUNK>(rg, nc0, 0, cb, cw);
  pipe_request_points<DIM, real, CFG::WAVES, CFG::CHUNK>(pos_in, pos_bytes, cb, cw, bufs, wave, lane);
  uint4 w[G::GPC][RPW];
#pragma unroll
  for (int g = 0; g < G::GPC; ++g) {
    const int off = g * 256 < cw ? enc_col_offset_bytes(cb + g * 256 + lane * 4) : 0x7ffffff0;
#pragma unroll
    for (int r = 0; r < RPW; ++r) w[g][r] = load_words(R.rsrc[r], off);
  }

  int thr_any = 0;  // wave-uniform: do any of this wave's rows hold threshold targets?
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    thr_any |= rowflags[rr[r] - row_begin];
    real p[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) p[d] = uniform(pos_in[(size_t)rr[r] * DIM + d]);  // scalar registers
    const real g = (real)gplus[rr[r]];
    R.set_row(r, p, uniform((real)(2.0 * k) / ((real)4 * g + (real)k)), uniform((real)(0.5 * c_rep) / g));
  }
  R.thr = ANYTHR && __builtin_amdgcn_readfirstlane(thr_any) != 0;
  pipe_await_points<0>();
  __syncthreads();

  // one loop body for both halves of the double buffer (selected by address, not by unrolling:
  // the prefetched words then stay in the same registers from one chunk to the next)
#pragma unroll 1
  for (int c = 0; c < nch; ++c) {
    int ncb, ncw;
    pipe_chunk_at<G::CHUNK>(rg, nc0, c + 1, ncb, ncw);
    if (c + 1 >= nch) ncw = 0;
    unsigned char* cur = bufs + (c & 1) * G::kBufBytes;
    unsigned char* oth = bufs + ((c & 1) ^ 1) * G::kBufBytes;
    if (CFG::PRIO == 1 && falling_priority) {
      // issue priority falls as a workgroup advances, so the workgroups sharing a CU finish
      // together instead of oldest-first (the last one would otherwise run alone, latency-bound).
      // Only when the whole grid is resident at once (the host decides): with several rounds of
      // workgroups it holds the late rounds back (N = 20 000: 94 instead of 86 us per stage)
      const int left = nch - c;
      if (left >= 4) __builtin_amdgcn_s_setprio(3);
      else if (left == 3) __builtin_amdgcn_s_setprio(2);
      else if (left == 2) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    pipe_chunk<DIM, real, CFG, ANYTHR, ERR>(R, pos_in, pos_bytes, cur, oth, cw, ncb, ncw, w, wave, lane);
    cw = ncw;
    if constexpr (ERR) err_d += R.take_err();   // fp32 partial of one chunk (<= 32 terms per lane) into f64
  }
  if constexpr (ERR) {
    // a wave whose rows lie past the block's end works on clamped copies of the last row: it contributes
    // nothing (the host fuses a check only into blocks of an even number of rows, so a wave's two rows are
    // valid or invalid together)
    double s = 0.0;
    unsigned long long c = 0;
    if (row0 + RPW - 1 < row_end) { s = wave_sum<double>(err_d); c = R.cnt_wave; }
    double* red_s = reinterpret_cast<double*>(bufs);                       // the point buffers are free now
    unsigned long long* red_c = reinterpret_cast<unsigned long long*>(bufs + 64);
    if (lane == 0) { red_s[wave] = s; red_c[wave] = c; }
    __syncthreads();
    if (tid == 0) {
      double ts = 0.0;
      unsigned long long tc = 0;
      for (int q = 0; q < CFG::WAVES; ++q) { ts += red_s[q]; tc += red_c[q]; }
      part_sum[blockIdx.x] = ts;
      // threshold-free block: the number of contributing ordered pairs is the host's (2 x measured pairs)
      part_cnt[blockIdx.x] = ANYTHR ? tc : (blockIdx.x == 0 ? fixed_cnt : 0ull);
    }
  }

#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int row = row0 + r;
    bool finite = true;
    real out[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      real total;
      if constexpr (sizeof(real) == 4) total = wave_sum_dpp(R.lane_sum(r, d));
      else total = wave_sum<real>(R.lane_sum(r, d));
      out[d] = R.origin(r, d) - total;
      finite = finite && isfinite(out[d]);
    }
    if (lane == 0 && row < row_end) {
#pragma unroll
      for (int d = 0; d < DIM; ++d) pos_out[(size_t)row * DIM + d] = out[d];
      if (!finite && st != nullptr) atomicMin(&st->first_nonfinite, iter1);
    }
    if (n_push > 0 && row < row_end && lane >= 1 && lane <= n_push) {   // lane q serves block q-1
      real* dst = push[lane - 1];
#pragma unroll
      for (int d = 0; d < DIM; ++d) dst[(size_t)row * DIM + d] = out[d];
    }
  }
  TL_WG_STAMP(1);
}

// ---------------------------------------------------------------------------------------
// Wide embeddings (ndim > 16; the reference accepts any ndim, src/optimization.cpp:129).  The tuned stage kernel
// keeps a wave's rows in scalar registers and is instantiated per coordinate count up to 16; beyond that the same
// stage -- same slabs, same row-owner update, same arithmetic per pair -- runs in this plain form, instantiated for
// 32 and 64 coordinates (other counts zero-padded, exact).  One wave per row; a lane walks every 64th column of the
// slab, reads the column's point from global memory (twice: once for the distance, once for the update, so that
// only the DIM accumulators live in registers) and the row's point from LDS.  Not tuned: such problems are rare
// (the reference's own parameter search covers ndim 2..10) and small.
// ---------------------------------------------------------------------------------------
template <int DIM, typename real>
__global__ __launch_bounds__(kThreads) void slab_stage_wide_kernel(
    const uint32_t* __restrict__ denc, int ld, int row_begin, int row_end, int n,
    const real* __restrict__ pos_in, real* __restrict__ pos_out, const float* __restrict__ gplus, RunState* st,
    SlabRanges rg, int iter1, double k, double c_rep, real* const* __restrict__ push, int n_push) {
  if (st != nullptr && st->stopped) return;
  __shared__ real pi_s[kWaves][DIM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = row_begin + blockIdx.x * kWaves + wave;
  const int rrow = row < row_end ? row : row_end - 1;      // clamp: result discarded below
  for (int d = lane; d < DIM; d += 64) pi_s[wave][d] = pos_in[(size_t)rrow * DIM + d];
  __syncthreads();
  const real* pi = pi_s[wave];
  const real g = (real)gplus[rrow];
  const real ks = (real)(2.0 * k) / ((real)4 * g + (real)k), cg = (real)(0.5 * c_rep) / g;
  real acc[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) acc[d] = 0;
  const uint32_t* wrow = denc + enc_index(rrow - row_begin, 0, ld);
  for (int part = 0; part < 2; ++part) {
    const int b = part == 0 ? rg.b0 : rg.b1, e = part == 0 ? rg.e0 : rg.e1;
    for (int c = b + lane; c < e; c += 64) {
      const real* pc = pos_in + (size_t)c * DIM;
      real s = 0;
#pragma unroll 8
      for (int d = 0; d < DIM; ++d) {
        const real dx = pc[d] - pi[d];
        s = fma(dx, dx, s);
      }
      const uint32_t w = wrow[enc_col_offset_bytes(c) / 4];
      const real r = Math<real>::sqrt(s);
      const real inv = Math<real>::rcp(r + (real)0.01);
      const real t = (real)bits_f32(w & ~kCodeMask);
      const uint32_t code = w & kCodeMask;
      const bool spring = (code == 0u) | ((code == 1u) & (r < t)) | ((code == 2u) & (r > t));
      const real coef = spring ? (t - r) * inv * ks : inv * inv * inv * cg;
#pragma unroll
      for (int d = 0; d < DIM; ++d) acc[d] = fma(pc[d] - pi[d], coef, acc[d]);
    }
  }
  bool finite = true;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const real out = pi[d] - wave_sum<real>(acc[d]);
    finite = finite && isfinite(out);
    if (lane == 0 && row < row_end) pos_out[(size_t)row * DIM + d] = out;
    if (n_push > 0 && row < row_end && lane >= 1 && lane <= n_push) push[lane - 1][(size_t)row * DIM + d] = out;
  }
  if (lane == 0 && row < row_end && !finite && st != nullptr) atomicMin(&st->first_nonfinite, iter1);
  (void)n;
}

// ---------------------------------------------------------------------------------------
// Dense MAE pass: the same quantity as the edge MAE (reference src/optimization.cpp:54-81), read
// from the encoded target block instead of the COO list -- coalesced 16-B target loads and LDS
// staged points instead of two gathers per edge.  Used when the session has verified that the
// caller's edge list is exactly the measured upper triangle of the block (it is, for everything
// the reference's R driver builds: R/core.R:383-402 and :429-436 come from one matrix).
//   PARITY = false: row i reduces the pairs (i, c) with c > i            (single GPU: reads half)
//   PARITY = true : row i reduces pair {i,c} when (c>i and i+c even) or (c<i and i+c odd), so
//                   every row -- hence every rank of a row-sharded run -- gets ~half its columns.
// ---------------------------------------------------------------------------------------
template <int DIM, typename real, bool THR>
__device__ __forceinline__ void pair_error(const real (&pc)[DIM], const real (&pi)[DIM],
                                           uint32_t w, bool take, float& err, unsigned& cnt) {
  real s = 0;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const real dx = pc[d] - pi[d];
    s = fma(dx, dx, s);
  }
  const real r = Math<real>::sqrt(s);
  const real t = (real)bits_f32(w & ~kCodeMask);
  bool contributes;
  if constexpr (THR) {
    const uint32_t code = w & kCodeMask;
    // unmeasured is (+Inf, "<"): r > +Inf never holds, so it never contributes
    contributes = (code == 0u) | ((code == 1u) & (r < t)) | ((code == 2u) & (r > t));
  } else {
    contributes = t < (real)INFINITY;
  }
  contributes = contributes & take;
  err += contributes ? (float)fabs(t - r) : 0.0f;
  cnt += contributes ? 1u : 0u;
}

// fp32, two rows per wave, every pair of the batch counted (tile right of the diagonal): the two
// rows ride in the halves of packed fp32 operations, as in the stage kernel.  Same operations per
// pair as pair_error; the count goes through the scalar unit (population count of the lane mask).
template <int DIM, bool THR>
__device__ __forceinline__ void pair_error_rows2(const float (&pc)[DIM], const f32x2_t (&pi2)[DIM],
                                                 uint32_t w0, uint32_t w1, float& err,
                                                 unsigned& cnt_wave) {
  f32x2_t s = {0.0f, 0.0f};
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const f32x2_t pcd = {pc[d], pc[d]};
    const f32x2_t dx = pcd - pi2[d];
    s = __builtin_elementwise_fma(dx, dx, s);
  }
  const f32x2_t r = {Math<float>::sqrt(s.x), Math<float>::sqrt(s.y)};
  const f32x2_t t = {bits_f32(THR ? (w0 & ~kCodeMask) : w0), bits_f32(THR ? (w1 & ~kCodeMask) : w1)};
  const f32x2_t e = t - r;
  bool c0, c1;
  if constexpr (THR) {
    const uint32_t k0 = w0 & kCodeMask, k1 = w1 & kCodeMask;
    c0 = (k0 == 0u) | ((k0 == 1u) & (r.x < t.x)) | ((k0 == 2u) & (r.x > t.x));
    c1 = (k1 == 0u) | ((k1 == 1u) & (r.y < t.y)) | ((k1 == 2u) & (r.y > t.y));
  } else {
    c0 = __builtin_amdgcn_classf(bits_f32(w0), 0x1f8);
    c1 = __builtin_amdgcn_classf(bits_f32(w1), 0x1f8);
  }
  err += fabsf(c0 ? e.x : 0.0f);
  err += fabsf(c1 ? e.y : 0.0f);
  cnt_wave += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(c0)) +
              (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(c1));
}

// Grid: x = column chunk (CFG::CHUNK columns), y = row tile (kErrTileRows rows).  A workgroup stages
// its chunk's points into LDS ONCE and sweeps the tile's rows over it (RPW rows per wave at a
// time), so the staging is amortised over kErrTileRows rows; tiles that hold no pair to reduce
// (left of the diagonal in upper-triangle mode) write a zero partial and leave.
constexpr int kErrTileRows = 64;

template <int DIM, typename real, typename CFG, bool PARITY>
__global__ __launch_bounds__(CFG::THREADS) void dense_error_kernel(
    const uint32_t* __restrict__ denc, int ld, int row_begin, int row_end, int n,
    const real* __restrict__ pos, const unsigned char* __restrict__ rowflags,
    double* __restrict__ part_sum, unsigned long long* __restrict__ part_cnt,
    const RunState* st) {
  if (st != nullptr && st->stopped) return;
  constexpr int kChunk = CFG::CHUNK;
  constexpr int RPW = CFG::RPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char slab_smem[];
  real* lds_pos = reinterpret_cast<real*>(slab_smem);
  __shared__ double sh_s[CFG::WAVES];
  __shared__ unsigned long long sh_c[CFG::WAVES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n4 = (n + 3) & ~3;
  const int part = blockIdx.y * gridDim.x + blockIdx.x;
  const int cb = blockIdx.x * kChunk;
  const int tile_row0 = row_begin + blockIdx.y * kErrTileRows;
  const int cw = min(kChunk, n4 - cb);
  // upper-triangle mode: the tile holds a pair (i, c > i) only if its last column lies right of its
  // first row
  if (cw <= 0 || tile_row0 >= row_end || (!PARITY && cb + cw - 1 <= tile_row0)) {
    if (tid == 0) { part_sum[part] = 0.0; part_cnt[part] = 0; }
    return;
  }
  stage_points<DIM, real, CFG::THREADS, kChunk>(pos, cb, cw, lds_pos, tid);
  __syncthreads();

  float err = 0.0f;
  unsigned cnt = 0;
  double err_d = 0.0;
#pragma unroll 1
  for (int sub = 0; sub < kErrTileRows; sub += CFG::ROWS) {
    const int row0 = tile_row0 + sub + wave * RPW;
    if (row0 >= row_end) break;                        // wave-uniform
    if (!PARITY && cb + cw - 1 <= row0) continue;      // these rows lie right of the whole chunk
    real pi[RPW][DIM];
    int rows[RPW];
    row_rsrc_t rsrc[RPW];
    int thr_any = 0;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int row = row0 + r;
      const int rr = row < row_end ? row : row_end - 1;
      rows[r] = row < row_end ? row : 0x7fffffff;      // out-of-range rows take no pair
      thr_any |= rowflags[rr - row_begin];
#pragma unroll
      for (int d = 0; d < DIM; ++d) pi[r][d] = uniform(pos[(size_t)rr * DIM + d]);
      rsrc[r] = make_row_rsrc(denc, rr - row_begin, ld);
    }
    const bool thr = __builtin_amdgcn_readfirstlane(thr_any) != 0;
    // all of this row batch's target words are requested before any is used (the points are
    // already in LDS, so nothing queues behind these loads)
    constexpr int kGroups = kChunk / 256;
    uint4 w4[kGroups][RPW];
#pragma unroll
    for (int t = 0; t < kGroups; ++t) {
      const int c4 = lane * 4 + t * 256;
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const bool need = c4 < cw && (PARITY || cb + c4 + 3 > rows[r]) && rows[r] != 0x7fffffff;
        w4[t][r] = need ? load_words(rsrc[r], enc_col_offset_bytes(cb + c4)) : make_uint4(kInfWord, kInfWord, kInfWord, kInfWord);
      }
    }
    if constexpr (sizeof(real) == 4 && RPW == 2 && !PARITY) {
      // both rows valid and the whole chunk right of them: every pair of the batch is counted
      if (row0 + 1 < row_end && cb > row0 + 1) {     // wave-uniform
        f32x2_t pi2[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) pi2[d] = (f32x2_t){pi[0][d], pi[1][d]};
        unsigned cnt_wave = 0;
#pragma unroll
        for (int t = 0; t < kGroups; ++t) {
          const int c4 = lane * 4 + t * 256;
          if (c4 < cw) {
            real pc[4][DIM];
            load_points<DIM, real>(lds_pos, c4, pc);
            if (thr) {
              pair_error_rows2<DIM, true>(pc[0], pi2, w4[t][0].x, w4[t][1].x, err, cnt_wave);
              pair_error_rows2<DIM, true>(pc[1], pi2, w4[t][0].y, w4[t][1].y, err, cnt_wave);
              pair_error_rows2<DIM, true>(pc[2], pi2, w4[t][0].z, w4[t][1].z, err, cnt_wave);
              pair_error_rows2<DIM, true>(pc[3], pi2, w4[t][0].w, w4[t][1].w, err, cnt_wave);
            } else {
              pair_error_rows2<DIM, false>(pc[0], pi2, w4[t][0].x, w4[t][1].x, err, cnt_wave);
              pair_error_rows2<DIM, false>(pc[1], pi2, w4[t][0].y, w4[t][1].y, err, cnt_wave);
              pair_error_rows2<DIM, false>(pc[2], pi2, w4[t][0].z, w4[t][1].z, err, cnt_wave);
              pair_error_rows2<DIM, false>(pc[3], pi2, w4[t][0].w, w4[t][1].w, err, cnt_wave);
            }
          }
        }
        if (lane == 0) cnt += cnt_wave;   // the wave's count rides in lane 0's counter
        err_d += (double)err;
        err = 0.0f;
        continue;
      }
    }
#pragma unroll
    for (int t = 0; t < kGroups; ++t) {
      const int c4 = lane * 4 + t * 256;
      if (c4 >= cw) continue;
      const int col = cb + c4;
      real pc[4][DIM];
      load_points<DIM, real>(lds_pos, c4, pc);
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int i = rows[r];
        bool take[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = col + q;
          if (PARITY) {
            const bool even = ((i + c) & 1) == 0;
            take[q] = (i != 0x7fffffff) & ((even & (c > i)) | (!even & (c < i)));
          } else {
            take[q] = c > i;
          }
        }
        if (thr) {
          pair_error<DIM, real, true>(pc[0], pi[r], w4[t][r].x, take[0], err, cnt);
          pair_error<DIM, real, true>(pc[1], pi[r], w4[t][r].y, take[1], err, cnt);
          pair_error<DIM, real, true>(pc[2], pi[r], w4[t][r].z, take[2], err, cnt);
          pair_error<DIM, real, true>(pc[3], pi[r], w4[t][r].w, take[3], err, cnt);
        } else {
          pair_error<DIM, real, false>(pc[0], pi[r], w4[t][r].x, take[0], err, cnt);
          pair_error<DIM, real, false>(pc[1], pi[r], w4[t][r].y, take[1], err, cnt);
          pair_error<DIM, real, false>(pc[2], pi[r], w4[t][r].z, take[2], err, cnt);
          pair_error<DIM, real, false>(pc[3], pi[r], w4[t][r].w, take[3], err, cnt);
        }
      }
    }
    // fold the fp32 running sum into f64 once per row batch (<= 16*RPW terms per lane before that)
    err_d += (double)err;
    err = 0.0f;
  }

  double s = wave_sum<double>(err_d);
  unsigned long long c64 = cnt;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) c64 += __shfl_xor(c64, m, 64);
  if (lane == 0) { sh_s[wave] = s; sh_c[wave] = c64; }
  __syncthreads();
  if (tid == 0) {
    double ts = 0.0;
    unsigned long long tc = 0;
    for (int w = 0; w < CFG::WAVES; ++w) { ts += sh_s[w]; tc += sh_c[w]; }
    part_sum[part] = ts;
    part_cnt[part] = tc;
  }
}

// ---------------------------------------------------------------------------------------
// Edge MAE (reference src/optimization.cpp:54-81): per-block partial (sum, count) in f64.
// ---------------------------------------------------------------------------------------
template <int DIM, typename real, typename tgt_t>
__global__ __launch_bounds__(kThreads) void edge_error_kernel(
    const real* __restrict__ pos, const int* __restrict__ ei, const int* __restrict__ ej,
    const tgt_t* __restrict__ et, const int8_t* __restrict__ ec, long long n_edges,
    double* __restrict__ part_sum, unsigned long long* __restrict__ part_cnt,
    const RunState* st) {
  if (st != nullptr && st->stopped) return;
  double s = 0.0;
  unsigned long long cnt = 0;
  for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < n_edges;
       e += (long long)gridDim.x * kThreads) {
    const int a = ei[e], b = ej[e];
    double q = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      const double diff = (double)pos[(size_t)b * DIM + d] - (double)pos[(size_t)a * DIM + d];
      q = fma(diff, diff, q);
    }
    const double r = ::sqrt(q);
    const double t = (double)et[e];
    const int c = ec[e];
    const bool contributes = (c == 0) || (c == 1 && r < t) || (c == -1 && r > t);
    if (contributes) { s += fabs(t - r); ++cnt; }
  }
  __shared__ double sh_s[kWaves];
  __shared__ unsigned long long sh_c[kWaves];
  s = wave_sum<double>(s);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh_s[wave] = s; sh_c[wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0;
    unsigned long long tc = 0;
    for (int w = 0; w < kWaves; ++w) { ts += sh_s[w]; tc += sh_c[w]; }
    part_sum[blockIdx.x] = ts;
    part_cnt[blockIdx.x] = tc;
  }
}

// Row-sharded runs: folds a block's per-workgroup partials into one (sum, count) in a fixed order
// and stores it into slot `slot` of every block's rank table (peer stores), so that after the
// cross-block barrier every block's controller reads the same n_ranks entries in the same order.
__global__ __launch_bounds__(1024) void reduce_push_kernel(
    const double* __restrict__ part_sum, const unsigned long long* __restrict__ part_cnt, int n_parts,
    double* const* __restrict__ dst_sum, unsigned long long* const* __restrict__ dst_cnt, int n_dst,
    int slot, const RunState* st) {
  if (st != nullptr && st->stopped) return;
  __shared__ double sh_s[1024];
  __shared__ unsigned long long sh_c[1024];
  double s = 0.0;
  unsigned long long c = 0;
  for (int p = threadIdx.x; p < n_parts; p += 1024) { s += part_sum[p]; c += part_cnt[p]; }
  sh_s[threadIdx.x] = s;
  sh_c[threadIdx.x] = c;
  __syncthreads();
  for (int half = 512; half >= 1; half >>= 1) {
    if (threadIdx.x < half) {
      sh_s[threadIdx.x] += sh_s[threadIdx.x + half];
      sh_c[threadIdx.x] += sh_c[threadIdx.x + half];
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < n_dst) {
    dst_sum[threadIdx.x][slot] = sh_s[0];
    dst_cnt[threadIdx.x][slot] = sh_c[0];
  }
}

// A block's partials folded into two doubles (sum, count) in a fixed order: the operand of the
// multi-process driver's all-reduce.
__global__ __launch_bounds__(1024) void reduce_total_kernel(
    const double* __restrict__ part_sum, const unsigned long long* __restrict__ part_cnt, int n_parts,
    double* __restrict__ out2, const RunState* st) {
  if (st != nullptr && st->stopped) { if (threadIdx.x == 0) { out2[0] = 0.0; out2[1] = 0.0; } return; }
  __shared__ double sh_s[1024];
  __shared__ unsigned long long sh_c[1024];
  double s = 0.0;
  unsigned long long c = 0;
  for (int p = threadIdx.x; p < n_parts; p += 1024) { s += part_sum[p]; c += part_cnt[p]; }
  sh_s[threadIdx.x] = s;
  sh_c[threadIdx.x] = c;
  __syncthreads();
  for (int half = 512; half >= 1; half >>= 1) {
    if (threadIdx.x < half) {
      sh_s[threadIdx.x] += sh_s[threadIdx.x + half];
      sh_c[threadIdx.x] += sh_c[threadIdx.x + half];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out2[0] = sh_s[0]; out2[1] = (double)sh_c[0]; }
}

// Single-block controller step: reduces the partials in a fixed order, runs the reference's
// three-way classification, snapshots positions when the error improved, advances the run
// state and mirrors it to the pinned host mailbox.
constexpr int kCtlThreads = 1024;

template <typename real>
__global__ __launch_bounds__(kCtlThreads) void controller_kernel(
    RunState* st, RunState* mailbox, const double* __restrict__ part_sum,
    const unsigned long long* __restrict__ part_cnt, int n_parts, const real* __restrict__ pos,
    real* __restrict__ best_pos, long long n_values, int iter1, double k_after,
    double* __restrict__ trace, int trace_cap, const double* __restrict__ total2) {
  // total2 != nullptr: the error sum and count arrive already reduced (two doubles: the row-sharded
  // multi-process driver all-reduces them over RCCL); the partial arrays are ignored
  if (st->stopped) return;
  __shared__ double sh_s[kCtlThreads];
  __shared__ unsigned long long sh_c[kCtlThreads];
  __shared__ int sh_action;
  double s = 0.0;
  unsigned long long c = 0;
  if (total2 == nullptr)
    for (int p = threadIdx.x; p < n_parts; p += kCtlThreads) { s += part_sum[p]; c += part_cnt[p]; }
  else if (threadIdx.x == 0) { s = total2[0]; c = (unsigned long long)total2[1]; }
  sh_s[threadIdx.x] = s;
  sh_c[threadIdx.x] = c;
  __syncthreads();
  for (int half = kCtlThreads / 2; half >= 1; half >>= 1) {
    if (threadIdx.x < half) {
      sh_s[threadIdx.x] += sh_s[threadIdx.x + half];
      sh_c[threadIdx.x] += sh_c[threadIdx.x + half];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double err = sh_c[0] > 0 ? sh_s[0] / (double)sh_c[0] : 0.0;  // reference :296
    const int action = st->ctl.observe(err, iter1, k_after);
    st->last_mae = err;
    if (trace != nullptr && st->n_checks < trace_cap) {   // (iteration, MAE, k) of every check
      trace[3 * st->n_checks + 0] = (double)iter1;
      trace[3 * st->n_checks + 1] = err;
      trace[3 * st->n_checks + 2] = k_after;
    }
    st->n_checks += 1;
    st->iter_base = iter1;
    st->k_base = k_after;
    if (action & 1) { st->stopped = 1; st->converged = 1; }
    sh_action = action;
  }
  __syncthreads();
  if (sh_action & 2) {
    // snapshot: 16-byte copies (both buffers come from hipMalloc, so they are 256-B aligned)
    const long long bytes = n_values * (long long)sizeof(real);
    const long long n16 = bytes / 16;
    const uint4* src = reinterpret_cast<const uint4*>(pos);
    uint4* dst = reinterpret_cast<uint4*>(best_pos);
    for (long long q = threadIdx.x; q < n16; q += kCtlThreads) dst[q] = src[q];
    for (long long q = n16 * 16 / (long long)sizeof(real) + threadIdx.x; q < n_values; q += kCtlThreads)
      best_pos[q] = pos[q];
  }
  __syncthreads();
  if (threadIdx.x == 0 && mailbox != nullptr) {
    *mailbox = *st;
    __threadfence_system();
  }
}

// ---------------------------------------------------------------------------------------
// Encoders: reference dense inputs (column-major f64 targets, i32 codes; the upper-triangle
// cell of each unordered pair is authoritative, src/optimization.cpp:217) -> fp32 words.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void encode_dense_kernel(
    const double* __restrict__ D, const int* __restrict__ T, int n, int row_begin, int row_end,
    int ld, uint32_t* __restrict__ out, const int* __restrict__ perm) {
  // grid: x = row of the block (no 65535 limit), y = 256-column group
  // perm (nullable): session label -> caller's label (the block is stored in session labels)
  const int c = blockIdx.y * kThreads + threadIdx.x;
  const int i = row_begin + blockIdx.x;
  if (c >= ld || i >= row_end) return;
  uint32_t w = kInfWord;  // diagonal and padding: see relax_common.h
  if (c < n && c != i) {
    const int oi = perm ? perm[i] : i, oc = perm ? perm[c] : c;
    const int lo = oi < oc ? oi : oc, hi = oi < oc ? oc : oi;
    const size_t cell = (size_t)lo + (size_t)hi * n;
    w = encode_target(D[cell], T[cell]);
  }
  out[enc_index(i - row_begin, c, ld)] = w;
}

// Order-independent fingerprint of the measured cells the dense MAE pass would reduce, used to
// verify that a caller's edge list is exactly that set (same pairs, same encoded targets).
TL_HD inline uint64_t cell_fingerprint(int lo, int hi, uint32_t word) {
  return mix64(((uint64_t)(uint32_t)lo << 32) ^ (uint64_t)(uint32_t)hi ^ ((uint64_t)word * 0x9e3779b97f4a7c15ull));
}
TL_HD inline bool dense_takes(int i, int c, bool parity) {
  if (!parity) return c > i;
  const bool even = ((i + c) & 1) == 0;
  return (even & (c > i)) | (!even & (c < i));
}

__global__ __launch_bounds__(kThreads) void upper_fingerprint_kernel(
    const uint32_t* __restrict__ enc, int n, int row_begin, int row_end, int ld, int parity,
    unsigned long long* __restrict__ out /* [0]=sum of fingerprints, [1]=count */) {
  const int i = row_begin + blockIdx.x;
  if (i >= row_end) return;
  unsigned long long fp = 0, cnt = 0;
  for (int c = threadIdx.x; c < n; c += kThreads) {
    if (c == i || !dense_takes(i, c, parity != 0)) continue;
    const uint32_t w = enc[enc_index(i - row_begin, c, ld)];
    if (w == kInfWord) continue;
    fp += cell_fingerprint(i < c ? i : c, i < c ? c : i, w);
    ++cnt;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    fp += __shfl_xor(fp, m, 64);
    cnt += __shfl_xor(cnt, m, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[0], fp);
    atomicAdd(&out[1], cnt);
  }
}

// rowflags[i] = 1 when encoded row i holds a ">" target or a measured "<" target; *measured += the row's
// measured cells (every word but the unmeasured one).
__global__ __launch_bounds__(kThreads) void row_flags_kernel(const uint32_t* __restrict__ enc,
                                                             int rows, int ld,
                                                             unsigned char* __restrict__ flags,
                                                             unsigned long long* __restrict__ measured) {
  const int i = blockIdx.x;
  if (i >= rows) return;
  int any = 0;
  unsigned cnt = 0;
  for (int c = threadIdx.x; c < ld; c += kThreads) {
    const uint32_t w = enc[enc_index(i, c, ld)];
    const uint32_t code = w & kCodeMask;
    any |= (code == 1u) | ((code == 2u) & (w != kInfWord));
    cnt += w != kInfWord ? 1u : 0u;
  }
  any = __syncthreads_or(any);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
  if ((threadIdx.x & 63) == 0 && cnt != 0) atomicAdd(measured, (unsigned long long)cnt);
  if (threadIdx.x == 0) flags[i] = any ? 1 : 0;
}

__global__ __launch_bounds__(kThreads) void fill_unmeasured_kernel(
    int n, int row_begin, int row_end, int ld, uint32_t* __restrict__ out) {
  const int c = blockIdx.y * kThreads + threadIdx.x;   // grid: x = row, y = 256-column group
  const int i = row_begin + blockIdx.x;
  if (c >= ld || i >= row_end) return;
  out[enc_index(i - row_begin, c, ld)] = kInfWord;
}

__global__ __launch_bounds__(kThreads) void scatter_edges_kernel(
    const int* __restrict__ ei, const int* __restrict__ ej, const double* __restrict__ ed,
    const int* __restrict__ ec, long long n_edges, int n, int row_begin, int row_end, int ld,
    uint32_t* __restrict__ out, const int* __restrict__ inv) {
  // inv (nullable): caller's label -> session label
  const long long e = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (e >= n_edges) return;
  int a = ei[e], b = ej[e];
  if (a == b || a < 0 || b < 0 || a >= n || b >= n) return;
  if (inv != nullptr) { a = inv[a]; b = inv[b]; }
  const uint32_t w = encode_target(ed[e], ec[e]);
  if (a >= row_begin && a < row_end) out[enc_index(a - row_begin, b, ld)] = w;
  if (b >= row_begin && b < row_end) out[enc_index(b - row_begin, a, ld)] = w;
}

// est_distances = as.matrix(dist(positions)) (reference R/core.R:474), f64, rows [row0, row0 + rows).
// positions: n x dim row-major f64; out: rows x n f64, row-major (the full matrix is symmetric, so
// R's column-major reading of an n x n result is the same matrix).
__global__ __launch_bounds__(kThreads) void pdist_kernel(const double* __restrict__ pos, int n,
                                                         int dim, int row0, int rows,
                                                         double* __restrict__ out) {
  const int j = blockIdx.y * kThreads + threadIdx.x;   // grid: x = row of the block, y = 256-column group
  const int r = blockIdx.x;
  if (j >= n || r >= rows) return;
  const int i = row0 + r;
  double s = 0.0;
  for (int d = 0; d < dim; ++d) {
    const double dev = pos[(size_t)i * dim + d] - pos[(size_t)j * dim + d];
    s += dev * dev;
  }
  out[(size_t)r * n + j] = ::sqrt(s);
}

}  // namespace topolow
