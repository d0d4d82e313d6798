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
#include "relax_common.h"

namespace topolow {

constexpr int kThreads = 256;       // 4 waves
constexpr int kWaves = kThreads / 64;
constexpr int kRowsPerWave = 4;
constexpr int kRowsPerWG = kWaves * kRowsPerWave;  // 16
// slab columns staged in LDS at a time (multiple of 256); LDS use stays <= 40 KiB
template <int DIM, typename real> struct ChunkOf {
  static constexpr int value = (sizeof(real) * DIM * 1024 <= 40960) ? 1024 : ((sizeof(real) * DIM * 512 <= 40960) ? 512 : 256);
};

template <typename real> struct Math;
template <> struct Math<float> {
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
};
template <> struct Math<double> {
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
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

template <int DIM, typename real>
__device__ __forceinline__ void pair_accum(const real (&pc)[DIM], const real (&pi)[DIM],
                                           uint32_t w, real k2, real chalf, real inv_ns,
                                           real inv_g, real (&acc)[DIM]) {
  real dx[DIM];
  real s = 0;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    dx[d] = pc[d] - pi[d];
    s = fma(dx[d], dx[d], s);
  }
  const real r = Math<real>::sqrt(s);
  const real rs = r + (real)0.01;
  const uint32_t code = w & kCodeMask;
  const real t = (real)bits_f32(w & ~kCodeMask);
  // branch-free classification (bitwise ops on purpose: no short-circuit control flow)
  const bool measured = (w & 0x7ffffffcu) != kInfWord;
  const bool spring = measured & ((code == 0u) | ((code == 1u) & (r < t)) | ((code == 2u) & (r > t)));
  const real inv = Math<real>::rcp(rs);
  const real fs = k2 * (t - r) * inv * inv_ns;
  const real fr = chalf * inv * inv * inv * inv_g;
  real coef = spring ? fs : fr;
  coef = (code == 3u) ? (real)0 : coef;
#pragma unroll
  for (int d = 0; d < DIM; ++d) acc[d] = fma(dx[d], coef, acc[d]);
}

// One slab stage for rows [row_begin,row_end).
//   denc    : (row_end-row_begin) x ld encoded targets, row-major, ld % 64 == 0
//   pos_in  : n x DIM row-major, all points, read-only in this launch
//   pos_out : n x DIM row-major; rows [row_begin,row_end) are written
//   st      : run state (nullable): launch is a no-op once st->stopped is set; non-finite
//             results are reported through st->first_nonfinite
template <int DIM, typename real>
__global__ __launch_bounds__(kThreads) void slab_stage_kernel(
    const uint32_t* __restrict__ denc, int ld, int row_begin, int row_end, int n,
    const real* __restrict__ pos_in, real* __restrict__ pos_out,
    const float* __restrict__ gplus, RunState* st, SlabRanges rg, int iter1, double k,
    double c_rep) {
  if (st != nullptr && st->stopped) return;

  constexpr int kChunk = ChunkOf<DIM, real>::value;
  __shared__ real lds_pos[DIM * kChunk];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = row_begin + blockIdx.x * kRowsPerWG + wave * kRowsPerWave;

  const real k2 = (real)(2.0 * k);
  const real chalf = (real)(0.5 * c_rep);

  real pi[kRowsPerWave][DIM];
  real acc[kRowsPerWave][DIM];
  real inv_ns[kRowsPerWave], inv_g[kRowsPerWave];
  const uint32_t* rowp[kRowsPerWave];
#pragma unroll
  for (int r = 0; r < kRowsPerWave; ++r) {
    const int row = row0 + r;
    const int rr = row < row_end ? row : row_end - 1;  // clamp: result discarded below
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      pi[r][d] = uniform(pos_in[(size_t)rr * DIM + d]);  // wave-uniform: lives in SGPRs
      acc[r][d] = 0;
    }
    const real g = (real)gplus[rr];
    inv_ns[r] = uniform((real)1 / ((real)4 * g + (real)k));
    inv_g[r] = uniform((real)1 / g);
    rowp[r] = denc + (size_t)(rr - row_begin) * ld;
  }

#pragma unroll 1
  for (int part = 0; part < 2; ++part) {
    const int rb = part == 0 ? rg.b0 : rg.b1;
    const int re = part == 0 ? rg.e0 : rg.e1;
#pragma unroll 1
    for (int cb = rb; cb < re; cb += kChunk) {
      const int cw = min(kChunk, re - cb);  // multiple of 4
      __syncthreads();  // previous chunk fully consumed
      // stage column points [cb, cb+cw) into LDS, structure-of-arrays
      for (int c = tid; c < cw; c += kThreads) {
        const int col = cb + c;
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          lds_pos[d * kChunk + c] = col < n ? pos_in[(size_t)col * DIM + d] : (real)0;
      }
      __syncthreads();
#pragma unroll 1
      for (int c4 = lane * 4; c4 < cw; c4 += 256) {
        uint4 w4[kRowsPerWave];
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r)
          w4[r] = *reinterpret_cast<const uint4*>(rowp[r] + cb + c4);
        real pc[4][DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
#pragma unroll
          for (int q = 0; q < 4; ++q) pc[q][d] = lds_pos[d * kChunk + c4 + q];
        }
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
          pair_accum<DIM, real>(pc[0], pi[r], w4[r].x, k2, chalf, inv_ns[r], inv_g[r], acc[r]);
          pair_accum<DIM, real>(pc[1], pi[r], w4[r].y, k2, chalf, inv_ns[r], inv_g[r], acc[r]);
          pair_accum<DIM, real>(pc[2], pi[r], w4[r].z, k2, chalf, inv_ns[r], inv_g[r], acc[r]);
          pair_accum<DIM, real>(pc[3], pi[r], w4[r].w, k2, chalf, inv_ns[r], inv_g[r], acc[r]);
        }
      }
    }
  }

#pragma unroll
  for (int r = 0; r < kRowsPerWave; ++r) {
    const int row = row0 + r;
    bool finite = true;
    real out[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      const real total = wave_sum<real>(acc[r][d]);
      out[d] = pi[r][d] - total;
      finite = finite && isfinite(out[d]);
    }
    if (lane == 0 && row < row_end) {
#pragma unroll
      for (int d = 0; d < DIM; ++d) pos_out[(size_t)row * DIM + d] = out[d];
      if (!finite && st != nullptr) atomicMin(&st->first_nonfinite, iter1);
    }
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

// Single-block controller step: reduces the partials in a fixed order, runs the reference's
// three-way classification, snapshots positions when the error improved, advances the run
// state and mirrors it to the pinned host mailbox.
template <typename real>
__global__ __launch_bounds__(kThreads) void controller_kernel(
    RunState* st, RunState* mailbox, const double* __restrict__ part_sum,
    const unsigned long long* __restrict__ part_cnt, int n_parts, const real* __restrict__ pos,
    real* __restrict__ best_pos, long long n_values, int iter1, double k_after) {
  if (st->stopped) return;
  __shared__ double sh_s[kThreads];
  __shared__ unsigned long long sh_c[kThreads];
  __shared__ int sh_action;
  double s = 0.0;
  unsigned long long c = 0;
  for (int p = threadIdx.x; p < n_parts; p += kThreads) { s += part_sum[p]; c += part_cnt[p]; }
  sh_s[threadIdx.x] = s;
  sh_c[threadIdx.x] = c;
  __syncthreads();
  for (int half = kThreads / 2; half >= 1; half >>= 1) {
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
    st->n_checks += 1;
    st->iter_base = iter1;
    st->k_base = k_after;
    if (action & 1) { st->stopped = 1; st->converged = 1; }
    sh_action = action;
  }
  __syncthreads();
  if (sh_action & 2) {
    for (long long q = threadIdx.x; q < n_values; q += kThreads) best_pos[q] = pos[q];
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
    int ld, uint32_t* __restrict__ out) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  const int i = row_begin + blockIdx.y;
  if (c >= ld || i >= row_end) return;
  uint32_t w = kSkipWord;
  if (c < n && c != i) {
    const int lo = i < c ? i : c, hi = i < c ? c : i;
    const size_t cell = (size_t)lo + (size_t)hi * n;
    w = encode_target(D[cell], T[cell]);
  }
  out[(size_t)(i - row_begin) * ld + c] = w;
}

__global__ __launch_bounds__(kThreads) void fill_unmeasured_kernel(
    int n, int row_begin, int row_end, int ld, uint32_t* __restrict__ out) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  const int i = row_begin + blockIdx.y;
  if (c >= ld || i >= row_end) return;
  out[(size_t)(i - row_begin) * ld + c] = (c < n && c != i) ? kInfWord : kSkipWord;
}

__global__ __launch_bounds__(kThreads) void scatter_edges_kernel(
    const int* __restrict__ ei, const int* __restrict__ ej, const double* __restrict__ ed,
    const int* __restrict__ ec, long long n_edges, int n, int row_begin, int row_end, int ld,
    uint32_t* __restrict__ out) {
  const long long e = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (e >= n_edges) return;
  const int a = ei[e], b = ej[e];
  if (a == b || a < 0 || b < 0 || a >= n || b >= n) return;
  const uint32_t w = encode_target(ed[e], ec[e]);
  if (a >= row_begin && a < row_end) out[(size_t)(a - row_begin) * ld + b] = w;
  if (b >= row_begin && b < row_end) out[(size_t)(b - row_begin) * ld + a] = w;
}

// est_distances = as.matrix(dist(positions)) (reference R/core.R:474), f64.
// positions: n x dim row-major f64; out: n x n f64 (symmetric, so layout-agnostic).
__global__ __launch_bounds__(kThreads) void pdist_kernel(const double* __restrict__ pos, int n,
                                                         int dim, double* __restrict__ out) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= n) return;
  double s = 0.0;
  for (int d = 0; d < dim; ++d) {
    const double dev = pos[(size_t)i * dim + d] - pos[(size_t)j * dim + d];
    s += dev * dev;
  }
  out[(size_t)i * n + j] = ::sqrt(s);
}

}  // namespace topolow
