// topolow_amd/csrc/relax_gs.h -- exact Gauss-Seidel relaxation, one workgroup per embedding.
//
// This is the reference's algorithm itself (src/optimization.cpp:193-374 of the reference):
// every unordered pair is visited once per iteration and BOTH endpoints move immediately.
// The only freedom taken is the visiting order.  The reference draws a uniformly random
// order (std::shuffle, :196); here each iteration visits the pairs in a randomised
// round-robin tournament order (circle method over a fresh random permutation of the
// points, random starting round).  A round holds floor(n/2) DISJOINT pairs, so its pairs
// commute exactly and one workgroup relaxes them in parallel out of LDS; rounds are
// separated by a workgroup barrier.  Any sequential replay of the same rounds -- e.g. the
// CPU oracle fed topolow_gs_pair_order() -- performs the identical floating-point
// operations (contraction is off in the pair update), so f64 results agree bit for bit.
//
// The whole embedding (all iterations, the edge-MAE checks of :54-81/:294-296, the
// three-way convergence controller of :303-357, the best-state snapshot/restore and the
// non-finite guard of :359-361) runs inside ONE launch; a grid of B workgroups relaxes B
// independent embeddings (the reference's only parallel mode: one embedding per process,
// R/adaptive_sampling.R:666).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/topolow_relax.h"
#include "relax_common.h"

namespace topolow {

constexpr uint64_t kGsKeyStream = 0x6500ull;
constexpr uint64_t kGsRoundStream = 0x6501ull;

TL_HD inline uint32_t gs_key(uint64_t seed, int iter, int i) {
  return (uint32_t)(rnd64(seed, kGsKeyStream, ((uint64_t)(uint32_t)iter << 32) | (uint32_t)i) >> 32);
}
TL_HD inline int gs_round0(uint64_t seed, int iter, int m1) {
  return m1 > 0 ? (int)rnd_below(rnd64(seed, kGsRoundStream, (uint64_t)(uint32_t)iter), (uint32_t)m1) : 0;
}
// Players of pair slot p in round rr (circle method over M = m1 + 1 players, M even).
TL_HD inline void gs_round_pair(int m1, int rr, int p, int* a, int* b) {
  if (p == 0) { *a = m1; *b = rr; return; }
  int x = rr + p; if (x >= m1) x -= m1;
  int y = rr - p; if (y < 0) y += m1;
  *a = x; *b = y;
}

// Host: the exact visiting order of iteration `iter` (pairs of point indices).
inline int64_t gs_pair_order(int n, uint64_t seed, int iter, int32_t* pairs_out) {
  if (n < 2) return 0;
  const int M = n + (n & 1), m1 = M - 1;
  std::vector<uint32_t> keys(n);
  for (int i = 0; i < n; ++i) keys[i] = gs_key(seed, iter, i);
  std::vector<int> perm(n);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return keys[x] < keys[y]; });
  const int r0 = gs_round0(seed, iter, m1);
  int64_t cnt = 0;
  for (int r = 0; r < m1; ++r) {
    int rr = r + r0; if (rr >= m1) rr -= m1;
    for (int p = 0; p < M / 2; ++p) {
      int a, b;
      gs_round_pair(m1, rr, p, &a, &b);
      if (a >= n || b >= n) continue;  // the bye of an odd field
      if (pairs_out) { pairs_out[2 * cnt] = perm[a]; pairs_out[2 * cnt + 1] = perm[b]; }
      ++cnt;
    }
  }
  return cnt;
}

// ---------------------------------------------------------------------------------------
struct GsOut {
  double final_mae;
  double final_k;
  int converged;
  int iterations;   // best iteration (reference :373,378)
  int iters_run;
  int n_checks;
  int nonfinite_iter;  // != 0: "Numerical instability at iteration %d"
  int pad;
};

template <typename real>
struct GsDev {
  const real* tm;        // n x n targets, column-major as R passes them; cell [lo + hi*n] is read
  const int8_t* cm;      // n x n threshold codes (0, 1, -1)
  const double* gplus;   // n: degree + 1
  const int* ei; const int* ej; const double* et; const int8_t* ec;  // MAE edge list
  // SPARSE kernels only: the measured pairs as a CSR table over rows i (entries j > i, ascending),
  // copied into LDS at start so the round loop touches no global memory at all
  const int* row_off;            // n + 1
  const unsigned short* ecol;    // n_edges
  const unsigned short* erow;    // n_edges
  const real* etgt;              // n_edges (targets in the kernel's precision)
  const int8_t* ecode;           // n_edges
  real* pos;             // n x dim row-major: in = initial positions, out = best positions
  real* best;            // n x dim scratch
  GsOut* out;
  long long n_edges;
  double k0, cooling, c_rep, eps;
  uint64_t seed;
  int n, n_iter, check_freq, window;
};

template <int DIM, typename real>
__device__ __forceinline__ void gs_pair_update(real* pi, real* pj, real target, int code,
                                               double gi, double gj, double k, double c_rep) {
#pragma clang fp contract(off)
  real dist_sq = 0;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const real diff = pj[d] - pi[d];
    dist_sq += diff * diff;
  }
  const real dist = sqrt(dist_sq);
  const real dist_stable = dist + (real)0.01;
  bool spring = false;
  if (isfinite(target)) {
    if (code == 0) spring = true;
    else if (code == 1) spring = dist < target;
    else spring = dist > target;
  }
  if (spring) {
    const real factor = (real)2.0 * (real)k * (target - dist) / dist_stable;
    const real norm_i = (real)4.0 * (real)gi + (real)k;
    const real norm_j = (real)4.0 * (real)gj + (real)k;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      const real delta = pj[d] - pi[d];
      const real f = delta * factor;
      pi[d] -= f / norm_i;
      pj[d] += f / norm_j;
    }
  } else {
    const real mag = (real)c_rep / ((real)2.0 * dist_stable * dist_stable * dist_stable);
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      const real delta = pj[d] - pi[d];
      const real f = delta * mag;
      pi[d] -= f / (real)gi;
      pj[d] += f / (real)gj;
    }
  }
}

// fp32 variant: not bit-comparable with anything anyway, so the 2*DIM divisions per pair of the
// reference's operation order collapse into two reciprocals (v_rcp_f32) and v_sqrt_f32.
template <int DIM>
__device__ __forceinline__ void gs_pair_update(float* pi, float* pj, float target, int code,
                                               double gi, double gj, double k, double c_rep) {
  float dx[DIM];
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    dx[d] = pj[d] - pi[d];
    s = fmaf(dx[d], dx[d], s);
  }
  const float dist = __builtin_amdgcn_sqrtf(s);
  const float inv = __builtin_amdgcn_rcpf(dist + 0.01f);
  bool spring = false;
  if (isfinite(target)) {
    if (code == 0) spring = true;
    else if (code == 1) spring = dist < target;
    else spring = dist > target;
  }
  float ci, cj;
  if (spring) {
    const float factor = 2.0f * (float)k * (target - dist) * inv;
    ci = factor * __builtin_amdgcn_rcpf(4.0f * (float)gi + (float)k);
    cj = factor * __builtin_amdgcn_rcpf(4.0f * (float)gj + (float)k);
  } else {
    const float mag = 0.5f * (float)c_rep * inv * inv * inv;
    ci = mag * __builtin_amdgcn_rcpf((float)gi);
    cj = mag * __builtin_amdgcn_rcpf((float)gj);
  }
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    pi[d] = fmaf(-dx[d], ci, pi[d]);
    pj[d] = fmaf(dx[d], cj, pj[d]);
  }
}

template <int DIM, typename real>
__device__ __forceinline__ void gs_pair_dispatch(real* pi, real* pj, real target, int code, double gi,
                                                 double gj, double k, double c_rep) {
  if constexpr (sizeof(real) == 4) gs_pair_update<DIM>(pi, pj, target, code, gi, gj, k, c_rep);
  else gs_pair_update<DIM, real>(pi, pj, target, code, gi, gj, k, c_rep);
}

// Block-wide sum of (double, u64) in a fixed order; result valid in every thread.
__device__ inline void gs_block_sum(double& s, unsigned long long& c, double* sh_s,
                                    unsigned long long* sh_c) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    s += __shfl_xor(s, m, 64);
    c += __shfl_xor(c, m, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) { sh_s[wave] = s; sh_c[wave] = c; }
  __syncthreads();
  double ts = 0.0;
  unsigned long long tc = 0;
  for (int w = 0; w < nw; ++w) { ts += sh_s[w]; tc += sh_c[w]; }
  s = ts;
  c = tc;
}

//   SPARSE = true: targets come from the LDS-resident CSR table (sparse panels: antigenic data is
//   > 90 % missing); false: from the dense n x n matrix in global memory.
template <int DIM, typename real, bool SPARSE>
__global__ __launch_bounds__(1024) void gs_embed_kernel(const GsDev<real>* __restrict__ problems) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gs_smem[];
  const GsDev<real> P = problems[blockIdx.x];
  const int n = P.n;
  const int M = n + (n & 1), m1 = M - 1, half = M / 2;
  const int tid = threadIdx.x, nthr = blockDim.x;

  // LDS carve: positions | perm | keys | reduction scratch
  real* pos = reinterpret_cast<real*>(gs_smem);
  size_t off = ((size_t)n * DIM * sizeof(real) + 15) & ~(size_t)15;
  int* perm = reinterpret_cast<int*>(gs_smem + off);
  off += ((size_t)(n + 1) * sizeof(int) + 15) & ~(size_t)15;
  uint32_t* keys = reinterpret_cast<uint32_t*>(gs_smem + off);
  off += ((size_t)n * sizeof(uint32_t) + 15) & ~(size_t)15;
  double* lds_g = reinterpret_cast<double*>(gs_smem + off);   // degree + 1 of every point
  off += ((size_t)n * sizeof(double) + 15) & ~(size_t)15;
  double* sh_s = reinterpret_cast<double*>(gs_smem + off);
  off += 16 * sizeof(double);
  unsigned long long* sh_c = reinterpret_cast<unsigned long long*>(gs_smem + off);
  off += 16 * sizeof(unsigned long long);
  int* sh_flag = reinterpret_cast<int*>(gs_smem + off);
  off += 16;
  // CSR table of the measured pairs (SPARSE only)
  const int ne = SPARSE ? (int)P.n_edges : 0;
  real* l_tgt = reinterpret_cast<real*>(gs_smem + off);
  off += ((size_t)ne * sizeof(real) + 7) & ~(size_t)7;
  // per-iteration round schedule of the measured pairs: b_start[r] .. b_start[r+1] index the
  // (slot, edge) entries of the pairs that meet in round r; rt[] are three rotating per-round
  // slot -> edge tables (read this round / filled for the next / being cleared)
  int* b_start = reinterpret_cast<int*>(gs_smem + off);
  off += ((size_t)(n + 2) * sizeof(int) + 7) & ~(size_t)7;
  int* b_cur = reinterpret_cast<int*>(gs_smem + off);
  off += ((size_t)(n + 2) * sizeof(int) + 7) & ~(size_t)7;
  unsigned short* l_col = reinterpret_cast<unsigned short*>(gs_smem + off);
  off += ((size_t)ne * 2 + 7) & ~(size_t)7;
  unsigned short* l_row = reinterpret_cast<unsigned short*>(gs_smem + off);
  off += ((size_t)ne * 2 + 7) & ~(size_t)7;
  unsigned short* b_slot = reinterpret_cast<unsigned short*>(gs_smem + off);
  off += ((size_t)ne * 2 + 7) & ~(size_t)7;
  unsigned short* b_edge = reinterpret_cast<unsigned short*>(gs_smem + off);
  off += ((size_t)ne * 2 + 7) & ~(size_t)7;
  unsigned short* rt = reinterpret_cast<unsigned short*>(gs_smem + off);   // [3][half]
  off += ((size_t)3 * half * 2 + 7) & ~(size_t)7;
  int8_t* l_code = reinterpret_cast<int8_t*>(gs_smem + off);
  if constexpr (SPARSE) {
    for (int q = tid; q < ne; q += nthr) {
      l_tgt[q] = P.etgt[q]; l_col[q] = P.ecol[q]; l_row[q] = P.erow[q]; l_code[q] = P.ecode[q];
    }
  }
  // round (in execution order) and slot in which the players a, b of a measured pair meet
  auto meet = [&](int a, int b, int r0, int& r, int& p) {
    int rr;
    if (a == m1) { rr = b; p = 0; }
    else if (b == m1) { rr = a; p = 0; }
    else {
      rr = (int)(((long long)(a + b) * ((m1 + 1) / 2)) % m1);
      p = a - rr; if (p < 0) p += m1;
      if (p >= half) { p = b - rr; if (p < 0) p += m1; }
    }
    r = rr - r0; if (r < 0) r += m1;
  };
  // dense path: target and code of the pair (i < j) from the n x n matrix
  auto lookup = [&](int i, int j, real& target, int& code) {
    const size_t cell = (size_t)i + (size_t)j * n;
    target = P.tm[cell];
    code = P.cm[cell];
  };

  for (int q = tid; q < n * DIM; q += nthr) {
    const real v = P.pos[q];
    pos[q] = v;
    P.best[q] = v;  // reference :171: best_pos starts as the initial positions
  }
  for (int q = tid; q < n; q += nthr) lds_g[q] = P.gplus[q];
  if (tid == 0) { perm[n] = n; sh_flag[0] = 0; sh_flag[1] = 0; }

  Controller ctl;
  ctl.init(P.k0, P.window, P.eps);
  double k = P.k0;
  int converged = 0, iters_run = 0, n_checks = 0, nonfinite_iter = 0;
  const int check_freq = P.check_freq < 1 ? 10 : P.check_freq;  // reference :181
  __syncthreads();

  for (int iter = 0; iter < P.n_iter; ++iter) {
    // ---- this iteration's random permutation: rank of a hashed key (stable) ----
    for (int i = tid; i < n; i += nthr) keys[i] = gs_key(P.seed, iter, i);
    __syncthreads();
    for (int i = tid; i < n; i += nthr) {
      const uint32_t ki = keys[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        const uint32_t kj = keys[j];
        rank += (kj < ki) || (kj == ki && j < i);
      }
      perm[rank] = i;
    }
    const int r0 = gs_round0(P.seed, iter, m1);
    __syncthreads();

    if constexpr (SPARSE) {
      // ---- bucket the measured pairs by the round in which they meet (host guarantees half <= nthr)
      int* pinv = reinterpret_cast<int*>(keys);   // keys[] is free again: player index of every point
      for (int a = tid; a < n; a += nthr) pinv[perm[a]] = a;
      for (int q = tid; q <= m1 + 1; q += nthr) b_start[q] = 0;
      for (int q = tid; q < 3 * half; q += nthr) rt[q] = 0;
      __syncthreads();
      for (int e = tid; e < ne; e += nthr) {
        int r, p;
        meet(pinv[l_row[e]], pinv[l_col[e]], r0, r, p);
        atomicAdd(&b_start[r + 1], 1);
      }
      __syncthreads();
      if (tid == 0) {
        int run = 0;
        for (int r = 0; r <= m1; ++r) { run += b_start[r]; b_start[r] = run; b_cur[r] = run; }
      }
      __syncthreads();
      for (int e = tid; e < ne; e += nthr) {
        int r, p;
        meet(pinv[l_row[e]], pinv[l_col[e]], r0, r, p);
        const int at = atomicAdd(&b_cur[r], 1);
        b_slot[at] = (unsigned short)p;
        b_edge[at] = (unsigned short)e;
      }
      __syncthreads();
      for (int q = b_start[0] + tid; q < b_start[1]; q += nthr) rt[b_slot[q]] = (unsigned short)(b_edge[q] + 1);
      __syncthreads();
      // ---- m1 rounds of disjoint pairs; one barrier per round ----
      for (int r = 0; r < m1; ++r) {
        int rr = r + r0; if (rr >= m1) rr -= m1;
        unsigned short* rt_now = rt + (r % 3) * half;
        unsigned short* rt_next = rt + ((r + 1) % 3) * half;
        unsigned short* rt_clear = rt + ((r + 2) % 3) * half;
        const int mine = tid < half ? rt_now[tid] : 0;
        if (r + 1 < m1)
          for (int q = b_start[r + 1] + tid; q < b_start[r + 2]; q += nthr)
            rt_next[b_slot[q]] = (unsigned short)(b_edge[q] + 1);
        if (tid < half) {
          rt_clear[tid] = 0;
          int a, b;
          gs_round_pair(m1, rr, tid, &a, &b);
          int i = perm[a], j = perm[b];
          if (i < n && j < n) {
            if (i > j) { const int t = i; i = j; j = t; }  // reference pairs have i<j
            const real target = mine ? l_tgt[mine - 1] : (real)INFINITY;   // unmeasured: R/core.R:345
            const int code = mine ? (int)l_code[mine - 1] : 0;
            gs_pair_dispatch<DIM, real>(pos + (size_t)i * DIM, pos + (size_t)j * DIM, target, code, lds_g[i],
                                        lds_g[j], k, P.c_rep);
          }
        }
        __syncthreads();
      }
    } else
    // ---- m1 rounds of disjoint pairs ----
    if (half <= nthr) {
      // one pair per thread per round: the NEXT round's target word is requested before the
      // current pair is relaxed, so the (random, L2-missing) target fetch overlaps the arithmetic
      // and the barrier instead of heading every round
      int ci = n, cj = n, ccode = 0;
      real ctarget = 0;
      auto fetch = [&](int r, int& fi, int& fj, real& ft, int& fc) {
        int rr = r + r0; if (rr >= m1) rr -= m1;
        fi = n; fj = n; ft = 0; fc = 0;
        if (tid < half) {
          int a, b;
          gs_round_pair(m1, rr, tid, &a, &b);
          int i = perm[a], j = perm[b];
          if (i < n && j < n) {
            if (i > j) { const int t = i; i = j; j = t; }  // reference pairs have i<j
            lookup(i, j, ft, fc);
            fi = i; fj = j;
          }
        }
      };
      fetch(0, ci, cj, ctarget, ccode);
      for (int r = 0; r < m1; ++r) {
        int ni = n, nj = n, ncode = 0;
        real ntarget = 0;
        if (r + 1 < m1) fetch(r + 1, ni, nj, ntarget, ncode);
        if (ci < n) {
          gs_pair_dispatch<DIM, real>(pos + (size_t)ci * DIM, pos + (size_t)cj * DIM, ctarget, ccode,
                                      lds_g[ci], lds_g[cj], k, P.c_rep);
        }
        ci = ni; cj = nj; ctarget = ntarget; ccode = ncode;
        __syncthreads();
      }
    } else {
      for (int r = 0; r < m1; ++r) {
        int rr = r + r0; if (rr >= m1) rr -= m1;
        for (int p = tid; p < half; p += nthr) {
          int a, b;
          gs_round_pair(m1, rr, p, &a, &b);
          int i = perm[a], j = perm[b];
          if (i < n && j < n) {
            if (i > j) { const int t = i; i = j; j = t; }  // reference pairs have i<j
            real target;
            int code;
            lookup(i, j, target, code);
            gs_pair_dispatch<DIM, real>(pos + (size_t)i * DIM, pos + (size_t)j * DIM, target, code,
                                        lds_g[i], lds_g[j], k, P.c_rep);
          }
        }
        __syncthreads();
      }
    }
    iters_run = iter + 1;
    k *= (1.0 - P.cooling);  // reference :289

    // ---- convergence check (reference :294-357) ----
    if ((iter + 1) % check_freq == 0 || iter == P.n_iter - 1) {
      double s = 0.0;
      unsigned long long c = 0;
      for (long long e = tid; e < P.n_edges; e += nthr) {
        const int a = SPARSE ? (int)l_row[e] : P.ei[e], b = SPARSE ? (int)l_col[e] : P.ej[e];
        double q = 0.0;
        {
#pragma clang fp contract(off)
#pragma unroll
          for (int d = 0; d < DIM; ++d) {
            const double diff = (double)pos[(size_t)b * DIM + d] - (double)pos[(size_t)a * DIM + d];
            q += diff * diff;
          }
        }
        const double rdist = sqrt(q);
        const double t = SPARSE ? (double)l_tgt[e] : P.et[e];
        const int cd = SPARSE ? (int)l_code[e] : (int)P.ec[e];
        if ((cd == 0) || (cd == 1 && rdist < t) || (cd == -1 && rdist > t)) {
          s += fabs(t - rdist);
          ++c;
        }
      }
      gs_block_sum(s, c, sh_s, sh_c);
      const double err = c > 0 ? s / (double)c : 0.0;
      ++n_checks;
      const int action = ctl.observe(err, iter + 1, k);  // identical in every thread
      if (action & 2) {
        for (int q = tid; q < n * DIM; q += nthr) P.best[q] = pos[q];
      }
      if (action & 1) { converged = 1; break; }
    }
    // ---- non-finite guard (reference :359-361) ----
    if ((iter + 1) % 10 == 0) {
      int bad = 0;
      for (int q = tid; q < n * DIM; q += nthr) bad |= !isfinite(pos[q]);
      if (bad) sh_flag[0] = 1;
      __syncthreads();
      if (sh_flag[0]) { nonfinite_iter = iter + 1; break; }
    }
  }

  __syncthreads();
  // restore the best snapshot (reference :324-327, :368-374)
  for (int q = tid; q < n * DIM; q += nthr) P.pos[q] = P.best[q];
  if (tid == 0) {
    GsOut o;
    o.final_mae = ctl.best_mae;
    o.final_k = ctl.best_k;
    o.converged = converged;
    o.iterations = ctl.best_iter;
    o.iters_run = iters_run;
    o.n_checks = n_checks;
    o.nonfinite_iter = nonfinite_iter;
    o.pad = 0;
    *P.out = o;
  }
}

// ---------------------------------------------------------------------------------------
// Host driver
// ---------------------------------------------------------------------------------------
struct GsProblem {
  const double* initial_positions;  // n x dim col-major
  const double* D;                  // n x n col-major, Inf = unmeasured
  const int32_t* T;                 // n x n col-major
  const int32_t* degrees;
  const int32_t* edge_i; const int32_t* edge_j; const double* edge_dist; const int32_t* edge_thresh;
  int64_t n_edges;
  int n, dim, n_iter, window, check_freq;
  double k0, cooling, c_rep, eps;
  uint64_t seed;
};

struct GsResult {
  double* positions;  // n x dim col-major (caller-owned)
  int converged, iterations, iters_run, n_checks;
  int nonfinite_iter = 0;  // != 0: the non-finite guard fired at this iteration
  double final_mae, final_k;
};

inline size_t gs_lds_bytes(int n, int dim, size_t real_size, long long csr_edges = 0) {
  size_t off = ((size_t)n * dim * real_size + 15) & ~(size_t)15;
  off += ((size_t)(n + 1) * 4 + 15) & ~(size_t)15;
  off += ((size_t)n * 4 + 15) & ~(size_t)15;
  off += ((size_t)n * 8 + 15) & ~(size_t)15;
  off += 16 * 8 + 16 * 8 + 16;
  if (csr_edges > 0) {
    const size_t half = (size_t)(n + (n & 1)) / 2;
    off += ((size_t)csr_edges * real_size + 7) & ~(size_t)7;
    off += 2 * (((size_t)(n + 2) * 4 + 7) & ~(size_t)7);
    off += 4 * (((size_t)csr_edges * 2 + 7) & ~(size_t)7);
    off += ((size_t)3 * half * 2 + 7) & ~(size_t)7;
    off += ((size_t)csr_edges + 15) & ~(size_t)15;
  }
  return off;
}

// LDS budget under which the sparse (LDS-resident) table is used; two workgroups per CU still fit.
constexpr size_t kGsSparseLdsBudget = 78 * 1024;

struct GsHipError { int code; std::string msg; };
#define GS_TRY(expr)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) throw GsHipError{TOPOLOW_ERR_HIP, std::string(#expr) + ": " +   \
                                                                hipGetErrorString(e_)};   \
  } while (0)

template <typename real>
struct GsDeviceProblem {
  real* tm = nullptr; int8_t* cm = nullptr; double* gplus = nullptr;
  int* ei = nullptr; int* ej = nullptr; double* et = nullptr; int8_t* ec = nullptr;
  int* row_off = nullptr; unsigned short* ecol = nullptr; unsigned short* erow = nullptr;
  real* etgt = nullptr;
  real* pos = nullptr; real* best = nullptr;
  void release() {
    (void)hipFree(tm); (void)hipFree(cm); (void)hipFree(gplus); (void)hipFree(ei); (void)hipFree(ej);
    (void)hipFree(et); (void)hipFree(ec); (void)hipFree(row_off); (void)hipFree(ecol); (void)hipFree(erow);
    (void)hipFree(etgt);
    (void)hipFree(pos); (void)hipFree(best);
  }
};

template <int DIM, typename real>
void gs_launch(const GsDev<real>* d_problems, int count, int threads, size_t lds, bool sparse,
               hipStream_t st) {
  auto go = [&](auto kern) {
    if (lds > 64 * 1024)
      GS_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(count), dim3(threads), lds, st, d_problems);
    GS_TRY(hipGetLastError());
  };
  if (sparse) go(&gs_embed_kernel<DIM, real, true>);
  else go(&gs_embed_kernel<DIM, real, false>);
}

// The measured upper-triangle cells of the dense inputs are exactly the caller's edge list (same
// pairs, targets and codes)?  True for everything the reference's R driver builds (R/core.R:383-402
// and :429-436 derive both from one matrix); required before the edge list may stand in for the
// matrix in the LDS-resident table.
inline bool gs_edges_match_matrix(const GsProblem& p) {
  const int n = p.n;
  long long measured = 0;
  for (int j = 1; j < n; ++j)
    for (int i = 0; i < j; ++i)
      if (std::isfinite(p.D[(size_t)i + (size_t)j * n])) ++measured;
  if (measured != p.n_edges) return false;
  for (long long e = 0; e < p.n_edges; ++e) {
    const int a = p.edge_i[e], b = p.edge_j[e];
    if (a < 0 || b <= a || b >= n) return false;
    const size_t cell = (size_t)a + (size_t)b * n;
    if (!(p.D[cell] == p.edge_dist[e])) return false;
    const int tc = p.T[cell], ec = p.edge_thresh[e];
    const int tn = tc == 0 ? 0 : (tc == 1 ? 1 : -1), en = ec == 0 ? 0 : (ec == 1 ? 1 : -1);
    if (tn != en) return false;
  }
  return true;
}

template <typename real>
int gs_run_batch_t(const GsProblem* pbs, GsResult* res, int count, double* device_seconds,
                   char* errbuf, size_t errlen) {
  std::vector<GsDeviceProblem<real>> dev(count);
  std::vector<GsDev<real>> h(count);
  GsOut* d_out = nullptr;
  GsDev<real>* d_problems = nullptr;
  int rc = TOPOLOW_OK;
  auto cleanup = [&] {
    for (auto& d : dev) d.release();
    (void)hipFree(d_out);
    (void)hipFree(d_problems);
  };
  try {
    const int dim = pbs[0].dim;
    size_t lds_max = 0;
    int n_max = 0;
    // one kernel instance per launch: the LDS-resident table is used when EVERY problem of the
    // batch qualifies (edge list == matrix, n < 65536, table within the LDS budget)
    bool sparse = getenv("TOPOLOW_GS_DENSE") == nullptr;
    for (int b = 0; b < count && sparse; ++b) {
      const GsProblem& p = pbs[b];
      sparse = p.n <= 2048 && p.n_edges > 0 && p.n_edges < 65535 &&
               gs_lds_bytes(p.n, dim, sizeof(real), p.n_edges) <= kGsSparseLdsBudget &&
               gs_edges_match_matrix(p);
    }
    GS_TRY(hipMalloc((void**)&d_out, sizeof(GsOut) * count));
    for (int b = 0; b < count; ++b) {
      const GsProblem& p = pbs[b];
      if (p.dim != dim) throw GsHipError{TOPOLOW_ERR_BAD_ARGUMENT, "batch must share ndim"};
      if (p.n < 2) throw GsHipError{TOPOLOW_ERR_TOO_FEW_POINTS, "Need at least 2 points for embedding"};
      const size_t lds = gs_lds_bytes(p.n, dim, sizeof(real), sparse ? p.n_edges : 0);
      if (lds > 160 * 1024)
        throw GsHipError{TOPOLOW_ERR_UNSUPPORTED,
                         "problem too large for the single-workgroup GS kernel (LDS); use the slab schedule"};
      lds_max = std::max(lds_max, lds);
      n_max = std::max(n_max, p.n);
      const size_t nn = (size_t)p.n * p.n, nd = (size_t)p.n * dim, ne = (size_t)p.n_edges;
      std::vector<real> tm(nn), pos(nd);
      std::vector<int8_t> cm(nn), ec(ne ? ne : 1);
      std::vector<double> g(p.n);
      for (size_t q = 0; q < nn; ++q) {
        tm[q] = (real)p.D[q];
        const int c = p.T[q];
        cm[q] = (int8_t)(c == 0 ? 0 : (c == 1 ? 1 : -1));  // reference else-branch = "<"
      }
      for (int i = 0; i < p.n; ++i) {
        g[i] = (double)p.degrees[i] + 1.0;
        for (int d = 0; d < dim; ++d) pos[(size_t)i * dim + d] = (real)p.initial_positions[i + (size_t)d * p.n];
      }
      for (size_t e = 0; e < ne; ++e) {
        const int c = p.edge_thresh[e];
        ec[e] = (int8_t)(c == 0 ? 0 : (c == 1 ? 1 : (c == -1 ? -1 : 2)));
      }
      GsDeviceProblem<real>& d = dev[b];
      GS_TRY(hipMalloc((void**)&d.tm, nn * sizeof(real)));
      GS_TRY(hipMalloc((void**)&d.cm, nn));
      GS_TRY(hipMalloc((void**)&d.gplus, p.n * 8));
      GS_TRY(hipMalloc((void**)&d.ei, (ne ? ne : 1) * 4));
      GS_TRY(hipMalloc((void**)&d.ej, (ne ? ne : 1) * 4));
      GS_TRY(hipMalloc((void**)&d.et, (ne ? ne : 1) * 8));
      GS_TRY(hipMalloc((void**)&d.ec, (ne ? ne : 1)));
      GS_TRY(hipMalloc((void**)&d.pos, nd * sizeof(real)));
      GS_TRY(hipMalloc((void**)&d.best, nd * sizeof(real)));
      GS_TRY(hipMemcpy(d.tm, tm.data(), nn * sizeof(real), hipMemcpyHostToDevice));
      GS_TRY(hipMemcpy(d.cm, cm.data(), nn, hipMemcpyHostToDevice));
      GS_TRY(hipMemcpy(d.gplus, g.data(), p.n * 8, hipMemcpyHostToDevice));
      if (ne) {
        GS_TRY(hipMemcpy(d.ei, p.edge_i, ne * 4, hipMemcpyHostToDevice));
        GS_TRY(hipMemcpy(d.ej, p.edge_j, ne * 4, hipMemcpyHostToDevice));
        GS_TRY(hipMemcpy(d.et, p.edge_dist, ne * 8, hipMemcpyHostToDevice));
        GS_TRY(hipMemcpy(d.ec, ec.data(), ne, hipMemcpyHostToDevice));
      }
      GS_TRY(hipMemcpy(d.pos, pos.data(), nd * sizeof(real), hipMemcpyHostToDevice));
      if (sparse) {
        // CSR over rows (entries sorted by row, then column); the caller's list is column-major
        std::vector<long long> order(ne);
        std::iota(order.begin(), order.end(), 0ll);
        std::sort(order.begin(), order.end(), [&](long long x, long long y) {
          return p.edge_i[x] != p.edge_i[y] ? p.edge_i[x] < p.edge_i[y] : p.edge_j[x] < p.edge_j[y];
        });
        std::vector<int> roff(p.n + 1, 0);
        std::vector<unsigned short> col(ne), row(ne);
        std::vector<real> tgt(ne);
        std::vector<int8_t> code(ne);
        for (size_t q = 0; q < ne; ++q) {
          const long long e = order[q];
          row[q] = (unsigned short)p.edge_i[e]; col[q] = (unsigned short)p.edge_j[e];
          tgt[q] = (real)p.edge_dist[e]; code[q] = ec[e];
          roff[p.edge_i[e] + 1] += 1;
        }
        for (int i = 0; i < p.n; ++i) roff[i + 1] += roff[i];
        GS_TRY(hipMalloc((void**)&d.row_off, (p.n + 1) * 4));
        GS_TRY(hipMalloc((void**)&d.ecol, ne * 2));
        GS_TRY(hipMalloc((void**)&d.erow, ne * 2));
        GS_TRY(hipMemcpy(d.row_off, roff.data(), (p.n + 1) * 4, hipMemcpyHostToDevice));
        GS_TRY(hipMemcpy(d.ecol, col.data(), ne * 2, hipMemcpyHostToDevice));
        GS_TRY(hipMemcpy(d.erow, row.data(), ne * 2, hipMemcpyHostToDevice));
        GS_TRY(hipMalloc((void**)&d.etgt, ne * sizeof(real)));
        GS_TRY(hipMemcpy(d.etgt, tgt.data(), ne * sizeof(real), hipMemcpyHostToDevice));
        GS_TRY(hipMemcpy(d.ec, code.data(), ne, hipMemcpyHostToDevice));       // reuse: sorted codes
      }
      GsDev<real>& k = h[b];
      k.tm = d.tm; k.cm = d.cm; k.gplus = d.gplus; k.ei = d.ei; k.ej = d.ej; k.et = d.et; k.ec = d.ec;
      k.row_off = d.row_off; k.ecol = d.ecol; k.erow = d.erow; k.etgt = d.etgt; k.ecode = d.ec;
      k.pos = d.pos; k.best = d.best; k.out = d_out + b; k.n_edges = p.n_edges;
      k.k0 = p.k0; k.cooling = p.cooling; k.c_rep = p.c_rep; k.eps = p.eps; k.seed = p.seed;
      k.n = p.n; k.n_iter = p.n_iter; k.check_freq = p.check_freq; k.window = p.window;
    }
    GS_TRY(hipMalloc((void**)&d_problems, sizeof(GsDev<real>) * count));
    GS_TRY(hipMemcpy(d_problems, h.data(), sizeof(GsDev<real>) * count, hipMemcpyHostToDevice));
    int threads = (((n_max + 1) / 2) + 63) & ~63;
    threads = std::max(128, std::min(1024, threads));
    hipEvent_t e0, e1;
    GS_TRY(hipEventCreate(&e0));
    GS_TRY(hipEventCreate(&e1));
    GS_TRY(hipEventRecord(e0, 0));
    switch (dim) {
#define GS_CASE(D) case D: gs_launch<D, real>(d_problems, count, threads, lds_max, sparse, 0); break;
      GS_CASE(1) GS_CASE(2) GS_CASE(3) GS_CASE(4) GS_CASE(5) GS_CASE(6) GS_CASE(7) GS_CASE(8)
      GS_CASE(9) GS_CASE(10)
#undef GS_CASE
      default: throw GsHipError{TOPOLOW_ERR_UNSUPPORTED, "ndim must be between 1 and 10"};
    }
    GS_TRY(hipEventRecord(e1, 0));
    GS_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    GS_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (device_seconds) *device_seconds = ms * 1e-3;
    std::vector<GsOut> outs(count);
    GS_TRY(hipMemcpy(outs.data(), d_out, sizeof(GsOut) * count, hipMemcpyDeviceToHost));
    for (int b = 0; b < count; ++b) {
      const GsProblem& p = pbs[b];
      const GsOut& o = outs[b];
      if (o.nonfinite_iter != 0 && rc == TOPOLOW_OK) {
        if (errbuf && errlen)
          snprintf(errbuf, errlen, "Numerical instability at iteration %d. Reduce k0 or c_repulsion.",
                   o.nonfinite_iter);
        rc = TOPOLOW_ERR_NONFINITE;
      }
      const size_t nd = (size_t)p.n * dim;
      std::vector<real> pos(nd);
      GS_TRY(hipMemcpy(pos.data(), dev[b].pos, nd * sizeof(real), hipMemcpyDeviceToHost));
      for (int i = 0; i < p.n; ++i)
        for (int d = 0; d < dim; ++d) res[b].positions[i + (size_t)d * p.n] = (double)pos[(size_t)i * dim + d];
      res[b].converged = o.converged; res[b].iterations = o.iterations; res[b].iters_run = o.iters_run;
      res[b].n_checks = o.n_checks; res[b].final_mae = o.final_mae; res[b].final_k = o.final_k;
      res[b].nonfinite_iter = o.nonfinite_iter;
    }
  } catch (const GsHipError& e) {
    if (errbuf && errlen) snprintf(errbuf, errlen, "%s", e.msg.c_str());
    rc = e.code;
  }
  cleanup();
  return rc;
}

inline int gs_run_batch(const GsProblem* pbs, GsResult* res, int count, int precision,
                        double* device_seconds, char* errbuf, size_t errlen) {
  if (precision == TOPOLOW_PRECISION_F32)
    return gs_run_batch_t<float>(pbs, res, count, device_seconds, errbuf, errlen);
  return gs_run_batch_t<double>(pbs, res, count, device_seconds, errbuf, errlen);
}

}  // namespace topolow
