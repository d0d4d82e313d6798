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
#include <cstring>
#include <numeric>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
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
  int aborted;         // != 0: the host raised the abort word (user interrupt)
  double hold_sum;               // sum |truth - distance| over the holdout pairs, final positions
  unsigned long long hold_cnt;
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
  // held-out pairs scored on the final positions (cross-validation; may be empty)
  const int* hi; const int* hj; const double* ht;
  long long n_hold;
  real* pos;             // n x dim row-major: in = initial positions, out = best positions
  real* best;            // n x dim scratch
  GsOut* out;
  // host mailbox (pinned memory, device alias; nullable): ctrl[0] = abort word the host may raise at
  // any time (read once per iteration: the interrupt poll of reference :364 reaches this one-launch
  // kernel through it), ctrl[1] = iterations completed (written every 10th iteration);
  // trace: (iteration, MAE, k) of every convergence check, up to trace_cap checks (verbose lines of
  // reference :298-301)
  int* ctrl;
  double* trace;
  int trace_cap, pad0;
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
  int converged = 0, iters_run = 0, n_checks = 0, nonfinite_iter = 0, aborted = 0;
  const int check_freq = P.check_freq < 1 ? 10 : P.check_freq;  // reference :181
  __syncthreads();

  for (int iter = 0; iter < P.n_iter; ++iter) {
    if (P.ctrl != nullptr && tid == 0) {
      sh_flag[1] = __hip_atomic_load(P.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (iter % 10 == 0) __hip_atomic_store(P.ctrl + 1, iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- this iteration's random permutation: rank of a hashed key (stable) ----
    for (int i = tid; i < n; i += nthr) keys[i] = gs_key(P.seed, iter, i);
    __syncthreads();
    if (P.ctrl != nullptr && sh_flag[1] != 0) { aborted = 1; break; }   // block-uniform
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
      if (P.trace != nullptr && tid == 0 && n_checks <= P.trace_cap) {
        P.trace[3 * (n_checks - 1) + 0] = (double)(iter + 1);
        P.trace[3 * (n_checks - 1) + 1] = err;
        P.trace[3 * (n_checks - 1) + 2] = k;
      }
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
  // Out-of-sample error of the returned map (what the reference gets from as.matrix(dist()) +
  // error_calculator_comparison, R/core.R:474, R/error_metrics.R:55-144): f64 distances of the
  // held-out pairs against their true values.
  double hs = 0.0;
  unsigned long long hc = 0;
  for (long long e = tid; e < P.n_hold; e += nthr) {
    const int a = P.hi[e], b = P.hj[e];
    double q = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      const double diff = (double)P.best[(size_t)a * DIM + d] - (double)P.best[(size_t)b * DIM + d];
      q += diff * diff;
    }
    hs += fabs(P.ht[e] - sqrt(q));
    ++hc;
  }
  if (P.n_hold > 0) gs_block_sum(hs, hc, sh_s, sh_c);   // block-uniform condition
  if (tid == 0) {
    GsOut o;
    o.hold_sum = hs;
    o.hold_cnt = hc;
    o.final_mae = ctl.best_mae;
    o.final_k = ctl.best_k;
    o.converged = converged;
    o.iterations = ctl.best_iter;
    o.iters_run = iters_run;
    o.n_checks = n_checks;
    o.nonfinite_iter = nonfinite_iter;
    o.aborted = aborted;
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
  // D == T == nullptr: the edge list DEFINES the matrix (every unlisted pair is unmeasured)
  const int32_t* hold_i = nullptr; const int32_t* hold_j = nullptr; const double* hold_truth = nullptr;
  int64_t n_hold = 0;
};

struct GsResult {
  double* positions;  // n x dim col-major (caller-owned)
  int converged, iterations, iters_run, n_checks;
  int nonfinite_iter = 0;  // != 0: the non-finite guard fired at this iteration
  int aborted = 0;
  double final_mae, final_k;
  double hold_sum = 0.0;
  long long hold_count = 0;
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

// An edge list that may stand for the matrix: 0 <= i < j < n, finite targets, no pair twice.
inline bool gs_edges_well_formed(const GsProblem& p) {
  const int n = p.n;
  std::vector<long long> key((size_t)p.n_edges);
  for (long long e = 0; e < p.n_edges; ++e) {
    const int a = p.edge_i[e], b = p.edge_j[e];
    if (a < 0 || b <= a || b >= n || !std::isfinite(p.edge_dist[e])) return false;
    key[(size_t)e] = (long long)a * n + b;
  }
  std::sort(key.begin(), key.end());
  return std::adjacent_find(key.begin(), key.end()) == key.end();
}

// Host-side loops over the problems of a batch are independent: spread them over a few threads (a
// sweep stages thousands of small problems; sorting and packing them serially took as long as the
// kernel).  `fn(b)` may throw GsHipError; the first one is rethrown on the caller's thread.
template <typename Fn>
void gs_parallel_for(int count, Fn fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  const int workers = std::max(1, std::min({count / 16, (int)(hw ? hw : 1), 16}));
  if (workers <= 1) {
    for (int b = 0; b < count; ++b) fn(b);
    return;
  }
  std::atomic<int> next{0};
  std::atomic<bool> failed{false};
  GsHipError first{TOPOLOW_OK, ""};
  std::vector<std::thread> pool;
  for (int w = 0; w < workers; ++w) {
    pool.emplace_back([&] {
      for (int b = next.fetch_add(1); b < count && !failed.load(); b = next.fetch_add(1)) {
        try {
          fn(b);
        } catch (const GsHipError& e) {
          if (!failed.exchange(true)) first = e;
        }
      }
    });
  }
  for (auto& t : pool) t.join();
  if (failed.load()) throw first;
}

// Host/device staging of a whole batch: every array of every problem lives in ONE host buffer that
// goes to the device with one copy (thousands of small embeddings per launch otherwise spend their
// time in hipMalloc / hipMemcpy calls); positions sit together at the front so they come back with
// one copy as well.
class GsArena {
 public:
  size_t reserve(size_t bytes) {
    const size_t at = size_;
    size_ += (bytes + 255) & ~(size_t)255;
    return at;
  }
  void commit() { host_.assign(size_, 0); }
  template <typename T> T* host(size_t off) { return reinterpret_cast<T*>(host_.data() + off); }
  template <typename T> T* dev(size_t off) const { return reinterpret_cast<T*>(dev_ + off); }
  void upload() {
    GS_TRY(hipMalloc((void**)&dev_, size_ ? size_ : 256));
    GS_TRY(hipMemcpy(dev_, host_.data(), size_, hipMemcpyHostToDevice));
  }
  void download_front(size_t bytes) { GS_TRY(hipMemcpy(host_.data(), dev_, bytes, hipMemcpyDeviceToHost)); }
  ~GsArena() { (void)hipFree(dev_); }
 private:
  size_t size_ = 0;
  std::vector<unsigned char> host_;
  unsigned char* dev_ = nullptr;
};

// One grid of the exact-GS kernel (all problems share ndim and precision), in three steps so that
// several grids can be staged, run side by side on their own streams, and collected afterwards.
struct GsBatchBase {
  virtual ~GsBatchBase() {}
  // host mailbox shared by every problem of the grid (see GsDev::ctrl / trace); call before stage()
  virtual void set_mailbox(int* ctrl_dev, double* trace_dev, int trace_cap) = 0;
  virtual void stage(const GsProblem* pbs, int count) = 0;   // validate, pack, upload
  virtual void launch(hipStream_t st) = 0;
  virtual int collect(GsResult* res, char* errbuf, size_t errlen) = 0;   // after the stream is idle
};

template <typename real>
class GsBatch : public GsBatchBase {
  struct Off {
    size_t pos, gplus, tm = 0, cm = 0, ei = 0, ej = 0, et = 0, ec, roff = 0, ecol = 0, erow = 0, etgt = 0,
           hi = 0, hj = 0, ht = 0, best;
  };
  const GsProblem* pbs = nullptr;
  int count = 0, dim = 0, udim = 0, n_max = 0;   // dim: coordinates the kernel carries (11 -> 12, 13..15 -> 16, zero-padded)
  size_t lds_max = 0, o_out = 0, o_prob = 0;
  bool sparse = false;
  std::vector<Off> off;
  GsArena A;
  unsigned char* d_scratch = nullptr;   // best snapshots: device only
  int* ctrl_dev = nullptr;
  double* trace_dev = nullptr;
  int trace_cap = 0;

 public:
  ~GsBatch() override { (void)hipFree(d_scratch); }
  void set_mailbox(int* c, double* t, int cap) override { ctrl_dev = c; trace_dev = t; trace_cap = cap; }

  void stage(const GsProblem* pbs_, int count_) override {
    pbs = pbs_;
    count = count_;
    udim = pbs[0].dim;
    dim = udim <= 10 ? udim : (udim <= 12 ? 12 : 16);
    lds_max = 0;
    n_max = 0;
    // one kernel instance per launch: the LDS-resident table is used when EVERY problem of the
    // batch qualifies (edge list == matrix, or edge list given as the matrix; n <= 2048; table
    // within the LDS budget)
    sparse = getenv("TOPOLOW_GS_DENSE") == nullptr;
    {
      std::vector<char> ok_sparse(count, 1);
      const bool want_sparse = sparse;
      gs_parallel_for(count, [&](int b) {
        const GsProblem& p = pbs[b];
        if (p.dim != udim) throw GsHipError{TOPOLOW_ERR_BAD_ARGUMENT, "batch must share ndim"};
        if (p.dim < 1 || p.dim > 16) throw GsHipError{TOPOLOW_ERR_UNSUPPORTED, "ndim must be between 1 and 16"};
        if (p.n < 2) throw GsHipError{TOPOLOW_ERR_TOO_FEW_POINTS, "Need at least 2 points for embedding"};
        if ((p.D == nullptr) != (p.T == nullptr))
          throw GsHipError{TOPOLOW_ERR_BAD_ARGUMENT, "dissimilarity and threshold matrices come together"};
        if (p.D == nullptr && !gs_edges_well_formed(p))
          throw GsHipError{TOPOLOW_ERR_BAD_ARGUMENT,
                           "edge list that stands for the matrix needs 0 <= i < j < n, finite targets, no pair twice"};
        for (long long e = 0; e < p.n_hold; ++e)
          if (p.hold_i[e] < 0 || p.hold_i[e] >= p.n || p.hold_j[e] < 0 || p.hold_j[e] >= p.n)
            throw GsHipError{TOPOLOW_ERR_BAD_ARGUMENT, "holdout pair out of range"};
        if (want_sparse)
          ok_sparse[b] = p.n <= 2048 && p.n_edges > 0 && p.n_edges < 65535 &&
                         gs_lds_bytes(p.n, dim, sizeof(real), p.n_edges) <= kGsSparseLdsBudget &&
                         (p.D == nullptr || gs_edges_match_matrix(p));
      });
      for (int b = 0; b < count; ++b) sparse = sparse && ok_sparse[b];
    }
    // ---- layout ----
    off.assign(count, Off{});
    size_t best_total = 0;
    for (int b = 0; b < count; ++b) off[b].pos = A.reserve((size_t)pbs[b].n * dim * sizeof(real));
    o_out = A.reserve(sizeof(GsOut) * count);
    o_prob = A.reserve(sizeof(GsDev<real>) * count);
    for (int b = 0; b < count; ++b) {
      const GsProblem& p = pbs[b];
      const size_t lds = gs_lds_bytes(p.n, dim, sizeof(real), sparse ? p.n_edges : 0);
      if (lds > 160 * 1024)
        throw GsHipError{TOPOLOW_ERR_UNSUPPORTED,
                         "problem too large for the single-workgroup GS kernel (LDS); use the slab schedule"};
      lds_max = std::max(lds_max, lds);
      n_max = std::max(n_max, p.n);
      const size_t nn = (size_t)p.n * p.n, ne = (size_t)p.n_edges, ne1 = ne ? ne : 1;
      Off& o = off[b];
      o.gplus = A.reserve(p.n * 8);
      o.ec = A.reserve(ne1);
      if (sparse) {
        o.roff = A.reserve((p.n + 1) * 4);
        o.ecol = A.reserve(ne1 * 2);
        o.erow = A.reserve(ne1 * 2);
        o.etgt = A.reserve(ne1 * sizeof(real));
      } else {
        o.tm = A.reserve(nn * sizeof(real));
        o.cm = A.reserve(nn);
        o.ei = A.reserve(ne1 * 4);
        o.ej = A.reserve(ne1 * 4);
        o.et = A.reserve(ne1 * 8);
      }
      if (p.n_hold > 0) {
        o.hi = A.reserve(p.n_hold * 4);
        o.hj = A.reserve(p.n_hold * 4);
        o.ht = A.reserve(p.n_hold * 8);
      }
      o.best = best_total;
      best_total += ((size_t)p.n * dim * sizeof(real) + 255) & ~(size_t)255;
    }
    A.commit();
    // ---- fill ----
    gs_parallel_for(count, [&](int b) {
      const GsProblem& p = pbs[b];
      const Off& o = off[b];
      const size_t nn = (size_t)p.n * p.n, ne = (size_t)p.n_edges;
      real* pos = A.host<real>(o.pos);
      double* g = A.host<double>(o.gplus);
      for (int i = 0; i < p.n; ++i) {
        g[i] = (double)p.degrees[i] + 1.0;   // reference :137-140
        for (int d = 0; d < udim; ++d) pos[(size_t)i * dim + d] = (real)p.initial_positions[i + (size_t)d * p.n];
      }
      auto code_of = [](int c) { return (int8_t)(c == 0 ? 0 : (c == 1 ? 1 : -1)); };  // else-branch = "<"
      // edge codes keep "neither 0, 1 nor -1" apart: such a pair moves like "<" (:236-242) but never
      // counts in the error (:68-76 compares with -1 exactly)
      auto edge_code = [](int c) { return (int8_t)(c == 0 ? 0 : (c == 1 ? 1 : (c == -1 ? -1 : 2))); };
      int8_t* ec = A.host<int8_t>(o.ec);
      if (sparse) {
        // CSR over rows (entries sorted by row, then column); the caller's list is column-major
        std::vector<long long> order(ne);
        std::iota(order.begin(), order.end(), 0ll);
        std::sort(order.begin(), order.end(), [&](long long x, long long y) {
          return p.edge_i[x] != p.edge_i[y] ? p.edge_i[x] < p.edge_i[y] : p.edge_j[x] < p.edge_j[y];
        });
        int* roff = A.host<int>(o.roff);
        unsigned short* col = A.host<unsigned short>(o.ecol);
        unsigned short* row = A.host<unsigned short>(o.erow);
        real* tgt = A.host<real>(o.etgt);
        for (size_t q = 0; q < ne; ++q) {
          const long long e = order[q];
          row[q] = (unsigned short)p.edge_i[e];
          col[q] = (unsigned short)p.edge_j[e];
          tgt[q] = (real)p.edge_dist[e];
          ec[q] = edge_code(p.edge_thresh[e]);
          roff[p.edge_i[e] + 1] += 1;
        }
        for (int i = 0; i < p.n; ++i) roff[i + 1] += roff[i];
      } else {
        real* tm = A.host<real>(o.tm);
        int8_t* cm = A.host<int8_t>(o.cm);
        if (p.D != nullptr) {
          for (size_t q = 0; q < nn; ++q) { tm[q] = (real)p.D[q]; cm[q] = code_of(p.T[q]); }
        } else {   // the edge list is the matrix: unlisted pairs are unmeasured, the diagonal is 0
          for (size_t q = 0; q < nn; ++q) tm[q] = (real)INFINITY;
          for (int i = 0; i < p.n; ++i) tm[(size_t)i * p.n + i] = 0;
          for (size_t e = 0; e < ne; ++e) {
            const size_t up = (size_t)p.edge_i[e] + (size_t)p.edge_j[e] * p.n;
            const size_t lo = (size_t)p.edge_j[e] + (size_t)p.edge_i[e] * p.n;
            tm[up] = tm[lo] = (real)p.edge_dist[e];
            cm[up] = cm[lo] = code_of(p.edge_thresh[e]);
          }
        }
        std::memcpy(A.host<int>(o.ei), p.edge_i, ne * 4);
        std::memcpy(A.host<int>(o.ej), p.edge_j, ne * 4);
        std::memcpy(A.host<double>(o.et), p.edge_dist, ne * 8);
        for (size_t e = 0; e < ne; ++e) ec[e] = edge_code(p.edge_thresh[e]);
      }
      if (p.n_hold > 0) {
        std::memcpy(A.host<int>(o.hi), p.hold_i, (size_t)p.n_hold * 4);
        std::memcpy(A.host<int>(o.hj), p.hold_j, (size_t)p.n_hold * 4);
        std::memcpy(A.host<double>(o.ht), p.hold_truth, (size_t)p.n_hold * 8);
      }
    });
    GS_TRY(hipMalloc((void**)&d_scratch, best_total ? best_total : 256));
    A.upload();   // device addresses exist from here on; the problem table follows with its own copy
    {
      std::vector<GsDev<real>> h(count);
      for (int b = 0; b < count; ++b) {
        const GsProblem& p = pbs[b];
        const Off& o = off[b];
        GsDev<real>& k = h[b];
        std::memset(&k, 0, sizeof k);
        k.gplus = A.dev<double>(o.gplus);
        k.ec = k.ecode = A.dev<int8_t>(o.ec);
        if (sparse) {
          k.row_off = A.dev<int>(o.roff); k.ecol = A.dev<unsigned short>(o.ecol);
          k.erow = A.dev<unsigned short>(o.erow); k.etgt = A.dev<real>(o.etgt);
        } else {
          k.tm = A.dev<real>(o.tm); k.cm = A.dev<int8_t>(o.cm);
          k.ei = A.dev<int>(o.ei); k.ej = A.dev<int>(o.ej); k.et = A.dev<double>(o.et);
        }
        if (p.n_hold > 0) { k.hi = A.dev<int>(o.hi); k.hj = A.dev<int>(o.hj); k.ht = A.dev<double>(o.ht); }
        k.n_hold = p.n_hold;
        k.pos = A.dev<real>(o.pos);
        k.best = reinterpret_cast<real*>(d_scratch + o.best);
        k.out = A.dev<GsOut>(o_out) + b;
        k.n_edges = p.n_edges;
        k.k0 = p.k0; k.cooling = p.cooling; k.c_rep = p.c_rep; k.eps = p.eps; k.seed = p.seed;
        k.n = p.n; k.n_iter = p.n_iter; k.check_freq = p.check_freq; k.window = p.window;
        k.ctrl = ctrl_dev;
        k.trace = b == 0 ? trace_dev : nullptr;   // the trace follows the grid's first problem
        k.trace_cap = trace_cap;
      }
      GS_TRY(hipMemcpy(A.dev<GsDev<real>>(o_prob), h.data(), sizeof(GsDev<real>) * count, hipMemcpyHostToDevice));
    }
  }

  void launch(hipStream_t st) override {
    const GsDev<real>* d_problems = A.dev<GsDev<real>>(o_prob);
    int threads = (((n_max + 1) / 2) + 63) & ~63;
    threads = std::max(128, std::min(1024, threads));
    switch (dim) {
#define GS_CASE(D) case D: gs_launch<D, real>(d_problems, count, threads, lds_max, sparse, st); break;
      GS_CASE(1) GS_CASE(2) GS_CASE(3) GS_CASE(4) GS_CASE(5) GS_CASE(6) GS_CASE(7) GS_CASE(8)
      GS_CASE(9) GS_CASE(10) GS_CASE(12) GS_CASE(16)
#undef GS_CASE
      default: throw GsHipError{TOPOLOW_ERR_UNSUPPORTED, "ndim must be between 1 and 16"};
    }
  }

  int collect(GsResult* res, char* errbuf, size_t errlen) override {
    int rc = TOPOLOW_OK;
    A.download_front(o_out + sizeof(GsOut) * count);   // positions and results
    const GsOut* outs = A.host<GsOut>(o_out);
    for (int b = 0; b < count; ++b) {
      const GsProblem& p = pbs[b];
      const GsOut& o = outs[b];
      if (o.nonfinite_iter != 0 && rc == TOPOLOW_OK) {
        if (errbuf && errlen)
          snprintf(errbuf, errlen, "Numerical instability at iteration %d. Reduce k0 or c_repulsion.",
                   o.nonfinite_iter);
        rc = TOPOLOW_ERR_NONFINITE;
      }
      const real* pos = A.host<real>(off[b].pos);
      for (int i = 0; i < p.n; ++i)
        for (int d = 0; d < udim; ++d) res[b].positions[i + (size_t)d * p.n] = (double)pos[(size_t)i * dim + d];
      res[b].converged = o.converged; res[b].iterations = o.iterations; res[b].iters_run = o.iters_run;
      res[b].n_checks = o.n_checks; res[b].final_mae = o.final_mae; res[b].final_k = o.final_k;
      res[b].nonfinite_iter = o.nonfinite_iter;
      res[b].aborted = o.aborted;
      res[b].hold_sum = o.hold_sum; res[b].hold_count = (long long)o.hold_cnt;
    }
    return rc;
  }
};

inline GsBatchBase* gs_new_batch(int precision) {
  if (precision == TOPOLOW_PRECISION_F32) return new GsBatch<float>();
  return new GsBatch<double>();
}

// Host mailbox of a one-launch GS grid: abort word + progress counter + check trace, in pinned memory.
struct GsMailbox {
  int* ctrl = nullptr;       // [0] abort, [1] iterations completed
  double* trace = nullptr;   // 3 doubles per check
  int* ctrl_dev = nullptr;
  double* trace_dev = nullptr;
  int trace_cap = 0;
  void alloc(int cap) {
    trace_cap = cap;
    GS_TRY(hipHostMalloc((void**)&ctrl, 64, hipHostMallocMapped));
    std::memset(ctrl, 0, 64);
    GS_TRY(hipHostGetDevicePointer((void**)&ctrl_dev, ctrl, 0));
    if (cap > 0) {
      GS_TRY(hipHostMalloc((void**)&trace, sizeof(double) * 3 * (size_t)cap, hipHostMallocMapped));
      std::memset(trace, 0, sizeof(double) * 3 * (size_t)cap);
      GS_TRY(hipHostGetDevicePointer((void**)&trace_dev, trace, 0));
    }
  }
  ~GsMailbox() {
    if (ctrl) (void)hipHostFree(ctrl);
    if (trace) (void)hipHostFree(trace);
  }
};

// One grid, start to finish (the single-embedding path of topolow_optimize_layout_exact).
//   interrupt_cb: polled while the launch runs, once per 50 iterations of progress (reference :364)
//   and at least every 50 ms; a non-zero return raises the kernel's abort word.
//   trace_out (3 doubles per check: iteration, MAE, k) / n_trace: the first problem's checks.
inline int gs_run_batch(const GsProblem* pbs, GsResult* res, int count, int precision,
                        double* device_seconds, char* errbuf, size_t errlen,
                        int32_t (*interrupt_cb)(void*) = nullptr, void* interrupt_user = nullptr,
                        std::vector<double>* trace_out = nullptr) {
  int rc = TOPOLOW_OK;
  GsBatchBase* batch = gs_new_batch(precision);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  GsMailbox mb;   // outlives the launch on every path (the kernel reads and writes it)
  try {
    if (interrupt_cb != nullptr || trace_out != nullptr) {
      const int freq = pbs[0].check_freq < 1 ? 10 : pbs[0].check_freq;
      mb.alloc(trace_out != nullptr ? pbs[0].n_iter / freq + 2 : 0);
      batch->set_mailbox(mb.ctrl_dev, mb.trace_dev, mb.trace_cap);
    }
    batch->stage(pbs, count);
    GS_TRY(hipEventCreate(&e0));
    GS_TRY(hipEventCreate(&e1));
    GS_TRY(hipEventRecord(e0, 0));
    batch->launch(0);
    GS_TRY(hipEventRecord(e1, 0));
    if (interrupt_cb != nullptr) {
      int polled_at = 0;
      auto last_poll = std::chrono::steady_clock::now();
      while (hipEventQuery(e1) == hipErrorNotReady) {
        std::this_thread::sleep_for(std::chrono::microseconds(200));
        const int progress = __atomic_load_n(mb.ctrl + 1, __ATOMIC_RELAXED);
        const auto now = std::chrono::steady_clock::now();
        if (progress - polled_at >= 50 || now - last_poll >= std::chrono::milliseconds(50)) {
          polled_at = progress;
          last_poll = now;
          if (interrupt_cb(interrupt_user)) __atomic_store_n(mb.ctrl, 1, __ATOMIC_RELAXED);
        }
      }
    }
    GS_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    GS_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (device_seconds) *device_seconds = ms * 1e-3;
    rc = batch->collect(res, errbuf, errlen);
    if (trace_out != nullptr) {
      const int nc = std::min(res[0].n_checks, mb.trace_cap);
      trace_out->assign(mb.trace, mb.trace + 3 * (size_t)nc);
    }
    for (int b = 0; b < count && rc == TOPOLOW_OK; ++b)
      if (res[b].aborted) {
        if (errbuf && errlen) snprintf(errbuf, errlen, "interrupted by the caller");
        rc = TOPOLOW_ERR_INTERRUPTED;
      }
  } catch (const GsHipError& e) {
    if (errbuf && errlen) snprintf(errbuf, errlen, "%s", e.msg.c_str());
    rc = e.code;
    (void)hipStreamSynchronize(0);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) { (void)hipEventSynchronize(e1); (void)hipEventDestroy(e1); }
  delete batch;
  return rc;
}

}  // namespace topolow
