// topolow_amd/csrc/relax_tilegs.h -- exact Gauss-Seidel relaxation across workgroups ("tile GS").
//
// The reference's algorithm (src/optimization.cpp:193-289 of the reference: every unordered pair
// once per iteration, both endpoints move immediately) for problems too large for the
// one-workgroup kernel of relax_gs.h.  As there, only the visiting ORDER is chosen:
//   * points are cut into blocks of 64 consecutive indices;
//   * an iteration is a round-robin tournament over the blocks (circle method over a fresh random
//     permutation of the block ids): a round holds disjoint block pairs, one wavefront each, one
//     kernel launch per round (the launch boundary is the global barrier between rounds);
//   * inside a block pair (I,J) the 64 x 64 pairs are visited in 64 steps of 64 disjoint pairs:
//     at step s lane l relaxes (I[l], J[(l + s + s0) mod 64]) -- the lane keeps its I point in
//     registers, the J points live in LDS, the 64 x 64 target words are staged in LDS once;
//   * a last launch relaxes the pairs inside each block (circle method over its 64 points).
// Disjoint pairs commute exactly, so the device result equals a sequential replay of
// tilegs_pair_order() -- which is how the tests check it against the CPU oracle.  Targets are the
// session's encoded fp32 words (4-ulp rounded; the oracle replay is given the same rounded
// targets), positions and arithmetic are the session's precision, contraction off in f64.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "relax_common.h"
#include "relax_gs.h"

namespace topolow {

constexpr int kTile = 64;
constexpr uint64_t kTileBlockStream = 0x7100ull;
constexpr uint64_t kTileShiftStream = 0x7101ull;
constexpr uint64_t kTileIntraStream = 0x7102ull;

TL_HD inline int tile_shift(uint64_t seed, int iter, int bi, int bj) {
  return (int)(rnd64(seed, kTileShiftStream,
                     ((uint64_t)(uint32_t)iter << 40) ^ ((uint64_t)(uint32_t)bi << 20) ^ (uint32_t)bj) >> 58);
}
TL_HD inline int tile_intra_round0(uint64_t seed, int iter, int b) {
  return (int)rnd_below(rnd64(seed, kTileIntraStream, ((uint64_t)(uint32_t)iter << 32) | (uint32_t)b), 63u);
}

// Host: block pairs of every round of iteration `iter`.  rounds[r] = list of (I, J) with I < J.
inline std::vector<std::vector<std::pair<int, int>>> tilegs_rounds(int n_blocks, uint64_t seed, int iter) {
  std::vector<std::vector<std::pair<int, int>>> rounds;
  if (n_blocks < 2) return rounds;
  std::vector<uint32_t> keys(n_blocks);
  for (int b = 0; b < n_blocks; ++b)
    keys[b] = (uint32_t)(rnd64(seed, kTileBlockStream, ((uint64_t)(uint32_t)iter << 32) | (uint32_t)b) >> 32);
  std::vector<int> perm(n_blocks);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return keys[x] < keys[y]; });
  const int M = n_blocks + (n_blocks & 1), m1 = M - 1;
  for (int r = 0; r < m1; ++r) {
    std::vector<std::pair<int, int>> rd;
    for (int p = 0; p < M / 2; ++p) {
      int a, b;
      gs_round_pair(m1, r, p, &a, &b);
      if (a >= n_blocks || b >= n_blocks) continue;
      int bi = perm[a], bj = perm[b];
      if (bi > bj) std::swap(bi, bj);
      rd.emplace_back(bi, bj);
    }
    rounds.push_back(rd);
  }
  return rounds;
}

// Host: the exact visiting order of iteration `iter` (for the oracle replay).
inline int64_t tilegs_pair_order(int n, uint64_t seed, int iter, int32_t* pairs_out) {
  const int nb = (n + kTile - 1) / kTile;
  int64_t cnt = 0;
  auto emit = [&](int a, int b) {
    if (a >= n || b >= n || a == b) return;
    if (pairs_out) { pairs_out[2 * cnt] = a; pairs_out[2 * cnt + 1] = b; }
    ++cnt;
  };
  for (const auto& rd : tilegs_rounds(nb, seed, iter)) {
    for (const auto& pr : rd) {
      const int s0 = tile_shift(seed, iter, pr.first, pr.second);
      for (int s = 0; s < kTile; ++s)
        for (int l = 0; l < kTile; ++l) emit(pr.first * kTile + l, pr.second * kTile + ((l + s + s0) & 63));
    }
  }
  for (int b = 0; b < nb; ++b) {
    const int r0 = tile_intra_round0(seed, iter, b);
    for (int r = 0; r < 63; ++r) {
      int rr = r + r0; if (rr >= 63) rr -= 63;
      for (int p = 0; p < 32; ++p) {
        int x, y;
        gs_round_pair(63, rr, p, &x, &y);
        emit(b * kTile + std::min(x, y), b * kTile + std::max(x, y));
      }
    }
  }
  return cnt;
}

// Pair update on an encoded target word (same operations as gs_pair_update / the reference).
template <int DIM, typename real>
__device__ __forceinline__ void tile_pair(real* pa, real* pb, uint32_t w, double ga, double gb, double k,
                                          double c_rep) {
  const uint32_t c = w & kCodeMask;
  const real target = (w == kInfWord) ? (real)INFINITY : (real)bits_f32(w & ~kCodeMask);
  const int code = c == 0 ? 0 : (c == 1 ? 1 : -1);
  gs_pair_dispatch<DIM, real>(pa, pb, target, code, ga, gb, k, c_rep);
}

// This iteration's random permutation of the block ids (rank of a hashed key, stable) -- one
// small workgroup, launched ahead of the rounds on the same stream.
__global__ __launch_bounds__(256) void tilegs_perm_kernel(uint64_t seed, int iter, int n_blocks,
                                                          int* __restrict__ bperm, const RunState* st) {
  if (st != nullptr && st->stopped) return;
  for (int b = threadIdx.x; b < n_blocks; b += 256) {
    const uint32_t kb =
        (uint32_t)(rnd64(seed, kTileBlockStream, ((uint64_t)(uint32_t)iter << 32) | (uint32_t)b) >> 32);
    int rank = 0;
    for (int c = 0; c < n_blocks; ++c) {
      const uint32_t kc =
          (uint32_t)(rnd64(seed, kTileBlockStream, ((uint64_t)(uint32_t)iter << 32) | (uint32_t)c) >> 32);
      rank += (kc < kb) || (kc == kb && c < b);
    }
    bperm[rank] = b;
  }
}

// One round: block pair (I < J) per 64-thread workgroup (one wavefront).
template <int DIM, typename real>
__global__ __launch_bounds__(kTile) void tilegs_pair_kernel(
    const uint32_t* __restrict__ enc, int ld, int n, real* __restrict__ pos,
    const float* __restrict__ gplus, const int* __restrict__ bperm, int n_blocks, int round,
    const RunState* st, uint64_t seed, int iter, double k, double c_rep) {
  if (st != nullptr && st->stopped) return;
  __shared__ real pj[kTile][DIM];
  __shared__ float gj[kTile];
  __shared__ uint32_t tile[kTile][kTile + 1];
  const int lane = threadIdx.x;
  int bi, bj;
  {
    const int M = n_blocks + (n_blocks & 1), m1 = M - 1;
    int a, b;
    gs_round_pair(m1, round, (int)blockIdx.x, &a, &b);
    if (a >= n_blocks || b >= n_blocks) return;  // the bye of an odd number of blocks
    bi = bperm[a];
    bj = bperm[b];
    if (bi > bj) { const int t = bi; bi = bj; bj = t; }
  }
  const int i = bi * kTile + lane, j_own = bj * kTile + lane;
  const int s0 = tile_shift(seed, iter, bi, bj);
  real pi[DIM];
  const bool vi = i < n;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    pi[d] = vi ? pos[(size_t)i * DIM + d] : (real)0;
    pj[lane][d] = j_own < n ? pos[(size_t)j_own * DIM + d] : (real)0;
  }
  const double gi = vi ? (double)gplus[i] : 1.0;
  gj[lane] = j_own < n ? gplus[j_own] : 1.0f;
  // row i of the tile: 64 consecutive words (16-byte loads), into LDS row `lane`
  if (vi) {
    const uint4* src = reinterpret_cast<const uint4*>(enc + enc_index(i, bj * kTile, ld));
#pragma unroll
    for (int q = 0; q < kTile / 4; ++q) {
      const uint4 v = src[q];
      tile[lane][4 * q + 0] = v.x; tile[lane][4 * q + 1] = v.y;
      tile[lane][4 * q + 2] = v.z; tile[lane][4 * q + 3] = v.w;
    }
  }
  __syncthreads();
  for (int s = 0; s < kTile; ++s) {
    const int t = (lane + s + s0) & 63;
    const int j = bj * kTile + t;
    if (vi && j < n) {
      real pjt[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d) pjt[d] = pj[t][d];
      tile_pair<DIM, real>(pi, pjt, tile[lane][t], gi, (double)gj[t], k, c_rep);
#pragma unroll
      for (int d = 0; d < DIM; ++d) pj[t][d] = pjt[d];
    }
    __syncthreads();  // one wavefront: this only orders the LDS traffic of consecutive steps
  }
  if (vi) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) pos[(size_t)i * DIM + d] = pi[d];
  }
  if (j_own < n) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) pos[(size_t)j_own * DIM + d] = pj[lane][d];
  }
}

// Last launch of an iteration: the pairs inside each block.
template <int DIM, typename real>
__global__ __launch_bounds__(kTile) void tilegs_intra_kernel(
    const uint32_t* __restrict__ enc, int ld, int n, real* __restrict__ pos,
    const float* __restrict__ gplus, const RunState* st, uint64_t seed, int iter, double k, double c_rep) {
  if (st != nullptr && st->stopped) return;
  __shared__ real pb[kTile][DIM];
  __shared__ float gb[kTile];
  __shared__ uint32_t tile[kTile][kTile + 1];
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  const int i = b * kTile + lane;
  const bool vi = i < n;
#pragma unroll
  for (int d = 0; d < DIM; ++d) pb[lane][d] = vi ? pos[(size_t)i * DIM + d] : (real)0;
  gb[lane] = vi ? gplus[i] : 1.0f;
  if (vi) {
    const uint4* src = reinterpret_cast<const uint4*>(enc + enc_index(i, b * kTile, ld));
#pragma unroll
    for (int q = 0; q < kTile / 4; ++q) {
      const uint4 v = src[q];
      tile[lane][4 * q + 0] = v.x; tile[lane][4 * q + 1] = v.y;
      tile[lane][4 * q + 2] = v.z; tile[lane][4 * q + 3] = v.w;
    }
  }
  __syncthreads();
  const int r0 = tile_intra_round0(seed, iter, b);
  for (int r = 0; r < 63; ++r) {
    int rr = r + r0; if (rr >= 63) rr -= 63;
    if (lane < 32) {
      int x, y;
      gs_round_pair(63, rr, lane, &x, &y);
      const int lo = x < y ? x : y, hi = x < y ? y : x;
      if (b * kTile + hi < n) {
        real pa[DIM], pc[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) { pa[d] = pb[lo][d]; pc[d] = pb[hi][d]; }
        tile_pair<DIM, real>(pa, pc, tile[lo][hi], (double)gb[lo], (double)gb[hi], k, c_rep);
#pragma unroll
        for (int d = 0; d < DIM; ++d) { pb[lo][d] = pa[d]; pb[hi][d] = pc[d]; }
      }
    }
    __syncthreads();
  }
  if (vi) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) pos[(size_t)i * DIM + d] = pb[lane][d];
  }
}

// Non-finite guard for the in-place schedule (reference :359-361): flags the iteration in st.
template <typename real>
__global__ __launch_bounds__(256) void tilegs_finite_kernel(const real* __restrict__ pos, long long n_values,
                                                            RunState* st, int iter1) {
  if (st->stopped) return;
  bool bad = false;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n_values; q += (long long)gridDim.x * 256)
    bad |= !isfinite(pos[q]);
  if (__syncthreads_or(bad) && threadIdx.x == 0) atomicMin(&st->first_nonfinite, iter1);
}

}  // namespace topolow
