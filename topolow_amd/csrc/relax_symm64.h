// topolow_amd/csrc/relax_symm64.h -- the symmetric sweep in f64 (the reference's arithmetic type)
//
// The same sweep as relax_symm.h -- one-stage iterations, every unordered pair met once from the tile-major copy of the
// upper triangle, both ends moved, the same plan (units / runs), the same tiles and word order, the same partial
// buffers and the same fixed-order sum in the apply kernel -- with positions, records, sums and the pair update in
// f64 (reference src/optimization.cpp:203-281 in double; sqrt and reciprocal to 1 ulp, see sym64_pair).  Against the
// row-owner f64 stage kernel it halves the pair evaluations; an f64 pair costs ~60 full-rate f64 instructions, so the
// sweep is VALU-bound and what matters is the count of pairs, not bytes.
//
// Differences from the fp32 kernel, all consequences of the type: a lane keeps its eight rows (coordinates, two
// constants: 14 doubles per row at ndim 5) in LDS and only their sums in registers -- 80 of the 254 the ndim-5
// instance uses, two waves per SIMD (ndim 6: one); nothing is packed; the column sums of a half tile are reduced over the 8 lanes of a
// column group with three lane exchanges per value (the fp32 kernel's one-instruction DPP adds have no f64 form).
//
// The fused convergence check (ERR instance) is EXACT.  The tiles hold the targets as 4-byte words (fp32 rounded to
// 4 ulp, 3e-7 relative): reduced from them the MAE would sit 3e-8 off the reference's edge MAE (measured), where the
// separate pass of f64 sessions over the f64 edge list is exact.  So the ERR instance also reads, tile-major like the
// words, what the rounding took away: delta = (exact f64 target) - (decoded word), stored as fp32 (symm64_delta_kernel,
// from the session's f64 edge list; 6e-8 of 3e-7 of the target: 2e-14 relative), and sums |t_word + delta - r| = the
// exact |t - r|.  Forces still come from the words, as in every other f64 kernel of the library.
#pragma once

#include "relax_kernels.h"
#include "relax_symm.h"

namespace topolow {

// One point as the sweep reads it: DIM coordinates, then ks = 2k / (4 g + k) and cg = (c_rep / 2) / g, padded to 16 bytes.
template <int DIM> struct SymRec64 { static constexpr int W = (DIM + 2 + 1) & ~1; };

__device__ __forceinline__ double sym64_xor(double v, int mask) { return __shfl_xor(v, mask, 64); }

// one row x one column: both halves of the pair.  base = (t - r) / (r + 0.01) for a spring, 1 / (r + 0.01)^3 otherwise;
// every endpoint multiplies it with its own constant of that kind.
template <int DIM, bool THR, bool ERR>
__device__ __forceinline__ void sym64_pair(const double (&pc)[DIM], double ksc, double cgc, const double (&pi)[DIM],
                                           double ksr, double cgr, uint32_t w, double (&racc)[DIM], double (&cacc)[DIM],
                                           uint32_t dl_bits, double& err, unsigned& cnt) {
  double dx[DIM];
  double s = 0.0;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    dx[d] = pc[d] - pi[d];
    s = fma(dx[d], dx[d], s);
  }
  const double r = Math<double>::sqrt(s);             // 1 ulp (relax_kernels.h): estimate + two Newton steps
  const double inv = Math<double>::rcp(r + 0.01);
  const double t = (double)bits_f32(THR ? (w & ~kCodeMask) : w);
  bool spring;
  if constexpr (THR) {
    // 0: exact target; 1: ">" -- a spring while r < t; 2: "<" -- while r > t: the sign of t - r, turned round for code 2,
    // and two comparisons per pair (one per code and relation keeps six lane masks per pair alive)
    const uint32_t code = w & kCodeMask;
    const double e = t - r;
    const double es = __hiloint2double(__double2hiint(e) ^ (int)((w << 30) & 0x80000000u), __double2loint(e));
    spring = (code == 0u) | (es > 0.0);
  } else {
    spring = __builtin_amdgcn_classf(bits_f32(w), 0x1f8);   // measured = finite
  }
  const double base = spring ? (t - r) * inv : inv * inv * inv;
  const double coef = base * (spring ? ksr : cgr);
  const double cc = base * (spring ? ksc : cgc);
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    racc[d] = fma(dx[d], coef, racc[d]);
    cacc[d] = fma(dx[d], cc, cacc[d]);
  }
  if constexpr (ERR) {   // the convergence MAE of the positions this sweep reads, against the EXACT target t + delta
    err += spring ? fabs((t - r) + (double)bits_f32(dl_bits)) : 0.0;
    if constexpr (THR) cnt += spring ? 1u : 0u;
  }
}

// delta tiles: for every edge of the session's f64 edge list (session labels; codes 0, 1, -1) the difference between the
// exact target and what its 4-byte word decodes to, at the cell(s) of the tile-major copy the sweep meets the pair at
// (a pair inside a diagonal square is met from both sides).  tdelta: zero-filled by the caller.
__global__ __launch_bounds__(256) void symm64_delta_kernel(const int* __restrict__ ei, const int* __restrict__ ej,
                                                          const double* __restrict__ et, const int8_t* __restrict__ ec,
                                                          long long n_edges, float* __restrict__ tdelta, int TC, int n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += (long long)gridDim.x * blockDim.x) {
    const int a = ei[e], b = ej[e], c = ec[e];
    if (a < 0 || b < 0 || a >= n || b >= n || a == b || c == 2) continue;
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    const double t = et[e];
    const uint32_t w = encode_target(t, c);
    if (w == kInfWord) continue;
    const float dl = (float)(t - (double)bits_f32(w & ~kCodeMask));
    const int R = lo / kSymRows;
    tdelta[(size_t)sym_tile_index(R, hi / kSymCols, TC) * kSymTileWords + sym_word_in_tile(lo % kSymRows, hi % kSymCols)] = dl;
    if (hi / kSymRows == R)
      tdelta[(size_t)sym_tile_index(R, lo / kSymCols, TC) * kSymTileWords + sym_word_in_tile(hi % kSymRows, lo % kSymCols)] = dl;
  }
}

// enc, units, runs, col_row0: as symm_sweep_kernel.  rec: npad records of SymRec64<DIM>::W doubles.
// rowpart [n_units][64][DIM], colpart [n_tile_rows][npad][DIM] in f64.  ERR: tdelta (tile-major like enc, see above);
// part_sum / part_cnt / fixed_cnt as symm_sweep_kernel leaves them (TWICE the sum and the count over the unit's
// contributing pairs; a pair of the diagonal square is met from both sides and counts once per visit).
template <int DIM, bool ANYTHR, bool ERR>
__global__ __launch_bounds__(64 * kSymWaves, ((DIM <= 3 || (DIM == 4 && !(ANYTHR && ERR)) || (DIM == 5 && !ERR)) ? 2 : 1)) void symm64_sweep_kernel(
    const uint32_t* __restrict__ enc, const double* __restrict__ rec, const SymUnit* __restrict__ units,
    const SymRun* __restrict__ runs, double* __restrict__ rowpart, double* __restrict__ colpart, int npad,
    const RunState* st, int col_row0, const float* __restrict__ tdelta, double* __restrict__ part_sum,
    unsigned long long* __restrict__ part_cnt, unsigned long long fixed_cnt) {
  if (st != nullptr && st->stopped) return;
  constexpr int W = SymRec64<DIM>::W;
  constexpr int kRecVec = W / 2;                   // 16-byte pieces per record
  constexpr int kTileVec = kSymCols * kRecVec;     // ... per column block (<= 128)
  static_assert(kTileVec <= 128, "two pieces per lane");
  // (two 16-byte pieces per lane: 128 slots per block whatever kTileVec is, so that the hand-over at the end of a tile
  //  is unconditional -- a divergent store there splits the tile into basic blocks, see the column stores below)
  __shared__ uint4 lds[kSymWaves][2][128];
  __shared__ uint4 rows_lds[kSymWaves][kSymRows * kRecVec + 8];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int a = lane & 7, b = lane >> 3;
  const int gw = blockIdx.x * kSymWaves + wave;
  const SymRun run = runs[gw];
  const int u_begin = __builtin_amdgcn_readfirstlane(run.u0);
  const int u_end = __builtin_amdgcn_readfirstlane(run.u1);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rec_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(rec), 0, npad * W * 8, 0x00020000);
  for (int u = u_begin; u < u_end; ++u) {
    const SymUnit U = u == u_begin ? run.first : units[u];
    const int R = __builtin_amdgcn_readfirstlane(U.tile_row);
    const int J0 = __builtin_amdgcn_readfirstlane(U.j0), J1 = __builtin_amdgcn_readfirstlane(U.j1);
    const int tile0 = __builtin_amdgcn_readfirstlane(U.tile0);

    // the tile-row's 64 row records go to LDS (a lane re-reads the two rows of a row pair whenever it meets them: kept
    // in registers, eight rows' coordinates and constants cost 112 of them and the kernel spilled); only the row sums
    // stay in registers.  Record r sits one 16-byte piece further for every 8 rows, so the 8 lane groups a read 8 banks
    double racc[8][DIM];
    double err_tile = 0.0, err_unit = 0.0;
    unsigned cnt_tile = 0, cnt_unit2 = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int d = 0; d < DIM; ++d) racc[q][d] = 0.0;
    {
      const uint4* rr = reinterpret_cast<const uint4*>(rec + (size_t)R * kSymRows * W);
#pragma unroll
      for (int q = lane; q < kSymRows * kRecVec; q += 64) rows_lds[wave][q + (q / (8 * kRecVec))] = rr[q];
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(enc) + (size_t)tile0 * kSymTileWords, 0, (J1 - J0) * kSymTileWords * 4, 0x00020000);
    const int swap = a & 1;                // this lane's q-th column of a half is column 2h + (q ^ swap)
    auto request = [&](int J, int h, u32x4 (&dst)[4]) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        dst[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16 + ((J - J0) * 8 + 4 * h + p) * 1024, 0, 0);
    };
    // the ERR instance's delta words: the same addresses in the tile-major delta array
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(ERR ? tdelta + (size_t)tile0 * kSymTileWords : nullptr), 0, ERR ? (J1 - J0) * kSymTileWords * 4 : 0, 0x00020000);
    auto request_delta = [&](int J, int h, u32x4 (&dst)[4]) {
      if constexpr (ERR) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          dst[p] = __builtin_amdgcn_raw_buffer_load_b128(drsrc, lane * 16 + ((J - J0) * 8 + 4 * h + p) * 1024, 0, 0);
      }
    };
    u32x4 wa[4], wb[4], da[4], db[4];
    request(J0, 0, wa);
    request_delta(J0, 0, da);
    const uint4* recv = reinterpret_cast<const uint4*>(rec);
    if (lane < kTileVec) lds[wave][J0 & 1][lane] = recv[(size_t)J0 * kTileVec + lane];
    if constexpr (kTileVec > 64) if (lane + 64 < kTileVec) lds[wave][J0 & 1][lane + 64] = recv[(size_t)J0 * kTileVec + lane + 64];
    const __amdgpu_buffer_rsrc_t col_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        colpart + (size_t)(R - col_row0) * npad * DIM, 0, npad * DIM * 8, 0x00020000);
    const int col_off = a < 2 ? (4 * b + a) * DIM * 8 : 0x40000000;
#pragma unroll 1
    for (int J = J0; J < J1; ++J) {
      const int Jn = J + 1 < J1 ? J + 1 : J;
      request(J, 1, wb);
      request_delta(J, 1, db);
      u32x4 rn0 = {0, 0, 0, 0}, rn1 = {0, 0, 0, 0};
      rn0 = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, (Jn * kTileVec + lane) * 16, 0, 0);
      if constexpr (kTileVec > 64) rn1 = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, (Jn * kTileVec + lane + 64) * 16, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      auto half = [&](auto hc, const u32x4 (&wc)[4], const u32x4 (&dc)[4]) {   // columns 2h, 2h + 1 of the lane's four x its eight rows
        constexpr int h = decltype(hc)::value;
        double cacc[2][DIM], pc[2][DIM], ksc[2], cgc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          // the lane's c-th column of the half (its order: see sym_word_in_tile) from the wave's LDS copy of the block
          const int col = 4 * b + 2 * h + (c ^ swap);
          const double* f = reinterpret_cast<const double*>(&lds[wave][J & 1][col * kRecVec]);
#pragma unroll
          for (int d = 0; d < DIM; ++d) {
            pc[c][d] = f[d];
            cacc[c][d] = 0.0;
          }
          ksc[c] = f[DIM];
          cgc[c] = f[DIM + 1];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          double pi[2][DIM], ks[2], cg[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int q = (8 * a + 2 * p + e) * kRecVec;
            const double* f = reinterpret_cast<const double*>(&rows_lds[wave][q + a]);     // (8a + 2p + e) / 8 == a
#pragma unroll
            for (int d = 0; d < DIM; ++d) pi[e][d] = f[d];
            ks[e] = f[DIM];
            cg[e] = f[DIM + 1];
          }
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const uint32_t w0 = c == 0 ? wc[p].x : wc[p].z, w1 = c == 0 ? wc[p].y : wc[p].w;
            const uint32_t d0 = ERR ? (c == 0 ? dc[p].x : dc[p].z) : 0u, d1 = ERR ? (c == 0 ? dc[p].y : dc[p].w) : 0u;
            sym64_pair<DIM, ANYTHR, ERR>(pc[c], ksc[c], cgc[c], pi[0], ks[0], cg[0], w0, racc[2 * p], cacc[c], d0, err_tile, cnt_tile);
            sym64_pair<DIM, ANYTHR, ERR>(pc[c], ksc[c], cgc[c], pi[1], ks[1], cg[1], w1, racc[2 * p + 1], cacc[c], d1, err_tile, cnt_tile);
          }
          // four pairs in flight, no more.  The sums are pinned here (empty statements that "use" them): a scheduling
          // barrier alone does not order pure arithmetic -- instruction selection had put every pair's distance and factor
          // first and all the updates of the sums last, with each pair's dx and factor alive in between (370 registers
          // at ndim 2, scratch from ndim 4)
#pragma unroll
          for (int d = 0; d < DIM; ++d)
            asm volatile("" : "+v"(racc[2 * p][d]), "+v"(racc[2 * p + 1][d]), "+v"(cacc[0][d]), "+v"(cacc[1][d]));
          if constexpr (ERR) asm volatile("" : "+v"(err_tile));
          __builtin_amdgcn_sched_barrier(0);
        }
        // column sums over the 8 lanes a = 0..7 of the column group: lanes a and a ^ 1 hold the two columns in opposite
        // order, so own first + the partner's second is one column's sum over both; then a ^ 2, a ^ 4.  Lane a = 0 ends
        // with column 2h, lane a = 1 with column 2h + 1.  (The diagonal square's sums land in slots nobody reads.)
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          double v = cacc[0][d] + sym64_xor(cacc[1][d], 1);
          v += sym64_xor(v, 2);
          v += sym64_xor(v, 4);
          cacc[0][d] = v;
        }
        // lanes a = 0 and a = 1 store a column each: a buffer store whose offset lies past the buffer's end for the other
        // lanes (dropped by the bounds check).  No branch: the tile stays ONE basic block -- with a divergent store the
        // optimiser sank the row-sum updates of the whole tile behind it and kept every pair's dx and factor alive
        const int off0 = col_off + ((J * kSymCols + 2 * h) * DIM) * 8;
#pragma unroll
        for (int d = 0; d < DIM; d += 2) {
          if (d + 2 <= DIM) {
            const uint4 pk = __builtin_bit_cast(uint4, (double2){cacc[0][d], cacc[0][d + 1]});
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){pk.x, pk.y, pk.z, pk.w}, col_rsrc, off0 + d * 8, 0, 0);
          } else {
            const uint2 pk = __builtin_bit_cast(uint2, cacc[0][d]);
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){pk.x, pk.y}, col_rsrc, off0 + d * 8, 0, 0);
          }
        }
      };
      half(std::integral_constant<int, 0>{}, wa, da);
      request(Jn, 0, wa);                  // the next tile's first half, while this tile's second half is computed
      request_delta(Jn, 0, da);
      __builtin_amdgcn_sched_barrier(0);
      half(std::integral_constant<int, 1>{}, wb, db);
      if constexpr (ERR) {
        const bool diag = J < 2 * R + 2;
        err_unit += diag ? err_tile : 2.0 * err_tile;
        cnt_unit2 += diag ? cnt_tile : 2u * cnt_tile;
        err_tile = 0.0;
        cnt_tile = 0;
      }
      __builtin_amdgcn_sched_barrier(0);
      lds[wave][(J + 1) & 1][lane] = make_uint4(rn0.x, rn0.y, rn0.z, rn0.w);
      if constexpr (kTileVec > 64) lds[wave][(J + 1) & 1][lane + 64] = make_uint4(rn1.x, rn1.y, rn1.z, rn1.w);   // (slots >= kTileVec: never read)
    }
    // row sums over the 8 lanes b = 0..7 of a row group (lane bits 3..5); lane b = 0 stores
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        double v = racc[q][d];
        v += sym64_xor(v, 8);
        v += sym64_xor(v, 16);
        v += sym64_xor(v, 32);
        racc[q][d] = v;
      }
    if (b == 0) {
      double* dst = rowpart + ((size_t)u * kSymRows + 8 * a) * DIM;
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int d = 0; d < DIM; ++d) dst[q * DIM + d] = racc[q][d];
    }
    if constexpr (ERR) {
      double es = err_unit;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) es += sym64_xor(es, m);
      if constexpr (ANYTHR) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) cnt_unit2 += __shfl_xor(cnt_unit2, m, 64);
      }
      if (lane == 0) {
        part_sum[u] = es;
        part_cnt[u] = ANYTHR ? (unsigned long long)cnt_unit2 : (u == 0 ? fixed_cnt : 0ull);
      }
    }
  }
}

// As symm_apply_kernel, in f64: p_i(new) = p_i - (row sums of i's units) + (column sums of the tile-rows above i's),
// summed in a fixed order; writes the positions and the records of the next iteration.
template <int DIM>
__global__ __launch_bounds__(32 * kSymApplyParts) void symm64_apply_kernel(
    const double* __restrict__ rec, double* __restrict__ rec_next, double* __restrict__ pos_out, const float* __restrict__ gplus,
    const double* __restrict__ rowpart, const double* __restrict__ colpart, const int2* __restrict__ row_units, int n,
    int npad, double k_next, double c_rep, int iter1, RunState* st, int rr_stages = 0, int rr_stage = 0) {
  if (st != nullptr && st->stopped) return;
  constexpr int W = SymRec64<DIM>::W;
  constexpr int kWavesA = kSymApplyParts / 2;
  __shared__ double red[kWavesA][kSymCols][DIM];
  const int R = blockIdx.x >> 1;
  const int part = threadIdx.x >> 5, pt = threadIdx.x & 31;
  const int i = blockIdx.x * kSymCols + pt;
  double acc[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) acc[d] = 0.0;
  int rp0 = 0, rp1 = R;           // (rr_stages, rr_stage: as symm_apply_kernel)
  if (rr_stages > 0) sym_rr_above(npad / kSymRows, rr_stages, rr_stage, R, rp0, rp1);
  for (int Rp = rp0 + part; Rp < rp1; Rp += kSymApplyParts) {
    const double* src = colpart + ((size_t)Rp * npad + i) * DIM;
#pragma unroll
    for (int d = 0; d < DIM; ++d) acc[d] += src[d];
  }
  const int2 ru = row_units[R];
  const int row_in_tile = i - R * kSymRows;
  for (int q = part; q < ru.y; q += kSymApplyParts) {
    const double* src = rowpart + ((size_t)(ru.x + q) * kSymRows + row_in_tile) * DIM;
#pragma unroll
    for (int d = 0; d < DIM; ++d) acc[d] -= src[d];
  }
#pragma unroll
  for (int d = 0; d < DIM; ++d) acc[d] += sym64_xor(acc[d], 32);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 32) == 0) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) red[wave][pt][d] = acc[d];
  }
  __syncthreads();
  if (part == 0 && i < n) {
    bool finite = true;
    double out[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      double t = red[0][pt][d];
#pragma unroll
      for (int p = 1; p < kWavesA; ++p) t += red[p][pt][d];
      out[d] = rec[(size_t)i * W + d] + t;
      finite = finite && isfinite(out[d]);
      pos_out[(size_t)i * DIM + d] = out[d];
      rec_next[(size_t)i * W + d] = out[d];
    }
    const double g = (double)gplus[i];
    rec_next[(size_t)i * W + DIM] = 2.0 * k_next / (4.0 * g + k_next);
    rec_next[(size_t)i * W + DIM + 1] = 0.5 * c_rep / g;
    if (!finite && st != nullptr) atomicMin(&st->first_nonfinite, iter1);
  }
}

// Records of iteration `k` from plain positions, and the phantom records [n, npad).
template <int DIM>
__global__ __launch_bounds__(256) void symm64_records_kernel(const double* __restrict__ pos, const float* __restrict__ gplus,
                                                            double* __restrict__ rec, int n, int npad, double k, double c_rep) {
  constexpr int W = SymRec64<DIM>::W;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  double* r = rec + (size_t)i * W;
  if (i < n) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) r[d] = pos[(size_t)i * DIM + d];
    const double g = (double)gplus[i];
    r[DIM] = 2.0 * k / (4.0 * g + k);
    r[DIM + 1] = 0.5 * c_rep / g;
  } else {
#pragma unroll
    for (int d = 0; d < DIM; ++d) r[d] = d == 0 ? kFarF64 : 0.0;
    r[DIM] = 0.0;
    r[DIM + 1] = 0.0;
  }
#pragma unroll
  for (int d = DIM + 2; d < W; ++d) r[d] = 0.0;
}

}  // namespace topolow
