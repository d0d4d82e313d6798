// topolow_amd/csrc/relax_symm.h -- the symmetric sweep: ONE-stage iterations of the slab path on one GPU (fp32)
//
// With one stage per iteration every point moves by the sum of its own halves of all its pairs, all taken from
// the positions the previous iteration left (relax_kernels.h, slab_stage_pipe_kernel with S' = 1).  The row-owner
// kernel meets the unordered pair {i, c} twice -- in row i and in row c -- and computes the same distance, the
// same classification and the same (t - r) / (r + 0.01) or 1 / (r + 0.01)^3 both times; only the endpoint's
// constants differ (reference src/optimization.cpp:245-281: the two halves of a pair update share everything but
// the degree term).  This kernel meets every unordered pair ONCE, from the upper triangle of the encoded block
// (2 N^2 instead of 4 N^2 bytes), and applies the shared factor to both endpoints.
//
// Tiling (fp32 only).  A wave owns a 32 x 32 tile of pairs: lane (a, b), a = lane & 7, b = lane >> 3, owns rows
// 4a..4a+3 and columns 4b..4b+3 of the tile -- 16 pairs, the four rows as two packed halves of v_pk_*_f32.
//   * row sums stay in registers along a unit (a run of tiles in one tile-row) and are reduced over the 8 lanes
//     that share a row once per unit;
//   * column sums are reduced over the 8 lanes that share a column after every tile (three DPP steps inside
//     8-lane groups) and stored as the partial of (tile-row, column);
//   * the diagonal tile is swept in both orders with the column side switched off.
// The tile-row-major list of upper-triangle tiles is cut into equal runs, one per wave of a grid that is resident
// at once; a run is one or more units (a unit never crosses a tile-row).  Partials are stored per unit / per
// tile-row and summed in a fixed order by symm_apply_kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "relax_common.h"

namespace topolow {

#ifndef TOPOLOW_SYM_MINW
#define TOPOLOW_SYM_MINW 2
#endif
constexpr int kSymTile = 32;
constexpr int kSymWaves = 4;   // waves per workgroup (independent of one another)

// One point as the sweep reads it: DIM coordinates, then ks = 2k / (4 g + k) and cg = (c_rep / 2) / g of THIS
// iteration (relax_kernels.h: pair_accum), padded to a multiple of 4 floats.
template <int DIM> struct SymRec { static constexpr int W = (DIM + 2 + 3) & ~3; };

struct SymUnit {
  int tile_row;   // I
  int j0, j1;     // tile columns [j0, j1), j0 >= I
  int tile0;      // index of tile (I, j0) in the tile-major copy of the upper triangle (sym_tile_index)
};   // unit u stores its row partial and its error partial in slot u; the units of a tile-row are consecutive

// The sweep reads the targets from a TILE-MAJOR copy of the upper triangle: tile (I, J), J >= I, is 4 KB at tile index
// I T - I (I - 1) / 2 + (J - I), T = n32 / 32, so a unit is one contiguous run; inside a tile the word of
// (row 4a + j, column 4b + q) sits at ((j * 64 + a + 8 b) * 4 + q): each of a wave's four loads covers 1 KB.  (Read from
// the row-major block the same tile is 32 segments of 128 bytes, rows ld * 4 bytes apart: 68.6 instead of 64.8 us per
// sweep at N = 10 000.)
inline long long sym_tile_index(int I, int J, int T) { return (long long)I * T - (long long)I * (I - 1) / 2 + (J - I); }
inline size_t sym_word_in_tile(int r, int c) { return (size_t)(((r & 3) * 64 + (r >> 2) + 8 * (c >> 2)) * 4 + (c & 3)); }

// Host: the plan of a sweep over n32 / 32 tile-rows for a grid of n_waves waves.
//   units      : tile-row-major; unit u covers tile columns [j0, j1) of tile-row I
//   wave_first : n_waves + 1 entries, wave w sweeps units [wave_first[w], wave_first[w + 1])
//   row_units  : per tile-row (first unit, number of units)
struct SymPlan {
  std::vector<SymUnit> units;
  std::vector<int> wave_first;
  std::vector<int2> row_units;
};
inline SymPlan relax_symm_plan(int n32, int n_waves) {
  SymPlan P;
  const int T = n32 / kSymTile;
  const long long total = (long long)T * (T + 1) / 2;
  P.row_units.resize(T);
  P.wave_first.assign(n_waves + 1, 0);
  long long done = 0;   // tiles handed out so far
  int w = 0;
  long long w_end = (total * (w + 1) + n_waves - 1) / n_waves;   // wave w's run ends at tile w_end (exclusive)
  for (int I = 0; I < T; ++I) {
    P.row_units[I].x = (int)P.units.size();
    int j = I;
    while (j < T) {
      while (done >= w_end && w + 1 < n_waves) {
        ++w;
        P.wave_first[w] = (int)P.units.size();
        w_end = (total * (w + 1) + n_waves - 1) / n_waves;
      }
      const int take = (int)std::min<long long>(T - j, std::max<long long>(w_end - done, 1));
      P.units.push_back({I, j, j + take, (int)sym_tile_index(I, j, T)});
      j += take;
      done += take;
    }
    P.row_units[I].y = (int)P.units.size() - P.row_units[I].x;
  }
  for (int q = w + 1; q <= n_waves; ++q) P.wave_first[q] = (int)P.units.size();
  return P;
}

typedef float symf2 __attribute__((ext_vector_type(2)));

// v + (v of the partner lane) as ONE instruction (v_add_f32 with a DPP operand).  Written as asm because the
// optimiser otherwise pairs the adds of two values into a v_pk_add_f32 fed by two v_mov_b32_dpp -- three
// instructions and 8 cycles where two 2-cycle ones do.  Hazard: a VGPR written by a VALU instruction may be read
// through DPP only two wait states later; sym_col_reduce orders the adds so that a value's next step comes at least
// two instructions after its previous one, and starts with an s_nop.
#define TL_SYM_DPP_ADD(v, ctrl) asm volatile("v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(v))
template <int NV>
__device__ __forceinline__ void sym_col_reduce(float (&v)[NV]) {
  static_assert(NV >= 3, "DPP hazard spacing");
  asm volatile("s_nop 1");
#pragma unroll
  for (int q = 0; q < NV; ++q) TL_SYM_DPP_ADD(v[q], "quad_perm:[1,0,3,2]");
#pragma unroll
  for (int q = 0; q < NV; ++q) TL_SYM_DPP_ADD(v[q], "quad_perm:[2,3,0,1]");
#pragma unroll
  for (int q = 0; q < NV; ++q) TL_SYM_DPP_ADD(v[q], "row_half_mirror");
}

// rows (i0, i1) packed x one column c: both halves of both pairs.
template <int DIM, bool THR, bool ERR, bool CNT, bool FIRST>
__device__ __forceinline__ void sym_pair(const float (&pc)[DIM], float ksc, float cgc, const symf2 (&pi2)[DIM],
                                         symf2 ks2, symf2 cg2, uint32_t w0, uint32_t w1, symf2 (&racc2)[DIM],
                                         symf2 (&cacc2)[DIM], symf2& err2, unsigned& cnt_wave) {
  symf2 dx[DIM];
  symf2 s = {0.0f, 0.0f};
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const symf2 pcd = {pc[d], pc[d]};
    dx[d] = pcd - pi2[d];
    s = __builtin_elementwise_fma(dx[d], dx[d], s);
  }
  const symf2 r = {__builtin_amdgcn_sqrtf(s.x), __builtin_amdgcn_sqrtf(s.y)};
  const symf2 rs = r + (symf2){0.01f, 0.01f};
  const symf2 inv = {__builtin_amdgcn_rcpf(rs.x), __builtin_amdgcn_rcpf(rs.y)};
  const symf2 t = {bits_f32(THR ? (w0 & ~kCodeMask) : w0), bits_f32(THR ? (w1 & ~kCodeMask) : w1)};
  bool sp0, sp1;
  if constexpr (THR) {
    const uint32_t c0 = w0 & kCodeMask, c1 = w1 & kCodeMask;
    sp0 = (c0 == 0u) | ((c0 == 1u) & (r.x < t.x)) | ((c0 == 2u) & (r.x > t.x));
    sp1 = (c1 == 0u) | ((c1 == 1u) & (r.y < t.y)) | ((c1 == 2u) & (r.y > t.y));
  } else {
    sp0 = __builtin_amdgcn_classf(bits_f32(w0), 0x1f8);
    sp1 = __builtin_amdgcn_classf(bits_f32(w1), 0x1f8);
  }
  const symf2 e = t - r;
  const symf2 bs = e * inv;
  const symf2 br = inv * inv * inv;
  const symf2 fs = bs * ks2, fr = br * cg2;
  const symf2 coef = {sp0 ? fs.x : fr.x, sp1 ? fs.y : fr.y};
#pragma unroll
  for (int d = 0; d < DIM; ++d) racc2[d] = __builtin_elementwise_fma(dx[d], coef, racc2[d]);
  const symf2 kc = {ksc, ksc}, gc = {cgc, cgc};
  const symf2 fsc = bs * kc, frc = br * gc;
  const symf2 cc = {sp0 ? fsc.x : frc.x, sp1 ? fsc.y : frc.y};
#pragma unroll
  for (int d = 0; d < DIM; ++d)   // (row 0's, row 1's) share; FIRST: the column's first contribution starts the sum
    cacc2[d] = FIRST ? dx[d] * cc : __builtin_elementwise_fma(dx[d], cc, cacc2[d]);
  if constexpr (ERR) {
    const symf2 a = {sp0 ? fabsf(e.x) : 0.0f, sp1 ? fabsf(e.y) : 0.0f};
    err2 += a;
    if constexpr (CNT)
      cnt_wave += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sp0)) +
                  (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sp1));
  }
}

// enc    : tile-major copy of the upper triangle of the encoded matrix (above; rows and columns >= n hold kInfWord)
// rec    : n32 records (SymRec<DIM>::W floats each); records >= n are the phantom point
// rowpart: [n_units][32][DIM]   row sums of a unit            (sum of dx * coef over the unit's columns)
// colpart: [n_tile_rows][n32][DIM]  column sums of a tile-row (sum of dx * coef_c over the tile-row's 32 rows),
//          written for columns right of the tile-row's diagonal tile only
// part_sum / part_cnt: [n_units]  ERR launches: sum |t - r| and count over the unit's contributing pairs (each
//          unordered pair once; the diagonal tile meets its pairs twice and is weighted 1/2); fixed_cnt: the count of
//          a threshold-free block (every measured pair contributes whatever the positions are), stored in slot 0
template <int DIM, bool ANYTHR, bool ERR>
__global__ __launch_bounds__(64 * kSymWaves, TOPOLOW_SYM_MINW) void symm_sweep_kernel(
    const uint32_t* __restrict__ enc, const float* __restrict__ rec, const SymUnit* __restrict__ units,
    const int* __restrict__ wave_first, float* __restrict__ rowpart, float* __restrict__ colpart, int n32,
    const RunState* st, double* __restrict__ part_sum, unsigned long long* __restrict__ part_cnt,
    unsigned long long fixed_cnt) {
  if (st != nullptr && st->stopped) return;
  constexpr int W = SymRec<DIM>::W;
  constexpr int kRecVec = W / 4;                   // 16-byte pieces per record
  constexpr int kTileVec = kSymTile * kRecVec;     // ... per tile column block (<= 64 * kRecVec)
  // a tile column block's 32 records in LDS, one 16-byte piece of skew after every 4 records: the 8 lane groups b
  // read records 4b + c at the same time, and 4 records are a multiple of the 128 bytes the banks span
  constexpr int kLdsVec = kTileVec + kSymTile / 4;
  __shared__ uint4 lds[kSymWaves][2][kLdsVec];
  auto lds_slot = [](int q) { return q + (q / (4 * kRecVec)); };   // q = record * kRecVec + piece
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int a = lane & 7, b = lane >> 3;

  // units are dealt statically: wave w of the grid sweeps units [wave_first[w], wave_first[w + 1]) -- the host
  // cuts the tile-row-major list of upper-triangle tiles into equal runs, one per wave (relax_symm_plan)
  const int gw = blockIdx.x * kSymWaves + wave;
  const int u_end = __builtin_amdgcn_readfirstlane(wave_first[gw + 1]);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rec_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rec), 0, n32 * W * 4, 0x00020000);
  for (int u = __builtin_amdgcn_readfirstlane(wave_first[gw]); u < u_end; ++u) {
    const SymUnit U = units[u];
    const int I = __builtin_amdgcn_readfirstlane(U.tile_row);
    const int J0 = __builtin_amdgcn_readfirstlane(U.j0), J1 = __builtin_amdgcn_readfirstlane(U.j1);
    const int slot = u;
    const int tile0 = __builtin_amdgcn_readfirstlane(U.tile0);

    // the lane's four rows
    symf2 pi2[2][DIM], ks2[2], cg2[2], racc2[2][DIM];
    {
      const uint4* rr = reinterpret_cast<const uint4*>(rec + (size_t)(I * kSymTile + 4 * a) * W);
      float f[4][W];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int v = 0; v < kRecVec; ++v) {
          const uint4 q = rr[j * kRecVec + v];
          f[j][4 * v + 0] = __builtin_bit_cast(float, q.x);
          f[j][4 * v + 1] = __builtin_bit_cast(float, q.y);
          f[j][4 * v + 2] = __builtin_bit_cast(float, q.z);
          f[j][4 * v + 3] = __builtin_bit_cast(float, q.w);
        }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          pi2[p][d] = (symf2){f[2 * p][d], f[2 * p + 1][d]};
          racc2[p][d] = (symf2){0.0f, 0.0f};
        }
        ks2[p] = (symf2){f[2 * p][DIM], f[2 * p + 1][DIM]};
        cg2[p] = (symf2){f[2 * p][DIM + 1], f[2 * p + 1][DIM + 1]};
      }
    }
    // the unit's tiles as one buffer (wave-uniform descriptor): tile J at (J - J0) * 4 KB, row j of a lane's four 1 KB in
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(enc) + (size_t)tile0 * (kSymTile * kSymTile), 0, (J1 - J0) * kSymTile * kSymTile * 4, 0x00020000);
    const int row_off = lane * 16 - J0 * kSymTile * kSymTile * 4;
    // the tile-row's column partials as one buffer; lanes a != 0 get an offset past its end
    const __amdgpu_buffer_rsrc_t col_rsrc = __builtin_amdgcn_make_buffer_rsrc(colpart + (size_t)I * n32 * DIM, 0, n32 * DIM * 4, 0x00020000);
    const int col_off = a == 0 ? 4 * b * DIM * 4 : 0x40000000;
    u32x4 w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, row_off + j * 1024 + J0 * kSymTile * kSymTile * 4, 0, 0);
    const uint4* recv = reinterpret_cast<const uint4*>(rec);
    if (lane < kTileVec) lds[wave][J0 & 1][lds_slot(lane)] = recv[(size_t)J0 * kTileVec + lane];
    if constexpr (kTileVec > 64) if (lane + 64 < kTileVec) lds[wave][J0 & 1][lds_slot(lane + 64)] = recv[(size_t)J0 * kTileVec + lane + 64];

    symf2 err2 = {0.0f, 0.0f};
    float err_unit = 0.0f;
    unsigned cnt_wave = 0, cnt_unit2 = 0;   // cnt_unit2: twice the count (the diagonal tile counts once per visit)
    // one tile: words of tile J in wc, the next tile's requested into wx (two register sets, used alternately)
    auto tile = [&](int J, const u32x4 (&wc)[4], u32x4 (&wx)[4]) {
      const int Jn = J + 1 < J1 ? J + 1 : J;
#pragma unroll
      for (int j = 0; j < 4; ++j) wx[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, row_off + j * 1024 + Jn * kSymTile * kSymTile * 4, 0, 0);
      // (buffer loads: the optimiser leaves them where they are written; a plain load of the records was sunk
      // down to its use at the end of the tile)
      u32x4 rn0 = {0, 0, 0, 0}, rn1 = {0, 0, 0, 0};
      rn0 = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, (Jn * kTileVec + lane) * 16, 0, 0);
      if constexpr (kTileVec > 64) rn1 = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, (Jn * kTileVec + lane + 64) * 16, 0, 0);
      __builtin_amdgcn_sched_barrier(0);   // the requests stay up here ...

      const bool diag = J == I;
      const float cscale = diag ? 0.0f : 1.0f;
      const uint4* cp = &lds[wave][J & 1][4 * b * kRecVec + b];
      // the lane's four column records are read from LDS one column ahead of their use
      auto read_rec = [&](int col, float (&f)[W]) {
#pragma unroll
        for (int v = 0; v < kRecVec; ++v) {
          const uint4 q = cp[col * kRecVec + v];
          f[4 * v + 0] = __builtin_bit_cast(float, q.x);
          f[4 * v + 1] = __builtin_bit_cast(float, q.y);
          f[4 * v + 2] = __builtin_bit_cast(float, q.z);
          f[4 * v + 3] = __builtin_bit_cast(float, q.w);
        }
      };
      float fq[4][W];
      read_rec(0, fq[0]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {   // columns 2h, 2h + 1 of the lane's four
        symf2 cacc2[2][DIM];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (2 * h + c + 1 < 4) {
            read_rec(2 * h + c + 1, fq[2 * h + c + 1]);
            __builtin_amdgcn_sched_barrier(0);   // the read is issued before this column's arithmetic, not after it
          }
          const float (&f)[W] = fq[2 * h + c];
          float pc[DIM];
#pragma unroll
          for (int d = 0; d < DIM; ++d) pc[d] = f[d];
          const float ksc = f[DIM] * cscale, cgc = f[DIM + 1] * cscale;
          const bool lo = (2 * h + c) == 0, second = (2 * h + c) == 1, third = (2 * h + c) == 2;
          const uint32_t w0 = lo ? wc[0].x : second ? wc[0].y : third ? wc[0].z : wc[0].w;
          const uint32_t w1 = lo ? wc[1].x : second ? wc[1].y : third ? wc[1].z : wc[1].w;
          const uint32_t w2 = lo ? wc[2].x : second ? wc[2].y : third ? wc[2].z : wc[2].w;
          const uint32_t w3 = lo ? wc[3].x : second ? wc[3].y : third ? wc[3].z : wc[3].w;
          sym_pair<DIM, ANYTHR, ERR, ANYTHR, true>(pc, ksc, cgc, pi2[0], ks2[0], cg2[0], w0, w1, racc2[0], cacc2[c], err2, cnt_wave);
          sym_pair<DIM, ANYTHR, ERR, ANYTHR, false>(pc, ksc, cgc, pi2[1], ks2[1], cg2[1], w2, w3, racc2[1], cacc2[c], err2, cnt_wave);
#ifdef TOPOLOW_SYM_SCHED_BARRIER
          __builtin_amdgcn_sched_barrier(0);   // one column's two packed pair updates at a time: bounds the live temporaries
#endif
        }
        {
          // column sums over the 8 lanes a = 0..7 of a column group (lane bits 0..2); all eight hold the sum, lane
          // a = 0 stores it: a buffer store whose offset lies past the buffer's end for the other lanes (dropped by
          // the bounds check) -- no branch, so the compiler's wait counts for the prefetched words stay exact.  The
          // diagonal tile stores zeros into a slot nobody reads (symm_apply_kernel sums the tile-rows strictly above
          // a point's own).
          float flat[2 * DIM];
#pragma unroll
          for (int q = 0; q < 2 * DIM; ++q) flat[q] = cacc2[q / DIM][q % DIM].x + cacc2[q / DIM][q % DIM].y;
          sym_col_reduce<2 * DIM>(flat);
          const int off0 = col_off + ((J * kSymTile) * DIM + h * 2 * DIM) * 4;
#pragma unroll
          for (int q = 0; q < 2 * DIM; q += 4) {
            if (q + 4 <= 2 * DIM) {
              const u32x4 pk = {__builtin_bit_cast(uint32_t, flat[q]), __builtin_bit_cast(uint32_t, flat[q + 1]),
                                __builtin_bit_cast(uint32_t, flat[q + 2]), __builtin_bit_cast(uint32_t, flat[q + 3])};
              __builtin_amdgcn_raw_buffer_store_b128(pk, col_rsrc, off0 + q * 4, 0, 0);
            } else {
#pragma unroll
              for (int t = q; t < 2 * DIM; ++t)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, flat[t]), col_rsrc, off0 + t * 4, 0, 0);
            }
          }
        }
      }
      if constexpr (ERR) {
        const float es = err2.x + err2.y;
        err_unit += diag ? 0.5f * es : es;
        cnt_unit2 += diag ? cnt_wave : 2u * cnt_wave;
        err2 = (symf2){0.0f, 0.0f};
        cnt_wave = 0;
      }
      // hand over: next tile's records into the other LDS half
      __builtin_amdgcn_sched_barrier(0);   // ... and their first use stays down here, a tile's arithmetic later
      if (lane < kTileVec) lds[wave][(J + 1) & 1][lds_slot(lane)] = make_uint4(rn0.x, rn0.y, rn0.z, rn0.w);
      if constexpr (kTileVec > 64) if (lane + 64 < kTileVec) lds[wave][(J + 1) & 1][lds_slot(lane + 64)] = make_uint4(rn1.x, rn1.y, rn1.z, rn1.w);
    };
    u32x4 w2nd[4];
#pragma unroll 1
    for (int J = J0; J < J1; J += 2) {
      tile(J, w, w2nd);
      if (J + 1 < J1) tile(J + 1, w2nd, w);
    }
    // row sums over the 8 lanes b = 0..7 of a row group (lane bits 3..5); lane b = 0 stores
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        symf2 v = racc2[p][d];
        v.x += __shfl_xor(v.x, 8, 64);  v.y += __shfl_xor(v.y, 8, 64);
        v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64);
        v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64);
        racc2[p][d] = v;
      }
    if (b == 0) {
      float* dst = rowpart + ((size_t)slot * kSymTile + 4 * a) * DIM;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          dst[(2 * p) * DIM + d] = racc2[p][d].x;
          dst[(2 * p + 1) * DIM + d] = racc2[p][d].y;
        }
    }
    if constexpr (ERR) {
      double s = (double)err_unit;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
      if (lane == 0) {
        part_sum[slot] = s;
        // threshold-free block: the number of contributing pairs is the host's; otherwise the wave's ballots
        // (cnt_unit2 counted every pair twice, the diagonal tile's two visits once each: always even)
        part_cnt[slot] = ANYTHR ? (unsigned long long)(cnt_unit2 >> 1) : (slot == 0 ? fixed_cnt : 0ull);
      }
    }
  }
}

// Sums the partials of one tile-row's 32 points in a fixed order and moves the points:
//   p_i(new) = p_i - (row sums of i's units) + (column sums of the tile-rows above i's)
// (relax_kernels.h: p_i(new) = p_i - sum over ALL c of (p_c - p_i) coef_i; for c in a tile-row above, the stored
// column sum is sum (p_i - p_c)... with dx = p_c' - p_i' taken row-side, hence the sign).
// Writes the positions (row-major n4 x DIM, as every other kernel reads them) and the records of the NEXT
// iteration (k_next).
constexpr int kSymApplyParts = 32;   // threads of the apply kernel: 32 parts x 32 points (a tile-row)
template <int DIM>
__global__ __launch_bounds__(32 * kSymApplyParts) void symm_apply_kernel(
    const float* __restrict__ rec, float* __restrict__ rec_next, float* __restrict__ pos_out, const float* __restrict__ gplus,
    const float* __restrict__ rowpart, const float* __restrict__ colpart, const int2* __restrict__ row_units, int n,
    int n32, double k_next, double c_rep, int iter1, RunState* st) {
  if (st != nullptr && st->stopped) return;
  constexpr int W = SymRec<DIM>::W;
  __shared__ float red[kSymApplyParts][kSymTile][DIM];
  const int I = blockIdx.x;
  const int part = threadIdx.x >> 5, pt = threadIdx.x & 31;
  const int i = I * kSymTile + pt;
  float acc[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) acc[d] = 0.0f;
  // column sums of the tile-rows above (added), this thread's share: I' = part, part + 32, ... (the last tile-rows
  // sum ~n/32 strips each: with 8 parts their serial chains of loads set the kernel's time, 15 us at n = 10 000)
  for (int Ip = part; Ip < I; Ip += kSymApplyParts) {
    const float* src = colpart + ((size_t)Ip * n32 + i) * DIM;
#pragma unroll
    for (int d = 0; d < DIM; ++d) acc[d] += src[d];
  }
  const int2 ru = row_units[I];
  for (int q = part; q < ru.y; q += kSymApplyParts) {
    const float* src = rowpart + ((size_t)(ru.x + q) * kSymTile + pt) * DIM;
#pragma unroll
    for (int d = 0; d < DIM; ++d) acc[d] -= src[d];
  }
#pragma unroll
  for (int d = 0; d < DIM; ++d) red[part][pt][d] = acc[d];
  __syncthreads();
  if (part == 0 && i < n) {
    bool finite = true;
    float out[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      float t = red[0][pt][d];
#pragma unroll
      for (int p = 1; p < kSymApplyParts; ++p) t += red[p][pt][d];
      out[d] = rec[(size_t)i * W + d] + t;
      finite = finite && isfinite(out[d]);
      pos_out[(size_t)i * DIM + d] = out[d];
      rec_next[(size_t)i * W + d] = out[d];
    }
    const float g = gplus[i];
    rec_next[(size_t)i * W + DIM] = (float)(2.0 * k_next) / (4.0f * g + (float)k_next);
    rec_next[(size_t)i * W + DIM + 1] = (float)(0.5 * c_rep) / g;
    if (!finite && st != nullptr) atomicMin(&st->first_nonfinite, iter1);
  }
}

// Records of iteration `k` from plain positions (the first symmetric iteration after multi-stage ones, and the
// phantom records [n, n32)).
template <int DIM>
__global__ __launch_bounds__(256) void symm_records_kernel(const float* __restrict__ pos, const float* __restrict__ gplus,
                                                          float* __restrict__ rec, int n, int n32, double k, double c_rep) {
  constexpr int W = SymRec<DIM>::W;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n32) return;
  float* r = rec + (size_t)i * W;
  if (i < n) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) r[d] = pos[(size_t)i * DIM + d];
    const float g = gplus[i];
    r[DIM] = (float)(2.0 * k) / (4.0f * g + (float)k);
    r[DIM + 1] = (float)(0.5 * c_rep) / g;
  } else {
#pragma unroll
    for (int d = 0; d < DIM; ++d) r[d] = kFarF32;
    r[DIM] = 0.0f;
    r[DIM + 1] = 0.0f;
  }
#pragma unroll
  for (int d = DIM + 2; d < W; ++d) r[d] = 0.0f;
}

// The tile-major copy of the upper triangle from the row-major encoded block (rows x ld words, rows >= n): one
// workgroup per tile, 4 words per thread; rows past the block's end read as unmeasured.
__global__ __launch_bounds__(256) void symm_tiles_kernel(const uint32_t* __restrict__ enc, int rows, int ld,
                                                        uint32_t* __restrict__ tenc, int T) {
  // tile index -> (I, J): tile-rows are T, T - 1, ... tiles long
  long long t = blockIdx.x;
  int I = 0;
  {
    // largest I with I T - I (I - 1) / 2 <= t  (closed form, then corrected for rounding)
    const double b = 2.0 * T + 1.0;
    I = (int)((b - sqrt(b * b - 8.0 * (double)t)) * 0.5);
    while (I > 0 && (long long)I * T - (long long)I * (I - 1) / 2 > t) --I;
    while ((long long)(I + 1) * T - (long long)(I + 1) * I / 2 <= t) ++I;
  }
  const int J = I + (int)(t - ((long long)I * T - (long long)I * (I - 1) / 2));
  uint32_t* dst = tenc + (size_t)t * (kSymTile * kSymTile);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int cell = threadIdx.x + q * 256;          // 32 x 32 cells, row-major: coalesced reads of 128 bytes per row
    const int r = cell >> 5, c = cell & 31;
    const int row = I * kSymTile + r, col = J * kSymTile + c;
    const uint32_t w = (row < rows && col < ld) ? enc[enc_index(row, col, ld)] : kInfWord;
    dst[((r & 3) * 64 + (r >> 2) + 8 * (c >> 2)) * 4 + (c & 3)] = w;
  }
}

}  // namespace topolow
