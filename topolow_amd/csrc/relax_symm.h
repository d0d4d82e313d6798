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
// Tiling (fp32 only).  A wave owns a 64 x 32 tile of pairs: lane (a, b), a = lane & 7, b = lane >> 3, owns rows
// 8a..8a+7 and columns 4b..4b+3 of the tile -- 32 pairs, the eight rows as four packed halves of v_pk_*_f32.
//   * row sums stay in registers along a unit (a run of tiles in one tile-row of 64 rows) and are reduced over the
//     8 lanes that share a row once per unit;
//   * column sums are reduced over the 8 lanes that share a column after every half tile (two columns x 64 rows;
//     three DPP steps inside 8-lane groups) and stored as the partial of (tile-row, column) -- a 32-row tile
//     (round 2) flushed twice as often per pair and left twice the partials;
//   * the 64 x 64 square on the diagonal (two tiles) is swept in both orders and only its row sums are used.
// The tile-row-major list of upper-triangle tiles is cut into equal runs, one per wave of a grid that is resident
// at once; a run is one or more units (a unit never crosses a tile-row).  Partials are stored per unit / per
// tile-row and summed in a fixed order by symm_apply_kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include <vector>
#include "relax_common.h"

namespace topolow {

#ifndef TOPOLOW_SYM_MINW
#define TOPOLOW_SYM_MINW 2
#endif
constexpr int kSymRows = 64;   // rows of a tile: 8 lane groups a x 8 rows
constexpr int kSymCols = 32;   // columns of a tile: 8 lane groups b x 4 columns
constexpr int kSymTileWords = kSymRows * kSymCols;
constexpr int kSymWaves = 4;   // waves per workgroup (independent of one another)

// One point as the sweep reads it: DIM coordinates, then ks = 2k / (4 g + k) and cg = (c_rep / 2) / g of THIS
// iteration (relax_kernels.h: pair_accum), padded to a multiple of 4 floats.
template <int DIM> struct SymRec { static constexpr int W = (DIM + 2 + 3) & ~3; };

struct SymUnit {
  int tile_row;   // R: rows [64 R, 64 R + 64)
  int j0, j1;     // column blocks [j0, j1) of 32 columns, j0 >= 2 R
  int tile0;      // index of tile (R, j0) in the tile-major copy of the upper triangle (sym_tile_index)
};   // unit u stores its row partial and its error partial in slot u; the units of a tile-row are consecutive

// The sweep reads the targets from a TILE-MAJOR copy of the upper triangle (incl. the whole 64 x 64 squares on the
// diagonal): with TC = npad / 32 column blocks, tile (R, J), J >= 2 R, is 8 KB at tile index R TC - R (R - 1) + (J - 2 R),
// so a unit is one contiguous run.  Inside a tile a lane's eight 16-byte loads are the (column half h, row pair p)
// groups in the order the sweep consumes them: the words of rows 8a + 2p + {0, 1} x columns 4b + 2h + {0, 1} sit at
// (((4 h + p) * 64 + a + 8 b) * 4 + 2 (f ^ (a & 1)) + e) -- a column's two rows adjacent, as the packed pair update
// reads them, and the two columns of a half in opposite order for odd and even a: a lane sweeps "its first" column
// then "its second", which for odd a are columns 2h + 1 and 2h, so that the first step of the column reduction --
// between the lanes a and a ^ 1 -- can add the one's first to the other's second with one instruction per
// coordinate (sym_col_reduce).  Each load of the wave covers 1 KB.
TL_HD inline long long sym_tile_index(int R, int J, int TC) { return (long long)R * TC - (long long)R * (R - 1) + (J - 2 * R); }
TL_HD inline size_t sym_word_in_tile(int r, int c) {   // r in [0, 64), c in [0, 32)
  const int a = r >> 3, p = (r & 7) >> 1, e = r & 1, b = c >> 2, h = (c & 3) >> 1, f = c & 1;
  return (size_t)((((4 * h + p) * 64) + a + 8 * b) * 4 + 2 * (f ^ (a & 1)) + e);
}

// Host: the plan of a sweep over npad / 64 tile-rows for a grid of n_waves waves.
//   units      : tile-row-major; unit u covers column blocks [j0, j1) of tile-row R
//   wave_first : n_waves + 1 entries, wave w sweeps units [wave_first[w], wave_first[w + 1]) (the kernel reads runs())
//   row_units  : per tile-row (first unit, number of units)
struct SymRun {      // one wave's run: units [u0, u1), the first of them inline (one load instead of two dependent ones)
  int u0, u1;
  SymUnit first;
};
struct SymPlan {
  std::vector<SymUnit> units;
  std::vector<int> wave_first;
  std::vector<int2> row_units;
  std::vector<SymRun> runs() const {       // what the kernel reads: wave_first with every wave's first unit beside it
    std::vector<SymRun> r(wave_first.size() - 1);
    for (size_t w = 0; w + 1 < wave_first.size(); ++w) {
      r[w].u0 = wave_first[w];
      r[w].u1 = wave_first[w + 1];
      r[w].first = r[w].u0 < r[w].u1 ? units[r[w].u0] : SymUnit{0, 0, 0, 0};
    }
    return r;
  }
};
// The plan of the tiles [t0, t1) of the tile-row-major list (a SEGMENT: the whole list on one GPU, one of P equal
// runs of it when the sweep is sharded over P row-block sessions): tile0 counts from t0, row_units is indexed by
// R - r_first and covers the tile-rows [r_first, r_last] that hold a tile of the segment.
inline SymPlan relax_symm_plan(int npad, int n_waves, long long t0 = 0, long long t1 = -1, int* r_first = nullptr,
                               int* r_last = nullptr) {
  SymPlan P;
  const int TR = npad / kSymRows, TC = npad / kSymCols;
  if (t1 < 0) t1 = (long long)TR * (TR + 1);
  const long long total = t1 - t0;
  int rf = -1, rl = -1;
  for (int R = 0; R < TR; ++R) {
    const long long a = sym_tile_index(R, 2 * R, TC), b = a + (TC - 2 * R);
    if (b > t0 && a < t1) { if (rf < 0) rf = R; rl = R; }
  }
  if (r_first) *r_first = rf;
  if (r_last) *r_last = rl;
  P.wave_first.assign(n_waves + 1, 0);
  if (rf < 0) return P;
  P.row_units.resize(rl - rf + 1);
  long long done = 0;   // tiles handed out so far
  int w = 0;
  long long w_end = (total * (w + 1) + n_waves - 1) / n_waves;   // wave w's run ends at tile w_end (exclusive)
  for (int R = rf; R <= rl; ++R) {
    P.row_units[R - rf].x = (int)P.units.size();
    const long long row0 = sym_tile_index(R, 2 * R, TC);
    int j = 2 * R + (int)std::max<long long>(0, t0 - row0);
    const int j_end = 2 * R + (int)std::min<long long>(TC - 2 * R, t1 - row0);
    while (j < j_end) {
      while (done >= w_end && w + 1 < n_waves) {
        ++w;
        P.wave_first[w] = (int)P.units.size();
        w_end = (total * (w + 1) + n_waves - 1) / n_waves;
      }
      const int take = (int)std::min<long long>(j_end - j, std::max<long long>(w_end - done, 1));
      P.units.push_back({R, j, j + take, (int)(sym_tile_index(R, j, TC) - t0)});
      j += take;
      done += take;
    }
    P.row_units[R - rf].y = (int)P.units.size() - P.row_units[R - rf].x;
  }
  for (int q = w + 1; q <= n_waves; ++q) P.wave_first[q] = (int)P.units.size();
  return P;
}

// ---- multi-stage iterations on the symmetric sweep: S stages that split the PAIRS (S = 2, 4, 8) -----------------------
// The tile-rows are cut into S slabs (tile-rows [q TR / S, (q + 1) TR / S)); stage st sweeps the pairs between slabs a and
// b with (a + b) mod S == st -- for every slab a exactly one partner slab b = (st - a) mod S, itself included -- so over
// the S stages every point meets each slab once, both ends of a pair move in the same stage, and every pair is evaluated
// once per iteration.  In the upper triangle the pairs (a, b) live in the tile-rows of min(a, b).
TL_HD inline int sym_rr_bound(int TR, int S, int q) { return (int)((long long)q * TR / S); }
TL_HD inline int sym_rr_slab(int TR, int S, int R) {
  int q = 0;
  while (q + 1 < S && R >= sym_rr_bound(TR, S, q + 1)) ++q;
  return q;
}
// The column blocks [j0, j1) of tile-row R that stage st sweeps (empty: j0 >= j1).
TL_HD inline void sym_rr_row(int TR, int S, int st, int R, int& j0, int& j1) {
  const int a = sym_rr_slab(TR, S, R), b = ((st - a) % S + S) % S;
  if (b < a) { j0 = j1 = 0; return; }
  j0 = b == a ? 2 * R : 2 * sym_rr_bound(TR, S, b);
  j1 = 2 * sym_rr_bound(TR, S, b + 1);
}
// The tile-rows [rp0, rp1) whose column sums of stage st belong to a point of tile-row R (symm_apply_kernel).
TL_HD inline void sym_rr_above(int TR, int S, int st, int R, int& rp0, int& rp1) {
  const int c = sym_rr_slab(TR, S, R), a = ((st - c) % S + S) % S;
  if (a > c) { rp0 = rp1 = 0; return; }
  rp0 = sym_rr_bound(TR, S, a);
  rp1 = a == c ? R : sym_rr_bound(TR, S, a + 1);
}
// The order of the stages in iteration `iter` (random per iteration, like the slabs of the row-owner form).
TL_HD inline void sym_rr_order(uint64_t seed, int iter, int S, int* perm) {
  for (int q = 0; q < S; ++q) perm[q] = q;
  for (int q = S - 1; q > 0; --q) {
    const uint32_t r = rnd_below(rnd64(seed, 0x2a1f5ull, ((uint64_t)iter << 8) | (uint64_t)q), (uint32_t)(q + 1));
    const int tmp = perm[q]; perm[q] = perm[r]; perm[r] = tmp;
  }
}

// The plan of an arbitrary set of tiles given per tile-row as ONE interval of column blocks: rows_j(R, j0, j1) sets the
// interval [j0, j1) of tile-row R (j0 >= 2 R; empty when j0 >= j1).  Equal runs per wave as above; row_units covers all
// tile-rows.  Used for the two halves of a two-stage iteration (topolow_relax.hip: sym_half_stage).
template <typename RowsJ>
inline SymPlan relax_symm_plan_rows(int npad, int n_waves, RowsJ rows_j) {
  SymPlan P;
  const int TR = npad / kSymRows, TC = npad / kSymCols;
  long long total = 0;
  for (int R = 0; R < TR; ++R) {
    int j0 = 0, j1 = 0;
    rows_j(R, j0, j1);
    if (j1 > j0) total += j1 - j0;
  }
  P.wave_first.assign(n_waves + 1, 0);
  P.row_units.assign(TR, int2{0, 0});
  if (total == 0) return P;
  long long done = 0;
  int w = 0;
  long long w_end = (total * (w + 1) + n_waves - 1) / n_waves;
  for (int R = 0; R < TR; ++R) {
    P.row_units[R].x = (int)P.units.size();
    int j = 0, j_end = 0;
    rows_j(R, j, j_end);
    while (j < j_end) {
      while (done >= w_end && w + 1 < n_waves) {
        ++w;
        P.wave_first[w] = (int)P.units.size();
        w_end = (total * (w + 1) + n_waves - 1) / n_waves;
      }
      const int take = (int)std::min<long long>(j_end - j, std::max<long long>(w_end - done, 1));
      P.units.push_back({R, j, j + take, (int)sym_tile_index(R, j, TC)});
      j += take;
      done += take;
    }
    P.row_units[R].y = (int)P.units.size() - P.row_units[R].x;
  }
  for (int q = w + 1; q <= n_waves; ++q) P.wave_first[q] = (int)P.units.size();
  return P;
}

typedef float symf2 __attribute__((ext_vector_type(2)));
#ifdef TOPOLOW_SYM_STAMPS
__device__ unsigned long long* g_sym_stamps = nullptr;   // [4 * waves]: shader cycles of the wave's lifetime, its start and end on the 100-MHz counter, (XCC_ID, HW_ID)
#endif

// v + (v of the partner lane) as ONE instruction (v_add_f32 with a DPP operand).  Written as asm because the
// optimiser otherwise pairs the adds of two values into a v_pk_add_f32 fed by two v_mov_b32_dpp -- three
// instructions and 8 cycles where two 2-cycle ones do.  Hazard: a VGPR written by a VALU instruction may be read
// through DPP only two wait states later; sym_col_reduce orders the adds so that a value's next step comes at least
// two instructions after its previous one, and starts with an s_nop.
#define TL_SYM_DPP_ADD(v, ctrl) asm volatile("v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(v))
// Column sums of a half tile over the 8 lanes a = 0..7 of a column group.  In: per lane the sums of ITS first and
// ITS second column (for even a columns 2h, 2h + 1, for odd a the other way round).  Step 1 (lanes a, a ^ 1): own
// first + the partner's second -- the same column in both -- so NV adds serve both columns; steps 2 and 3 (a ^ 2, then
// a + 4 through row_shl:4, which is right for a < 4) on NV values.  Out: first[] holds the total of column 2h in lane
// a = 0 and of column 2h + 1 in lane a = 1 (round 2 reduced both columns in every lane: 3 x 2 NV adds).
template <int NV>
__device__ __forceinline__ void sym_col_reduce(float (&first)[NV], float (&second)[NV]) {
  // every operand is pinned in its register BEFORE the first DPP read: the empty statements make the compiler finish
  // the sums that feed them here (left free it sinks one of them between two of the adds below, whose DPP read of the
  // register just written then returns the old value -- the hazard is invisible to it inside inline asm)
#pragma unroll
  for (int q = 0; q < NV; ++q) asm volatile("" : "+v"(first[q]), "+v"(second[q]));
  asm volatile("s_nop 1");
#pragma unroll
  for (int q = 0; q < NV; ++q)
    asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(first[q]) : "v"(second[q]));
  if constexpr (NV < 3) asm volatile("s_nop 1");
#pragma unroll
  for (int q = 0; q < NV; ++q) TL_SYM_DPP_ADD(first[q], "quad_perm:[2,3,0,1]");
  if constexpr (NV < 3) asm volatile("s_nop 1");
#pragma unroll
  for (int q = 0; q < NV; ++q) TL_SYM_DPP_ADD(first[q], "row_shl:4");
}

// rows (i0, i1) packed x one column c: both halves of both pairs.  The pair's shared factor is selected ONCE --
// u = (t - r) for a spring, (r + 0.01)^-2 otherwise; base = u / (r + 0.01) -- and each endpoint multiplies it with
// its own constant of the same kind (ks: 2k / (4 g + k), cg: c / 2g): 5 packed multiplies and 6 selects where
// computing both kinds for both endpoints took 8 and 4; the products are the same, bit for bit.
template <int DIM, bool THR, bool ERR, bool CNT, bool FIRST>
__device__ __forceinline__ void sym_pair(const float (&pc)[DIM], float ksc, float cgc, const symf2 (&pi2)[DIM],
                                         symf2 ks2, symf2 cg2, uint32_t w0, uint32_t w1, symf2 (&racc2)[DIM],
                                         symf2 (&cacc2)[DIM], symf2& err2, unsigned& cnt_wave /* per lane */) {
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
    // (code 0: exact target, always a spring; 1: ">" -- a spring while r < t; 2: "<" -- a spring while r > t).  Decided
    // from the sign of t - r with integer operations on the word and ONE comparison per pair, as in the threshold-free
    // instance: comparisons per code keep a dozen lane masks per pair in flight and the masks of a half tile no longer
    // fit the scalar registers (the instance then spills both register files).
    const symf2 e0 = t - r;
    auto decide = [](uint32_t w, float e) {
      const uint32_t flip = (w << 30) & 0x80000000u;                         // code 2: the sign of t - r turned round
      const int32_t some = (int32_t)(((w >> 1) | w) << 31) >> 31;           // -1 for codes 1 and 2, 0 for an exact target
      const uint32_t q = ((f32_bits(e) ^ flip) & (uint32_t)some) | (0x3f800000u & ~(uint32_t)some);
      return bits_f32(q) > 0.0f;
    };
    sp0 = decide(w0, e0.x);
    sp1 = decide(w1, e0.y);
  } else {
    sp0 = __builtin_amdgcn_classf(bits_f32(w0), 0x1f8);
    sp1 = __builtin_amdgcn_classf(bits_f32(w1), 0x1f8);
  }
  const symf2 e = t - r;
  const symf2 inv2 = inv * inv;
  const symf2 u = {sp0 ? e.x : inv2.x, sp1 ? e.y : inv2.y};
  const symf2 base = u * inv;
  const symf2 mr = {sp0 ? ks2.x : cg2.x, sp1 ? ks2.y : cg2.y};
  const symf2 coef = base * mr;
#pragma unroll
  for (int d = 0; d < DIM; ++d) racc2[d] = __builtin_elementwise_fma(dx[d], coef, racc2[d]);
  const symf2 mc = {sp0 ? ksc : cgc, sp1 ? ksc : cgc};
  const symf2 cc = base * mc;
#pragma unroll
  for (int d = 0; d < DIM; ++d)   // (row 0's, row 1's) share; FIRST: the column's first contribution starts the sum
    cacc2[d] = FIRST ? dx[d] * cc : __builtin_elementwise_fma(dx[d], cc, cacc2[d]);
  if constexpr (ERR) {
    const symf2 a = {sp0 ? fabsf(e.x) : 0.0f, sp1 ? fabsf(e.y) : 0.0f};
    err2 += a;
    if constexpr (CNT) cnt_wave += (sp0 ? 1u : 0u) + (sp1 ? 1u : 0u);   // per lane (ballots would keep the masks in SGPRs)
  }
}

// enc    : tile-major copy of the upper triangle of the encoded matrix (above; rows and columns >= n hold kInfWord)
// rec    : npad records (SymRec<DIM>::W floats each), npad = roundup(n, 64); records >= n are the phantom point
// rowpart: [n_units][64][DIM]   row sums of a unit            (sum of dx * coef over the unit's columns)
// colpart: [n_tile_rows][npad][DIM]  column sums of a tile-row (sum of dx * coef_c over the tile-row's 64 rows),
//          meaningful for the columns right of the tile-row's diagonal square only; tile-row R is row R - col_row0
//          (a segment's first tile-row; 0 for the whole triangle)
// part_sum / part_cnt: [n_units]  ERR launches: TWICE the sum |t - r| and TWICE the count over the unit's contributing
//          pairs, as the row-owner ERR instance leaves them (it meets every pair twice; the ratio is the MAE): a pair of
//          the diagonal square is met from both sides, possibly by two units, and counts once per visit, every other pair
//          twice at its one visit; fixed_cnt: twice the count of a threshold-free block (every measured pair contributes
//          whatever the positions are: the number of measured cells), stored in slot 0
template <int DIM, bool ANYTHR, bool ERR>
__global__ __launch_bounds__(64 * kSymWaves, TOPOLOW_SYM_MINW) void symm_sweep_kernel(
    const uint32_t* __restrict__ enc, const float* __restrict__ rec, const SymUnit* __restrict__ units,
    const SymRun* __restrict__ runs, float* __restrict__ rowpart, float* __restrict__ colpart, int npad,
    const RunState* st, double* __restrict__ part_sum, unsigned long long* __restrict__ part_cnt,
    unsigned long long fixed_cnt, int col_row0) {
  if (st != nullptr && st->stopped) return;
#ifdef TOPOLOW_SYM_STAMPS   // diagnostic build (tools/symm_probe.hip): shader clock against the 100-MHz real-time counter, per wave
  const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int W = SymRec<DIM>::W;
  constexpr int kRecVec = W / 4;                   // 16-byte pieces per record
  constexpr int kTileVec = kSymCols * kRecVec;     // ... per column block (<= 64 * kRecVec)
  // a column block's 32 records in LDS, one 16-byte piece of skew after every 4 records: the 8 lane groups b
  // read records 4b + c at the same time, and 4 records are a multiple of the 128 bytes the banks span
  constexpr int kLdsVec = kTileVec + kSymCols / 4;
  __shared__ uint4 lds[kSymWaves][2][kLdsVec];
  auto lds_slot = [](int q) { return q + (q / (4 * kRecVec)); };   // q = record * kRecVec + piece
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int a = lane & 7, b = lane >> 3;

  // units are dealt statically: wave w of the grid sweeps units [runs[w].u0, runs[w].u1) -- the host
  // cuts the tile-row-major list of upper-triangle tiles into equal runs, one per wave (relax_symm_plan)
  const int gw = blockIdx.x * kSymWaves + wave;
  const SymRun run = runs[gw];
  const int u_begin = __builtin_amdgcn_readfirstlane(run.u0);
  const int u_end = __builtin_amdgcn_readfirstlane(run.u1);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rec_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rec), 0, npad * W * 4, 0x00020000);
  for (int u = u_begin; u < u_end; ++u) {
    const SymUnit U = u == u_begin ? run.first : units[u];
    const int R = __builtin_amdgcn_readfirstlane(U.tile_row);
    const int J0 = __builtin_amdgcn_readfirstlane(U.j0), J1 = __builtin_amdgcn_readfirstlane(U.j1);
    const int slot = u;
    const int tile0 = __builtin_amdgcn_readfirstlane(U.tile0);

    // the lane's eight rows
    symf2 pi2[4][DIM], ks2[4], cg2[4], racc2[4][DIM];
    {
      const uint4* rr = reinterpret_cast<const uint4*>(rec + (size_t)(R * kSymRows + 8 * a) * W);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float f[2][W];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int v = 0; v < kRecVec; ++v) {
            const uint4 q = rr[(2 * p + e) * kRecVec + v];
            f[e][4 * v + 0] = __builtin_bit_cast(float, q.x);
            f[e][4 * v + 1] = __builtin_bit_cast(float, q.y);
            f[e][4 * v + 2] = __builtin_bit_cast(float, q.z);
            f[e][4 * v + 3] = __builtin_bit_cast(float, q.w);
          }
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          pi2[p][d] = (symf2){f[0][d], f[1][d]};
          racc2[p][d] = (symf2){0.0f, 0.0f};
        }
        ks2[p] = (symf2){f[0][DIM], f[1][DIM]};
        cg2[p] = (symf2){f[0][DIM + 1], f[1][DIM + 1]};
      }
    }
    // the unit's tiles as one buffer (wave-uniform descriptor): tile J at (J - J0) * 8 KB; a lane's load (h, p) 1 KB apart
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(enc) + (size_t)tile0 * kSymTileWords, 0, (J1 - J0) * kSymTileWords * 4, 0x00020000);
    // the tile-row's column partials as one buffer; lane a = 0 stores the first column of a half, a = 1 the second,
    // the other lanes get an offset past its end
    const __amdgpu_buffer_rsrc_t col_rsrc = __builtin_amdgcn_make_buffer_rsrc(colpart + (size_t)(R - col_row0) * npad * DIM, 0, npad * DIM * 4, 0x00020000);
    const int col_off = a < 2 ? (4 * b + a) * DIM * 4 : 0x40000000;
    const int swap = a & 1;                // this lane's q-th column of a half is column 2h + (q ^ swap)
    // the words of half tile (J, h): four 16-byte loads, one per row pair
    auto request = [&](int J, int h, u32x4 (&dst)[4]) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        dst[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16 + ((J - J0) * 8 + 4 * h + p) * 1024, 0, 0);
    };
    u32x4 wa[4], wb[4];
    request(J0, 0, wa);
    const uint4* recv = reinterpret_cast<const uint4*>(rec);
    if (lane < kTileVec) lds[wave][J0 & 1][lds_slot(lane)] = recv[(size_t)J0 * kTileVec + lane];
    if constexpr (kTileVec > 64) if (lane + 64 < kTileVec) lds[wave][J0 & 1][lds_slot(lane + 64)] = recv[(size_t)J0 * kTileVec + lane + 64];

    symf2 err2 = {0.0f, 0.0f};
    float err_unit = 0.0f;
    unsigned cnt_wave = 0, cnt_unit2 = 0;   // err_unit, cnt_unit2: twice the sum / count (the diagonal square counts once per visit)
    // the lane's four column records come from LDS one column ahead of their use -- the first column's at the end of
    // the previous tile, behind the write that hands its block over
    auto read_rec = [&](int J, int idx, float (&f)[W]) {   // the lane's idx-th column of the tile, in ITS order
      const int col = idx ^ swap;
      const uint4* cp = &lds[wave][J & 1][4 * b * kRecVec + b];
#pragma unroll
      for (int v = 0; v < kRecVec; ++v) {
        const uint4 q = cp[col * kRecVec + v];
        f[4 * v + 0] = __builtin_bit_cast(float, q.x);
        f[4 * v + 1] = __builtin_bit_cast(float, q.y);
        f[4 * v + 2] = __builtin_bit_cast(float, q.z);
        f[4 * v + 3] = __builtin_bit_cast(float, q.w);
      }
    };
    float f_first[W];
    read_rec(J0, 0, f_first);
#pragma unroll 1
    for (int J = J0; J < J1; ++J) {
      const int Jn = J + 1 < J1 ? J + 1 : J;
      // requests first: the second half's words, the next column block's records (buffer loads: the optimiser leaves
      // them where they are written; a plain load of the records was sunk down to its use at the end of the tile)
      request(J, 1, wb);
      u32x4 rn0 = {0, 0, 0, 0}, rn1 = {0, 0, 0, 0};
      rn0 = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, (Jn * kTileVec + lane) * 16, 0, 0);
      if constexpr (kTileVec > 64) rn1 = __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, (Jn * kTileVec + lane + 64) * 16, 0, 0);
      __builtin_amdgcn_sched_barrier(0);   // the requests stay up here ...

      const bool diag = J < 2 * R + 2;   // (its column sums go to slots nobody reads: no need to switch the column side off)
      float fq[4][W];
#pragma unroll
      for (int q = 0; q < W; ++q) fq[0][q] = f_first[q];
      auto half = [&](auto hc, const u32x4 (&wc)[4]) {   // columns 2h, 2h + 1 of the lane's four x its eight rows
        constexpr int h = decltype(hc)::value;
        symf2 cacc2[2][DIM];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          if (2 * h + c + 1 < 4) {
            read_rec(J, 2 * h + c + 1, fq[2 * h + c + 1]);
            __builtin_amdgcn_sched_barrier(0);   // the read is issued before this column's arithmetic, not after it
          }
          const float (&f)[W] = fq[2 * h + c];
          float pc[DIM];
#pragma unroll
          for (int d = 0; d < DIM; ++d) pc[d] = f[d];
          const float ksc = f[DIM], cgc = f[DIM + 1];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const uint32_t w0 = c == 0 ? wc[p].x : wc[p].z, w1 = c == 0 ? wc[p].y : wc[p].w;
            if (p == 0)
              sym_pair<DIM, ANYTHR, ERR, ANYTHR, true>(pc, ksc, cgc, pi2[p], ks2[p], cg2[p], w0, w1, racc2[p], cacc2[c], err2, cnt_wave);
            else
              sym_pair<DIM, ANYTHR, ERR, ANYTHR, false>(pc, ksc, cgc, pi2[p], ks2[p], cg2[p], w0, w1, racc2[p], cacc2[c], err2, cnt_wave);
          }
        }
        // column sums over the 8 lanes a = 0..7 of a column group (lane bits 0..2; sym_col_reduce); lanes a = 0 and
        // a = 1 store a column each: a buffer store whose offset lies past the buffer's end for the other lanes
        // (dropped by the bounds check) -- no branch, so the compiler's wait counts for the prefetched words stay
        // exact.  The diagonal square's sums land in slots nobody reads (symm_apply_kernel sums the tile-rows strictly
        // above a point's own): there every pair is met from both sides and only the row side counts.
        float first[DIM], second[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          first[d] = cacc2[0][d].x + cacc2[0][d].y;
          second[d] = cacc2[1][d].x + cacc2[1][d].y;
        }
        sym_col_reduce<DIM>(first, second);
        const int off0 = col_off + ((J * kSymCols) * DIM + h * 2 * DIM) * 4;
#pragma unroll
        for (int q = 0; q < DIM; q += 4) {
          if (q + 4 <= DIM) {
            const u32x4 pk = {__builtin_bit_cast(uint32_t, first[q]), __builtin_bit_cast(uint32_t, first[q + 1]),
                              __builtin_bit_cast(uint32_t, first[q + 2]), __builtin_bit_cast(uint32_t, first[q + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(pk, col_rsrc, off0 + q * 4, 0, 0);
          } else {
#pragma unroll
            for (int t = q; t < DIM; ++t)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, first[t]), col_rsrc, off0 + t * 4, 0, 0);
          }
        }
      };
      half(std::integral_constant<int, 0>{}, wa);
      request(Jn, 0, wa);                  // the next tile's first half, while this tile's second half is computed
      __builtin_amdgcn_sched_barrier(0);
      half(std::integral_constant<int, 1>{}, wb);
      if constexpr (ERR) {
        const float es = err2.x + err2.y;
        err_unit += diag ? es : 2.0f * es;
        cnt_unit2 += diag ? cnt_wave : 2u * cnt_wave;
        err2 = (symf2){0.0f, 0.0f};
        cnt_wave = 0;
      }
      // hand over: next column block's records into the other LDS half
      __builtin_amdgcn_sched_barrier(0);   // ... and their first use stays down here, a tile's arithmetic later
      if (lane < kTileVec) lds[wave][(J + 1) & 1][lds_slot(lane)] = make_uint4(rn0.x, rn0.y, rn0.z, rn0.w);
      if constexpr (kTileVec > 64) if (lane + 64 < kTileVec) lds[wave][(J + 1) & 1][lds_slot(lane + 64)] = make_uint4(rn1.x, rn1.y, rn1.z, rn1.w);
      read_rec(J + 1, 0, f_first);         // (a wave's LDS operations complete in order: this read sees the write above)
    }
    // row sums over the 8 lanes b = 0..7 of a row group (lane bits 3..5); lane b = 0 stores
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        symf2 v = racc2[p][d];
        v.x += __shfl_xor(v.x, 8, 64);  v.y += __shfl_xor(v.y, 8, 64);
        v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64);
        v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64);
        racc2[p][d] = v;
      }
    if (b == 0) {
      float* dst = rowpart + ((size_t)slot * kSymRows + 8 * a) * DIM;
#pragma unroll
      for (int p = 0; p < 4; ++p)
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
      if constexpr (ANYTHR) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) cnt_unit2 += __shfl_xor(cnt_unit2, m, 64);
      }
      if (lane == 0) {
        part_sum[slot] = s;
        // threshold-free block: the number of contributing cells is the host's; otherwise the wave's ballots
        part_cnt[slot] = ANYTHR ? (unsigned long long)cnt_unit2 : (slot == 0 ? fixed_cnt : 0ull);
      }
    }
  }
#ifdef TOPOLOW_SYM_STAMPS
  if (lane == 0 && g_sym_stamps != nullptr) {
    g_sym_stamps[4 * gw] = __builtin_amdgcn_s_memtime() - stamp_c0;
    g_sym_stamps[4 * gw + 1] = stamp_r0;
    g_sym_stamps[4 * gw + 2] = __builtin_amdgcn_s_memrealtime();
    unsigned hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    g_sym_stamps[4 * gw + 3] = ((unsigned long long)xcc_id << 32) | hw_id;
  }
#endif
}

// Sums the partials of one column block's 32 points in a fixed order and moves the points:
//   p_i(new) = p_i - (row sums of i's units) + (column sums of the tile-rows above i's)
// (relax_kernels.h: p_i(new) = p_i - sum over ALL c of (p_c - p_i) coef_i; for c in a tile-row above, the stored
// column sum is sum (p_i - p_c)... with dx = p_c' - p_i' taken row-side, hence the sign).
// Writes the positions (row-major n4 x DIM, as every other kernel reads them) and the records of the NEXT
// iteration (k_next).
#ifndef TL_APPLY_PARTS
#define TL_APPLY_PARTS 16
#endif
constexpr int kSymApplyParts = TL_APPLY_PARTS;   // threads of the apply kernel: 16 parts x 32 points (a column block); a wave = 2 parts
template <int DIM>
__global__ __launch_bounds__(32 * kSymApplyParts) void symm_apply_kernel(
    const float* __restrict__ rec, float* __restrict__ rec_next, float* __restrict__ pos_out, const float* __restrict__ gplus,
    const float* __restrict__ rowpart, const float* __restrict__ colpart, const int2* __restrict__ row_units, int n,
    int npad, double k_next, double c_rep, int iter1, RunState* st, int rr_stages = 0, int rr_stage = 0) {
  if (st != nullptr && st->stopped) return;
  constexpr int W = SymRec<DIM>::W;
  constexpr int kWavesA = kSymApplyParts / 2;
  __shared__ float red[kWavesA][kSymCols][DIM];
  const int R = blockIdx.x >> 1;                   // the tile-row of this column block's points
  const int part = threadIdx.x >> 5, pt = threadIdx.x & 31;
  const int i = blockIdx.x * kSymCols + pt;
  float acc[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) acc[d] = 0.0f;
  // column sums of the tile-rows above (added), this thread's share: R' = part, part + 16, ... (independent loads: all
  // of a thread's strips are in flight together).  rr_stages > 0: the sweep was ONE STAGE of a multi-stage iteration
  // (sym_rr_*: the tiles that pair this point's slab with its partner slab of the stage): only those tile-rows' sums are
  // this sweep's, the other slots hold an older sweep's
  int rp0 = 0, rp1 = R;
  if (rr_stages > 0) sym_rr_above(npad / kSymRows, rr_stages, rr_stage, R, rp0, rp1);
  for (int Rp = rp0 + part; Rp < rp1; Rp += kSymApplyParts) {
    const float* src = colpart + ((size_t)Rp * npad + i) * DIM;
#pragma unroll
    for (int d = 0; d < DIM; ++d) acc[d] += src[d];
  }
  const int2 ru = row_units[R];
  const int row_in_tile = i - R * kSymRows;
  for (int q = part; q < ru.y; q += kSymApplyParts) {
    const float* src = rowpart + ((size_t)(ru.x + q) * kSymRows + row_in_tile) * DIM;
#pragma unroll
    for (int d = 0; d < DIM; ++d) acc[d] -= src[d];
  }
  // fixed order: the two parts of a wave (lanes l, l + 32), then the waves, then onto the point
#pragma unroll
  for (int d = 0; d < DIM; ++d) acc[d] += __shfl_xor(acc[d], 32, 64);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 32) == 0) {
#pragma unroll
    for (int d = 0; d < DIM; ++d) red[wave][pt][d] = acc[d];
  }
  __syncthreads();
  if (part == 0 && i < n) {
    bool finite = true;
    float out[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      float t = red[0][pt][d];
#pragma unroll
      for (int p = 1; p < kWavesA; ++p) t += red[p][pt][d];
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
// phantom records [n, npad)).
template <int DIM>
__global__ __launch_bounds__(256) void symm_records_kernel(const float* __restrict__ pos, const float* __restrict__ gplus,
                                                          float* __restrict__ rec, int n, int npad, double k, double c_rep) {
  constexpr int W = SymRec<DIM>::W;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
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

// The tile-major copy of tiles [t_first, t_first + gridDim.x) of the upper triangle from the row-major encoded matrix,
// which lies in n_src row blocks (one on a single GPU; the row-block sessions of a sharded run otherwise -- on other
// GPUs of the node their words arrive as peer reads, once per loaded matrix): block q holds rows [row0[q], row0[q + 1])
// of ld words each.  One workgroup per tile, 8 words per thread; rows and columns past the matrix read as unmeasured.
__global__ __launch_bounds__(256) void symm_tiles_kernel(const uint32_t* const* __restrict__ src, const int* __restrict__ row0,
                                                        int n_src, int ld, uint32_t* __restrict__ tenc, int TC,
                                                        long long t_first) {
  // tile index -> (R, J): tile-row R starts at R TC - R (R - 1) and is TC - 2 R tiles long
  const long long t = t_first + blockIdx.x;
  int R = 0;
  {
    // largest R with R (TC + 1) - R^2 <= t  (closed form, then corrected for rounding)
    const double b = (double)TC + 1.0;
    const double disc = b * b - 4.0 * (double)t;
    R = (int)((b - sqrt(disc > 0.0 ? disc : 0.0)) * 0.5);
    while (R > 0 && (long long)R * TC - (long long)R * (R - 1) > t) --R;
    while ((long long)(R + 1) * TC - (long long)(R + 1) * R <= t) ++R;
  }
  const int J = 2 * R + (int)(t - ((long long)R * TC - (long long)R * (R - 1)));
  uint32_t* dst = tenc + (size_t)blockIdx.x * kSymTileWords;
  const int rows = row0[n_src];
#pragma unroll
  for (int q = 0; q < kSymTileWords / 256; ++q) {
    const int cell = threadIdx.x + q * 256;          // 64 x 32 cells, row-major: coalesced reads of 128 bytes per row
    const int r = cell >> 5, c = cell & 31;
    const int row = R * kSymRows + r, col = J * kSymCols + c;
    uint32_t w = kInfWord;
    if (row < rows && col < ld) {
      int blk = 0;
      while (blk + 1 < n_src && row >= row0[blk + 1]) ++blk;
      w = src[blk][enc_index(row - row0[blk], col, ld)];
    }
    const int a = r >> 3, p = (r & 7) >> 1, e = r & 1, b = c >> 2, h = (c & 3) >> 1, f = c & 1;
    dst[(((4 * h + p) * 64) + a + 8 * b) * 4 + 2 * (f ^ (a & 1)) + e] = w;
  }
}

// ---- the sweep sharded over P row-block sessions (relax_sharded_engine.h) --------------------------------------
// Session b sweeps segment b of the tile list (equal tile counts, so equal work) into its own partials; a point's
// move needs the partials of every segment, so each session folds ITS partials per point (symm_partial_kernel: what
// symm_apply_kernel sums, restricted to one segment) and stores the result into slot b of the inbox of the session
// that OWNS the point (its row block: a peer store when that is another GPU) -- npad x DIM floats per session and
// iteration, whatever the size of the partials; behind the engine's barrier the owner adds the P slots in the
// order of the sessions (symm_owner_apply_kernel), moves its points and stores them into every session's positions.

// grid: npad / 32 workgroups of 32 parts x 32 points.  inbox[q]: owner q's inbox, [P][npad][DIM]; own0: first row of
// every owner (n_own + 1 entries); tile-rows [r_first, r_last] hold this segment's tiles (r_first > r_last: none).
template <int DIM>
__global__ __launch_bounds__(32 * 32) void symm_partial_kernel(
    const float* __restrict__ rowpart, const float* __restrict__ colpart, const int2* __restrict__ row_units, int r_first,
    int r_last, int n, int npad, float* const* __restrict__ inbox, const int* __restrict__ own0, int n_own, int slot,
    const RunState* st) {
  if (st != nullptr && st->stopped) return;
  __shared__ float red[32][kSymCols][DIM];
  const int R = blockIdx.x >> 1;
  const int part = threadIdx.x >> 5, pt = threadIdx.x & 31;
  const int i = blockIdx.x * kSymCols + pt;
  float acc[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) acc[d] = 0.0f;
  const int r_stop = R < r_last + 1 ? R : r_last + 1;     // tile-rows of this segment strictly above the point's own
  for (int Rp = r_first + part; Rp < r_stop; Rp += 32) {
    const float* src = colpart + ((size_t)(Rp - r_first) * npad + i) * DIM;
#pragma unroll
    for (int d = 0; d < DIM; ++d) acc[d] += src[d];
  }
  if (R >= r_first && R <= r_last) {
    const int2 ru = row_units[R - r_first];
    const int row_in_tile = i - R * kSymRows;
    for (int q = part; q < ru.y; q += 32) {
      const float* src = rowpart + ((size_t)(ru.x + q) * kSymRows + row_in_tile) * DIM;
#pragma unroll
      for (int d = 0; d < DIM; ++d) acc[d] -= src[d];
    }
  }
#pragma unroll
  for (int d = 0; d < DIM; ++d) red[part][pt][d] = acc[d];
  __syncthreads();
  if (part == 0 && i < n) {
    int q = 0;
    while (q + 1 < n_own && i >= own0[q + 1]) ++q;
    float* dst = inbox[q] + ((size_t)slot * npad + i) * DIM;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      float t = red[0][pt][d];
#pragma unroll
      for (int p = 1; p < 32; ++p) t += red[p][pt][d];
      dst[d] = t;
    }
  }
}

// The owner's points [row_begin, row_end): new position = old + the P slots of its inbox in session order; stored into
// its own next-position buffer and into the other sessions' (push / n_push, as the stage kernel's epilogue does).
template <int DIM>
__global__ __launch_bounds__(256) void symm_owner_apply_kernel(
    const float* __restrict__ pos_in, float* __restrict__ pos_out, const float* __restrict__ inbox, int n_slots, int npad,
    int row_begin, int row_end, float* const* __restrict__ push, int n_push, int iter1, RunState* st) {
  if (st != nullptr && st->stopped) return;
  const int i = row_begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= row_end) return;
  bool finite = true;
  float out[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    float t = 0.0f;
    for (int b = 0; b < n_slots; ++b) t += inbox[((size_t)b * npad + i) * DIM + d];
    out[d] = pos_in[(size_t)i * DIM + d] + t;
    finite = finite && isfinite(out[d]);
    pos_out[(size_t)i * DIM + d] = out[d];
  }
  for (int q = 0; q < n_push; ++q) {
    float* dst = push[q];
#pragma unroll
    for (int d = 0; d < DIM; ++d) dst[(size_t)i * DIM + d] = out[d];
  }
  if (!finite && st != nullptr) atomicMin(&st->first_nonfinite, iter1);
}

}  // namespace topolow
