// Probe of the symmetric sweep (topolow_amd/csrc/relax_symm.h) on synthetic data of config 3's shape:
// checks one sweep + apply against a plain row-owner evaluation of the same update and times them.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I topolow_amd/csrc -o tools/symm_probe tools/symm_probe.hip
// Run:   tools/symm_probe [n=10000] [workgroups per CU, 0 = occupancy] [reps=50]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>
#include "relax_symm.h"

using namespace topolow;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int DIM>
__global__ void ref_kernel(const uint32_t* enc, int ld, const float* pos, const float* gplus, int n, double k, double c_rep,
                           float* out, double* err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double acc[DIM] = {};
  double es = 0;
  const float g = gplus[i];
  const float ks = (float)(2.0 * k) / (4.0f * g + (float)k), cg = (float)(0.5 * c_rep) / g;
  for (int c = 0; c < n; ++c) {
    float dx[DIM], s = 0;
    for (int d = 0; d < DIM; ++d) { dx[d] = pos[(size_t)c * DIM + d] - pos[(size_t)i * DIM + d]; s = fmaf(dx[d], dx[d], s); }
    const float r = sqrtf(s), inv = 1.0f / (r + 0.01f);
    const uint32_t w = enc[(size_t)i * ld + c];
    const bool sp = (w & 0x7f800000u) != 0x7f800000u;
    const float t = bits_f32(w);
    const float coef = sp ? (t - r) * inv * ks : inv * inv * inv * cg;
    for (int d = 0; d < DIM; ++d) acc[d] += (double)dx[d] * coef;
    if (sp && c > i) es += fabs((double)t - r);
  }
  for (int d = 0; d < DIM; ++d) out[(size_t)i * DIM + d] = (float)((double)pos[(size_t)i * DIM + d] - acc[d]);
  err[i] = es;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  constexpr int DIM = 5;
  constexpr int W = SymRec<DIM>::W;
  const int n = argc > 1 ? atoi(argv[1]) : 10000;
  const int KT = argc > 2 ? atoi(argv[2]) : 0;
  const int reps = argc > 3 ? atoi(argv[3]) : 50;
  const int npad = (n + 63) & ~63, ld = (n + 63) & ~63, TR = npad / kSymRows, TC = npad / kSymCols;
  const double k = 2.3, c_rep = 0.01;
  std::mt19937_64 rng(1);
  std::normal_distribution<float> nd(0.f, 3.f);
  std::uniform_real_distribution<float> ud(0.f, 1.f);
  std::vector<float> pos((size_t)n * DIM), g(n);
  for (auto& v : pos) v = nd(rng);
  std::vector<uint32_t> enc((size_t)npad * ld, kInfWord);
  std::vector<int> deg(n, 0);
  for (int i = 0; i < n; ++i)
    for (int c = i + 1; c < n; ++c)
      if (ud(rng) < 0.3f) {
        float s = 0;
        for (int d = 0; d < DIM; ++d) { const float q = pos[(size_t)i * DIM + d] - pos[(size_t)c * DIM + d]; s += q * q; }
        const uint32_t w = encode_target(std::sqrt(s) * (1.0 + 0.05 * nd(rng) / 3.0), 0);
        enc[(size_t)i * ld + c] = w;
        enc[(size_t)c * ld + i] = w;
        ++deg[i]; ++deg[c];
      }
  for (int i = 0; i < n; ++i) g[i] = (float)(deg[i] + 2);
  for (auto& v : pos) v += 0.3f * nd(rng) / 3.0f;   // perturb so that the springs are loaded

  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, symm_sweep_kernel<DIM, false, false>, 64 * kSymWaves, 0));
  const int grid = (KT > 0 ? KT : occ) * cus;
  const SymPlan plan = relax_symm_plan(npad, grid * kSymWaves);
  const std::vector<SymUnit>& units = plan.units;
  const std::vector<int2>& row_units = plan.row_units;
  const int n_units = (int)units.size();
  printf("n %d  tile-rows %d  occupancy %d WG/CU  grid %d  units %d  colpart %.1f MB rowpart %.1f MB\n", n, TR, occ, grid, n_units,
         (double)TR * npad * DIM * 4 / 1e6, (double)n_units * kSymRows * DIM * 4 / 1e6);

  // tile-major copy of the upper triangle
  std::vector<uint32_t> tenc((size_t)TR * (TR + 1) * kSymTileWords, kInfWord);
  for (int R = 0; R < TR; ++R)
    for (int J = 2 * R; J < TC; ++J) {
      uint32_t* t = tenc.data() + (size_t)sym_tile_index(R, J, TC) * kSymTileWords;
      for (int r = 0; r < kSymRows; ++r)
        for (int c = 0; c < kSymCols; ++c) t[sym_word_in_tile(r, c)] = enc[(size_t)(R * kSymRows + r) * ld + J * kSymCols + c];
    }
  uint32_t* d_tenc;
  CK(hipMalloc(&d_tenc, tenc.size() * 4)); CK(hipMemcpy(d_tenc, tenc.data(), tenc.size() * 4, hipMemcpyHostToDevice));
  uint32_t* d_enc; float *d_pos, *d_g, *d_rec, *d_rec2, *d_out, *d_ref, *d_rowp, *d_colp; double *d_err, *d_psum; unsigned long long* d_pcnt;
  SymUnit* d_units; int2* d_ru; SymRun* d_wf;
  CK(hipMalloc(&d_enc, enc.size() * 4)); CK(hipMemcpy(d_enc, enc.data(), enc.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_pos, pos.size() * 4)); CK(hipMemcpy(d_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_g, n * 4)); CK(hipMemcpy(d_g, g.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_rec, (size_t)npad * W * 4)); CK(hipMalloc(&d_rec2, (size_t)npad * W * 4));
  CK(hipMalloc(&d_out, pos.size() * 4)); CK(hipMalloc(&d_ref, pos.size() * 4));
  CK(hipMalloc(&d_rowp, (size_t)n_units * kSymRows * DIM * 4)); CK(hipMalloc(&d_colp, (size_t)TR * npad * DIM * 4));
  CK(hipMalloc(&d_err, n * 8)); CK(hipMalloc(&d_psum, n_units * 8)); CK(hipMalloc(&d_pcnt, n_units * 8));
  CK(hipMalloc(&d_units, n_units * sizeof(SymUnit))); CK(hipMemcpy(d_units, units.data(), n_units * sizeof(SymUnit), hipMemcpyHostToDevice));
  CK(hipMalloc(&d_ru, TR * sizeof(int2))); CK(hipMemcpy(d_ru, row_units.data(), TR * sizeof(int2), hipMemcpyHostToDevice));
  const std::vector<SymRun> runs = plan.runs();
  CK(hipMalloc(&d_wf, runs.size() * sizeof(SymRun))); CK(hipMemcpy(d_wf, runs.data(), runs.size() * sizeof(SymRun), hipMemcpyHostToDevice));

  hipLaunchKernelGGL(symm_records_kernel<DIM>, dim3((npad + 255) / 256), dim3(256), 0, 0, d_pos, d_g, d_rec, n, npad, k, c_rep);
  hipLaunchKernelGGL(symm_records_kernel<DIM>, dim3((npad + 255) / 256), dim3(256), 0, 0, d_pos, d_g, d_rec2, n, npad, k, c_rep);
  CK(hipDeviceSynchronize()); printf("records done\n");
  hipLaunchKernelGGL(ref_kernel<DIM>, dim3((n + 63) / 64), dim3(64), 0, 0, d_enc, ld, d_pos, d_g, n, k, c_rep, d_ref, d_err);
  CK(hipDeviceSynchronize()); printf("ref done\n");
  hipLaunchKernelGGL((symm_sweep_kernel<DIM, false, true>), dim3(grid), dim3(64 * kSymWaves), 0, 0, d_tenc, d_rec, d_units,
                     d_wf, d_rowp, d_colp, npad, (const RunState*)nullptr, d_psum, d_pcnt, 0ull, 0);
  CK(hipDeviceSynchronize()); printf("sweep done\n");
  hipLaunchKernelGGL(symm_apply_kernel<DIM>, dim3(TC), dim3(32 * kSymApplyParts), 0, 0, d_rec, d_rec2, d_out, d_g, d_rowp, d_colp, d_ru, n, npad,
                     k * 0.99, c_rep, 1, (RunState*)nullptr);
  CK(hipDeviceSynchronize()); printf("apply done\n");
  std::vector<float> out(pos.size()), ref(pos.size());
  std::vector<double> err(n), psum(n_units);
  CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(ref.data(), d_ref, ref.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(err.data(), d_err, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(psum.data(), d_psum, n_units * 8, hipMemcpyDeviceToHost));
  double maxd = 0, move = 0, e_ref = 0, e_sym = 0;
  for (size_t q = 0; q < out.size(); ++q) {
    maxd = std::max(maxd, (double)std::fabs(out[q] - ref[q]));
    move = std::max(move, (double)std::fabs(ref[q] - pos[q]));
  }
  for (double v : err) e_ref += v;
  for (double v : psum) e_sym += 0.5 * v;   // (the ERR launch leaves twice the sum)
  printf("max |sym - ref| %.3g (largest move %.3g)   err sum ref %.9g sym %.9g\n", maxd, move, e_ref, e_sym);

  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto&& fn, const char* what) {
    for (int q = 0; q < 5; ++q) fn();
    CK(hipEventRecord(e0));
    for (int q = 0; q < reps; ++q) fn();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %8.2f us\n", what, 1e3 * ms / reps);
  };
  auto sweep = [&](auto err_tag) {
    hipLaunchKernelGGL((symm_sweep_kernel<DIM, false, decltype(err_tag)::value>), dim3(grid), dim3(64 * kSymWaves), 0, 0, d_tenc, d_rec,
                       d_units, d_wf, d_rowp, d_colp, npad, (const RunState*)nullptr, d_psum, d_pcnt, 0ull, 0);
  };
  auto apply = [&]() {
    hipLaunchKernelGGL(symm_apply_kernel<DIM>, dim3(TC), dim3(32 * kSymApplyParts), 0, 0, d_rec, d_rec2, d_out, d_g, d_rowp, d_colp, d_ru, n, npad,
                       k * 0.99, c_rep, 1, (RunState*)nullptr);
  };
  time([&]() { sweep(std::false_type{}); }, "sweep");
  time([&]() { apply(); }, "apply");
  time([&]() { sweep(std::false_type{}); apply(); }, "sweep + apply");
  time([&]() { sweep(std::true_type{}); apply(); }, "sweep<ERR> + apply");
#ifdef TOPOLOW_SYM_STAMPS
  {   // in-kernel clock and wave timeline of the sweep in steady state: ~1.5 s of back-to-back sweep + apply, the last sweep's stamps
    const int n_waves = grid * kSymWaves;
    unsigned long long* d_st; CK(hipMalloc(&d_st, (size_t)n_waves * 32)); CK(hipMemset(d_st, 0, (size_t)n_waves * 32));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_sym_stamps), &d_st, sizeof d_st));
    for (int q = 0; q < 25000; ++q) { sweep(std::false_type{}); apply(); }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> stv((size_t)n_waves * 4);
    CK(hipMemcpy(stv.data(), d_st, stv.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk, life, t0s, t1s;
    unsigned long long first = ~0ull;
    for (int w = 0; w < n_waves; ++w) if (stv[4 * w + 2] > 0) first = std::min(first, stv[4 * w + 1]);
    std::vector<int> tiles_of(n_waves, 0), units_of(n_waves, 0);
    for (int w = 0; w < n_waves; ++w) {
      units_of[w] = plan.wave_first[w + 1] - plan.wave_first[w];
      for (int u = plan.wave_first[w]; u < plan.wave_first[w + 1]; ++u) tiles_of[w] += units[u].j1 - units[u].j0;
    }
    std::map<unsigned long long, std::vector<int>> by_simd;
    for (int w = 0; w < n_waves; ++w) if (stv[4 * w + 2] > 0) {
      const double dt = (double)(stv[4 * w + 2] - stv[4 * w + 1]);
      clk.push_back(100.0 * stv[4 * w] / dt); life.push_back(dt / 100.0);
      t0s.push_back((stv[4 * w + 1] - first) / 100.0); t1s.push_back((stv[4 * w + 2] - first) / 100.0);
      const unsigned long long id = stv[4 * w + 3];
      const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 0xf;
      // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
      by_simd[((unsigned long long)xcc << 16) | (hw & 0xff30)].push_back(w);
    }
    auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    printf("in-kernel clock (last sweep of 25000 back-to-back iterations): median %.0f MHz [%.0f, %.0f]\n", pct(clk, .5), pct(clk, 0), pct(clk, 1));
    printf("wave start us: p50 %.2f p99 %.2f max %.2f | end us: p1 %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | lifetime p50 %.1f max %.1f\n",
           pct(t0s, .5), pct(t0s, .99), pct(t0s, 1), pct(t1s, .01), pct(t1s, .1), pct(t1s, .5), pct(t1s, .9), pct(t1s, .99), pct(t1s, 1), pct(life, .5), pct(life, 1));
    std::map<int, int> hist; double worst = 0; int worst_tiles = 0, worst_n = 0;
    std::map<int, std::vector<double>> end_by_load;
    for (auto& kv : by_simd) {
      hist[(int)kv.second.size()]++;
      int tl = 0; double e = 0;
      for (int w : kv.second) { tl += tiles_of[w]; e = std::max(e, (stv[4 * w + 2] - first) / 100.0); }
      end_by_load[tl].push_back(e);
      if (e > worst) { worst = e; worst_tiles = tl; worst_n = (int)kv.second.size(); }
    }
    printf("SIMDs seen %zu; waves per SIMD:", by_simd.size());
    for (auto& h : hist) printf("  %d x%d", h.first, h.second);
    printf("\nlast SIMD ends at %.1f us with %d waves / %d tiles; end time by tiles per SIMD:\n", worst, worst_n, worst_tiles);
    for (auto& kv : end_by_load) printf("   %3d tiles: %4zu SIMDs, end p50 %.1f max %.1f us\n", kv.first, kv.second.size(), pct(kv.second, .5), pct(kv.second, 1));
    std::map<int, std::vector<double>> end_by_units;      // does a SIMD whose waves hold more units (runs that cross tile-rows) end later?
    for (auto& kv : by_simd) {
      int un = 0; double e = 0;
      for (int w : kv.second) { un += units_of[w]; e = std::max(e, (stv[4 * w + 2] - first) / 100.0); }
      end_by_units[un].push_back(e);
    }
    for (auto& kv : end_by_units) printf("   %3d units: %4zu SIMDs, end p50 %.1f max %.1f us\n", kv.first, kv.second.size(), pct(kv.second, .5), pct(kv.second, 1));
    {   // ... and by XCD
      std::map<int, std::vector<double>> end_by_xcc;
      for (auto& kv : by_simd) {
        double e = 0;
        for (int w : kv.second) e = std::max(e, (stv[4 * w + 2] - first) / 100.0);
        end_by_xcc[(int)(kv.first >> 16)].push_back(e);
      }
      for (auto& kv : end_by_xcc) printf("   XCD %d: %4zu SIMDs, end p50 %.1f max %.1f us\n", kv.first, kv.second.size(), pct(kv.second, .5), pct(kv.second, 1));
    }
    time([&]() { sweep(std::false_type{}); apply(); }, "sweep + apply (steady state)");
    time([&]() { sweep(std::false_type{}); }, "sweep (steady state)");
  }
#endif
  return 0;
}
