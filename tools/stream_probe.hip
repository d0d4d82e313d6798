// Micro-benchmark: how fast can MI355X stream one slab of the encoded target matrix in the stage
// kernel's access shape -- 8 rows per 256-thread workgroup, 2 rows per wave, 1 KB of a row per
// wave-instruction, 2500 of 10048 columns -- as a function of the number of 16-byte loads each lane
// keeps in flight (DEPTH groups x 2 rows) and of the waves per SIMD?  No arithmetic, no LDS: this is
// the memory floor the stage kernel is compared with in DESIGN.md section 6.  A second set of runs
// reads the same bytes from a panel-major layout (256-column panels of n x 1 KB): +11 %.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/stream_probe tools/stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int MINW>
__global__ __launch_bounds__(256, MINW) void probe(const unsigned* __restrict__ m, int ld, int n_rows,
                                                   int cb, int cw, unsigned* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * 8 + wave * 2;
  if (row0 >= n_rows) return;
  __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(m) + (size_t)row0 * ld, 0, ld * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(m) + (size_t)(row0 + 1) * ld, 0, ld * 4, 0x00020000);
  const int ng = (cw + 255) / 256;
  u32x4 w[DEPTH][2];
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int p = 0; p < DEPTH; ++p) {
    const int off = p < ng ? (cb + p * 256 + lane * 4) * 4 : 0x7ffffff0;
    w[p][0] = __builtin_amdgcn_raw_buffer_load_b128(r0, off, 0, 0);
    w[p][1] = __builtin_amdgcn_raw_buffer_load_b128(r1, off, 0, 0);
  }
  for (int g = 0; g < ng; g += DEPTH) {
#pragma unroll
    for (int p = 0; p < DEPTH; ++p) {
      const u32x4 a = w[p][0], b = w[p][1];
      // past the slab's last group: an out-of-range offset (returns 0, moves no bytes)
      const int off = (g + p + DEPTH) < ng ? (cb + (g + p + DEPTH) * 256 + lane * 4) * 4 : 0x7ffffff0;
      w[p][0] = __builtin_amdgcn_raw_buffer_load_b128(r0, off, 0, 0);
      w[p][1] = __builtin_amdgcn_raw_buffer_load_b128(r1, off, 0, 0);
      if (g + p < ng) acc ^= a ^ b;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[blockIdx.x * 256 + threadIdx.x] = acc.x;
}

template <int DEPTH, int MINW>
void run(const unsigned* m, int ld, int n, unsigned* out, int extra_lds) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = (n + 7) / 8;
  const int w = 2500;
  float best = 1e9f, sum = 0;
  const int reps = 40;
  for (int it = 0; it < reps + 5; ++it) {
    const int cb = ((it * 2500) % (n - w)) & ~3;
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<DEPTH, MINW>), dim3(blocks), dim3(256), extra_lds, 0, m, ld, n, cb, w, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it >= 5) { sum += ms; best = ms < best ? ms : best; }
  }
  const double bytes = (double)n * w * 4;
  printf("depth %d  launch_bounds waves %d  extra_lds %6d : mean %.2f us  best %.2f us  -> %.2f TB/s (best %.2f)\n",
         DEPTH, MINW, extra_lds, sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
}


// Same traffic with the matrix stored in 256-column PANELS (panel p = n_rows x 1 KB, contiguous): a
// wave's two rows are 2 KB contiguous and consecutive waves read consecutive memory.
template <int DEPTH, int MINW>
__global__ __launch_bounds__(256, MINW) void probe_panels(const unsigned* __restrict__ m, int n_rows_pad,
                                                          int n_rows, int panel0, int n_panels,
                                                          unsigned* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * 8 + wave * 2;
  if (row0 >= n_rows) return;
  u32x4 w[DEPTH][2];
  u32x4 acc = {0, 0, 0, 0};
  auto addr = [&](int g, int r) {
    const size_t panel = (size_t)(panel0 + g);
    return reinterpret_cast<const u32x4*>(m + panel * (size_t)n_rows_pad * 256 + (size_t)(row0 + r) * 256) + lane;
  };
#pragma unroll
  for (int p = 0; p < DEPTH; ++p) {
    const int g = p < n_panels ? p : 0;
    w[p][0] = *addr(g, 0);
    w[p][1] = *addr(g, 1);
  }
  for (int g = 0; g < n_panels; g += DEPTH) {
#pragma unroll
    for (int p = 0; p < DEPTH; ++p) {
      const u32x4 a = w[p][0], b = w[p][1];
      const int gg = g + p + DEPTH;
      if (gg < n_panels) { w[p][0] = *addr(gg, 0); w[p][1] = *addr(gg, 1); }
      if (g + p < n_panels) acc ^= a ^ b;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[blockIdx.x * 256 + threadIdx.x] = acc.x;
}

template <int DEPTH, int MINW>
void run_panels(const unsigned* m, int n, unsigned* out, int extra_lds) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = (n + 7) / 8, n_pad = (n + 7) & ~7, total_panels = 39, np = 10;
  float best = 1e9f, sum = 0;
  const int reps = 40;
  for (int it = 0; it < reps + 5; ++it) {
    const int p0 = (it * 7) % (total_panels - np);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_panels<DEPTH, MINW>), dim3(blocks), dim3(256), extra_lds, 0, m, n_pad, n, p0, np, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it >= 5) { sum += ms; best = ms < best ? ms : best; }
  }
  const double bytes = (double)n * np * 256 * 4;
  printf("PANELS depth %d  waves %d  extra_lds %6d : mean %.2f us  best %.2f us  -> %.2f TB/s (best %.2f)\n",
         DEPTH, MINW, extra_lds, sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
}

int main() {
  const int n = 10000, ld = 10048;
  unsigned* m;
  unsigned* out;
  hipMalloc(&m, (size_t)n * ld * 4);
  hipMalloc(&out, (size_t)1250 * 256 * 4);
  hipMemset(m, 1, (size_t)n * ld * 4);
  // extra dynamic LDS limits the workgroups per CU: 32 KB -> 5, 40 KB -> 4, 52 KB -> 3
  run<1, 5>(m, ld, n, out, 0);
  run<1, 5>(m, ld, n, out, 32 * 1024);
  run<2, 5>(m, ld, n, out, 32 * 1024);
  run<2, 4>(m, ld, n, out, 40 * 1024);
  run<3, 4>(m, ld, n, out, 40 * 1024);
  run<4, 4>(m, ld, n, out, 40 * 1024);
  run<5, 4>(m, ld, n, out, 40 * 1024);
  run<5, 3>(m, ld, n, out, 52 * 1024);
  run<10, 3>(m, ld, n, out, 52 * 1024);
  run<10, 2>(m, ld, n, out, 0);
  run<1, 8>(m, ld, n, out, 0);
  run<2, 8>(m, ld, n, out, 0);
  run<4, 8>(m, ld, n, out, 0);
  run_panels<1, 5>(m, n, out, 32 * 1024);
  run_panels<2, 5>(m, n, out, 32 * 1024);
  run_panels<3, 4>(m, n, out, 40 * 1024);
  run_panels<5, 4>(m, n, out, 40 * 1024);
  run_panels<2, 8>(m, n, out, 0);
  return 0;
}
