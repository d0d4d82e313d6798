// Micro-benchmark: issue cost on gfx950 of the stage kernel's instruction classes in MIXED streams --
// do transcendentals (v_sqrt/v_rcp) overlap with packed / plain fp32 VALU work of the same wave or
// of other waves, and what does a v_pk_fma_f32 cost next to them?  In-kernel clock measured with
// s_memtime / s_memrealtime, so the cycles are real cycles.
// Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_mix tools/valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE: 0 = NPK v_pk_fma per step; 1 = NPK v_pk_fma + NT transcendental; 2 = NS v_fma + NT trans;
//       3 = NT trans only; 4 = NS v_fma only
template <int MODE, int NPK, int NS, int NT>
__global__ void k(float* out, unsigned long long* clk, int iters, float a, float b) {
  f2 x[8]; float y[8]; float z[4];
  for (int i = 0; i < 8; ++i) { x[i] = (f2){threadIdx.x * 0.001f + i, 1.0f + i}; y[i] = 0.5f + i + threadIdx.x * 0.01f; }
  for (int i = 0; i < 4; ++i) z[i] = 1.5f + i;
  f2 aa = {a, a}, bb = {b, b};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 1) {
#pragma unroll
      for (int i = 0; i < NPK; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i % 8]) : "v"(aa), "v"(bb));
    }
    if (MODE == 2 || MODE == 4) {
#pragma unroll
      for (int i = 0; i < NS; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[i % 8]) : "v"(a), "v"(b));
    }
    if (MODE == 1 || MODE == 2 || MODE == 3) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        if (i & 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(z[i % 4]));
        else asm volatile("v_sqrt_f32 %0, %0" : "+v"(z[i % 4]));
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y + y[i];
  for (int i = 0; i < 4; ++i) s += z[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int MODE, int NPK, int NS, int NT>
void run(const char* name, int wps) {
  const int blocks = 256 * wps;
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
  const int iters = 20000;
  k<MODE, NPK, NS, NT><<<blocks, 256>>>(out, clk, 100, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  k<MODE, NPK, NS, NT><<<blocks, 256>>>(out, clk, iters, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  unsigned long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
  printf("%-34s waves/SIMD %d : %.1f cycles per step per wave-slot (clock %.2f GHz)\n", name, wps,
         (double)h[0] / iters / wps, ghz);
  hipFree(out); hipFree(clk);
}

int main() {
  for (int w : {1, 2, 5}) {
    run<0, 22, 0, 0>("22 pk_fma", w);
    run<4, 0, 22, 0>("22 v_fma", w);
    run<3, 0, 0, 4>("4 trans", w);
    run<1, 22, 0, 4>("22 pk_fma + 4 trans", w);
    run<2, 0, 22, 4>("22 v_fma + 4 trans", w);
    run<2, 0, 44, 4>("44 v_fma + 4 trans", w);
  }
  return 0;
}
