// Micro-benchmark: fp32 VALU issue rate on gfx950 -- scalar v_fma_f32 vs v_pk_fma_f32 vs
// transcendental, at 1/2/4 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        float2v v = {x[i], x[i + 1]};
        float2v aa = {a, a}, bb = {b, b};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(aa), "v"(bb));
        x[i] = v.x; x[i + 1] = v.y;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, double ops_per_inst) {
  int blocks = 256 * waves_per_simd;  // 256-thread blocks: 4 waves = one per SIMD
  float* out;
  hipMalloc(&out, blocks * 256 * 4);
  int iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 16, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double insts_per_wave = (double)iters * (MODE == 1 ? 8 : (MODE == 3 ? 32 : 16));
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-14s waves/SIMD=%d  time %.3f ms  cycles/wave-inst/SIMD @2.4GHz = %.2f  (%.1f Tops/s)\n", name,
         waves_per_simd, ms, cyc / (insts_per_wave * waves_per_simd),
         insts_per_wave * waves_per_simd * 1024 * 64 * ops_per_inst / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", w, 2);
    run<1>("v_pk_fma_f32", w, 4);
    run<2>("v_rcp_f32", w, 1);
    run<3>("cmp+cndmask", w, 1);
  }
  return 0;
}
