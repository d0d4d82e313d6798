// Micro-benchmark: cycles per wave-instruction of the packed fp32 forms the stage kernel uses, by
// operand shape (how many distinct VGPR pairs / SGPR pairs / broadcasts), 1 and 5 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_ops tools/valu_ops.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE>
__global__ void k(float* out, unsigned long long* clk, int iters, float a, float b) {
  f2 x0 = {threadIdx.x * 0.001f, 1.0f}, x1 = {2.f, 3.f}, x2 = {4.f, 5.f}, x3 = {6.f, 7.f};
  f2 aa = {a, a}, bb = {b, b};
  f2 sa = {a, a};  // uniform -> SGPR pair when passed with "s"
  float y0 = threadIdx.x * 0.5f, y1 = 1.f, y2 = 2.f, y3 = 3.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // pk_fma, 3 distinct VGPR pairs
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(aa), "v"(bb));)
    } else if (MODE == 1) {  // pk_fma x = a*a + x  (2 distinct pairs, as in dx*dx+s)
      REP16(asm volatile("v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(aa), "v"(bb));)
    } else if (MODE == 2) {  // pk_mul 2 VGPR pairs
      REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(aa), "v"(bb));)
    } else if (MODE == 3) {  // pk_add VGPR pair + SGPR pair
      REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(sa), "v"(bb));)
    } else if (MODE == 4) {  // pk_add with broadcast of the low half (op_sel_hi 0 on src0)
      REP16(asm volatile("v_pk_add_f32 %0, %4, %0 op_sel_hi:[0,1]\n v_pk_add_f32 %1, %4, %1 op_sel_hi:[0,1]\n v_pk_add_f32 %2, %4, %2 op_sel_hi:[0,1]\n v_pk_add_f32 %3, %4, %3 op_sel_hi:[0,1]"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(aa), "v"(bb));)
    } else if (MODE == 5) {  // plain v_fma, 3 distinct VGPRs
      REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(a), "v"(b));)
    } else if (MODE == 6) {  // plain v_fmac (2 sources + dst)
      REP16(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(a), "v"(b));)
    } else if (MODE == 7) {  // plain v_mul 2 sources
      REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(a), "v"(b));)
    } else if (MODE == 8) {  // cndmask + class
      REP16(asm volatile("v_cmp_class_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_class_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %0, vcc"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(0x1f8), "v"(b) : "vcc");)
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0.x + x1.y + x2.x + x3.y + y0 + y1 + y2 + y3;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

template <int MODE>
void run(const char* name, int wps) {
  const int blocks = 256 * wps;
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
  const int iters = 2000;
  k<MODE><<<blocks, 256>>>(out, clk, 10, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  k<MODE><<<blocks, 256>>>(out, clk, iters, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, clk, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  printf("%-44s waves/SIMD %2d : wave 0 sees %.2f cycles per instruction; kernel %.3f ms = %.2f ns per wave-instruction per SIMD\n",
         name, wps, (double)h[0] / iters / 64.0, ms, ms * 1e6 / ((double)iters * 64.0 * wps));
  hipFree(out); hipFree(clk);
}

int main() {
  for (int w : {1, 2, 4, 8, 16}) {
    run<0>("v_pk_fma_f32  3 VGPR pairs", w);
    run<1>("v_pk_fma_f32  a*a+x (2 pairs)", w);
    run<2>("v_pk_mul_f32  2 VGPR pairs", w);
    run<3>("v_pk_add_f32  VGPR pair + SGPR pair", w);
    run<4>("v_pk_add_f32  broadcast + VGPR pair", w);
    run<5>("v_fma_f32     3 VGPRs", w);
    run<6>("v_fmac_f32    2 VGPRs + dst", w);
    run<7>("v_mul_f32     2 VGPRs", w);
    run<8>("v_cmp_class + v_cndmask (per instruction)", w);
  }
  return 0;
}
