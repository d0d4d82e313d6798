// Developer aid: which lane a DPP operand reads from on gfx950 (row_shl / row_shr / row_ror / quad_perm), behind
// csrc/relax_symm.h: sym_col_reduce.  Build: hipcc -O2 --offload-arch=gfx950 -o tools/dpp_probe tools/dpp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  float v = (float)lane, z = 1000.0f, a, b, c, d;
  a = z; b = z; c = z; d = z;
  asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %0 row_shl:4 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(v));
  asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(v));
  asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(v));
  asm volatile("s_nop 1\n v_add_f32_dpp %0, %1, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(v));
  out[lane] = a; out[64 + lane] = b; out[128 + lane] = c; out[192 + lane] = d;
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[4] = {"row_shl:4", "row_shr:4", "quad[1,0,3,2]", "row_ror:4"};
  for (int q = 0; q < 4; ++q) { printf("%s:", nm[q]); for (int i = 0; i < 20; ++i) printf(" %g", h[q * 64 + i] - 1000); printf("\n"); }
  return 0;
}
