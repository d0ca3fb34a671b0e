// Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs on gfx950, and does v_cvt_pk_f16_f32 produce them?
// (The f16x2 split GEMM relies on both when a lo piece falls below 2^-14: migan_kernels.hpp split2_f16.)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_subnormal.hip -o scripts/ubench/_bin/mfma_subnormal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float tiny_f32) {
  // A = tiny (an fp16 subnormal after conversion), B = 1: D = 16 * tiny unless the matrix core flushes subnormal inputs
  const h2 c = __builtin_convertvector(f2{tiny_f32, tiny_f32}, h2);   // v_cvt_pk_f16_f32
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = c.x; b[i] = (_Float16)1.0f; }
  f16v acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  f16v acc2 = {0};
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc2, 0, 0, 0);   // subnormal on the B side
  // subnormal x subnormal-scale: (2^-20)(as fp16) x 2^-10
  h8 s; for (int i = 0; i < 8; ++i) s[i] = (_Float16)0.0009765625f;
  f16v acc3 = {0};
  acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, s, acc3, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = acc[0]; out[1] = acc2[0]; out[2] = acc3[0];
    out[3] = (float)c.x;                                              // v_cvt_f32_f16 of the subnormal
    const unsigned bits = __builtin_bit_cast(unsigned, c) & 0xffffu; out[4] = (float)bits;
  }
}
int main() {
  float* d; (void)hipMalloc(&d, 64);
  for (int e : {-15, -20, -24}) {
    const float tiny = std::ldexp(1.0f, e);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, tiny);
    float h[5]; (void)hipMemcpy(h, d, 20, hipMemcpyDeviceToHost);
    printf("tiny=2^%d: fp16 bits 0x%04x back-converted %g | mfma(A=tiny,B=1)=%g (want %g) mfma(A=1,B=tiny)=%g  mfma(tiny, 2^-10)=%g (want %g)\n", e, (unsigned)h[4], h[3],
           h[0], 16.0 * tiny, h[1], h[2], 16.0 * tiny / 1024);
  }
  return 0;
}
