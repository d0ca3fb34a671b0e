// Issue cost (cycles per wave64 instruction, one SIMD) of the VALU forms the streaming kernel is made of, on gfx950:
// plain v_fmac_f32, v_fmac_f32 with a DPP source (wave_shr:1, row_shr:1), v_med3_f32, v_cvt_pk_f16_f32, v_cvt_f32_f16 (SDWA),
// v_pk_fma_f32, ds_read_b64 / b128 of a wave-uniform address.  One wave per SIMD and two.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rates.hip -o scripts/ubench/_bin/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(256, 8) k(float* out, long long* cyc, int iters) {
  __shared__ float lds[1024];
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, x = 1.0001f, w = 0.999f;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const float* lp = lds + (threadIdx.x >> 5) * 4;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      REP16(asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                   "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));)
    } else if (KIND == 1) {
#define D " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      REP16(asm volatile("v_fmac_f32_dpp %0, %8, %9" D "v_fmac_f32_dpp %1, %8, %9" D "v_fmac_f32_dpp %2, %8, %9" D "v_fmac_f32_dpp %3, %8, %9" D
                   "v_fmac_f32_dpp %4, %8, %9" D "v_fmac_f32_dpp %5, %8, %9" D "v_fmac_f32_dpp %6, %8, %9" D "v_fmac_f32_dpp %7, %8, %9" D
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));)
#undef D
    } else if (KIND == 2) {
#define D " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      REP16(asm volatile("v_fmac_f32_dpp %0, %8, %9" D "v_fmac_f32_dpp %1, %8, %9" D "v_fmac_f32_dpp %2, %8, %9" D "v_fmac_f32_dpp %3, %8, %9" D
                   "v_fmac_f32_dpp %4, %8, %9" D "v_fmac_f32_dpp %5, %8, %9" D "v_fmac_f32_dpp %6, %8, %9" D "v_fmac_f32_dpp %7, %8, %9" D
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));)
#undef D
    } else if (KIND == 3) {
      REP16(asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                   "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));)
    } else if (KIND == 4) {
      REP16(asm volatile("v_cvt_pk_f16_f32 %0, %8, %9\n v_cvt_pk_f16_f32 %1, %8, %9\n v_cvt_pk_f16_f32 %2, %8, %9\n v_cvt_pk_f16_f32 %3, %8, %9\n"
                   "v_cvt_pk_f16_f32 %4, %8, %9\n v_cvt_pk_f16_f32 %5, %8, %9\n v_cvt_pk_f16_f32 %6, %8, %9\n v_cvt_pk_f16_f32 %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));)
    } else if (KIND == 5) {
      REP16(asm volatile("v_cvt_f32_f16_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                   "v_cvt_f32_f16_sdwa %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                   "v_cvt_f32_f16_sdwa %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                   "v_cvt_f32_f16_sdwa %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));)
    } else if (KIND == 6) {
      float2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, xx = {x, x}, ww = {w, w};
      REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                   "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(xx), "v"(ww));)
      a0 = p0.x + p1.y; a1 = p2.x + p3.y;
    } else if (KIND == 7 || KIND == 8) {
      float4 r0, r1, r2, r3, r4, r5, r6, r7;
      if (KIND == 7) {
        REP16(asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                     "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"((unsigned)(size_t)lp) : "memory");)
      } else {
        float2 q0, q1, q2, q3, q4, q5, q6, q7;
        REP16(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:16\n ds_read_b64 %2, %8 offset:32\n ds_read_b64 %3, %8 offset:48\n"
                     "ds_read_b64 %4, %8 offset:64\n ds_read_b64 %5, %8 offset:80\n ds_read_b64 %6, %8 offset:96\n ds_read_b64 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7) : "v"((unsigned)(size_t)lp) : "memory");)
        r0.x = q0.x + q7.y;
      }
      a0 += r0.x;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, int wpb, int per_iter) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 4096 * 512 * 4); (void)hipMalloc(&cyc, 4096 * 8);
  for (int blocks : {256, 512, 768, 1024, 1536, 2048}) {      // 256 workgroups of 4 waves = 1 wave per SIMD; 512 = 2 per SIMD ...
    const int iters = 200;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    static long long h[4096]; (void)hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
    printf("%-28s %d waves/SIMD: %.2f cycles per instruction per wave (wall cycles of a wave / its instructions)\n", name, blocks / 256, s / blocks / iters / per_iter);
  }
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>("v_fmac_f32_e32", 4, 128);
  run<1>("v_fmac_f32_dpp wave_shr:1", 4, 128);
  run<2>("v_fmac_f32_dpp row_shr:1", 4, 128);
  run<3>("v_med3_f32", 4, 128);
  run<4>("v_cvt_pk_f16_f32", 4, 128);
  run<5>("v_cvt_f32_f16_sdwa", 4, 128);
  run<6>("v_pk_fma_f32", 4, 128);
  run<7>("ds_read_b128 (2 addresses)", 4, 128);
  run<8>("ds_read_b64 (2 addresses)", 4, 128);
  return 0;
}
