// Microbenchmark: does VALU work overlap with fp32 MFMA (v_mfma_f32_32x32x2_f32) on gfx950?
//   role 0: MFMA only, role 1: packed-f32 VALU only, role 2: LDS reads only.
// A workgroup has 256 or 512 threads; the role of a wave is picked from its wave id so that on each
// SIMD (waves are placed round-robin over the 4 SIMDs) roles can be mixed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(512, 2) k(float* out, int iters, int role_lo, int role_hi, int mfma_per_iter, int valu_per_iter) {
  __shared__ float lds[8192];
  const int wave = threadIdx.x >> 6;
  const int role = (wave < 4) ? role_lo : role_hi;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 0.001f;
  __syncthreads();
  float r = 0.f;
  if (role == 0) {
    f16v a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 0.01f, y = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 4
      for (int j = 0; j < mfma_per_iter; j += 4) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
      }
    }
    for (int q = 0; q < 16; ++q) r += a0[q] + a1[q] + a2[q] + a3[q];
  } else if (role == 3) {
    // bf16 MFMA 32x32x16 (8 bf16 per lane for A and B)
    f16v a0 = {}, a1 = {}, a2 = {}, a3 = {};
    bf16x8 x, y;
    for (int q = 0; q < 8; ++q) { x[q] = (__bf16)(threadIdx.x * 0.01f + q); y[q] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll 4
      for (int j = 0; j < mfma_per_iter; j += 4) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
      }
    }
    for (int q = 0; q < 16; ++q) r += a0[q] + a1[q] + a2[q] + a3[q];
  } else if (role == 1) {
    f4 v0 = {1.f, 2.f, 3.f, 4.f}, v1 = v0 * 0.5f, v2 = v0 * 0.25f, v3 = v0 * 0.125f;
    const f4 m = {1.0001f, 0.9999f, 1.0002f, 0.9998f}, c = {1e-6f, 2e-6f, 3e-6f, 4e-6f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll 4
      for (int j = 0; j < valu_per_iter; j += 4) {
        v0 = v0 * m + c; v1 = v1 * m + c; v2 = v2 * m + c; v3 = v3 * m + c;
      }
    }
    r = v0.x + v1.y + v2.z + v3.w;
  } else if (role == 2) {
    f4 s = {0.f, 0.f, 0.f, 0.f};
    const float* p = lds + (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
      for (int j = 0; j < valu_per_iter; ++j) s += *(const f4*)(p + ((j * 256) & 4095));
    }
    r = s.x + s.y + s.z + s.w;
  }
  if (r == 12345.678f) out[0] = r;
}

static float run(int threads, int role_lo, int role_hi, int iters, int mpi, int vpi) {
  float* out; hipMalloc(&out, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 256;  // one workgroup per CU
  k<<<blocks, threads>>>(out, 10, role_lo, role_hi, mpi, vpi);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    k<<<blocks, threads>>>(out, iters, role_lo, role_hi, mpi, vpi);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  hipFree(out);
  return best;
}

int main() {
  const int iters = 2000, mpi = 64, vpi = 256;   // per wave per iter: 64 MFMA (4096 cyc) ; 256 f4-FMAs
  const double mfma_flop_wave = 2.0 * 32 * 32 * 2 * mpi * iters, valu_flop_wave = 2.0 * 4 * 64 * vpi * iters;
  struct T { const char* name; int threads, lo, hi; } tests[] = {
      {"MFMA only, 1 wave/SIMD", 256, 0, 0}, {"MFMA only, 2 waves/SIMD", 512, 0, 0},
      {"VALU only, 1 wave/SIMD", 256, 1, 1}, {"VALU only, 2 waves/SIMD", 512, 1, 1},
      {"MFMA wave + VALU wave per SIMD", 512, 0, 1}, {"LDS only, 1 wave/SIMD", 256, 2, 2},
      {"MFMA wave + LDS wave per SIMD", 512, 0, 2}, {"VALU wave + LDS wave per SIMD", 512, 1, 2},
      {"bf16 MFMA only, 1 wave/SIMD", 256, 3, 3}, {"bf16 MFMA only, 2 waves/SIMD", 512, 3, 3},
      {"bf16 MFMA wave + VALU wave per SIMD", 512, 3, 1}, {"bf16 MFMA wave + fp32 MFMA wave", 512, 3, 0},
  };
  for (auto& t : tests) {
    float ms = run(t.threads, t.lo, t.hi, iters, mpi, vpi);
    int nw_lo = 4, nw_hi = t.threads == 512 ? 4 : 0;
    double mf = 0, vf = 0, lb = 0;
    auto add = [&](int role, int nw) { if (role == 3) mf += nw * mfma_flop_wave * 8; if (role == 0) mf += nw * mfma_flop_wave; if (role == 1) vf += nw * valu_flop_wave; if (role == 2) lb += nw * 1024.0 * vpi * iters; };
    add(t.lo, nw_lo); add(t.hi, nw_hi);
    printf("%-34s %8.3f ms  MFMA %7.1f TF/s  VALU %7.1f TF/s  LDS %7.1f TB/s\n", t.name, ms, 256 * mf / ms / 1e9, 256 * vf / ms / 1e9, 256 * lb / ms / 1e9);
  }
  return 0;
}
