// HBM write bandwidth of the store patterns an epilogue can produce, 256 workgroups x 8 waves each writing a 256 KB tile (256 pixels x 256
// channels fp32, pixel pitch 1 KiB) per iteration to its own region of a 4 GiB buffer:
//   0  global_store_dwordx4, a wave instruction = 64 lanes x 16 B = ONE pixel's 1 KiB                       (sepconv_wide_kernel's epilogue)
//   1  global_store_dword,   a wave instruction = 2 pixels x 128 B (half-wave = 32 consecutive channels)     (W2: straight from the MFMA C layout)
//   2  global_store_dwordx4, a wave instruction = 8 pixels x 128 B (8 lanes x 16 B per pixel)                (pipe kernels; W2 after a quad transpose)
//   3  global_store_dwordx4, a wave instruction = 2 pixels x 512 B
// nontemporal and plain.  hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o _bin/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PAT, bool NT>
__global__ void __launch_bounds__(512) writer(float* __restrict__ dst, int iters, int tiles_per_wg) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f4 v = {1.f, 2.f, 3.f, (float)tid};
  for (int it = 0; it < iters; ++it) {
    float* tile = dst + ((size_t)blockIdx.x * tiles_per_wg + (it % tiles_per_wg)) * 65536;       // 256 px x 256 ch
    // wave w owns pixel rows [32 w, 32 w + 32) x all 256 channels = 32 KB
    float* wbase = tile + (size_t)wave * 32 * 256;
    if constexpr (PAT == 0) {
#pragma unroll 8
      for (int p = 0; p < 32; ++p) {
        f4* q = reinterpret_cast<f4*>(wbase + p * 256 + lane * 4);
        if (NT) __builtin_nontemporal_store(v, q); else *q = v;
      }
    } else if constexpr (PAT == 1) {
      // 128 instructions: (pixel pair pp, channel block cb of 32): lane -> pixel 2 pp + (lane >> 5), channel 32 cb + (lane & 31)
#pragma unroll 8
      for (int i = 0; i < 128; ++i) {
        const int pp = i >> 3, cb = i & 7;
        float* q = wbase + (2 * pp + (lane >> 5)) * 256 + cb * 32 + (lane & 31);
        if (NT) __builtin_nontemporal_store(v.x, q); else *q = v.x;
      }
    } else if constexpr (PAT == 2) {
      // 32 instructions: (pixel group of 8, channel block of 32): lane -> pixel 8 g + (lane >> 3), channels 32 cb + 4 (lane & 7)
#pragma unroll 8
      for (int i = 0; i < 32; ++i) {
        const int g = i >> 3, cb = i & 7;
        f4* q = reinterpret_cast<f4*>(wbase + (8 * g + (lane >> 3)) * 256 + cb * 32 + (lane & 7) * 4);
        if (NT) __builtin_nontemporal_store(v, q); else *q = v;
      }
    } else {
      // 32 instructions: (pixel pair, half row of 128 channels): lane -> pixel 2 pp + (lane >> 5), channels 128 h + 4 (lane & 31)
#pragma unroll 8
      for (int i = 0; i < 32; ++i) {
        const int pp = i >> 1, h = i & 1;
        f4* q = reinterpret_cast<f4*>(wbase + (2 * pp + (lane >> 5)) * 256 + h * 128 + (lane & 31) * 4);
        if (NT) __builtin_nontemporal_store(v, q); else *q = v;
      }
    }
  }
}

template <int PAT, bool NT>
void run(float* dst, const char* what) {
  const int grid = 256, iters = 64, tiles = 16;            // 256 x 16 x 256 KB = 1 GiB region, each tile rewritten 4 times
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((writer<PAT, NT>), dim3(grid), dim3(512), 0, 0, dst, 8, tiles);
  CHECK(hipEventRecord(a, 0));
  hipLaunchKernelGGL((writer<PAT, NT>), dim3(grid), dim3(512), 0, 0, dst, iters, tiles);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  printf("%-62s %-4s %8.0f GB/s   %6.2f us per 256 KB tile and CU\n", what, NT ? "nt" : "", (double)grid * iters * 262144.0 / (ms * 1e-3) / 1e9, ms * 1e3 / iters);
}

int main() {
  float* dst;
  CHECK(hipMalloc(&dst, (size_t)1 << 30));
  run<0, true>(dst, "dwordx4: 1 pixel x 1 KiB per wave instruction");
  run<0, false>(dst, "dwordx4: 1 pixel x 1 KiB per wave instruction");
  run<1, true>(dst, "dword:   2 pixels x 128 B per wave instruction");
  run<1, false>(dst, "dword:   2 pixels x 128 B per wave instruction");
  run<2, true>(dst, "dwordx4: 8 pixels x 128 B per wave instruction");
  run<2, false>(dst, "dwordx4: 8 pixels x 128 B per wave instruction");
  run<3, true>(dst, "dwordx4: 2 pixels x 512 B per wave instruction");
  run<3, false>(dst, "dwordx4: 2 pixels x 512 B per wave instruction");
  return 0;
}
