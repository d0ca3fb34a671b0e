// Per-CU load throughput of the paths a workgroup can use to bring L2-resident data (the fp16 weight planes) and streamed data (activation
// tiles) into a CU on MI355X: LDS-DMA (buffer_load_dwordx4 ... lds), global_load_dwordx4 -> VGPR (-> ds_write_b128), with 1, 2, 4 or 8
// waves per CU issuing.  One workgroup per CU (LDS-limited), every workgroup reads the SAME `span` bytes over and over (span = 1 MiB:
// L2-resident after the first pass) or its own slice of a large buffer (streamed from HBM).
//   hipcc --offload-arch=gfx950 -O3 cu_load_paths.hip -o _bin/cu_load_paths && _bin/cu_load_paths
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: LDS-DMA; 1: global_load -> VGPR, summed (no LDS); 2: global_load -> VGPR -> ds_write_b128
template <int MODE, int UNROLL>
__global__ void __launch_bounds__(512) loader(const char* __restrict__ src, size_t span, size_t per_wg, int iters, int waves, float* sink, int shared) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave >= waves) return;
  const char* base = shared ? src : src + (size_t)blockIdx.x * per_wg;
  const size_t mine = shared ? span : per_wg;
  const unsigned step = (unsigned)waves * 1024u * UNROLL;                 // bytes per iteration of the workgroup
  __amdgpu_buffer_rsrc_t buf = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)mine, 0x00020000);
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned off = (unsigned)wave * 1024u * UNROLL + (unsigned)lane * 16u;
  float* dst = smem + wave * (256 * UNROLL);
  for (int it = 0; it < iters; ++it) {
    if (off + 1024u * UNROLL > (unsigned)mine) off = (unsigned)wave * 1024u * UNROLL + (unsigned)lane * 16u;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(buf, (__attribute__((address_space(3))) void*)(dst + u * 256), 16, (int)(off + u * 1024u), 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UNROLL) : "memory");      // one batch stays in flight
    } else {
      f4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = *reinterpret_cast<const f4*>(base + off + u * 1024u);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if constexpr (MODE == 1) acc += v[u];
        else *reinterpret_cast<f4*>(dst + u * 256 + lane * 4) = v[u];
      }
    }
    off += step;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE == 2) acc = *reinterpret_cast<f4*>(dst + lane * 4);
  if (acc.x == 123.456f) sink[tid] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE, int UNROLL>
double run(const char* src, size_t span, size_t per_wg, int waves, int shared, int grid, float* sink) {
  const int iters = 4000;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const size_t lds = 150 * 1024;                                            // one workgroup per CU
  CHECK(hipFuncSetAttribute((const void*)loader<MODE, UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((loader<MODE, UNROLL>), dim3(grid), dim3(512), lds, 0, src, span, per_wg, 200, waves, sink, shared);
  CHECK(hipEventRecord(a, 0));
  hipLaunchKernelGGL((loader<MODE, UNROLL>), dim3(grid), dim3(512), lds, 0, src, span, per_wg, iters, waves, sink, shared);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)grid * iters * waves * 1024.0 * UNROLL;
  return bytes / (ms * 1e-3) / 1e9;                                         // GB/s, whole chip
}

int main() {
  const size_t total = (size_t)2 << 30;
  char* src; float* sink;
  CHECK(hipMalloc(&src, total)); CHECK(hipMemset(src, 1, total)); CHECK(hipMalloc(&sink, 4096));
  const int grid = 256;
  const char* names[3] = {"LDS-DMA 16 B/lane", "global_load_dwordx4 -> VGPR", "global_load_dwordx4 -> VGPR -> ds_write_b128"};
  for (int shared = 1; shared >= 0; --shared) {
    printf("## %s\n", shared ? "every workgroup re-reads the same 1 MiB (L2-resident)" : "every workgroup streams its own 8 MiB slice (HBM / Infinity Cache)");
    printf("%-46s %6s %10s %12s %10s\n", "path", "waves", "GB/s chip", "GB/s per CU", "B/clk/CU@2.1GHz");
    for (int mode = 0; mode < 3; ++mode)
      for (int waves : {1, 2, 4, 8}) {
        const size_t span = (size_t)1 << 20, per = (size_t)8 << 20;
        double g = mode == 0 ? run<0, 4>(src, span, per, waves, shared, grid, sink) : mode == 1 ? run<1, 4>(src, span, per, waves, shared, grid, sink)
                                                                                              : run<2, 4>(src, span, per, waves, shared, grid, sink);
        printf("%-46s %6d %10.0f %12.1f %10.1f\n", names[mode], waves, g, g / grid, g / grid / 2.1);
      }
  }
  // fewer CUs active: is the cap per CU or shared?
  printf("## 64 workgroups only (L2-resident)\n");
  for (int mode = 0; mode < 2; ++mode)
    for (int waves : {4, 8}) {
      double g = mode == 0 ? run<0, 4>(src, (size_t)1 << 20, (size_t)8 << 20, waves, 1, 64, sink) : run<1, 4>(src, (size_t)1 << 20, (size_t)8 << 20, waves, 1, 64, sink);
      printf("%-46s %6d %10.0f %12.1f %10.1f\n", names[mode], waves, g, g / 64, g / 64 / 2.1);
    }
  return 0;
}
