// How long does a wave spend ISSUING k back-to-back LDS-DMAs (buffer_load_dwordx4 ... lds, 1 KiB per instruction) whose data comes from HBM
// (every workgroup streams its own slice) or from L2 (all re-read 1 MiB), with 1..8 waves of the CU doing the same at the same time?
// s_memtime around the issue sequence, then s_waitcnt vmcnt(0); averaged over iterations and waves.  One workgroup per CU (LDS-limited).
//   hipcc --offload-arch=gfx950 -O3 dma_issue_cost.hip -o _bin/dma_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int K>
__global__ void __launch_bounds__(512) issue_cost(const char* __restrict__ src, size_t per_wg, int iters, int waves, int shared, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave >= waves) return;
  const size_t mine = shared ? ((size_t)1 << 20) : per_wg;
  const char* base = shared ? src : src + (size_t)blockIdx.x * per_wg;
  __amdgpu_buffer_rsrc_t buf = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)mine, 0x00020000);
  unsigned off = (unsigned)wave * 1024u * K + (unsigned)lane * 16u;
  const unsigned step = (unsigned)waves * 1024u * K;
  float* dst = smem + wave * (256 * K);
  unsigned long long t_issue = 0, t_wait = 0;
  for (int it = 0; it < iters; ++it) {
    if (off + 1024u * K > (unsigned)mine) off = (unsigned)wave * 1024u * K + (unsigned)lane * 16u;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int u = 0; u < K; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(buf, (__attribute__((address_space(3))) void*)(dst + u * 256), 16, (int)(off + u * 1024u), 0, 0, 0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    t_issue += t1 - t0; t_wait += t2 - t1;
    off += step;
  }
  if (lane == 0) { atomicAdd(out, t_issue); atomicAdd(out + 1, t_wait); atomicAdd(out + 2, 1ull); }
}

template <int K>
void run(const char* src, int waves, int shared, unsigned long long* dev) {
  const int iters = 2000, grid = 256;
  CHECK(hipMemset(dev, 0, 32));
  CHECK(hipFuncSetAttribute((const void*)issue_cost<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  hipLaunchKernelGGL((issue_cost<K>), dim3(grid), dim3(512), 150 * 1024, 0, src, (size_t)8 << 20, iters, waves, shared, dev);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[3];
  CHECK(hipMemcpy(h, dev, 24, hipMemcpyDeviceToHost));
  const double n = (double)h[2] * iters;
  printf("%-6s waves %d  k %2d : issue %7.0f cycles (%5.0f per DMA)   then wait %7.0f\n", shared ? "L2" : "HBM", waves, K, h[0] / n, h[0] / n / K, h[1] / n);
}

int main() {
  char* src; unsigned long long* dev;
  CHECK(hipMalloc(&src, (size_t)2 << 30)); CHECK(hipMemset(src, 1, (size_t)2 << 30)); CHECK(hipMalloc(&dev, 64));
  for (int shared = 0; shared < 2; ++shared)
    for (int waves : {1, 4, 8}) {
      run<1>(src, waves, shared, dev); run<2>(src, waves, shared, dev); run<3>(src, waves, shared, dev); run<4>(src, waves, shared, dev);
      run<6>(src, waves, shared, dev); run<8>(src, waves, shared, dev); run<12>(src, waves, shared, dev); run<16>(src, waves, shared, dev);
    }
  return 0;
}
