// A stand-in for the kernel of an overlapped collective (RCCL's all-gather: a few dozen long-lived workgroups that need a CU each): `wgs`
// workgroups of `threads` threads that hold their CU slots (and `lds` bytes of LDS) for `us` microseconds without using memory bandwidth.
// bench.py --occupy launches one per step on its own stream while the forward runs: what do the forward's one-workgroup-per-CU persistent
// launches lose when some CUs are taken -- with and without --reserve-cus?  (the build box has one GPU: RCCL with > 1 rank cannot be run)
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC occupy.hip -o _bin/liboccupy.so
#include <hip/hip_runtime.h>

// (RCCL's kernels use on the order of a hundred VGPRs: a wave of theirs does not fit beside the 504 of 512 registers per lane that three
// 168-register waves of the persistent kernels hold on a SIMD.  The clobber makes this kernel allocate 128 registers, so that it needs CUs
// -- or at least SIMD register space -- of its own like they do; the first form, with 8 registers, slipped in beside the persistent waves.)
__global__ void __launch_bounds__(512) occupy_kernel(long long ticks) {
  extern __shared__ float lds[];
  if (ticks < 0) lds[threadIdx.x] = 0.f;                     // (keeps the dynamic LDS allocation)
  asm volatile("v_mov_b32 v127, 0" ::: "v127");
  const long long t0 = wall_clock64();                       // constant 100 MHz
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int occupy_launch(void* stream, int wgs, int threads, int lds, int us) {
  hipLaunchKernelGGL(occupy_kernel, dim3(wgs), dim3(threads), (size_t)lds, (hipStream_t)stream, (long long)us * 100);
  return (int)hipGetLastError();
}
