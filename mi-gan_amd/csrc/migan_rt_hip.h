// HIP runtime shim for the product build (hipcc --offload-arch=gfx950).  The same `rt` interface
// is implemented by tests/emu/hip_emu.h for the CPU test harness.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#define MIGAN_DEVICE __device__
#define MIGAN_INLINE __forceinline__
#define MIGAN_GLOBAL __global__
#define MIGAN_LAUNCH_BOUNDS(threads, waves_per_simd) __launch_bounds__(threads, waves_per_simd)
#define MIGAN_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) float name[]
// v_mfma_f32_32x32x2_f32: exact-f32 matrix FMA (D = A[32x2] * B[2x32] + C), 64 cycles per SIMD
#define MIGAN_MFMA_F32_32X32X2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MIGAN_FMUL_RN(a, b) __fmul_rn((a), (b))
#define MIGAN_FADD_RN(a, b) __fadd_rn((a), (b))
#define MIGAN_FSUB_RN(a, b) __fsub_rn((a), (b))
// v_mfma_f32_32x32x16_bf16: D = A[32x16] * B[16x32] + C, operands as 8 bf16 (16 bytes) per lane
typedef __bf16 migan_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 migan_bf16x2 __attribute__((ext_vector_type(2)));
#define MIGAN_MFMA_BF16_32X32X16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(migan_bf16x8, (a)), __builtin_bit_cast(migan_bf16x8, (b)), (c), 0, 0, 0)
// two fp32 -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32); low half = first argument
#define MIGAN_PACK_BF16(lo, hi) __builtin_bit_cast(unsigned, migan_bf16x2{(__bf16)(lo), (__bf16)(hi)})
// fp16 pieces (GEMMV 2): v_cvt_pk_f16_f32 (round to nearest even), v_cvt_f32_f16, v_mfma_f32_32x32x16_f16
typedef _Float16 migan_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 migan_f16x2 __attribute__((ext_vector_type(2)));
typedef float migan_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned migan_pack_f16(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(migan_f32x2{lo, hi}, migan_f16x2));
}
__device__ __forceinline__ float migan_f16lo_f32(unsigned pk) { return (float)__builtin_bit_cast(migan_f16x2, pk).x; }
__device__ __forceinline__ float migan_f16hi_f32(unsigned pk) { return (float)__builtin_bit_cast(migan_f16x2, pk).y; }
#define MIGAN_PACK_F16(lo, hi) migan_pack_f16((lo), (hi))
#define MIGAN_F16LO_F32(pk) migan_f16lo_f32(pk)
#define MIGAN_F16HI_F32(pk) migan_f16hi_f32(pk)
#define MIGAN_MFMA_F16_32X32X16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(migan_f16x8, (a)), __builtin_bit_cast(migan_f16x8, (b)), (c), 0, 0, 0)
// The clamp of lrelu_agc (reference :21-23) is v_med3_f32, which returns lo for a NaN (what the reference's CUDA plugin does, bias_act.cu:139),
// whereas Tensor.clamp in the reference module keeps a NaN a NaN.  The DEFAULT build follows the module ("follow torch"): clamp4 /
// clamp1 (migan_kernels.hpp) repair the NaNs behind a wave-uniform branch that finite data never takes -- one v_cmp_u_f32 per TWO values.
// -DMIGAN_NAN_CLAMP (libmigan_hip_nanclamp.so, Generator(nan_policy="clamp")): the bare v_med3_f32, the fastest form, NaN -> -256.
// MIGAN_ANY_LANE(p): true in every lane of the wave when p holds in any (s_cbranch on the ballot: a wave-uniform branch around code that
// is lane-wise a no-op where p is false -- the NaN fix-up of clamp4 / clamp1, migan_kernels.hpp)
#define MIGAN_ANY_LANE(p) (__builtin_amdgcn_ballot_w64(p) != 0ull)
// keeps the optimiser from turning a branch into speculated selects (an empty asm with a side effect cannot be hoisted)
#define MIGAN_COLD_PATH() asm volatile("" ::: "memory")
#define MIGAN_CLAMP(v, lo, hi) __builtin_amdgcn_fmed3f((v), (lo), (hi))     // v_med3_f32
// ds_swizzle bit mode: lane' = ((lane & and_mask) | or_mask) ^ xor_mask inside groups of 32 lanes
// (a function, not a macro body: __builtin_bit_cast applied directly to a vector element lvalue such as `v.y`
// reads element 0 with this compiler; passing the float by value is safe)
template <int M>
__device__ __forceinline__ float migan_swizzle_xor(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), ((M << 10) | 0x1f)));
}
#define MIGAN_SWIZZLE_XOR(v, m) migan_swizzle_xor<(m)>(v)
// sum over aligned groups of 8 lanes, result in every lane of the group: three v_add_f32_dpp (quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_half_mirror) -- VALU only, no LDS crossbar; same summation tree as an xor-1/2/4 butterfly
template <int CTRL>
__device__ __forceinline__ float migan_dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float migan_sum8(float v) {
  v += migan_dpp_get<0xB1>(v);
  v += migan_dpp_get<0x4E>(v);
  v += migan_dpp_get<0x141>(v);
  return v;
}
#define MIGAN_SUM8(v) migan_sum8(v)
#define MIGAN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// instruction-class pipeline hints for the machine scheduler: the next `n` instructions of class `mask` (0x8 MFMA, 0x20 VMEM
// read, 0x100 DS read, 0x200 DS write) form one group; groups are laid out in the order these calls appear
#define MIGAN_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#define MIGAN_STORE_NT(ptr, v) __builtin_nontemporal_store((v), (ptr))
#define MIGAN_LOAD_NT(ptr) __builtin_nontemporal_load(ptr)
#define MIGAN_OPAQUE(x) asm volatile("" : "+v"(x))
// the same for a float: an optimisation barrier on one value (keeps scalar FMA chains scalar)
#define MIGAN_OPAQUE_F(x) asm volatile("" : "+v"(x))
#define MIGAN_CLOCK() __builtin_readcyclecounter()
#define MIGAN_ATOMIC_ADD_U64(p, v) atomicAdd((p), (v))

// ---- LDS-DMA staging (sepconv_wide_kernel<..., DMA>) ---------------------------------------------------------------------------
// wave-uniform value the compiler can keep in an SGPR (threadIdx-derived values are divergent to it even when they are not)
#define MIGAN_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// raw buffer descriptor over [ptr, ptr + bytes): an access whose lane byte offset is >= bytes returns 0 (the hardware range check
// is the zero padding of the convolution)
typedef __amdgpu_buffer_rsrc_t MIGAN_BUF;
__device__ __forceinline__ MIGAN_BUF migan_make_buf(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
#define MIGAN_MAKE_BUF(ptr, bytes) migan_make_buf((ptr), (bytes))
// buffer_load_dwordx4 ... lds: 16 bytes per active lane from buf[voff + soff] straight into LDS at (wave-uniform) ldsp + 16 * lane,
// no VGPR and no ds_write in between.  Completion is tracked by vmcnt; hipcc waits for it (vmcnt(0)) at the next __syncthreads().
#define MIGAN_LDS_DMA16(buf, voff, soff, ldsp) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((buf), (__attribute__((address_space(3))) void*)(ldsp), 16, (int)(voff), (int)(soff), 0, 0)
// buffer_load_dword ... lds: 4 bytes per lane, LDS destination ldsp + 4 * lane
#define MIGAN_LDS_DMA4(buf, voff, soff, ldsp) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((buf), (__attribute__((address_space(3))) void*)(ldsp), 4, (int)(voff), (int)(soff), 0, 0)
// the same under a lane predicate (every wave must keep at least one active lane, so that all waves of a group issue the same number of
// vector-memory instructions and one counted s_waitcnt holds for all of them; the CPU emulator, which counts per lane, records a null
// operation for the inactive lanes)
#define MIGAN_LDS_DMA16_IF(cond, buf, voff, soff, ldsp) do { if (cond) MIGAN_LDS_DMA16((buf), (voff), (soff), (ldsp)); } while (0)
#define MIGAN_LDS_DMA4_IF(cond, buf, voff, soff, ldsp) do { if (cond) MIGAN_LDS_DMA4((buf), (voff), (soff), (ldsp)); } while (0)
// value of lane k (compile-time constant) of this wave, in a scalar register: v_readlane_b32
#define MIGAN_READLANE(v, k) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (v)), (k)))
// optimisation barrier on a wave-uniform integer (stays in a scalar register)
#define MIGAN_OPAQUE_S(x) asm volatile("" : "+s"(x))

// ---- hand-placed synchronisation of the LDS-DMA pipelines (sepconv_pipe_kernel, sepconv_wide_kernel<..., DMA>) --------------------
// s_waitcnt vmcnt(n): at most n of this wave's vector-memory operations (LDS-DMAs, loads, stores: issue order) still outstanding.
// Inline asm on purpose: the compiler's own waitcnt pass neither sees nor removes it (MI355X_MICROARCH.md, "Compiler hazard").
#define MIGAN_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// workgroup barrier that publishes this wave's LDS writes but leaves its vector-memory operations in flight: __syncthreads() would drain
// them (an LDS-DMA is a pending LDS write on the VM counter; a store tail would be waited for as well)
#define MIGAN_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// the lanes of one wave exchange data through LDS: in-order LDS + lockstep execution need no instruction, only a fence the scheduler
// will not move LDS accesses across (the CPU emulator, whose lanes are independent fibers, synchronises the wave here)
#define MIGAN_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// s_setprio: issue priority of this wave among the waves of its SIMD (0 = default .. 3)
#ifdef MIGAN_NO_PRIO
#define MIGAN_SETPRIO(n) do {} while (0)
#else
#define MIGAN_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif

namespace rt {
typedef hipStream_t stream_t;
typedef hipEvent_t event_t;

inline const char* backend_name() { return "hip:gfx950"; }
inline std::string error_string(int rc) { return hipGetErrorString((hipError_t)rc); }
inline int set_device(int dev) { return (int)hipSetDevice(dev); }
inline int get_device(int* dev) { return (int)hipGetDevice(dev); }
inline int stream_create(stream_t* s) { return (int)hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
inline int stream_destroy(stream_t s) { return (int)hipStreamDestroy(s); }
// event used only to order streams (no timestamps)
inline int event_create_sync(event_t* e) { return (int)hipEventCreateWithFlags(e, hipEventDisableTiming); }
inline int stream_wait_event(stream_t s, event_t e) { return (int)hipStreamWaitEvent(s, e, 0); }
inline int allow_dynamic_lds(const void* fn, size_t bytes) {
  return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
template <class Args>
inline int launch(void (*kernel)(const Args), const Args& a, unsigned grid, unsigned block, size_t lds, stream_t s) {
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, s, a);
  return (int)hipGetLastError();
}
inline int memcpy_d2h(void* dst, const void* src, size_t bytes, stream_t s) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
  if (e != hipSuccess) return (int)e;
  return (int)hipStreamSynchronize(s);
}
inline int prof_alloc(unsigned long long** p, int n) {
  hipError_t e = hipMalloc((void**)p, n * sizeof(unsigned long long));
  if (e != hipSuccess) return (int)e;
  return (int)hipMemset(*p, 0, n * sizeof(unsigned long long));
}
inline int prof_read(unsigned long long* dev, unsigned long long* out, int n, bool reset) {
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return (int)e;
  e = hipMemcpy(out, dev, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return (int)e;
  return reset ? (int)hipMemset(dev, 0, n * sizeof(unsigned long long)) : 0;
}
inline int stream_sync(stream_t s) { return (int)hipStreamSynchronize(s); }
inline int event_create(event_t* e) { return (int)hipEventCreate(e); }
inline int event_destroy(event_t e) { return (int)hipEventDestroy(e); }
inline int event_record(event_t e, stream_t s) { return (int)hipEventRecord(e, s); }
inline int event_elapsed(float* ms, event_t a, event_t b) { return (int)hipEventElapsedTime(ms, a, b); }
}  // namespace rt
