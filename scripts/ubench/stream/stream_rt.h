// Device-side helpers of the register-streaming experiment (scripts/ubench/stream/migan_stream.hpp): buffer descriptors with
// hardware range checks, wave-level LDS ordering, and the DPP depthwise taps.  Experiment only -- not part of libmigan_hip.so.
#pragma once
// wave-uniform value the compiler can keep in an SGPR (threadIdx-derived values are divergent to it even when they are not)
#ifndef MIGAN_UNIFORM
#define MIGAN_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif
// lanes of ONE wave exchanging data through LDS: a wave's LDS instructions execute in order, so no barrier instruction is needed,
// only the compiler must not move the accesses across this point
#define MIGAN_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// raw buffer descriptor over [ptr, ptr + bytes): accesses at byte offset voffset + soffset >= bytes return 0 / are dropped
// (the hardware range check is the zero padding of the convolution and the mask of partial strips)
#ifndef MIGAN_MAKE_BUF
typedef __amdgpu_buffer_rsrc_t MIGAN_BUF;
#endif
typedef unsigned migan_u4 __attribute__((ext_vector_type(4)));
typedef float migan_f4 __attribute__((ext_vector_type(4)));
#ifndef MIGAN_MAKE_BUF
__device__ __forceinline__ MIGAN_BUF migan_make_buf(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
#define MIGAN_MAKE_BUF(ptr, bytes) migan_make_buf((ptr), (bytes))
#endif
#define MIGAN_BUF_LOAD4(buf, voff, soff) __builtin_bit_cast(migan_f4, __builtin_amdgcn_raw_buffer_load_b128((buf), (int)(voff), (int)(soff), 0))
#define MIGAN_BUF_STORE4(buf, voff, soff, v) \
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(migan_u4, (v)), (buf), (int)(voff), (int)(soff), 2 /* nt */)
#define MIGAN_BUF_STORE1(buf, voff, soff, v) \
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(v)), (buf), (int)(voff), (int)(soff), 0)
#define MIGAN_BUF_LOAD1(buf, voff, soff) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32((buf), (int)(voff), (int)(soff), 0))
// eight taps of a depthwise 3x3 whose x - 1 / x + 1 neighbours are the adjacent lanes: v_fmac_f32 with a DPP source
// (wave_shr:1 = the value of lane - 1, wave_shl:1 = lane + 1, 0 beyond the wave).
//   acc = bias + w1 t + w7 b + shr(t) w0 + shl(t) w2 + shr(m) w3 + shl(m) w5 + shr(b) w6 + shl(b) w8      (the centre tap w4 m is the caller's)
// The two plain instructions come first: a DPP read needs two wait states after a VALU write of its source register, and nothing
// inside an asm statement is padded by the compiler.
#if defined(MIGAN_STREAM_ABL) && (MIGAN_STREAM_ABL & 1)
// measurement build: the same eight FMAs without the DPP lane shifts (wrong results, same instruction count)
#define MIGAN_DW8(acc, bias, t, m, b, w0, w1, w2, w3, w5, w6, w7, w8)                                                 \
  asm("v_fma_f32 %0, %1, %5, %12\n\t"                                                                                  \
      "v_fmac_f32_e32 %0, %3, %10\n\t"                                                                                 \
      "v_fmac_f32_e32 %0, %1, %4\n\t"                               \
      "v_fmac_f32_e32 %0, %1, %6\n\t"                               \
      "v_fmac_f32_e32 %0, %2, %7\n\t"                               \
      "v_fmac_f32_e32 %0, %2, %8\n\t"                               \
      "v_fmac_f32_e32 %0, %3, %9\n\t"                               \
      "v_fmac_f32_e32 %0, %3, %11"                                  \
      : "=&v"(acc)                                                                                                     \
      : "v"(t), "v"(m), "v"(b), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w5), "v"(w6), "v"(w7), "v"(w8), "v"(bias))
#elif defined(MIGAN_STREAM_ABL) && (MIGAN_STREAM_ABL & 2)
// measurement build: row_shr / row_shl (16-lane rows) instead of the wave-wide shifts
#define MIGAN_DW8(acc, bias, t, m, b, w0, w1, w2, w3, w5, w6, w7, w8)                                                 \
  asm("v_fma_f32 %0, %1, %5, %12\n\t"                                                                                  \
      "v_fmac_f32_e32 %0, %3, %10\n\t"                                                                                 \
      "v_fmac_f32_dpp %0, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
      "v_fmac_f32_dpp %0, %1, %6 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
      "v_fmac_f32_dpp %0, %2, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
      "v_fmac_f32_dpp %0, %2, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
      "v_fmac_f32_dpp %0, %3, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
      "v_fmac_f32_dpp %0, %3, %11 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"                                  \
      : "=&v"(acc)                                                                                                     \
      : "v"(t), "v"(m), "v"(b), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w5), "v"(w6), "v"(w7), "v"(w8), "v"(bias))
#else
#define MIGAN_DW8(acc, bias, t, m, b, w0, w1, w2, w3, w5, w6, w7, w8)                                                 \
  asm("v_fma_f32 %0, %1, %5, %12\n\t"                                                                                  \
      "v_fmac_f32_e32 %0, %3, %10\n\t"                                                                                 \
      "v_fmac_f32_dpp %0, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
      "v_fmac_f32_dpp %0, %1, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
      "v_fmac_f32_dpp %0, %2, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
      "v_fmac_f32_dpp %0, %2, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
      "v_fmac_f32_dpp %0, %3, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                               \
      "v_fmac_f32_dpp %0, %3, %11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"                                  \
      : "=&v"(acc)                                                                                                     \
      : "v"(t), "v"(m), "v"(b), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w5), "v"(w6), "v"(w7), "v"(w8), "v"(bias))
#endif

