// MI-GAN generator forward: gfx950 (MI355X / CDNA4) device kernels.
//
// One fused kernel per SeparableConv2d (reference lib/model_zoo/migan_inference.py:154-170):
//
//   HBM (NHWC fp32) --coalesced float4, register-prefetched one K chunk ahead--> LDS input tile
//     -> depthwise 3x3 + bias            (VALU on LDS data; column strips with rotating accumulators)
//     -> lrelu*sqrt2, clamp              (reference :20-28)
//     -> [4x4 FIR, stride 2]             (reference Downsample2d :58-76; separable, vertical pass fused in the strip)
//     -> 1x1 conv as an fp32 MFMA GEMM   (v_mfma_f32_32x32x2_f32, exact f32; M = tile pixels, K = Cin chunk, N = Cout tile)
//     -> LDS result tile
//     -> [polyphase 2x FIR upsample]     (reference Upsample2d :79-103; zero-insertion folded into 2x2 taps)
//     -> [+ noise_const*noise_strength]  (reference :165-167)
//     -> lrelu*sqrt2, clamp -> [+ skip]  (reference :272 / :305)
//     -> [ToRGB 1x1 + bias + upsampled previous image, wave-shuffle channel reduction] (reference :308-313)
//     -> HBM (NHWC fp32), float4 stores, 256..1024 B contiguous per pixel
//
// K loop software pipeline (per workgroup, 4 wave64):
//     regs <- global(chunk c+1)   issued right after the barrier that publishes chunk c, so HBM/L2
//                                 latency runs under the depthwise stage and the MFMAs of chunk c
//     LDS  <- regs(chunk c)       input tile, 1x1 weight tile (double buffered when it fits: 2 barriers
//                                 per chunk instead of 3), depthwise weights
//
// Everything in the encoder/decoder runs through `sepconv_kernel`; `torgb_kernel` is the un-fused
// ToRGB for layers whose Cout is split over several workgroups.
//
// The file is compiled twice: by hipcc for gfx950 (the product, libmigan_hip.so) and by the host
// compiler against tests/emu/hip_emu.h (a fiber-per-lane SIMT emulator used only by the CPU test
// suite to check indexing/LDS/MFMA-fragment logic without a GPU).  The emulator is test
// infrastructure; nothing in the product loads it.
#pragma once

namespace migan {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;          // 4 wave64 per workgroup, one per SIMD

enum : int { MODE_NORMAL = 0, MODE_DOWN = 1, MODE_UP = 2, MODE_PW = 3 };
struct TrueT { static constexpr bool value = true; };
struct FalseT { static constexpr bool value = false; };

struct SepArgs {
  // activations
  // activation tensors are stored as the kernel's STV template parameter says: 0 = fp32, 1 = bf16, 2 = fp16 (all
  // arithmetic is fp32 either way; 16-bit storage rounds once, when a layer output is written)
  const void* x;               // NHWC [B][H][W][CI]  (FROMRGB: network input, fp32 NCHW [B][4][H][W]; MODE_PW: fp32, dwfir_kernel's output)
  void* y;                     // NHWC [B][HO][WO][CO]
  const void* skip;            // NHWC like y, added after the activation, or null
  // SeparableConv2d parameters in the reference's native layouts
  const float* wdw;            // conv1.weight [CI][1][3][3]
  const float* bdw;            // conv1.bias   [CI]
  const float* wpw;            // conv2.weight [CO][CI][1][1]
  const unsigned short* wsplit;// GEMMV 1/2: conv2.weight as 16-bit planes, chunk-major [planes][CI/32][CO][32] (split_weights_kernel), else null
  const float* noise;          // noise_const [HO][WO] or null
  const float* noise_strength; // scalar
  // EncoderBlock.fromrgb (reference :186,:194-195), only for FROMRGB instantiations
  const float* frgb_w;         // [CI][4][1][1]
  const float* frgb_b;         // [CI]
  // Synthesis torgb (reference :268,:300), fused only when one workgroup owns all of CO
  const float* trgb_w;         // [3][CO][1][1] or null
  const float* trgb_b;         // [3]
  const float* img_prev;       // planar [B][3][HO/2][WO/2] or null
  float* img_out;              // planar [B][3][HO][WO]
  unsigned long long* prof;    // phase-cycle accumulators (MIGAN_PHASE_PROF builds only), else null
  // uint8 network I/O (migan_forward_u8; reference scripts/demo.py:56-66,135-140), else null
  const unsigned char* u8_img; // [B][H][W][3] HWC: FROMRGB builds x = cat([mask - 0.5, img * mask]) from it instead of reading p.x;
  const unsigned char* u8_mask;// [B][H][W], 255 = known pixel      the fused ToRGB tail composes its output with it
  unsigned char* u8_out;       // [B][HO][WO][3]: written by the fused ToRGB tail instead of img_out
  int B, H, W, CI, CO, HO, WO;
  // GEMM pixel grid of one workgroup: IMGS images x GH x GW (all powers of two), MT = IMGS*GH*GW rows
  int lgGH, lgGW, lgIMGS;
  int tiles_x, tiles_y, nchunks;   // workgroup grid: tiles_x * tiles_y * ceil(B/IMGS) * nchunks
  int sy, sx, off;                 // tile pitch in GEMM-resolution pixels; off = 1 for MODE_UP (recomputed halo)
  int lgRS;                        // log2(row segments per column) of the depthwise stage (NORMAL/UP)
  int off_a, off_b, off_v, off_rgb, off_w;   // LDS carve, in floats
  int b_stride;                    // floats between the two 1x1-weight buffers (0 = single buffered)
  int a_stride;                    // MODE_PW: floats between the two A-operand buffers
};

struct RgbArgs {
  const void* x;         // NHWC [B][H][W][C], stored as Io<STV>
  const float* w;        // [3][C]
  const float* b;        // [3]
  const float* img_prev; // planar [B][3][H/2][W/2] or null
  float* img_out;        // planar [B][3][H][W]
  const unsigned char* u8_img;   // uint8 network output (see SepArgs), else null
  const unsigned char* u8_mask;
  unsigned char* u8_out;
  int B, H, W, C;
};

struct NoiseArgs {
  const float* src;      // noise_const [r][r]
  float* dst;            // [h][w] = src tiled periodically and cropped
  int r, h, w;
};

// ------------------------------------------------------------------------------------------------
// small device helpers

// lrelu_agc(alpha=0.2, gain=sqrt(2), clamp=256), reference :20-28: leaky_relu, fp32 multiply by
// float(np.sqrt(2)), clamp.  max(v, 0.2v) == (v > 0 ? v : 0.2v) for every finite v; the two
// multiplies stay separate so results round exactly like the reference's three passes.
// The clamp of lrelu_agc (reference :21-23) on four values / one value.  v_med3_f32 returns the lower bound for a NaN; Tensor.clamp in the
// reference module keeps it a NaN, and so does this (default) build, at half an instruction per value: one v_cmp_u_f32 tests TWO values
// (unordered(a, b) holds iff a or b is a NaN), the wave branches on the ballot, and only a wave that holds a NaN runs the per-value
// compare + select, in a block laid out of line (__builtin_expect): finite data falls through.  Measured on MI355X, migan-512 forward, each row
// against the bare v_med3_f32 of -DMIGAN_NAN_CLAMP on the same box (profiles/r06_nan_policy.md): compare + select on every value (rounds 4-5)
// -4.1 %; this form with the repair block inline -1.7 ... -2.1 %; out of line -1.2 % (shipped); testing first and clamping in each arm -4.6 %;
// three values folded by v_maximum3_f32 (NaN-propagating, gfx950) + one compare per four -1.65 %.
#ifdef MIGAN_NAN_NOEXPECT          // (measurement builds: the repair block inline behind a taken branch, the first form of round 6)
#define MIGAN_UNLIKELY(x) (x)
#else                              // the repair block out of line: the fast path falls through
#define MIGAN_UNLIKELY(x) __builtin_expect((x), 0)
#endif
MIGAN_DEVICE MIGAN_INLINE f4 clamp4(f4 t, float lo, float hi) {
  // (the clamps are issued BEFORE the branch on purpose: they run in the shadow of the compare -> ballot -> s_cbranch latency.  Testing first
  // and clamping in each arm measured -4.6 % against -2.1 % on the forward: profiles/r06_nan_policy.md)
  f4 c = f4{MIGAN_CLAMP(t.x, lo, hi), MIGAN_CLAMP(t.y, lo, hi), MIGAN_CLAMP(t.z, lo, hi), MIGAN_CLAMP(t.w, lo, hi)};
#ifndef MIGAN_NAN_CLAMP
  if (MIGAN_UNLIKELY(MIGAN_ANY_LANE(__builtin_isunordered(t.x, t.y) | __builtin_isunordered(t.z, t.w)))) {
    MIGAN_COLD_PATH();
    c.x = t.x != t.x ? t.x : c.x;
    c.y = t.y != t.y ? t.y : c.y;
    c.z = t.z != t.z ? t.z : c.z;
    c.w = t.w != t.w ? t.w : c.w;
  }
#endif
  return c;
}
MIGAN_DEVICE MIGAN_INLINE float clamp1(float t, float lo, float hi) {
  float c = MIGAN_CLAMP(t, lo, hi);
#ifndef MIGAN_NAN_CLAMP
  if (MIGAN_UNLIKELY(MIGAN_ANY_LANE(t != t))) {
    MIGAN_COLD_PATH();
    c = t != t ? t : c;
  }
#endif
  return c;
}
MIGAN_DEVICE MIGAN_INLINE float act1(float v) {
  float t = fmaxf(v, v * 0.2f);
  t = t * 1.41421356237309515f;
  return clamp1(t, -256.0f, 256.0f);
}
MIGAN_DEVICE MIGAN_INLINE f4 act4(f4 v) {
  // vector form so the two multiplies become v_pk_mul_f32 (2 instead of 4 VALU issues each)
  f4 t = __builtin_elementwise_max(v, v * 0.2f);
  t = t * 1.41421356237309515f;
  return clamp4(t, -256.0f, 256.0f);
}

// act4 with the gain pre-multiplied by a power of two s: act4g(v, fl(sqrt2 * s)) == act4(v * s) exactly
MIGAN_DEVICE MIGAN_INLINE f4 act4g(f4 v, float gain) {
  f4 t = __builtin_elementwise_max(v, v * 0.2f);
  t = t * gain;
  return clamp4(t, -256.0f, 256.0f);
}

// ---- error-compensated bf16 GEMM operands -----------------------------------------------------
// x = h1 + h2 + h3 (+ <= 2^-24 |x|) with h_i bf16: h1 = bf16(x), h2 = bf16(x - h1), h3 = bf16(x - h1 - h2)
// (both subtractions are exact in fp32).  a*b is then summed from the six bf16 x bf16 products of
// order <= 2^-16 (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1), each exact in the MFMA's fp32 accumulator:
// fp32-grade accuracy (whole-generator error vs fp64 7.7e-6, same as the fp32 path) at 6/16 of the
// fp32-MFMA cost.
MIGAN_DEVICE MIGAN_INLINE float bf16lo_f32(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
MIGAN_DEVICE MIGAN_INLINE float bf16hi_f32(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }
// split 4 floats into three planes of 4 bf16 (8 bytes each)
MIGAN_DEVICE MIGAN_INLINE void split3_bf16(f4 v, u2v& h1, u2v& h2, u2v& h3) {
  const unsigned a01 = MIGAN_PACK_BF16(v.x, v.y), a23 = MIGAN_PACK_BF16(v.z, v.w);
  const f4 r1 = f4{v.x - bf16lo_f32(a01), v.y - bf16hi_f32(a01), v.z - bf16lo_f32(a23), v.w - bf16hi_f32(a23)};
  const unsigned b01 = MIGAN_PACK_BF16(r1.x, r1.y), b23 = MIGAN_PACK_BF16(r1.z, r1.w);
  const f4 r2 = f4{r1.x - bf16lo_f32(b01), r1.y - bf16hi_f32(b01), r1.z - bf16lo_f32(b23), r1.w - bf16hi_f32(b23)};
  h1 = u2v{a01, a23};
  h2 = u2v{b01, b23};
  h3 = u2v{MIGAN_PACK_BF16(r2.x, r2.y), MIGAN_PACK_BF16(r2.z, r2.w)};
}

// ---- error-compensated fp16 GEMM operands (GEMMV 2) ----------------------------------------------
// x*s = h1 + h2 (+ <= 2^-22 |x*s|) with h_i fp16 (round to nearest even, 11-bit significands; the
// subtraction is exact in fp32); a*b is summed from a2b1 + a1b2 + a1b1 (a2b2 <= 2^-22 |ab| is dropped),
// each product exact in the MFMA's fp32 accumulator.  Power-of-two scales keep both operands inside
// fp16's normal range: activations are bounded by the +-256 clamp of lrelu_agc (reference :21-23), so
// kF16AScale = 2^7 maps them into +-2^15; every weight tensor is scaled so that its largest magnitude
// lands in [2^13, 2^14) (weight_absmax_kernel).  The accumulators are multiplied by the exact inverse.
// Whole-generator error vs float64: same as the fp32 path (scripts/split_accuracy.py), at 3/16 of
// the fp32-MFMA cost.
constexpr float kF16AScale = 128.0f;
MIGAN_DEVICE MIGAN_INLINE void split2_f16(f4 v, u2v& h1, u2v& h2) {
  const unsigned a01 = MIGAN_PACK_F16(v.x, v.y), a23 = MIGAN_PACK_F16(v.z, v.w);
  const f4 r = f4{v.x - MIGAN_F16LO_F32(a01), v.y - MIGAN_F16HI_F32(a01), v.z - MIGAN_F16LO_F32(a23), v.w - MIGAN_F16HI_F32(a23)};
  h1 = u2v{a01, a23};
  h2 = u2v{MIGAN_PACK_F16(r.x, r.y), MIGAN_PACK_F16(r.z, r.w)};
}
// act4(v) * 2^S, exactly (a power-of-two scale commutes with every rounding of act4)
template <int S>
MIGAN_DEVICE MIGAN_INLINE f4 act4_scaled(f4 v) {
  constexpr float sc = (float)(1 << S);
  f4 t = __builtin_elementwise_max(v, v * 0.2f);
  t = t * (1.41421356237309515f * sc);
  return clamp4(t, -256.0f * sc, 256.0f * sc);
}

MIGAN_DEVICE MIGAN_INLINE f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
MIGAN_DEVICE MIGAN_INLINE void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
// store of a layer output (streamed: the consumer is a later kernel, after far more than L2 has been written)
// (nontemporal: measured +1.5 % end to end, profiles/)
MIGAN_DEVICE MIGAN_INLINE void st4o(float* p, f4 v) { MIGAN_STORE_NT(reinterpret_cast<f4*>(p), v); }
// load of a tensor that is read exactly once (the skip connection in the epilogue)
// uniform base pointer (SGPR pair) + 32-bit lane BYTE offset: the form the global_load/store saddr + voffset
// encoding takes directly, so no 64-bit VALU address add per access
MIGAN_DEVICE MIGAN_INLINE float* at_bytes(float* base, unsigned byte_off) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off);
}
MIGAN_DEVICE MIGAN_INLINE const float* at_bytes(const float* base, unsigned byte_off) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
MIGAN_DEVICE MIGAN_INLINE f4 ld4once(const float* p) { return MIGAN_LOAD_NT(reinterpret_cast<const f4*>(p)); }

// ---- activation storage formats -------------------------------------------------------------------
// Io<STV>: four consecutive channels of an NHWC activation tensor <-> one f4 of fp32 values.  `raw4` is what a
// thread keeps in its prefetch registers (16 bytes for fp32 storage, 8 bytes for the 16-bit formats); every
// address is (wave-uniform base pointer) + (32-bit lane BYTE offset), the saddr + voffset form of global_load.
//   STV 0: fp32
//   STV 1: bf16, round to nearest even on store (v_cvt_pk_bf16_f32), exact widening on load
//   STV 2: fp16, round to nearest even on store (v_cvt_pk_f16_f32); layer outputs are bounded by the +-256 clamp of
//          lrelu_agc (+ one skip tensor), far inside fp16's range
template <int STV> struct Io;
template <> struct Io<0> {
  typedef f4 raw4;
  static constexpr unsigned ESZ = 4;
  static MIGAN_DEVICE MIGAN_INLINE raw4 ld(const char* base, unsigned off) { return *reinterpret_cast<const f4*>(base + off); }
  static MIGAN_DEVICE MIGAN_INLINE raw4 ld_once(const char* base, unsigned off) { return MIGAN_LOAD_NT(reinterpret_cast<const f4*>(base + off)); }
  static MIGAN_DEVICE MIGAN_INLINE f4 cvt(raw4 r) { return r; }
  static MIGAN_DEVICE MIGAN_INLINE f4 rounded(f4 v) { return v; }      // the value a consumer of the stored tensor reads back
  static MIGAN_DEVICE MIGAN_INLINE raw4 zero() { return f4{0.f, 0.f, 0.f, 0.f}; }
  static MIGAN_DEVICE MIGAN_INLINE void st(char* base, unsigned off, f4 v) { MIGAN_STORE_NT(reinterpret_cast<f4*>(base + off), v); }
};
template <> struct Io<1> {
  typedef u2v raw4;
  static constexpr unsigned ESZ = 2;
  static MIGAN_DEVICE MIGAN_INLINE raw4 ld(const char* base, unsigned off) { return *reinterpret_cast<const u2v*>(base + off); }
  static MIGAN_DEVICE MIGAN_INLINE raw4 ld_once(const char* base, unsigned off) { return MIGAN_LOAD_NT(reinterpret_cast<const u2v*>(base + off)); }
  static MIGAN_DEVICE MIGAN_INLINE f4 cvt(raw4 r) { return f4{bf16lo_f32(r.x), bf16hi_f32(r.x), bf16lo_f32(r.y), bf16hi_f32(r.y)}; }
  static MIGAN_DEVICE MIGAN_INLINE f4 rounded(f4 v) { return cvt(u2v{MIGAN_PACK_BF16(v.x, v.y), MIGAN_PACK_BF16(v.z, v.w)}); }
  static MIGAN_DEVICE MIGAN_INLINE raw4 zero() { return u2v{0u, 0u}; }
  static MIGAN_DEVICE MIGAN_INLINE void st(char* base, unsigned off, f4 v) {
    MIGAN_STORE_NT(reinterpret_cast<u2v*>(base + off), (u2v{MIGAN_PACK_BF16(v.x, v.y), MIGAN_PACK_BF16(v.z, v.w)}));
  }
};
template <> struct Io<2> {
  typedef u2v raw4;
  static constexpr unsigned ESZ = 2;
  static MIGAN_DEVICE MIGAN_INLINE raw4 ld(const char* base, unsigned off) { return *reinterpret_cast<const u2v*>(base + off); }
  static MIGAN_DEVICE MIGAN_INLINE raw4 ld_once(const char* base, unsigned off) { return MIGAN_LOAD_NT(reinterpret_cast<const u2v*>(base + off)); }
  static MIGAN_DEVICE MIGAN_INLINE f4 cvt(raw4 r) { return f4{MIGAN_F16LO_F32(r.x), MIGAN_F16HI_F32(r.x), MIGAN_F16LO_F32(r.y), MIGAN_F16HI_F32(r.y)}; }
  static MIGAN_DEVICE MIGAN_INLINE f4 rounded(f4 v) { return cvt(u2v{MIGAN_PACK_F16(v.x, v.y), MIGAN_PACK_F16(v.z, v.w)}); }
  static MIGAN_DEVICE MIGAN_INLINE raw4 zero() { return u2v{0u, 0u}; }
  static MIGAN_DEVICE MIGAN_INLINE void st(char* base, unsigned off, f4 v) {
    MIGAN_STORE_NT(reinterpret_cast<u2v*>(base + off), (u2v{MIGAN_PACK_F16(v.x, v.y), MIGAN_PACK_F16(v.z, v.w)}));
  }
};

// One lane's share of the three ToRGB dot products (4 of the CO channels of a pixel; reference :312).
// Written as three scalar FMA chains the compiler cannot re-associate into packed-fp32 instructions: with 16-bit activation
// storage the SLP-vectorised form of this expression (v_pk_fma_f32 / v_pk_add_f32 with op_sel on operands unpacked from the
// rounded activations) returned wrong lower-half results in lanes 32-63, a few dozen pixels per launch and different ones
// every launch, on MI355X (hipcc 7.2; fp32 storage, whose operands come straight from v_med3_f32, never showed it; the
// un-fused torgb_kernel showed it only while a second stream's kernels shared the GPU).
// Measurements and the variants tried: profiles/r02_torgb_packed_f32_hazard.md.
MIGAN_DEVICE MIGAN_INLINE void torgb_partial(f4 v, f4 tw0, f4 tw1, f4 tw2, float& r0, float& r1, float& r2) {
  r0 = v.x * tw0.x; r1 = v.x * tw1.x; r2 = v.x * tw2.x;
  MIGAN_OPAQUE_F(r0); MIGAN_OPAQUE_F(r1); MIGAN_OPAQUE_F(r2);
  r0 = r0 + v.y * tw0.y; MIGAN_OPAQUE_F(r0); r1 = r1 + v.y * tw1.y; MIGAN_OPAQUE_F(r1); r2 = r2 + v.y * tw2.y; MIGAN_OPAQUE_F(r2);
  r0 = r0 + v.z * tw0.z; MIGAN_OPAQUE_F(r0); r1 = r1 + v.z * tw1.z; MIGAN_OPAQUE_F(r1); r2 = r2 + v.z * tw2.z; MIGAN_OPAQUE_F(r2);
  r0 = r0 + v.w * tw0.w; MIGAN_OPAQUE_F(r0); r1 = r1 + v.w * tw1.w; MIGAN_OPAQUE_F(r1); r2 = r2 + v.w * tw2.w; MIGAN_OPAQUE_F(r2);
}

// Four output channels of EncoderBlock.fromrgb (reference :186, :194): bias + sum over the 4 input planes, as scalar FMA chains
// separated by optimisation barriers for the same reason as torgb_partial (the operands raw.y / raw.w live in the high halves of
// register pairs: exactly what the hazardous op_sel form of v_pk_fma_f32 reads).
MIGAN_DEVICE MIGAN_INLINE f4 fromrgb_quad(f4 raw, f4 w0, f4 w1, f4 w2, f4 w3, f4 bb) {
  float t0 = w0.x * raw.x, t1 = w0.y * raw.x, t2 = w0.z * raw.x, t3 = w0.w * raw.x;
  MIGAN_OPAQUE_F(t0); MIGAN_OPAQUE_F(t1); MIGAN_OPAQUE_F(t2); MIGAN_OPAQUE_F(t3);
  t0 = t0 + w1.x * raw.y; MIGAN_OPAQUE_F(t0); t1 = t1 + w1.y * raw.y; MIGAN_OPAQUE_F(t1);
  t2 = t2 + w1.z * raw.y; MIGAN_OPAQUE_F(t2); t3 = t3 + w1.w * raw.y; MIGAN_OPAQUE_F(t3);
  t0 = t0 + w2.x * raw.z; MIGAN_OPAQUE_F(t0); t1 = t1 + w2.y * raw.z; MIGAN_OPAQUE_F(t1);
  t2 = t2 + w2.z * raw.z; MIGAN_OPAQUE_F(t2); t3 = t3 + w2.w * raw.z; MIGAN_OPAQUE_F(t3);
  t0 = t0 + w3.x * raw.w; MIGAN_OPAQUE_F(t0); t1 = t1 + w3.y * raw.w; MIGAN_OPAQUE_F(t1);
  t2 = t2 + w3.z * raw.w; MIGAN_OPAQUE_F(t2); t3 = t3 + w3.w * raw.w; MIGAN_OPAQUE_F(t3);
  return f4{bb.x + t0, bb.y + t1, bb.z + t2, bb.w + t3};
}

// XCD-aware workgroup order (MI355X: block b runs on XCD b%8, each XCD has a private 4 MiB L2):
// give every XCD one contiguous range of logical tiles so halo rows shared by neighbouring tiles
// and the Cout chunks of one tile hit the same L2.  Bijective for any grid size.
MIGAN_DEVICE MIGAN_INLINE int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// 2x polyphase FIR of Upsample2d at one output pixel of a planar image (reference :95-103, taps
// [1,3,3,1]/4 per axis on a zero-inserted signal): out[2i] = g[i-1]/4 + 3g[i]/4,
// out[2i+1] = 3g[i]/4 + g[i+1]/4, zeros outside.
MIGAN_DEVICE MIGAN_INLINE float up_prev3(const float* plane, int hp, int wp, int oy, int ox) {
  const int iy = oy >> 1, ix = ox >> 1;
  const int y0 = (oy & 1) ? iy : iy - 1, x0 = (ox & 1) ? ix : ix - 1;   // first of the two taps
  const float wy0 = (oy & 1) ? 0.75f : 0.25f, wx0 = (ox & 1) ? 0.75f : 0.25f;
  const int y1 = y0 + 1, x1 = x0 + 1;
  const bool vy0 = y0 >= 0, vy1 = y1 < hp, vx0 = x0 >= 0, vx1 = x1 < wp;
  const int cy0 = vy0 ? y0 : 0, cy1 = vy1 ? y1 : hp - 1, cx0 = vx0 ? x0 : 0, cx1 = vx1 ? x1 : wp - 1;
  const float p00 = plane[(size_t)cy0 * wp + cx0], p01 = plane[(size_t)cy0 * wp + cx1];
  const float p10 = plane[(size_t)cy1 * wp + cx0], p11 = plane[(size_t)cy1 * wp + cx1];
  const float r0 = (vx0 ? wx0 * p00 : 0.0f) + (vx1 ? (1.0f - wx0) * p01 : 0.0f);
  const float r1 = (vx0 ? wx0 * p10 : 0.0f) + (vx1 ? (1.0f - wx0) * p11 : 0.0f);
  return (vy0 ? wy0 * r0 : 0.0f) + (vy1 ? (1.0f - wy0) * r1 : 0.0f);
}

// the same in two steps: load the (clamped) 2x2 taps early, combine them later
MIGAN_DEVICE MIGAN_INLINE void up_taps(const float* plane, int hp, int wp, int oy, int ox, float (&t)[4]) {
  const int iy = oy >> 1, ix = ox >> 1;
  const int y0 = (oy & 1) ? iy : iy - 1, x0 = (ox & 1) ? ix : ix - 1;
  const int cy0 = y0 >= 0 ? y0 : 0, cy1 = y0 + 1 < hp ? y0 + 1 : hp - 1, cx0 = x0 >= 0 ? x0 : 0, cx1 = x0 + 1 < wp ? x0 + 1 : wp - 1;
  t[0] = plane[(size_t)cy0 * wp + cx0]; t[1] = plane[(size_t)cy0 * wp + cx1];
  t[2] = plane[(size_t)cy1 * wp + cx0]; t[3] = plane[(size_t)cy1 * wp + cx1];
}
MIGAN_DEVICE MIGAN_INLINE float up_combine(const float (&t)[4], int oy, int ox, int hp, int wp) {
  const int iy = oy >> 1, ix = ox >> 1;
  const int y0 = (oy & 1) ? iy : iy - 1, x0 = (ox & 1) ? ix : ix - 1;
  const float wy0 = (oy & 1) ? 0.75f : 0.25f, wx0 = (ox & 1) ? 0.75f : 0.25f;
  const bool vy0 = y0 >= 0, vy1 = y0 + 1 < hp, vx0 = x0 >= 0, vx1 = x0 + 1 < wp;
  const float r0 = (vx0 ? wx0 * t[0] : 0.0f) + (vx1 ? (1.0f - wx0) * t[1] : 0.0f);
  const float r1 = (vx0 ? wx0 * t[2] : 0.0f) + (vx1 ? (1.0f - wx0) * t[3] : 0.0f);
  return (vy0 ? wy0 * r0 : 0.0f) + (vy1 ? (1.0f - wy0) * r1 : 0.0f);
}

// scripts/demo.py of the reference, at network resolution
MIGAN_DEVICE MIGAN_INLINE float unit_image(unsigned b) {
  // demo.py:61: torch.Tensor(img).float() * 2 / 255 - 1, three fp32 roundings in that order
  float v = (float)b;
  v = v * 2.0f;
  v = v / 255.0f;
  return v - 1.0f;
}
MIGAN_DEVICE MIGAN_INLINE unsigned char unit_to_u8(float v) {
  // demo.py:135-136: (y * 0.5 + 0.5).clamp(0, 1) * 255 -> .to(torch.uint8) (truncation)
  float t = v * 0.5f + 0.5f;                       // y * 0.5 is exact, so a contracted FMA rounds identically
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  return (unsigned char)(int)(t * 255.0f);
}
// x = cat([mask - 0.5, img * mask]) of one pixel (demo.py:56-66)
MIGAN_DEVICE MIGAN_INLINE f4 pack_pixel(const unsigned char* img, const unsigned char* mask, size_t pix) {
  const float mk = mask[pix] == 255 ? 1.0f : 0.0f;                               // demo.py:60: np.array(mask) // 255
  const unsigned char* q = img + pix * 3;
  return f4{mk - 0.5f, unit_image(q[0]) * mk, unit_image(q[1]) * mk, unit_image(q[2]) * mk};
}
// the same in two halves, for kernels that request a pixel's bytes early and use them later: the four loads without any arithmetic on
// them (a use would make the wave wait for the loads -- and, vmcnt retiring in order, for every store it issued before them) ...
MIGAN_DEVICE MIGAN_INLINE u4v fetch_pixel_bytes(const unsigned char* img, const unsigned char* mask, size_t pix) {
  const unsigned char* q = img + pix * 3;
  return u4v{(unsigned)q[0], (unsigned)q[1], (unsigned)q[2], (unsigned)mask[pix]};
}
// ... and the arithmetic of pack_pixel on them
MIGAN_DEVICE MIGAN_INLINE f4 pack_pixel_bytes(u4v b) {
  const float mk = b.w == 255u ? 1.0f : 0.0f;
  return f4{mk - 0.5f, unit_image(b.x) * mk, unit_image(b.y) * mk, unit_image(b.z) * mk};
}
// composed = img * mask + result * (1 - mask) of one pixel, mask in {0, 1} (demo.py:139-140)
MIGAN_DEVICE MIGAN_INLINE void compose_pixel(const unsigned char* img, const unsigned char* mask, unsigned char* out, size_t pix, float y0,
                                             float y1, float y2) {
  const bool keep = mask[pix] == 255;
  const unsigned char* q = img + pix * 3;
  unsigned char* o = out + pix * 3;
  o[0] = keep ? q[0] : unit_to_u8(y0);
  o[1] = keep ? q[1] : unit_to_u8(y1);
  o[2] = keep ? q[2] : unit_to_u8(y2);
}
// ... of compose_pixel on bytes fetched earlier (fetch_pixel_bytes)
MIGAN_DEVICE MIGAN_INLINE void compose_pixel_bytes(u4v b, unsigned char* out, size_t pix, float y0, float y1, float y2) {
  const bool keep = b.w == 255u;
  unsigned char* o = out + pix * 3;
  o[0] = keep ? (unsigned char)b.x : unit_to_u8(y0);
  o[1] = keep ? (unsigned char)b.y : unit_to_u8(y1);
  o[2] = keep ? (unsigned char)b.z : unit_to_u8(y2);
}

#ifdef MIGAN_PHASE_PROF
#define PROF_BEGIN() long long prof_t = (long long)MIGAN_CLOCK(); long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_MARK(i) do { const long long n_ = (long long)MIGAN_CLOCK(); prof_acc[i] += n_ - prof_t; prof_t = n_; } while (0)
#define PROF_END_WIDE() do { if (p.prof && (tid == 0 || tid == 256)) { for (int i_ = 0; i_ < 8; ++i_) MIGAN_ATOMIC_ADD_U64(p.prof + i_, (unsigned long long)prof_acc[i_]); if (tid == 0) MIGAN_ATOMIC_ADD_U64(p.prof + 8, 1ull); } } while (0)
#define PROF_END() do { if (p.prof && tid == 0) { for (int i_ = 0; i_ < 8; ++i_) MIGAN_ATOMIC_ADD_U64(p.prof + i_, (unsigned long long)prof_acc[i_]); MIGAN_ATOMIC_ADD_U64(p.prof + 8, 1ull); } } while (0)
#else
#define PROF_BEGIN() do {} while (0)
#define PROF_MARK(i) do {} while (0)
#define PROF_END() do {} while (0)
#define PROF_END_WIDE() do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
// fused SeparableConv2d
//
//   MODE   : NORMAL (down=1, up=1) | DOWN (FIR stride 2 before the 1x1) | UP (FIR x2 after the 1x1)
//   MT     : GEMM rows (pixels) per workgroup, 128 or 64
//   NT     : GEMM columns (output channels) per workgroup, 64 / 128 / 256
//   KC     : input-channel chunk staged per K step, 32 or 16
//   FROMRGB: input tile is act(fromrgb(network input)) computed on the fly (encoder first block)
//   NI     : float4 input-tile items per thread per K chunk (prefetch registers)
//   MINW   : launch bound, minimum waves per SIMD (= workgroups per CU)
//   MAING  : compile-time tile geometry (8x16 pixels, one image per tile)
//   PERSIST: workgroups walk several tiles and prefetch the next tile during the epilogue
//   GEMMV  : 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 1 = error-compensated bf16 MFMA
//            (v_mfma_f32_32x32x16_bf16 x 6 on 3-way split operands, fp32 accumulate); 2 = error-
//            compensated fp16 MFMA (v_mfma_f32_32x32x16_f16 x 3 on scaled 2-way split operands)
//
// Waves are laid out 2x2 over the MT x NT tile; each wave owns (MT/2)x(NT/2) as 32x32 MFMA tiles.
//   STV    : activation storage format (Io<STV>): 0 fp32, 1 bf16, 2 fp16
#define BF_OF(G) ((G) >= 1)
template <int MODE, int MT, int NT, int KC, bool FROMRGB, int NI, int MINW, bool MAING, bool PERSIST, int GEMMV, bool TORGB, int STV = 0>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, MINW) sepconv_kernel(const SepArgs p) {
  static_assert(MODE != MODE_DOWN, "FIR-down layers run as dwfir_kernel + a MODE_PW pointwise GEMM");
  MIGAN_DYN_SMEM(smem);

  constexpr int QC = KC / 4;                       // float4 groups per pixel in a K chunk
  constexpr int LG_QC = (QC == 16) ? 4 : ((QC == 8) ? 3 : 2);
  static_assert(QC == 16 || QC == 8 || QC == 4, "KC must be 64, 32 or 16");
  constexpr int AS = KC + 4;                       // A/B row pitch (floats): odd number of 16-B slots -> conflict-free b128
  constexpr int GS = NT + 4;                       // result tile row pitch
  constexpr int QN = NT / 4;
  constexpr int LG_QN = (QN == 64) ? 6 : ((QN == 32) ? 5 : ((QN == 16) ? 4 : 3));
  static_assert(QN == 64 || QN == 32 || QN == 16 || QN == 8, "NT must be 256, 128, 64 or 32");
  // KSPLIT (NT == 32; the smallest launches): all four waves own the SAME 32 x 32 output tile and split the K steps of a chunk among
  // themselves (wave w takes 16-channel step w of the 64-channel chunk); the four partial tiles are summed through LDS before the
  // epilogue.  A quarter of the weight panel per workgroup (the 1x1 weights are what a small launch streams), four times the workgroups.
  constexpr bool KSPLIT = (NT == 32);
  static_assert(!KSPLIT || (MT == 32 && KC == 64 && GEMMV >= 1 && !TORGB && !PERSIST && !FROMRGB), "K-split tiles: 32 x 32, 64-channel chunks, split GEMM variants");
  constexpr int WM = KSPLIT ? 1 : ((MT >= 128) ? 2 : 1), WN = KSPLIT ? 1 : 4 / WM;   // wave grid over the MT x NT tile: 2x2, or 1x4 for the 64 x 256 tile
  constexpr int WROWS = MT / WM, WCOLS = NT / WN;  // per-wave tile
  constexpr int MTI = WROWS / 32, NTI = WCOLS / 32;
  static_assert(MTI >= 1 && NTI >= 1, "wave tile must be at least 32x32");
  constexpr bool BF = (GEMMV >= 1);                // operands are planes of 16-bit pieces
  constexpr bool F16 = (GEMMV >= 2);               // fp16 pieces of scaled operands
  constexpr bool X1 = (GEMMV == 3);                // one fp16 piece per operand (operands rounded to 11-bit significands)
  constexpr int NPL = X1 ? 1 : (F16 ? 2 : 3);      // planes per operand
  constexpr int PB = KC * 2;                       // BF: bytes per row of one 16-bit operand plane (XOR-swizzled 16-B slots, no padding)
  constexpr int NSLOT = PB / 16;
  constexpr int NB = BF ? NPL * NT * NSLOT / kThreads : NT * QC / kThreads;   // float4 items of the 1x1 weight tile per thread
  static_assert(NB >= 1 && NB * kThreads == (BF ? NPL * NT * NSLOT : NT * QC), "weight tile must split evenly over the threads");
  static_assert(!BF || KC == 64 || KC == 32 || KC == 16, "the split GEMM variants are built for 64-, 32- or 16-channel chunks");
  static_assert(KC != 64 || (BF && !FROMRGB && !PERSIST && !TORGB), "64-channel chunks: the small-launch tiles of the split GEMM variants");
  // input tensor format: the network input planes (FROMRGB) and dwfir_kernel's output (MODE_PW) are always fp32
  // ... except for the "f16" GEMM variant, whose dwfir_kernel<.., 3|4> writes the A operand itself (fp16 of value x 2^7)
  constexpr bool PWH = MODE == MODE_PW && GEMMV == 3;
  typedef Io<PWH ? 2 : ((FROMRGB || MODE == MODE_PW) ? 0 : STV)> IoIn;
  typedef Io<STV> IoOut;
  constexpr int NW4 = KC * 10 / 4;                 // float4s of depthwise weights (9 taps) + bias per chunk
  constexpr int NF4 = FROMRGB ? KC * 5 / 4 : 0;    // float4s of fromrgb weights (4 per channel) + bias
  constexpr int SEGH = (MT >= 128) ? 4 : 2;        // output rows per depthwise strip (NORMAL / UP)
  // The plain 3-workgroup tiles (MINW 3, host: Cin == 64) know their two K chunks at compile time: the loop unrolls, the prefetch
  // registers are dead in the second chunk and the accumulators in the first (its first product reads C = 0), so the register
  // peak drops from 184 to what three waves per SIMD allow (168) without spilling.
  constexpr int NKC = (MINW == 3 && KC == 32 && NT == 64 && MODE == MODE_NORMAL && !FROMRGB && BF_OF(GEMMV)) ? 2 : 0;
  constexpr bool ZEROC = NKC != 0;

  const char* __restrict__ gx_ = reinterpret_cast<const char*>(p.x);
  char* __restrict__ gy_ = reinterpret_cast<char*>(p.y);
  const char* __restrict__ gskip = reinterpret_cast<const char*>(p.skip);
  const float* __restrict__ gwdw = p.wdw;
  const float* __restrict__ gbdw = p.bdw;
  const float* __restrict__ gwpw = p.wpw;
  const float* __restrict__ gnoise = p.noise;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = KSPLIT ? 0 : wave / WN, wn = KSPLIT ? 0 : wave % WN;
  const int l31 = lane & 31, half = lane >> 5;
  PROF_BEGIN();

  // Main tiles (every layer at >= 16x16 output) have compile-time geometry so the index math below
  // folds to shifts and multiply-highs; the NI == 9 instantiations serve the small-resolution layers
  // (several images per tile) with run-time geometry.
  constexpr bool MAINGEO = MAING;
  const int lgGH = MAINGEO ? (MT >= 128 ? 3 : 2) : p.lgGH;   // main tiles: 8x16 pixels (MT 128) or 4x16 (MT 64)
  const int lgGW = MAINGEO ? 4 : p.lgGW;
  const int lgIMGS = MAINGEO ? 0 : p.lgIMGS;
  const int lgRS = MAINGEO ? 1 : p.lgRS;
  const int GH = 1 << lgGH, GW = 1 << lgGW, IMGS = 1 << lgIMGS;

  // ---- persistent tile schedule -------------------------------------------------------------
  // Logical tiles (n-chunk fastest, then x, y, image group) are split into 8 contiguous ranges, one
  // per XCD (block b runs on XCD b%8, each XCD has a private L2: halo rows shared by neighbouring
  // tiles and the Cout chunks of one tile then hit the same L2).  The workgroups of an XCD walk
  // their range with a stride equal to their count, so at any moment an XCD works on consecutive
  // tiles.  With gridDim == #tiles every workgroup does exactly one tile.
  const int ntiles = p.tiles_x * p.tiles_y * p.nchunks * ((p.B + IMGS - 1) >> lgIMGS);
  const int xcd = (int)blockIdx.x & 7;
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int tcnt = tq + (xcd < tr ? 1 : 0);                                  // tiles of this XCD
  const int tbase = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int tstep = ((int)gridDim.x + 7 - xcd) >> 3;                         // workgroups on this XCD
  int tl = (int)blockIdx.x >> 3;                                             // my first tile in the range
  auto decode = [&](int t, int& n0_, int& b0_, int& gy0_, int& gx0_) {
    const int nch = t % p.nchunks; t /= p.nchunks;
    const int tx = t % p.tiles_x;  t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    n0_ = nch * NT;
    b0_ = (t / p.tiles_y) << lgIMGS;
    gy0_ = ty * p.sy - p.off;                        // GEMM grid origin (GEMM-resolution image coords)
    gx0_ = tx * p.sx - p.off;
  };
  int n0, b0, gy0, gx0;
  decode(tbase + tl, n0, b0, gy0, gx0);

  // ---- LDS carve --------------------------------------------------------------------------
  float* in_s = smem;                       // [npix_in][KC]
  float* a_s = smem + p.off_a;              // [MT][AS]
  float* b_s = smem + p.off_b;              // [NT][AS] x (1 or 2 buffers)
  float* v_s = smem + p.off_v;              // DOWN: depthwise grid [IMGS][2GH+2][2GW+2][KC]
  static_assert(!TORGB || (MODE == MODE_NORMAL && !PERSIST), "ToRGB is fused into plain, non-persistent layers only");
  float* rgb_s = smem + p.off_rgb;          // FROMRGB: [npix_in][4]
  float* w_s = smem + p.off_w;              // [KC*9] depthwise taps, [KC] bias, (FROMRGB: [KC*4] + [KC])
  float* g_s = smem;                        // after the K loop: [MT][GS], aliases the buffers above
  // A-operand row m, channels 4*c4..4*c4+3 of the current chunk
  // XOR swizzle of the 16-byte slots of a row of a 16-bit operand plane (conflict-free ds_read_b128 of 32 consecutive
  // rows): 64-byte rows (KC 32) rotate through their 4 slots every 4 rows, 32-byte rows (KC 16) swap their 2 slots every 8 rows,
  // 128-byte rows (KC 64) rotate through their 8 slots every 2 rows: in each case 16 consecutive rows cover every 16-byte slot
  // of a 256-byte LDS bank row exactly once
  auto swz = [](int row) { return NSLOT == 8 ? ((row >> 1) & 7) : (NSLOT == 4 ? ((row >> 2) & 3) : ((row >> 3) & 1)); };
  auto emit_a = [&](float* abase, int m, int c4, f4 v) {
    if constexpr (X1) {
      char* d = reinterpret_cast<char*>(abase) + m * PB + (((c4 >> 1) ^ swz(m)) << 4) + ((c4 & 1) << 3);
      *reinterpret_cast<u2v*>(d) = u2v{MIGAN_PACK_F16(v.x, v.y), MIGAN_PACK_F16(v.z, v.w)};
    } else if constexpr (F16) {
      // v already carries the 2^7 activation scale
      u2v h1, h2;
      split2_f16(v, h1, h2);
      char* d = reinterpret_cast<char*>(abase) + m * PB + (((c4 >> 1) ^ swz(m)) << 4) + ((c4 & 1) << 3);
      *reinterpret_cast<u2v*>(d) = h1;
      *reinterpret_cast<u2v*>(d + MT * PB) = h2;
    } else if constexpr (BF) {
      u2v h1, h2, h3;
      split3_bf16(v, h1, h2, h3);
      char* d = reinterpret_cast<char*>(abase) + m * PB + (((c4 >> 1) ^ swz(m)) << 4) + ((c4 & 1) << 3);
      *reinterpret_cast<u2v*>(d) = h1;
      *reinterpret_cast<u2v*>(d + MT * PB) = h2;
      *reinterpret_cast<u2v*>(d + 2 * MT * PB) = h3;
    } else {
      st4(abase + m * AS + c4 * 4, v);
    }
  };

  // input-tile geometry (input-resolution coordinates)
  constexpr int HALO = (MODE == MODE_PW) ? 0 : 1;
  const int IGH = GH + 2 * HALO, IGW = GW + 2 * HALO;
  const int npix_in = IMGS * IGH * IGW;
  const int nitems_in = npix_in * QC;

  // per-thread descriptors of its input items (constant across the K chunks of a tile).  Item
  // i = tid + j*256 is float4 number i of the LDS tile ([pixel][KC/4]); goff = element offset of
  // its source (in BYTES) inside image group b0 (0 when the pixel is padding: loaded anyway, zeroed on the way
  // to LDS, which is the conv zero padding of reference :126); bit j of `vmask` = real pixel, of
  // `emask` = item exists.
  unsigned emask = 0;
#pragma unroll
  for (int j = 0; j < NI; ++j)
    if (tid + j * kThreads < nitems_in) emask |= 1u << j;
  auto make_items = [&](int b0_, int gy0_, int gx0_, unsigned (&goff_)[NI], unsigned& vmask_) {
    vmask_ = 0;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int i = tid + j * kThreads;
      unsigned g = 0;
      if (i < nitems_in) {
        const int c4 = i & (QC - 1);
        const int pix = i >> LG_QC;
        const int ix = pix % IGW;
        const int r = pix / IGW;
        const int iy = r % IGH, img = r / IGH;
        const int yy = gy0_ - HALO + iy, xx = gx0_ - HALO + ix;
        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && (b0_ + img) < p.B) {
          vmask_ |= 1u << j;
          g = FROMRGB ? 0u : (unsigned)(((img * p.H + yy) * p.W + xx) * p.CI + c4 * 4) * IoIn::ESZ;     // bytes
        }
      }
      goff_[j] = g;
    }
  };
  // 1x1 weight tile items: n = i / QC rows of conv2.weight, 4 consecutive input channels
  auto make_boff = [&](int n0_, unsigned (&boff_)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int i = tid + j * kThreads;
      if constexpr (BF) {
        // item = (plane, row n, 16-byte slot of 8 pieces); offset in 16-bit elements inside one 32-channel block of the
        // chunk-major planes [NPL][CI/32][CO][32] (the weight tile of a 32-channel K chunk is one contiguous run)
        const int plane = i / (NT * NSLOT), rem = i % (NT * NSLOT);
        // (a 64-channel chunk spans two 32-channel blocks: slots 4-7 come from the next block, 32 * CO elements further on)
        const int slot = rem % NSLOT;
        boff_[j] = (unsigned)(plane * p.CO * p.CI + (slot >> 2) * 32 * p.CO + (n0_ + rem / NSLOT) * 32 + (slot & 3) * 8) * 2u;    // bytes
      } else {
        boff_[j] = (unsigned)((n0_ + (i >> LG_QC)) * p.CI + (i & (QC - 1)) * 4) * 4u;                      // bytes
      }
    }
  };
  // FROMRGB: raw 4-channel network input (NCHW) of this thread's halo pixel (npix_in <= 256)
  auto load_raw = [&](int b0_, int gy0_, int gx0_) -> f4 {
    f4 v = {0.f, 0.f, 0.f, 0.f};
    const int pix = tid;
    if (pix < npix_in) {
      const int ix = pix % IGW;
      const int r = pix / IGW;
      const int iy = r % IGH, img = r / IGH;
      const int yy = gy0_ - HALO + iy, xx = gx0_ - HALO + ix;
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && (b0_ + img) < p.B) {
        if (p.u8_img) {
          v = pack_pixel(p.u8_img, p.u8_mask, ((size_t)(b0_ + img) * p.H + yy) * p.W + xx);
        } else {
          const float* src = reinterpret_cast<const float*>(gx_) + ((size_t)(b0_ + img) * 4 * p.H + yy) * p.W + xx;
          const size_t plane = (size_t)p.H * p.W;
          v = f4{src[0], src[plane], src[2 * plane], src[3 * plane]};
        }
      }
    }
    return v;
  };
  unsigned goff[NI], vmask, boff[NB];
  make_items(b0, gy0, gx0, goff, vmask);
  make_boff(n0, boff);

  f16v acc[MTI][NTI];

  // prefetch registers: one K chunk of the input tile, of the 1x1 weights and of the small weights
  typename IoIn::raw4 rin[NI];
  f4 rb[NB], rw, rraw;
  // every global access below is (wave-uniform base pointer, held in SGPRs) + (32-bit lane offset):
  // no 64-bit VALU address arithmetic in the K loop.
  auto issue_loads = [&](int b0_, const unsigned (&goff_)[NI], const unsigned (&boff_)[NB], int k0) {
    if constexpr (!FROMRGB) {
      const char* __restrict__ xk = gx_ + ((size_t)b0_ * p.H * p.W * p.CI + k0) * IoIn::ESZ;
#pragma unroll
      for (int j = 0; j < NI; ++j) rin[j] = IoIn::ld(xk, goff_[j]);
    }
    if constexpr (BF) {
      const unsigned short* __restrict__ wk = p.wsplit + (size_t)(k0 >> 5) * 32 * p.CO + (k0 & 31);   // 32-channel block k0/32, column k0%32
#pragma unroll
      for (int j = 0; j < NB; ++j) rb[j] = ld4(at_bytes(reinterpret_cast<const float*>(wk), boff_[j]));
    } else {
      const float* __restrict__ wk = gwpw + k0;
#pragma unroll
      for (int j = 0; j < NB; ++j) rb[j] = ld4(at_bytes(wk, boff_[j]));
    }
    // depthwise taps of channels [k0,k0+KC) are KC*9 contiguous floats of conv1.weight, then the bias
    if constexpr (MODE != MODE_PW) {
      if (tid < KC * 9 / 4) rw = ld4(gwdw + (size_t)k0 * 9 + (unsigned)(tid * 4));
      else if (tid < NW4) rw = ld4(gbdw + k0 + (unsigned)((tid - KC * 9 / 4) * 4));
    }
    if constexpr (FROMRGB) {
      if (tid >= NW4 && tid < NW4 + KC) rw = ld4(p.frgb_w + (size_t)k0 * 4 + (unsigned)((tid - NW4) * 4));
      else if (tid >= NW4 + KC && tid < NW4 + NF4) rw = ld4(p.frgb_b + k0 + (unsigned)((tid - NW4 - KC) * 4));
    }
  };

  PROF_MARK(0);
  const int nkc = NKC ? NKC : p.CI / KC;
  issue_loads(b0, goff, boff, 0);
  if constexpr (FROMRGB) rraw = load_raw(b0, gy0, gx0);

  // =================================== tile loop ===========================================
  for (;;) {
  const bool has_next = PERSIST && (tl + tstep < tcnt);   // PERSIST = false: exactly one tile per workgroup
  int n0n = 0, b0n = 0, gy0n = 0, gx0n = 0;
  unsigned goffn[NI], vmaskn = 0, boffn[NB];
  if constexpr (!ZEROC) {
#pragma unroll
    for (int i = 0; i < MTI; ++i)
#pragma unroll
      for (int j = 0; j < NTI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  }
  // interior tiles (no padding anywhere in the wave's items) skip the zero-fill selects
  const bool wave_all_valid = __all(vmask == emask);

  // ======================================= K loop ==========================================
#pragma unroll(NKC ? NKC : 1)
  for (int c = 0; c < nkc; ++c) {
    const int k0 = c * KC;
    float* bcur = b_s + (c & 1) * p.b_stride;
    if (p.b_stride == 0 || (MODE == MODE_PW && p.a_stride == 0)) __syncthreads();   // single buffers: wait for the MFMAs of chunk c-1

    // ---- S1: prefetched chunk -> LDS -----------------------------------------------------------
    // depthwise taps go to LDS tap-major ([9][KC]) so the strip below reads one float4 per tap
    if constexpr (MODE == MODE_PW) {
    } else if (tid < KC * 9 / 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = tid * 4 + e;                   // flat index into [KC][9]
        w_s[(f % 9) * KC + f / 9] = rw[e];
      }
    } else if (tid < NW4) {
      st4(w_s + tid * 4, rw);                        // depthwise bias
    } else if (FROMRGB && tid < NW4 + KC) {
      // fromrgb.weight row of channel ch = tid - NW4 (4 inputs) -> input-major [4][KC] so the tile builder below
      // reads one float4 of 4 channels per input (packed FMAs)
      const int ch = tid - NW4;
#pragma unroll
      for (int e = 0; e < 4; ++e) w_s[KC * 10 + e * KC + ch] = rw[e];
    } else if (tid < NW4 + NF4) {
      st4(w_s + tid * 4, rw);                        // fromrgb bias
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int i = tid + j * kThreads;
      if constexpr (BF) {
        const int plane = i / (NT * NSLOT), rem = i % (NT * NSLOT);
        const int n = rem / NSLOT, slot = rem % NSLOT;
        char* dst = reinterpret_cast<char*>(bcur) + (plane * NT + n) * PB + ((slot ^ swz(n)) << 4);
        st4(reinterpret_cast<float*>(dst), rb[j]);
      } else {
        st4(bcur + (i >> LG_QC) * AS + (i & (QC - 1)) * 4, rb[j]);
      }
    }
    if constexpr (FROMRGB) {
      if (c == 0 && tid < npix_in) st4(rgb_s + tid * 4, rraw);
      __syncthreads();                              // w_s (fromrgb weights of this chunk) and rgb_s visible
      // x = act(fromrgb(img)) (reference :194-195), 4 -> CI pointwise with bias, per halo pixel.  Scalar FMA chains the vectoriser
      // cannot fuse (fromrgb_quad): the packed form of this dot product is where hipcc 7.2 emitted v_pk_fma_f32 with low-lane
      // op_sel swizzles for the raw.y products -- the instruction form that gives intermittently wrong results on MI355X
      // (profiles/r02_torgb_packed_f32_hazard.md; measured: 1-2 % of the first layer's outputs wrong, different ones every
      // launch).  mi-gan_amd/build.py scans every freshly linked library for that form and refuses it.
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (emask & (1u << j)) {
          const int i = tid + j * kThreads;
          const int c4 = i & (QC - 1);
          const int pix = i >> LG_QC;
          f4 v = {0.f, 0.f, 0.f, 0.f};
          if (vmask & (1u << j)) {
            const f4 raw = ld4(rgb_s + pix * 4);
            const float* wr = w_s + KC * 10 + c4 * 4;
            const f4 w0 = ld4(wr), w1 = ld4(wr + KC), w2 = ld4(wr + 2 * KC), w3 = ld4(wr + 3 * KC);   // per input, 4 channels
            const f4 bb = ld4(w_s + KC * 14 + c4 * 4);
            v = act4(fromrgb_quad(raw, w0, w1, w2, w3, bb));
          }
          st4(in_s + i * 4, v);
        }
      }
    } else {
      if constexpr (MODE == MODE_PW) {
        // pointwise GEMM: the input pixels ARE the A operand rows (item i = row i/QC, k-quad i%QC)
        float* acur = a_s + (c & 1) * p.a_stride;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int i = tid + j * kThreads;
          if constexpr (PWH) {
            // the input already is the operand: 4 scaled fp16 values per item, straight into the swizzled plane
            u2v h = rin[j];
            if (!(vmask & (1u << j))) h = u2v{0u, 0u};
            const int m = i >> LG_QC, c4 = i & (QC - 1);
            char* d = reinterpret_cast<char*>(acur) + m * PB + (((c4 >> 1) ^ swz(m)) << 4) + ((c4 & 1) << 3);
            *reinterpret_cast<u2v*>(d) = h;
          } else {
            f4 v = IoIn::cvt(rin[j]);
            if (!(vmask & (1u << j))) v = f4{0.f, 0.f, 0.f, 0.f};
            if constexpr (F16) v = v * kF16AScale;
            emit_a(acur, i >> LG_QC, i & (QC - 1), v);
          }
        }
      } else if (wave_all_valid) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
          if (emask & (1u << j)) st4(in_s + (tid + j * kThreads) * 4, IoIn::cvt(rin[j]));
      } else {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if (emask & (1u << j)) {
            f4 v = IoIn::cvt(rin[j]);
            if (!(vmask & (1u << j))) v = f4{0.f, 0.f, 0.f, 0.f};
            st4(in_s + (tid + j * kThreads) * 4, v);
          }
        }
      }
    }
    __syncthreads();
    PROF_MARK(1);
    if (c + 1 < nkc) {
      issue_loads(b0, goff, boff, k0 + KC);         // in flight during the depthwise stage and the MFMAs
    } else if (has_next) {
      // last chunk: prefetch the first chunk of my NEXT tile; it lands during the MFMAs and the epilogue
      decode(tbase + tl + tstep, n0n, b0n, gy0n, gx0n);
      make_items(b0n, gy0n, gx0n, goffn, vmaskn);
      make_boff(n0n, boffn);
      issue_loads(b0n, goffn, boffn, 0);
      if constexpr (FROMRGB) rraw = load_raw(b0n, gy0n, gx0n);
    }

    // ---- S2: depthwise 3x3 + bias + act (+ FIR down) -> A operand in LDS --------------------
    // One thread walks a column of the tile for 4 channels.  Every input row it reads (3 float4
    // from LDS) is scattered into three running sums (the outputs it is the bottom / middle / top
    // tap row of), so each LDS value is read once per column and no register window is kept.
    if constexpr (MODE != MODE_PW) {
      const int RS = 1 << lgRS;
      const int ncols = (IMGS * GW * QC) << lgRS;
      for (int it = tid; it < ncols; it += kThreads) {
        const int c4 = it & (QC - 1);
        int r = it >> LG_QC;
        const int gx = r & (GW - 1); r >>= lgGW;
        const int seg = r & (RS - 1);
        const int img = r >> lgRS;
        const int r0 = seg * SEGH;
        f4 w[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) w[tap] = ld4(w_s + tap * KC + c4 * 4);
        const f4 bias = ld4(w_s + KC * 9 + c4 * 4);
        // sliding 3x3 register window down the strip: 3 LDS reads and 9 float4 FMAs per output
        const float* ip = in_s + ((img * IGH + r0) * IGW + gx) * KC + c4 * 4;
        const int mbase = (img << (lgGH + lgGW)) + (r0 << lgGW) + gx;
        f4 win[3][3];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          win[rr][0] = ld4(ip); win[rr][1] = ld4(ip + KC); win[rr][2] = ld4(ip + 2 * KC);
          ip += IGW * KC;
        }
#pragma unroll
        for (int o = 0; o < SEGH; ++o) {
          const int nr = (o + 2) % 3;
          win[nr][0] = ld4(ip); win[nr][1] = ld4(ip + KC); win[nr][2] = ld4(ip + 2 * KC);
          ip += IGW * KC;
          f4 sacc = bias;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) sacc += w[ky * 3 + kx] * win[(o + ky) % 3][kx];
          if constexpr (F16) emit_a(a_s, mbase + (o << lgGW), c4, act4_scaled<7>(sacc));
          else emit_a(a_s, mbase + (o << lgGW), c4, act4(sacc));
        }
      }
      __syncthreads();
    }
    PROF_MARK(2);

    // ---- S3: acc += A[MT x KC] * W^T[KC x NT] on the matrix cores ---------------------------
    // v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31].  Each lane
    // reads 4 consecutive k with one ds_read_b128; the two lane halves take k = 8kk+4*half+t, the
    // same for A and B, so any assignment of k to (half,t) sums the full K.
    if constexpr (BF) {
      // v_mfma_f32_32x32x16_{bf16,f16}: lane l supplies A[i=l&31][k=8*(l>>5)..+7] and B[k=8*(l>>5)..+7][j=l&31]
      // as one 16-byte LDS read each; NPL planes per operand; six (bf16x3) or three (f16x2) MFMAs per
      // 32x32 tile and k-step, smallest products first.
      const char* ab = reinterpret_cast<const char*>(a_s + (MODE == MODE_PW ? (c & 1) * p.a_stride : 0));
      const char* bb = reinterpret_cast<const char*>(bcur);
#pragma unroll
      for (int ks0 = 0; ks0 < (KSPLIT ? 1 : KC / 16); ++ks0) {
        const int ks = KSPLIT ? wave : ks0;          // K-split tiles: this wave's 16-channel step of the chunk
        f4 av[MTI][NPL], bv[NTI][NPL];
#pragma unroll
        for (int i = 0; i < MTI; ++i) {
          const int row = wm * WROWS + i * 32 + l31;
          const char* q = ab + row * PB + (((2 * ks + half) ^ swz(row)) << 4);
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) av[i][pl] = ld4(reinterpret_cast<const float*>(q + pl * MT * PB));
        }
#pragma unroll
        for (int j = 0; j < NTI; ++j) {
          const int row = wn * WCOLS + j * 32 + l31;
          const char* q = bb + row * PB + (((2 * ks + half) ^ swz(row)) << 4);
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) bv[j][pl] = ld4(reinterpret_cast<const float*>(q + pl * NT * PB));
        }
#pragma unroll
        for (int i = 0; i < MTI; ++i)
#pragma unroll
          for (int j = 0; j < NTI; ++j) {
            if (ZEROC && c == 0 && ks0 == 0) acc[i][j] = f16v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // (folds into the first product's C operand)
            if constexpr (X1) {
              acc[i][j] = MIGAN_MFMA_F16_32X32X16(av[i][0], bv[j][0], acc[i][j]);
            } else if constexpr (F16) {
              acc[i][j] = MIGAN_MFMA_F16_32X32X16(av[i][1], bv[j][0], acc[i][j]);
              acc[i][j] = MIGAN_MFMA_F16_32X32X16(av[i][0], bv[j][1], acc[i][j]);
              acc[i][j] = MIGAN_MFMA_F16_32X32X16(av[i][0], bv[j][0], acc[i][j]);
            } else {
              acc[i][j] = MIGAN_MFMA_BF16_32X32X16(av[i][NPL - 1], bv[j][0], acc[i][j]);
              acc[i][j] = MIGAN_MFMA_BF16_32X32X16(av[i][1], bv[j][1], acc[i][j]);
              acc[i][j] = MIGAN_MFMA_BF16_32X32X16(av[i][0], bv[j][NPL - 1], acc[i][j]);
              acc[i][j] = MIGAN_MFMA_BF16_32X32X16(av[i][1], bv[j][0], acc[i][j]);
              acc[i][j] = MIGAN_MFMA_BF16_32X32X16(av[i][0], bv[j][1], acc[i][j]);
              acc[i][j] = MIGAN_MFMA_BF16_32X32X16(av[i][0], bv[j][0], acc[i][j]);
            }
          }
      }
    } else
    {
      const float* ap = a_s + (MODE == MODE_PW ? (c & 1) * p.a_stride : 0) + (wm * WROWS + l31) * AS + 4 * half;
      const float* bp = bcur + (wn * WCOLS + l31) * AS + 4 * half;
#pragma unroll
      for (int kk = 0; kk < KC / 8; ++kk) {
        f4 av[MTI], bv[NTI];
#pragma unroll
        for (int i = 0; i < MTI; ++i) av[i] = ld4(ap + i * 32 * AS + kk * 8);
#pragma unroll
        for (int j = 0; j < NTI; ++j) bv[j] = ld4(bp + j * 32 * AS + kk * 8);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int i = 0; i < MTI; ++i)
#pragma unroll
            for (int j = 0; j < NTI; ++j)
              acc[i][j] = MIGAN_MFMA_F32_32X32X2(av[i][tt], bv[j][tt], acc[i][j]);
      }
    }
    PROF_MARK(3);
  }

  // ======================================= epilogue ========================================
  // `tide` is the thread id laundered through an empty asm: everything the epilogue derives from it
  // is then recomputed per tile instead of being hoisted above the K loop by LICM (which would keep
  // ~100 loop-invariant epilogue addresses live in VGPRs across the MFMA loop and force spills).
  int tide = tid;
  MIGAN_OPAQUE(tide);
  const int lanee = tide & 63, wavee = tide >> 6;
  const int wme = KSPLIT ? 0 : wavee / WN, wne = KSPLIT ? 0 : wavee % WN, l31e = lanee & 31, halfe = lanee >> 5;
  __syncthreads();                       // all waves done with a_s/b_s before g_s overwrites them
  // F16: 1 / (activation scale * weight scale), a power of two written next to the weight planes by weight_absmax_kernel
  // The scale is applied where the epilogue touches each value anyway: folded into the noise add (one FMA) or,
  // without noise, into the activation gain (both exact: power-of-two scaling commutes with every rounding).
  float acc_scale = 1.0f;
  if constexpr (F16) acc_scale = reinterpret_cast<const float*>(p.wsplit)[-4];
  const float gain_s = 1.41421356237309515f * acc_scale;
  // accumulator fragment -> LDS result tile.  C/D layout of the 32x32 MFMA: lane holds column
  // l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
#pragma unroll
  for (int i = 0; i < MTI; ++i)
#pragma unroll
    for (int j = 0; j < NTI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wme * WROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * halfe;
        const int col = wne * WCOLS + j * 32 + l31e;
        float v = acc[i][j][r];
        if constexpr (MODE == MODE_UP) {
          // halo pixels outside the low-resolution image contribute zeros to the upsampling FIR
          // (reference pads with zeros :101), not the conv of a zero-padded input
          const int gx = row & (GW - 1);
          const int gy = (row >> lgGW) & (GH - 1);
          const int ly = gy0 + gy, lx = gx0 + gx;
          if (ly < 0 || ly >= p.H || lx < 0 || lx >= p.W) v = 0.0f;
        }
        g_s[(KSPLIT ? wavee * MT * GS : 0) + row * GS + col] = v;          // K-split: one partial tile per wave
      }
  __syncthreads();
  if constexpr (KSPLIT) {
    // sum the four partial tiles into the first (one float4 per thread: 32 rows x 8 quads), fixed order
    const int c4r = tide & (QN - 1), mr = tide >> LG_QN;
    float* gp = g_s + mr * GS + c4r * 4;
    st4(gp, (ld4(gp) + ld4(gp + MT * GS)) + (ld4(gp + 2 * MT * GS) + ld4(gp + 3 * MT * GS)));
    __syncthreads();
  }
  PROF_MARK(4);

  const bool has_noise = gnoise != nullptr;
  const float ns = has_noise ? p.noise_strength[0] : 0.0f;
  constexpr int ITEMS = MT * QN / kThreads;          // epilogue items per thread
  constexpr int UB = ITEMS < 4 ? ITEMS : 4;          // items whose global loads are issued together
  static_assert(ITEMS >= 1 && ITEMS % UB == 0, "epilogue batches must divide the per-thread items");
  // Items of a thread are GEMM rows m0 + k*STEP (k = 0..ITEMS-1), always the same 4 channels.  With
  // the main geometry the pixel offset of item k is a compile-time function of k, so every global
  // address is (uniform pointer advanced per item in SGPRs) + (one 32-bit lane offset computed once).
  constexpr int STEP = kThreads >> LG_QN;
  const int c4 = tide & (QN - 1);
  const int m0 = tide >> LG_QN;
  const int gxt = m0 & (GW - 1), gyt = (m0 >> lgGW) & (GH - 1);
  const size_t img_elems = (size_t)p.HO * p.WO * p.CO;
  char* __restrict__ yb = gy_ + (size_t)b0 * img_elems * IoOut::ESZ;
  const char* __restrict__ sb = gskip ? gskip + (size_t)b0 * img_elems * IoOut::ESZ : nullptr;
  constexpr unsigned OE = IoOut::ESZ;

  if constexpr (MODE != MODE_UP) {
    constexpr bool do_rgb = TORGB;          // ToRGB fused into this epilogue (host: CO == NT, trgb_w set)
    f4 tw0 = {0.f, 0.f, 0.f, 0.f}, tw1 = tw0, tw2 = tw0;
    if constexpr (do_rgb) {
      tw0 = ld4(p.trgb_w + n0 + c4 * 4);
      tw1 = ld4(p.trgb_w + p.CO + n0 + c4 * 4);
      tw2 = ld4(p.trgb_w + 2 * p.CO + n0 + c4 * 4);
    }
    // tail-pass pixel of this thread pair and the taps of the previous (half resolution) RGB image under it
    int rgb_oy = 0, rgb_ox = 0, rgb_b = 0;
    bool rgb_ok = false;
    float pv[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if constexpr (do_rgb) {
      const int m = tide >> 1;
      rgb_oy = gy0 + ((m >> lgGW) & (GH - 1));
      rgb_ox = gx0 + (m & (GW - 1));
      rgb_b = b0 + (m >> (lgGW + lgGH));
      rgb_ok = (tide & 1) == 0 && rgb_b < p.B && tide < 2 * MT && rgb_oy < p.HO && rgb_ox < p.WO;
      if (rgb_ok && p.img_prev) {
        const size_t plane4 = ((size_t)p.HO * p.WO) >> 2;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) up_taps(p.img_prev + ((size_t)rgb_b * 3 + ch) * plane4, p.HO >> 1, p.WO >> 1, rgb_oy, rgb_ox, pv[ch]);
      }
    }
    const unsigned pix_t = (unsigned)((gy0 + gyt) * p.WO + gx0 + gxt);   // first pixel of this thread
    const unsigned off_t = pix_t * (unsigned)p.CO + (unsigned)(n0 + c4 * 4);
    auto epi_items = [&](auto hn_, auto hs_) {
      constexpr bool HN = decltype(hn_)::value, HS = decltype(hs_)::value;
#pragma unroll
    for (int it0 = 0; it0 < ITEMS; it0 += UB) {
      f4 val[UB];
      typename IoOut::raw4 sk[UB];
      float nz[UB];
      unsigned loff[UB];     // lane part of the offset, in BYTES
      int upix[UB];          // uniform part, in pixels
      bool ok[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int dm = (it0 + u) * STEP;
        const int m = m0 + dm;
        val[u] = ld4(g_s + m * GS + c4 * 4);
        if constexpr (MAINGEO) {
          upix[u] = (dm >> lgGW) * p.WO + (dm & (GW - 1));
          loff[u] = off_t * OE;
          ok[u] = true;
          if constexpr (HN) nz[u] = *at_bytes(gnoise + upix[u], pix_t * 4u);
        } else {
          const int gx = m & (GW - 1), gy = (m >> lgGW) & (GH - 1), img = m >> (lgGW + lgGH);
          ok[u] = (b0 + img) < p.B && (gy0 + gy) < p.HO && (gx0 + gx) < p.WO;       // ragged batch / ragged image edge
          const unsigned pix = ok[u] ? (unsigned)((gy0 + gy) * p.WO + gx0 + gx) : 0u;
          upix[u] = 0;
          loff[u] = ((unsigned)(ok[u] ? img : 0) * (unsigned)img_elems + pix * (unsigned)p.CO + (unsigned)(n0 + c4 * 4)) * OE;
          if constexpr (HN) nz[u] = gnoise[pix];
        }
        if constexpr (HS) sk[u] = IoOut::ld_once(sb + (size_t)upix[u] * p.CO * OE, loff[u]);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        f4 v = val[u];
        if constexpr (HN) {
          if constexpr (F16) v = v * acc_scale + MIGAN_FMUL_RN(nz[u], ns);
          else v += MIGAN_FMUL_RN(nz[u], ns);                         // product rounded first, reference :166
          v = act4(v);
        } else {
          v = F16 ? act4g(v, gain_s) : act4(v);
        }
        f4 outv = v;
        if constexpr (HS) outv = v + IoOut::cvt(sk[u]);
        if (ok[u]) IoOut::st(yb + (size_t)upix[u] * p.CO * OE, loff[u], outv);
        if constexpr (do_rgb) {
          // ToRGB (reference :312): this lane's share of the 3 dot products over the CO channels of the pixel
          // goes into the g_s slot the item just consumed; the per-pixel sums are formed in the tail pass below.
          // (16-bit storage: ToRGB sees the stored, i.e. rounded, activations, like torgb_kernel reading the tensor back)
          v = IoOut::rounded(outv);
          float r0, r1, r2;
          torgb_partial(v, tw0, tw1, tw2, r0, r1, r2);
          st4(g_s + (m0 + (it0 + u) * STEP) * GS + c4 * 4, f4{r0, r1, r2, 0.0f});
        }
      }
    }
    };
    if (has_noise) { if (sb) epi_items(TrueT{}, TrueT{}); else epi_items(TrueT{}, FalseT{}); }
    else { if (sb) epi_items(FalseT{}, TrueT{}); else epi_items(FalseT{}, FalseT{}); }
    if constexpr (do_rgb) {
      // tail pass: two threads per pixel each sum half of the QN partials of its row of g_s, one ds_swizzle
      // exchange combines them; the even thread adds bias + the 2x-upsampled previous image (its 4 taps per
      // channel were loaded before the item loop) and writes the three planes (consecutive x per lane pair).
      __syncthreads();
      const int m = (tide >> 1) & (MT - 1), hsel = tide & 1;      // MT < 128: the upper threads repeat rows, results unused
      f4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < QN / 2; ++q) sum += ld4(g_s + m * GS + (hsel * (QN / 2) + q) * 4);
      sum.x += MIGAN_SWIZZLE_XOR(sum.x, 1);
      sum.y += MIGAN_SWIZZLE_XOR(sum.y, 1);
      sum.z += MIGAN_SWIZZLE_XOR(sum.z, 1);
      if (rgb_ok) {   // (tide < 2 * MT is part of rgb_ok)
        const float rgb[3] = {sum.x + p.trgb_b[0], sum.y + p.trgb_b[1], sum.z + p.trgb_b[2]};
        const size_t plane = (size_t)p.HO * p.WO;
        float o3[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o3[ch] = up_combine(pv[ch], rgb_oy, rgb_ox, p.HO >> 1, p.WO >> 1) + rgb[ch];
        if (p.u8_out) {
          compose_pixel(p.u8_img, p.u8_mask, p.u8_out, (size_t)rgb_b * plane + (size_t)rgb_oy * p.WO + rgb_ox, o3[0], o3[1], o3[2]);
        } else {
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) p.img_out[((size_t)rgb_b * 3 + ch) * plane + (size_t)rgb_oy * p.WO + rgb_ox] = o3[ch];
        }
      }
    }
  } else {
    // UP: each item owns one interior low-resolution pixel x 4 channels and produces its 2x2
    // output pixels from the 3x3 neighbourhood in g_s (separable polyphase taps 1/4, 3/4).
    // lane part of the output address: output pixel (2*(gy0+gyt), 2*(gx0+gxt)) of this thread's first item
    // (signed: the first item of a thread may be a halo pixel above/left of the image)
    const int opix_t = 2 * (gy0 + gyt) * p.WO + 2 * (gx0 + gxt);
    const int ooff_t = opix_t * p.CO + n0 + c4 * 4;
    auto epi_items = [&](auto hn_, auto hs_) {
      constexpr bool HN = decltype(hn_)::value, HS = decltype(hs_)::value;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int dm = k * STEP;
      const int m = m0 + dm;
      int gx, gy, img;
      if constexpr (MAINGEO) {
        gy = dm >> lgGW;                     // compile-time: halo rows of the GEMM grid drop out statically
        gx = gxt + (dm & (GW - 1));
        img = 0;
        if (gy < 1 || gy > GH - 2) continue;
      } else {
        gx = m & (GW - 1); gy = (m >> lgGW) & (GH - 1); img = m >> (lgGW + lgGH);
        if (gy < 1 || gy > GH - 2) continue;
      }
      const int ly = gy0 + gy, lx = gx0 + gx, b = b0 + img;
      if (gx < 1 || gx > GW - 2 || ly >= p.H || lx >= p.W || b >= p.B) continue;     // halo columns, ragged edge
      // uniform part (pixels) and lane part (elements) of the four output addresses
      int upix, loff, lpix;
      if constexpr (MAINGEO) {
        upix = 2 * (dm >> lgGW) * p.WO + 2 * (dm & (GW - 1));
        loff = ooff_t;
        lpix = opix_t;
      } else {
        upix = 0;
        lpix = (2 * ly) * p.WO + 2 * lx;
        loff = img * (int)img_elems + lpix * p.CO + n0 + c4 * 4;
      }
      typename IoOut::raw4 sk[2][2];
      float nz[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const int dp = upix + a * p.WO + bb;                      // uniform
          if constexpr (HN) nz[a][bb] = gnoise[(unsigned)(lpix + dp)];
          if constexpr (HS) sk[a][bb] = IoOut::ld_once(sb, (unsigned)(loff + dp * p.CO) * OE);
        }
      f4 e[3], o[3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const float* gp = g_s + (m + (dy - 1) * GW) * GS + c4 * 4;
        const f4 l = ld4(gp - GS), ctr = ld4(gp), rgt = ld4(gp + GS);
        e[dy] = 0.25f * l + 0.75f * ctr;
        o[dy] = 0.75f * ctr + 0.25f * rgt;
      }
      f4 out[2][2];
      out[0][0] = 0.25f * e[0] + 0.75f * e[1];
      out[0][1] = 0.25f * o[0] + 0.75f * o[1];
      out[1][0] = 0.75f * e[1] + 0.25f * e[2];
      out[1][1] = 0.75f * o[1] + 0.25f * o[2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          f4 v = out[a][bb];
          if constexpr (HN) {
            if constexpr (F16) v = v * acc_scale + MIGAN_FMUL_RN(nz[a][bb], ns);
            else v += MIGAN_FMUL_RN(nz[a][bb], ns);
            v = act4(v);
          } else {
            v = F16 ? act4g(v, gain_s) : act4(v);
          }
          if constexpr (HS) v += IoOut::cvt(sk[a][bb]);
          IoOut::st(yb, (unsigned)(loff + (upix + a * p.WO + bb) * p.CO) * OE, v);
        }
    }
    };
    if (has_noise) { if (sb) epi_items(TrueT{}, TrueT{}); else epi_items(TrueT{}, FalseT{}); }
    else { if (sb) epi_items(FalseT{}, TrueT{}); else epi_items(FalseT{}, FalseT{}); }
  }
  PROF_MARK(5);
  if (!has_next) break;
  __syncthreads();                       // epilogue reads of g_s done before the next tile refills LDS
  tl += tstep;
  n0 = n0n; b0 = b0n; gy0 = gy0n; gx0 = gx0n; vmask = vmaskn;
#pragma unroll
  for (int j = 0; j < NI; ++j) goff[j] = goffn[j];
#pragma unroll
  for (int j = 0; j < NB; ++j) boff[j] = boffn[j];
  }  // tile loop
  PROF_END();
}

// ------------------------------------------------------------------------------------------------
// Wide plain layers (Cout % 256 == 0, output >= 16x16, f16x2 GEMM): one workgroup of 8 waves owns
// 8x16 pixels x 256 output channels, so the depthwise stage of a pixel runs Cout/256 instead of Cout/128
// times, and the two halves of the workgroup are specialised so that stage never stalls the matrix cores:
//
//   waves 0-3 ("A"): depthwise 3x3 + act + fp16 split of chunk c+1 (-> a_s[(c+1)&1]), then their MFMAs of chunk c
//   waves 4-7 ("B"): MFMAs of chunk c (they run on the same four SIMDs while A is in its VALU/LDS stage)
//   all 512 threads: global -> register prefetch (input tile 3 chunks ahead, 1x1 weight planes 2 ahead),
//                    register -> LDS stores, epilogue
//
// Everything the K loop touches is double buffered (input tile, taps, A planes, B planes: 145 KB), so one
// barrier per chunk suffices.  MFMA wave grid 2 x 4, each wave 64 x 64 (same fragments as sepconv_kernel).
constexpr int kWideThreads = 512;
// BALL (round 3): waves 4-7 run ALL the MFMAs (each 64 x 128 of the tile, 128 accumulator registers), waves 0-3 only the depthwise stage.
// The ablation of the original split (profiles/r03_wide_ablation.txt) showed its A waves run depthwise and their half of the MFMAs
// back to back while the B waves wait at the barrier: 2.5k cycles per K chunk where the matrix pipe needs 1.5k and the depthwise
// stage 1.6k.  The two K loops are separate code paths so that the accumulators and the depthwise temporaries share registers.
// DMA (round 3, fp32 storage): the input tile and the weight planes of a K chunk go from global memory straight into LDS
// (buffer_load_dwordx4 ... lds) instead of through 32 prefetch registers and seven ds_write_b128 per thread -- the phase profile of
// the register path (profiles/r03_wide_phase_profile.txt) has 60 % of a chunk in "registers -> LDS + wait for the loads".  The
// image is a buffer descriptor, so pixels outside it (conv zero padding) are lane offsets beyond its range and arrive as zeros;
// the XOR swizzle of the weight planes is applied to the SOURCE address (the LDS destination of a DMA is linear in the lane).
// Every wave waits for its own DMAs (explicit s_waitcnt vmcnt(0)) ahead of the barrier that publishes them.
// UP (round 4, on the DMA form): FIR-up layers with Cout % 256 == 0 (synthesis conv1 at 128x128 and below in migan-512) -- the 8x16 grid of GEMM
// pixels overlaps its neighbours by one pixel (tile pitch 6x14, as sepconv_kernel<MODE_UP>), the epilogue turns every interior pixel of
// the result tile into its 2x2 output pixels (polyphase 1/4, 3/4 taps, reference Upsample2d :79-104) + noise + activation + skip.  One
// workgroup per 256 columns instead of one per 128: the depthwise stage and the input tile are read half as often.
template <bool TORGB, int STV = 0, bool X1 = false, bool BALL = false, bool DMA = false, bool UP = false>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(kWideThreads, 2) sepconv_wide_kernel(const SepArgs p) {
  static_assert(!DMA || (STV == 0 && BALL), "the LDS-DMA staging is built for fp32 storage on the dedicated-MFMA-wave form");
  static_assert(!UP || (DMA && !TORGB), "the FIR-up epilogue is built on the LDS-DMA form");
  typedef Io<STV> IoT;
  constexpr unsigned OE = IoT::ESZ;
  MIGAN_DYN_SMEM(smem);
  constexpr int MT = 128, NT = 256, KC = 32, QC = 8, LG_QC = 3;
  constexpr int GS = NT + 4, QN = NT / 4, LG_QN = 6;
  constexpr int GH = 8, GW = 16, lgGW = 4;
  constexpr int IGH = GH + 2, IGW = GW + 2, NPIX = IGH * IGW, NITEMS = NPIX * QC;
  constexpr int NI = (NITEMS + kWideThreads - 1) / kWideThreads;            // 3 float4 input items per thread and chunk
  constexpr int PB = KC * 2, NSLOT = PB / 16, NPL = X1 ? 1 : 2;      // X1: one fp16 piece per operand (GEMM variant "f16")
  constexpr int NB = NPL * NT * NSLOT / kWideThreads;                       // 4 float4 weight-plane items per thread and chunk
  static_assert(NB * kWideThreads == NPL * NT * NSLOT, "weight tile must split evenly over the threads");
  constexpr int NW4 = KC * 10 / 4;
  constexpr int WN = BALL ? 2 : 4, WROWS = 64, WCOLS = BALL ? 128 : 64, MTI = 2, NTI = BALL ? 4 : 2;
  constexpr int SEGH = 4;
  // LDS carve (floats)
  constexpr int IN_SZ = NPIX * KC, W_SZ = KC * 10, A_SZ = NPL * MT * PB / 4, B_SZ = NPL * NT * PB / 4;
  constexpr int OFF_IN = 0, OFF_W = OFF_IN + 2 * IN_SZ, OFF_A = OFF_W + 2 * W_SZ, OFF_B = OFF_A + 2 * A_SZ;
  static_assert((OFF_B + 2 * B_SZ) * 4 <= 160 * 1024 && MT * GS * 4 <= 160 * 1024, "LDS budget");
  float* g_s = smem;                                                        // after the K loop: [MT][GS]

  const char* __restrict__ gx_ = reinterpret_cast<const char*>(p.x);
  char* __restrict__ gy_ = reinterpret_cast<char*>(p.y);
  const char* __restrict__ gskip = reinterpret_cast<const char*>(p.skip);
  const float* __restrict__ gnoise = p.noise;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = BALL ? ((wave & 3) >> 1) : wave / WN, wn = BALL ? (wave & 1) : wave % WN;     // BALL: the 2 x 2 grid of waves 4-7
  const int l31 = lane & 31, half = lane >> 5;
  const bool groupA = tid < 256;
  PROF_BEGIN();
  const int pslot = groupA ? 0 : 4;   // phase profile (debug builds): A waves -> slots 0..3, B waves -> 4..7 [stores+load waits, depthwise, MFMA, barrier]
  (void)pslot;

  // tile schedule: same XCD-contiguous order as sepconv_kernel, one tile per workgroup
  const int ntiles = p.tiles_x * p.tiles_y * p.nchunks * p.B;
  const int xcd = (int)blockIdx.x & 7;
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int tbase = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  int t = tbase + ((int)blockIdx.x >> 3);
  const int nch = t % p.nchunks; t /= p.nchunks;
  const int tx = t % p.tiles_x;  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n0 = nch * NT, b0 = t / p.tiles_y, gy0 = ty * p.sy - p.off, gx0 = tx * p.sx - p.off;     // (plain: pitch 8 x 16, off 0)

  // per-thread item descriptors (constant across K chunks)
  unsigned goff[NI], boff[NB], vmask = 0, emask = 0;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int i = tid + j * kWideThreads;
    unsigned g = 0;
    if (i < NITEMS) {
      emask |= 1u << j;
      const int c4 = i & (QC - 1);
      const int pix = i >> LG_QC;
      const int ix = pix % IGW, iy = pix / IGW;
      const int yy = gy0 - 1 + iy, xx = gx0 - 1 + ix;
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
        vmask |= 1u << j;
        g = (unsigned)((yy * p.W + xx) * p.CI + c4 * 4) * OE;      // bytes
      }
    }
    goff[j] = g;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int i = tid + j * kWideThreads;
    const int plane = i / (NT * NSLOT), rem = i % (NT * NSLOT);
    boff[j] = (unsigned)(plane * p.CO * p.CI + (n0 + rem / NSLOT) * KC + (rem % NSLOT) * 8) * 2u;    // bytes
  }
  const char* __restrict__ xb = gx_ + (size_t)b0 * p.H * p.W * p.CI * OE;
  typename IoT::raw4 rin[NI];
  f4 rb[NB], rw;
  // ---- DMA form: lane offsets into the image / the weight planes, wave-uniform LDS destinations -----------------------------------
  // MIGAN_DMA_WHO (measurement builds): which waves issue the DMAs -- 0 all eight, 1 the MFMA waves (4-7), 2 the depthwise waves (0-3)
#ifndef MIGAN_DMA_WHO
#define MIGAN_DMA_WHO 0
#endif
  constexpr int LT = MIGAN_DMA_WHO == 0 ? kWideThreads : 256;                 // loader threads
  constexpr int DNI = DMA ? (NITEMS + LT - 1) / LT : 1, DNB = DMA ? NPL * NT * NSLOT / LT : 1;
  const int wave_u = MIGAN_UNIFORM(wave);
  const bool loader = MIGAN_DMA_WHO == 0 || (MIGAN_DMA_WHO == 1 ? wave_u >= 4 : wave_u < 4);
  const int lt = MIGAN_DMA_WHO == 1 ? tid - 256 : tid, lwave = MIGAN_DMA_WHO == 1 ? wave_u - 4 : wave_u;
  unsigned dgoff[DNI], dboff[DNB], demask = 0;
  if constexpr (DMA) {
#pragma unroll
    for (int j = 0; j < DNI; ++j) {
      const int i = lt + j * LT;
      unsigned g = 0xfffff000u;                        // padding pixel: beyond the buffer -> zeros
      if (i < NITEMS && i >= 0) {
        demask |= 1u << j;
        const int c4 = i & (QC - 1), pix = i >> LG_QC;
        const int ix = pix % IGW, iy = pix / IGW;
        const int yy = gy0 - 1 + iy, xx = gx0 - 1 + ix;
        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) g = (unsigned)((yy * p.W + xx) * p.CI + c4 * 4) * OE;
      }
      dgoff[j] = g;
    }
#pragma unroll
    for (int j = 0; j < DNB; ++j) {
      const int i = (lt & (LT - 1)) + j * LT;          // 16-byte unit number inside the LDS image of the weight tile (linear)
      const int plane = i / (NT * NSLOT), rem = i % (NT * NSLOT);
      const int n = rem / NSLOT, sp = rem % NSLOT;     // LDS row n, stored slot sp holds source slot sp ^ swizzle(n)
      dboff[j] = (unsigned)(plane * p.CO * p.CI + (n0 + n) * KC + ((sp ^ ((n >> 2) & (NSLOT - 1))) * 8)) * 2u;
    }
  }
  const MIGAN_BUF xbuf = MIGAN_MAKE_BUF(xb, (unsigned)(p.H * p.W * p.CI) * OE);
  const MIGAN_BUF wbuf = MIGAN_MAKE_BUF(p.wsplit, (unsigned)(NPL * p.CO * p.CI) * 2u);     // the planes' real extent: a wrong lane offset reads zeros, not memory
  auto load_taps = [&](int k0) {
    if (tid < KC * 9 / 4) rw = ld4(p.wdw + (size_t)k0 * 9 + (unsigned)(tid * 4));
    else if (tid < NW4) rw = ld4(p.bdw + k0 + (unsigned)((tid - KC * 9 / 4) * 4));
  };
  auto store_taps = [&](int buf) {
    float* w_s = smem + OFF_W + buf * W_SZ;
    if (tid < KC * 9 / 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = tid * 4 + e;                   // flat index into [KC][9] -> tap-major [9][KC]
        w_s[(f % 9) * KC + f / 9] = rw[e];
      }
    } else if (tid < NW4) {
      st4(w_s + tid * 4, rw);
    }
  };
  auto dma_in = [&](int k0, int buf) {               // input tile of one chunk -> in_s[buf]
    float* in_s = smem + OFF_IN + buf * IN_SZ;
    if (!loader) return;
#pragma unroll
    for (int j = 0; j < DNI; ++j)
      if (demask & (1u << j)) MIGAN_LDS_DMA16(xbuf, dgoff[j], (unsigned)k0 * OE, in_s + (j * LT + lwave * 64) * 4);
  };
  auto dma_b = [&](int k0, int buf) {                // fp16 planes of the 1x1 weights of one chunk -> b_s[buf]
    float* bb = smem + OFF_B + buf * B_SZ;
    if (!loader) return;
#pragma unroll
    for (int j = 0; j < DNB; ++j) MIGAN_LDS_DMA16(wbuf, dboff[j], (unsigned)k0 * (unsigned)p.CO * 2u, bb + (j * LT + lwave * 64) * 4);
  };
  auto load_in = [&](int k0) {                     // input tile + depthwise taps of one chunk
#pragma unroll
    for (int j = 0; j < NI; ++j) rin[j] = IoT::ld(xb + (size_t)k0 * OE, goff[j]);
    if (tid < KC * 9 / 4) rw = ld4(p.wdw + (size_t)k0 * 9 + (unsigned)(tid * 4));
    else if (tid < NW4) rw = ld4(p.bdw + k0 + (unsigned)((tid - KC * 9 / 4) * 4));
  };
  auto load_b = [&](int k0) {                      // fp16 planes of the 1x1 weights of one chunk
    const unsigned short* __restrict__ wk = p.wsplit + (size_t)k0 * p.CO;
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[j] = ld4(at_bytes(reinterpret_cast<const float*>(wk), boff[j]));
  };
  auto store_in = [&](int buf) {
    float* in_s = smem + OFF_IN + buf * IN_SZ;
    float* w_s = smem + OFF_W + buf * W_SZ;
    if (tid < KC * 9 / 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = tid * 4 + e;                   // flat index into [KC][9] -> tap-major [9][KC]
        w_s[(f % 9) * KC + f / 9] = rw[e];
      }
    } else if (tid < NW4) {
      st4(w_s + tid * 4, rw);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (emask & (1u << j)) {
        f4 v = IoT::cvt(rin[j]);
        if (!(vmask & (1u << j))) v = f4{0.f, 0.f, 0.f, 0.f};
        st4(in_s + (tid + j * kWideThreads) * 4, v);
      }
    }
  };
  auto store_b = [&](int buf) {
    char* bb = reinterpret_cast<char*>(smem + OFF_B + buf * B_SZ);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int i = tid + j * kWideThreads;
      const int plane = i / (NT * NSLOT), rem = i % (NT * NSLOT);
      const int n = rem / NSLOT, slot = rem % NSLOT;
      st4(reinterpret_cast<float*>(bb + (plane * NT + n) * PB + ((slot ^ ((n >> 2) & (NSLOT - 1))) << 4)), rb[j]);
    }
  };
  // depthwise 3x3 + bias + act (x 2^7) + fp16 split of one chunk: threads 0..255, one 4-row strip x 4 channels each
  auto depthwise = [&](int buf, int abuf) {
    const float* in_s = smem + OFF_IN + buf * IN_SZ;
    const float* w_s = smem + OFF_W + buf * W_SZ;
    char* a_b = reinterpret_cast<char*>(smem + OFF_A + abuf * A_SZ);
    const int c4 = tid & (QC - 1);
    const int gx = (tid >> LG_QC) & (GW - 1);
    const int r0 = (tid >> (LG_QC + lgGW)) * SEGH;
    f4 w[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) w[tap] = ld4(w_s + tap * KC + c4 * 4);
    const f4 bias = ld4(w_s + KC * 9 + c4 * 4);
    const float* ip = in_s + (r0 * IGW + gx) * KC + c4 * 4;
    // The input row an output needs last is read one output ahead (`nxt`) and pinned there: its LDS latency
    // runs under the previous output's 18 packed FMAs + activation + split (these waves are alone in this
    // stage: nothing else would hide it).
    f4 win[3][3], nxt[3];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      win[rr][0] = ld4(ip); win[rr][1] = ld4(ip + KC); win[rr][2] = ld4(ip + 2 * KC);
      ip += IGW * KC;
    }
    nxt[0] = ld4(ip); nxt[1] = ld4(ip + KC); nxt[2] = ld4(ip + 2 * KC);
    ip += IGW * KC;
#pragma unroll
    for (int o = 0; o < SEGH; ++o) {
      const int nr = (o + 2) % 3;
      win[nr][0] = nxt[0]; win[nr][1] = nxt[1]; win[nr][2] = nxt[2];
      if (o + 1 < SEGH) {
        nxt[0] = ld4(ip); nxt[1] = ld4(ip + KC); nxt[2] = ld4(ip + 2 * KC);
        ip += IGW * KC;
      }
      MIGAN_SCHED_FENCE();
      f4 sacc = bias;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) sacc += w[ky * 3 + kx] * win[(o + ky) % 3][kx];
      const int m = ((r0 + o) << lgGW) + gx;
      char* d = a_b + m * PB + (((c4 >> 1) ^ ((m >> 2) & (NSLOT - 1))) << 4) + ((c4 & 1) << 3);
      if constexpr (X1) {
        const f4 av_ = act4_scaled<7>(sacc);
        *reinterpret_cast<u2v*>(d) = u2v{MIGAN_PACK_F16(av_.x, av_.y), MIGAN_PACK_F16(av_.z, av_.w)};
      } else {
        u2v h1, h2;
        split2_f16(act4_scaled<7>(sacc), h1, h2);
        *reinterpret_cast<u2v*>(d) = h1;
        *reinterpret_cast<u2v*>(d + MT * PB) = h2;
      }
    }
  };
  f16v acc[MTI][NTI];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MTI; ++i)
#pragma unroll
      for (int j = 0; j < NTI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  };
  auto mfma_chunk = [&](int buf) {
    const char* ab = reinterpret_cast<const char*>(smem + OFF_A + buf * A_SZ);
    const char* bb = reinterpret_cast<const char*>(smem + OFF_B + buf * B_SZ);
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      f4 av[MTI][NPL], bv[NTI][NPL];
#pragma unroll
      for (int i = 0; i < MTI; ++i) {
        const int row = wm * WROWS + i * 32 + l31;
        const char* q = ab + row * PB + (((2 * ks + half) ^ ((row >> 2) & (NSLOT - 1))) << 4);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) av[i][pl] = ld4(reinterpret_cast<const float*>(q + pl * MT * PB));
      }
#pragma unroll
      for (int j = 0; j < NTI; ++j) {
        const int row = wn * WCOLS + j * 32 + l31;
        const char* q = bb + row * PB + (((2 * ks + half) ^ ((row >> 2) & (NSLOT - 1))) << 4);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) bv[j][pl] = ld4(reinterpret_cast<const float*>(q + pl * NT * PB));
      }
#pragma unroll
      for (int i = 0; i < MTI; ++i)
#pragma unroll
        for (int j = 0; j < NTI; ++j) {
          if constexpr (!X1) {
            acc[i][j] = MIGAN_MFMA_F16_32X32X16(av[i][1], bv[j][0], acc[i][j]);
            acc[i][j] = MIGAN_MFMA_F16_32X32X16(av[i][0], bv[j][1], acc[i][j]);
          }
          acc[i][j] = MIGAN_MFMA_F16_32X32X16(av[i][0], bv[j][0], acc[i][j]);
        }
    }
  };
  // accumulator fragment -> LDS result tile (C/D layout of the 32x32 MFMA: lane holds column l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5))
  auto acc_to_lds = [&]() {
    int tw_ = tid;
    MIGAN_OPAQUE(tw_);
    const int lanee = tw_ & 63, wavee = tw_ >> 6;
    const int wme = BALL ? ((wavee & 3) >> 1) : wavee / WN, wne = BALL ? (wavee & 1) : wavee % WN, l31e = lanee & 31, halfe = lanee >> 5;
#pragma unroll
    for (int i = 0; i < MTI; ++i)
#pragma unroll
      for (int j = 0; j < NTI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wme * WROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * halfe;
          const int col = wne * WCOLS + j * 32 + l31e;
          float v = acc[i][j][r];
          if constexpr (UP) {
            // halo pixels outside the low-resolution image contribute zeros to the upsampling FIR (reference pads with zeros :101)
            const int ly = gy0 + (row >> lgGW), lx = gx0 + (row & (GW - 1));
            if (ly < 0 || ly >= p.H || lx < 0 || lx >= p.W) v = 0.0f;
          }
          g_s[row * GS + col] = v;
        }
  };

  // ---- prologue: chunk 0 complete in LDS (input, taps, A planes, B planes), chunk 1 input in LDS ----------
  const int nkc = p.CI / KC;
  // (Starting the K loop at a tile-position-dependent chunk, so that the workgroups of an XCD do not ask the L2 for the same weight planes
  // at the same moment, measured -1..-3 % on these layers -- but a layer must sum its K chunks in ONE order in every kernel form, or an image
  // is no longer bit-identical between a batch that takes this kernel and one that takes the small-launch tiles: not done.)
  const int krot = 0;
  auto kof = [&](int c) { int k = c + krot; if (k >= nkc) k -= nkc; return k * KC; };      // (c < nkc)
  if constexpr (DMA) {
    dma_in(kof(0), 0);
    dma_b(kof(0), 0);
    load_taps(kof(0));
    store_taps(0);
    if (1 < nkc) {
      dma_in(kof(1), 1);
      load_taps(kof(1));
      store_taps(1);
    }
    // every wave waits for ITS OWN DMAs before the barrier that publishes them: gfx950 barriers do not drain the VM counter, and
    // the wait hipcc 7.2 happens to place there is not a guarantee (a workgroup-scope release only needs lgkmcnt)
    MIGAN_WAIT_VMCNT(0);
    __syncthreads();
    if (groupA) depthwise(0, 0);
    __syncthreads();
  } else {
    load_in(0);
    load_b(0);
    store_in(0);
    store_b(0);
    if (1 < nkc) { load_in(KC); load_b(KC); }
    __syncthreads();
    if (groupA) depthwise(0, 0);
    if (1 < nkc) store_in(1);
    if (2 < nkc) load_in(2 * KC);
    __syncthreads();
  }
  // ---- K loop: one barrier per chunk ------------------------------------------------------------------------
  // at the top of iteration c: a_s[c&1], b_s[c&1] = chunk c; in_s/w_s[(c+1)&1] = chunk c+1;
  // registers: weight planes of chunk c+1, input tile of chunk c+2
  auto stage = [&](int c) {                          // all 512 threads: registers -> LDS, next global loads
    if constexpr (DMA) {
      // in_s / w_s[c&1] were last read by the depthwise stage of chunk c (previous iteration), b_s[(c+1)&1] by the MFMAs of chunk c-1
      if (c + 2 < nkc) {
        dma_in(kof(c + 2), c & 1);
        load_taps(kof(c + 2));
      }
      if (c + 1 < nkc) dma_b(kof(c + 1), (c + 1) & 1);
      PROF_MARK(pslot + 0);
      return;
    }
    if (c + 1 < nkc) store_b((c + 1) & 1);        // last read by the MFMAs of chunk c-1
    if constexpr (BALL) PROF_MARK(pslot + 0);        // (phase profile of the BALL form: [weight tile -> LDS, input tile -> LDS + loads, depthwise | MFMA, barrier])
    if (c + 2 < nkc) {
      store_in(c & 1);       // last read by the depthwise stage of chunk c
      load_b((c + 2) * KC);
    }
    if (c + 3 < nkc) load_in((c + 3) * KC);
  };
  if constexpr (BALL) {
    if (groupA) {
      for (int c = 0; c < nkc; ++c) {
        stage(c);
        PROF_MARK(pslot + 1);
        if (c + 1 < nkc) depthwise((c + 1) & 1, (c + 1) & 1);
        PROF_MARK(pslot + 2);
        if constexpr (DMA) { if (c + 2 < nkc) store_taps(c & 1); MIGAN_WAIT_VMCNT(0); }
        __syncthreads();
        PROF_MARK(pslot + 3);
      }
    } else {
      zero_acc();
      for (int c = 0; c < nkc; ++c) {
        stage(c);
        PROF_MARK(pslot + 1);
        mfma_chunk(c & 1);
        PROF_MARK(pslot + 2);
        if constexpr (DMA) { if (c + 2 < nkc) store_taps(c & 1); MIGAN_WAIT_VMCNT(0); }
        __syncthreads();
        PROF_MARK(pslot + 3);
      }
      acc_to_lds();                               // (every wave is past the last chunk's barrier: a_s / b_s are dead)
    }
  } else {
    zero_acc();
    for (int c = 0; c < nkc; ++c) {
      stage(c);
      PROF_MARK(pslot + 0);
      if (groupA && c + 1 < nkc) depthwise((c + 1) & 1, (c + 1) & 1);   // a_s[(c+1)&1] last read by the MFMAs of chunk c-1
      PROF_MARK(pslot + 1);
      mfma_chunk(c & 1);
      PROF_MARK(pslot + 2);
      __syncthreads();
      PROF_MARK(pslot + 3);
    }
    acc_to_lds();
  }
  PROF_END_WIDE();

  // ======================================= epilogue ========================================
  int tide = tid;
  MIGAN_OPAQUE(tide);
  const float acc_scale = reinterpret_cast<const float*>(p.wsplit)[-4];
  const float gain_s = 1.41421356237309515f * acc_scale;
  __syncthreads();

  const bool has_noise = gnoise != nullptr;
  const float ns = has_noise ? p.noise_strength[0] : 0.0f;
  constexpr int ITEMS = MT * QN / kWideThreads;      // 16
  constexpr int UB = 4;
  constexpr int STEP = kWideThreads >> LG_QN;        // 8 GEMM rows between the items of a thread
  const int c4 = tide & (QN - 1);
  const int m0 = tide >> LG_QN;
  const int gxt = m0 & (GW - 1), gyt = m0 >> lgGW;
  const size_t img_elems = (size_t)p.HO * p.WO * p.CO;
  char* __restrict__ yb = gy_ + (size_t)b0 * img_elems * OE;
  const char* __restrict__ sb = gskip ? gskip + (size_t)b0 * img_elems * OE : nullptr;
  f4 tw0 = {0.f, 0.f, 0.f, 0.f}, tw1 = tw0, tw2 = tw0;
  int rgb_oy = 0, rgb_ox = 0;
  bool rgb_ok = false;
  float pv[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if constexpr (TORGB) {
    tw0 = ld4(p.trgb_w + n0 + c4 * 4);
    tw1 = ld4(p.trgb_w + p.CO + n0 + c4 * 4);
    tw2 = ld4(p.trgb_w + 2 * p.CO + n0 + c4 * 4);
    const int m = tide >> 1;
    rgb_oy = gy0 + ((m >> lgGW) & (GH - 1));
    rgb_ox = gx0 + (m & (GW - 1));
    rgb_ok = (tide & 1) == 0 && tide < 2 * MT;
    if (rgb_ok && p.img_prev) {
      const size_t plane4 = ((size_t)p.HO * p.WO) >> 2;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) up_taps(p.img_prev + ((size_t)b0 * 3 + ch) * plane4, p.HO >> 1, p.WO >> 1, rgb_oy, rgb_ox, pv[ch]);
    }
  }
  if constexpr (UP) {
    // each item owns one interior low-resolution pixel x 4 channels and produces its 2x2 output pixels from the 3x3 neighbourhood in g_s.
    // Items of a thread: GEMM rows m0 + 8 k -> grid row k >> 1 (compile time: the halo rows drop out statically), column m0 + 8 (k & 1)
    const int opix_t = 2 * gy0 * p.WO + 2 * (gx0 + gxt);                      // (signed: the first column may be a halo pixel left of the image)
    const int ooff_t = opix_t * p.CO + n0 + c4 * 4;
    auto up_items = [&](auto hn_, auto hs_) {
      constexpr bool HN = decltype(hn_)::value, HS = decltype(hs_)::value;
#pragma unroll
      for (int kk = 0; kk < ITEMS; ++kk) {
        const int dm = kk * STEP;
        const int gy = dm >> lgGW, gx = gxt + (dm & (GW - 1));
        if (gy < 1 || gy > GH - 2) continue;
        const int ly = gy0 + gy, lx = gx0 + gx;
        if (gx < 1 || gx > GW - 2 || ly >= p.H || lx >= p.W) continue;          // halo columns, ragged edge
        const int m = m0 + dm;
        const int upix = 2 * (dm >> lgGW) * p.WO + 2 * (dm & (GW - 1));         // uniform part (pixels)
        typename IoT::raw4 sk[2][2];
        float nz[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const int dp = upix + a * p.WO + bb;
            if constexpr (HN) nz[a][bb] = gnoise[(unsigned)(opix_t + dp)];
            if constexpr (HS) sk[a][bb] = IoT::ld_once(sb, (unsigned)(ooff_t + dp * p.CO) * OE);
          }
        f4 e[3], o[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const float* gp = g_s + (m + (dy - 1) * GW) * GS + c4 * 4;
          const f4 l = ld4(gp - GS), ctr = ld4(gp), rgt = ld4(gp + GS);
          e[dy] = 0.25f * l + 0.75f * ctr;
          o[dy] = 0.75f * ctr + 0.25f * rgt;
        }
        f4 out[2][2];
        out[0][0] = 0.25f * e[0] + 0.75f * e[1];
        out[0][1] = 0.25f * o[0] + 0.75f * o[1];
        out[1][0] = 0.75f * e[1] + 0.25f * e[2];
        out[1][1] = 0.75f * o[1] + 0.25f * o[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            f4 v = out[a][bb];
            if constexpr (HN) {
              v = v * acc_scale + MIGAN_FMUL_RN(nz[a][bb], ns);
              v = act4(v);
            } else {
              v = act4g(v, gain_s);
            }
            if constexpr (HS) v += IoT::cvt(sk[a][bb]);
            IoT::st(yb, (unsigned)(ooff_t + (upix + a * p.WO + bb) * p.CO) * OE, v);
          }
      }
    };
    if (has_noise) { if (sb) up_items(TrueT{}, TrueT{}); else up_items(TrueT{}, FalseT{}); }
    else { if (sb) up_items(FalseT{}, TrueT{}); else up_items(FalseT{}, FalseT{}); }
    return;
  }
  const unsigned pix_t = (unsigned)((gy0 + gyt) * p.WO + gx0 + gxt);
  const unsigned off_t = (pix_t * (unsigned)p.CO + (unsigned)(n0 + c4 * 4)) * OE;      // lane byte offset
  auto epi_items = [&](auto hn_, auto hs_) {
    constexpr bool HN = decltype(hn_)::value, HS = decltype(hs_)::value;
#pragma unroll
    for (int it0 = 0; it0 < ITEMS; it0 += UB) {
      f4 val[UB];
      typename IoT::raw4 sk[UB];
      float nz[UB];
      int upix[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int dm = (it0 + u) * STEP;
        val[u] = ld4(g_s + (m0 + dm) * GS + c4 * 4);
        upix[u] = (dm >> lgGW) * p.WO + (dm & (GW - 1));
        if constexpr (HN) nz[u] = *at_bytes(gnoise + upix[u], pix_t * 4u);
        if constexpr (HS) sk[u] = IoT::ld_once(sb + (size_t)upix[u] * p.CO * OE, off_t);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        f4 v = val[u];
        if constexpr (HN) {
          v = v * acc_scale + MIGAN_FMUL_RN(nz[u], ns);              // product rounded first, reference :166
          v = act4(v);
        } else {
          v = act4g(v, gain_s);
        }
        f4 outv = v;
        if constexpr (HS) outv = v + IoT::cvt(sk[u]);
        IoT::st(yb + (size_t)upix[u] * p.CO * OE, off_t, outv);
        if constexpr (TORGB) {
          v = IoT::rounded(outv);
          float r0, r1, r2;
          torgb_partial(v, tw0, tw1, tw2, r0, r1, r2);
          st4(g_s + (m0 + (it0 + u) * STEP) * GS + c4 * 4, f4{r0, r1, r2, 0.0f});
        }
      }
    }
  };
  if (has_noise) { if (sb) epi_items(TrueT{}, TrueT{}); else epi_items(TrueT{}, FalseT{}); }
  else { if (sb) epi_items(FalseT{}, TrueT{}); else epi_items(FalseT{}, FalseT{}); }
  if constexpr (TORGB) {
    __syncthreads();
    const int m = (tide >> 1) & (MT - 1), hsel = tide & 1;
    f4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < QN / 2; ++q) sum += ld4(g_s + m * GS + (hsel * (QN / 2) + q) * 4);
    sum.x += MIGAN_SWIZZLE_XOR(sum.x, 1);
    sum.y += MIGAN_SWIZZLE_XOR(sum.y, 1);
    sum.z += MIGAN_SWIZZLE_XOR(sum.z, 1);
    if (rgb_ok) {
      const float rgb[3] = {sum.x + p.trgb_b[0], sum.y + p.trgb_b[1], sum.z + p.trgb_b[2]};
      const size_t plane = (size_t)p.HO * p.WO;
      float o3[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) o3[ch] = up_combine(pv[ch], rgb_oy, rgb_ox, p.HO >> 1, p.WO >> 1) + rgb[ch];
      if (p.u8_out) {
        compose_pixel(p.u8_img, p.u8_mask, p.u8_out, (size_t)b0 * plane + (size_t)rgb_oy * p.WO + rgb_ox, o3[0], o3[1], o3[2]);
      } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) p.img_out[((size_t)b0 * 3 + ch) * plane + (size_t)rgb_oy * p.WO + rgb_ox] = o3[ch];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// First half of a down=2 SeparableConv2d (reference :155-160): depthwise 3x3 + bias, lrelu_agc, then
// Downsample2d (4x4 FIR [1,3,3,1]x[1,3,3,1]/64, stride 2, zero pad 1; reference :58-76).  Writes the
// half-resolution NHWC tensor the pointwise GEMM (sepconv_kernel<MODE_PW>) consumes.
//
// Un-fused from the GEMM on purpose: on gfx950 FP32 VALU work and v_mfma_f32 share one issue budget
// (profiles/r01_ubench_mfma_valu_overlap.md), so the 4x larger high-resolution depthwise grid is
// computed exactly once per pixel here (memory-bound, 3 workgroups per CU) instead of once per
// Cout tile inside the MFMA loop.
struct DwFirArgs {
  const void* x;       // NHWC [B][H][W][C], stored as Io<STV>
  float* y;            // NHWC [B][H/2][W/2][C], always fp32 (intermediate of one SeparableConv2d, never rounded)
  const float* wdw;    // conv1.weight [C][1][3][3]
  const float* bdw;    // conv1.bias [C]
  int B, H, W, C;
  int lgGH, lgGW, lgIMGS;          // output tile: IMGS images x GH x GW low-resolution pixels (GH*GW*IMGS = 64)
  int tiles_x, tiles_y, nkg, kpw;  // grid = tiles_x * tiles_y * ceil(B/IMGS) * nkg; a workgroup walks kpw 16-channel chunks (nkg*kpw = C/16)
  int off_d, off_w;                // LDS carve (floats)
};

// STV 0/1/2: input stored as fp32 / bf16 / fp16, fp32 output.  STV 3/4: bf16 / fp16 input and the output written as the
// pointwise GEMM's A operand of the "f16" variant -- fp16(RNE) of value x 2^7, exactly what that kernel would make of the
// fp32 value (bit-identical results, half the bytes of the intermediate, no conversion in the GEMM kernel).
template <int NI, bool MAING, int STV = 0>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 3) dwfir_kernel(const DwFirArgs p) {
  constexpr bool OUT16 = STV >= 3;
  typedef Io<OUT16 ? STV - 2 : STV> IoIn;
  MIGAN_DYN_SMEM(smem);
  constexpr int KC = 16, QC = 4, LG_QC = 2, MT = 64;
  const int tid = threadIdx.x;
  const int lgGH = MAING ? 2 : p.lgGH, lgGW = MAING ? 4 : p.lgGW, lgIMGS = MAING ? 0 : p.lgIMGS;
  const int GH = 1 << lgGH, GW = 1 << lgGW, IMGS = 1 << lgIMGS;
  int t = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int kgrp = t % p.nkg; t /= p.nkg;                      // group of kpw consecutive 16-channel chunks
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b0 = (t / p.tiles_y) << lgIMGS;
  const int gy0 = ty * GH, gx0 = tx * GW;                      // output tile origin
  const int HO = p.H >> 1, WO = p.W >> 1;
  const int IGH = 2 * GH + 4, IGW = 2 * GW + 4;
  const int iy0 = 2 * gy0 - 2, ix0 = 2 * gx0 - 2;
  const int npix_in = IMGS * IGH * IGW, nitems_in = npix_in * QC;
  const int DH = 2 * GH + 2, DW = 2 * GW + 2, DH2 = GH + 1;
  float* in_s = smem;
  float* d_s = smem + p.off_d;
  float* w_s = smem + p.off_w;
  const char* __restrict__ xb = reinterpret_cast<const char*>(p.x) + (size_t)b0 * p.H * p.W * p.C * IoIn::ESZ;
  float* __restrict__ yb = OUT16 ? reinterpret_cast<float*>(reinterpret_cast<char*>(p.y) + (size_t)b0 * HO * WO * p.C * 2)
                                : p.y + (size_t)b0 * HO * WO * p.C;

  // per-thread item descriptors (constant across the channel chunks this workgroup walks)
  unsigned goff[NI];
  unsigned vmask = 0, emask = 0;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int i = tid + j * kThreads;
    unsigned g = 0;
    if (i < nitems_in) {
      emask |= 1u << j;
      const int c4 = i & (QC - 1);
      const int pix = i >> LG_QC;
      const int ix = pix % IGW;
      const int r = pix / IGW;
      const int iy = r % IGH, img = r / IGH;
      const int yy = iy0 + iy, xx = ix0 + ix;
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && (b0 + img) < p.B) {
        vmask |= 1u << j;
        g = (unsigned)(((img * p.H + yy) * p.W + xx) * p.C + c4 * 4) * IoIn::ESZ;     // bytes
      }
    }
    goff[j] = g;
  }
  typename IoIn::raw4 rin[NI];
  f4 rw;
  auto issue_loads = [&](int k0) {
    const char* __restrict__ xk = xb + (size_t)k0 * IoIn::ESZ;
#pragma unroll
    for (int j = 0; j < NI; ++j) rin[j] = IoIn::ld(xk, goff[j]);
    if (tid < KC * 9 / 4) rw = ld4(p.wdw + (size_t)k0 * 9 + (unsigned)(tid * 4));
    else if (tid < KC * 10 / 4) rw = ld4(p.bdw + k0 + (unsigned)((tid - KC * 9 / 4) * 4));
  };

  const int kfirst = kgrp * p.kpw;
  issue_loads(kfirst * KC);
  for (int kk = 0; kk < p.kpw; ++kk) {
    const int k0 = (kfirst + kk) * KC;
    if (kk > 0) __syncthreads();                 // previous chunk's FIR pass is done with d_s (and in_s, w_s)
    // prefetched chunk -> LDS (input tile with +2 halo, tap-major depthwise weights)
    if (tid < KC * 9 / 4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = tid * 4 + e;
        w_s[(f % 9) * KC + f / 9] = rw[e];
      }
    } else if (tid < KC * 10 / 4) {
      st4(w_s + tid * 4, rw);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (emask & (1u << j)) {
        f4 v = IoIn::cvt(rin[j]);
        if (!(vmask & (1u << j))) v = f4{0.f, 0.f, 0.f, 0.f};
        st4(in_s + (tid + j * kThreads) * 4, v);
      }
    }
    __syncthreads();
    if (kk + 1 < p.kpw) issue_loads(k0 + KC);    // next chunk in flight during both compute stages

    // stage 1: depthwise 3x3 + bias + act on the (2GH+2)x(2GW+2) high-resolution grid this tile's FIR
    // window touches -> d_s.  One item = 2 vertically adjacent grid pixels x 4 channels (4x3 register
    // window: 12 LDS reads, 18 float4 FMAs).
    const int nstrips = IMGS * DH2 * DW * QC;
    const int yim0 = 2 * gy0 - 1, xim0 = 2 * gx0 - 1;
    // every strip of a thread has the same channel quad (kThreads % QC == 0): its taps are read once per chunk
    const int c4 = tid & (QC - 1);
    f4 w[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) w[tap] = ld4(w_s + tap * KC + c4 * 4);
    const f4 bias = ld4(w_s + KC * 9 + c4 * 4);
    for (int it = tid; it < nstrips; it += kThreads) {
      int r = it >> LG_QC;
      const int dx = r % DW; r /= DW;
      const int sy2 = r % DH2, img = r / DH2;
      const int dy0 = 2 * sy2;
      const float* ip = in_s + ((img * IGH + dy0) * IGW + dx) * KC + c4 * 4;
      f4 win[4][3];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        win[rr][0] = ld4(ip); win[rr][1] = ld4(ip + KC); win[rr][2] = ld4(ip + 2 * KC);
        ip += IGW * KC;
      }
      const int xim = xim0 + dx;
      const bool colin = xim >= 0 && xim < p.W;
      float* dp = d_s + (((img * DH + dy0) * DW) + dx) * KC + c4 * 4;
      f4 dv[2];
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        f4 sacc = bias;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) sacc += w[ky * 3 + kx] * win[o + ky][kx];
        const int yim = yim0 + dy0 + o;
        dv[o] = f4{0.f, 0.f, 0.f, 0.f};                       // FIR zero padding outside the image (reference :67)
        if (colin && yim >= 0 && yim < p.H) dv[o] = act4(sacc);
      }
      // separable FIR, vertical half here (round 5; same order as sepconv_pipedown_kernel): row slot dy0 = (1, 3)/8 partial sum of this row
      // pair (for the output whose window starts with it), row slot dy0 + 1 = (3, 1)/8 (for the output whose window ends with it)
      st4(dp, 0.125f * dv[0] + 0.375f * dv[1]);
      st4(dp + DW * KC, 0.375f * dv[0] + 0.125f * dv[1]);
    }
    __syncthreads();
    // stage 2: the horizontal half of the 4x4 FIR, stride 2, taps outer([1,3,3,1])/64 (reference Downsample2d :58-76) -> HBM
    for (int it = tid; it < MT * QC; it += kThreads) {
      const int c4 = it & (QC - 1);
      const int m = it >> LG_QC;
      const int ox = m & (GW - 1), oy = (m >> lgGW) & (GH - 1), img = m >> (lgGW + lgGH);
      if (b0 + img >= p.B || gy0 + oy >= HO || gx0 + ox >= WO) continue;          // ragged batch / ragged image edge
      const float* dp = d_s + ((img * DH + 2 * oy) * DW + 2 * ox) * KC + c4 * 4;
      f4 a = {0.f, 0.f, 0.f, 0.f};
      // horizontal half: rows 2 oy (partial sum of the pair that opens the window) and 2 oy + 3 (of the pair that closes it), four columns
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const float fx = (kx == 0 || kx == 3) ? 0.125f : 0.375f;
        a += fx * (ld4(dp + kx * KC) + ld4(dp + (3 * DW + kx) * KC));
      }
      const unsigned oel = (unsigned)((((img * HO) + gy0 + oy) * WO + gx0 + ox) * p.C + k0 + c4 * 4);
      if constexpr (OUT16) Io<2>::st(reinterpret_cast<char*>(yb), oel * 2u, a * kF16AScale);
      else st4o(yb + oel, a);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv2.weight [CO][CI] fp32 -> three bf16 planes [3][CO*CI] for the error-compensated bf16 GEMM.
// One launch per forward covers every layer (table of up to 40 tensors passed by value).
struct SplitArgs {
  const float* src[40];
  unsigned long long dst_off[40];   // element (16-bit) offset of plane 0 inside `dst`; a 16-byte header precedes it
  unsigned count[40];               // CO*CI
  unsigned ci[40];                  // CI (the planes are written chunk-major: [plane][CI/32][CO][32])
  unsigned short* dst;
  int n;
  int f16;                          // 0: three bf16 planes; 1: two fp16 planes of the scaled weights; 2: one fp16 plane (GEMM variant "f16")
};
constexpr int kSplitHeader = 8;     // 16-bit elements of header in front of the planes: float[0] = accumulator scale, float[2] = weight scale
constexpr int kSplitBlocksPerTensor = 32;
#ifndef MIGAN_TEMPLATE_KERNELS_ONLY   // (the kernel-table slice translation units only instantiate sepconv_kernel)
// f16x2 only: one workgroup per tensor finds max|w| and derives the power-of-two scales.
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) weight_absmax_kernel(const SplitArgs p) {
  MIGAN_DYN_SMEM(red);
  const int t = (int)blockIdx.x;
  const unsigned cnt = p.count[t];
  const float* __restrict__ src = p.src[t];
  float m = 0.0f;
  for (unsigned i = threadIdx.x * 4; i < cnt; i += kThreads * 4) {
    const f4 v = ld4(src + i);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;       // floor(log2(max|w|))
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    float* hdr = reinterpret_cast<float*>(p.dst + p.dst_off[t] - kSplitHeader);
    hdr[2] = __builtin_bit_cast(float, (unsigned)(127 + 13 - e) << 23);           // weight scale: max|w| -> [2^13, 2^14)
    hdr[0] = __builtin_bit_cast(float, (unsigned)(127 + e - 13 - 7) << 23);       // 1 / (weight scale * kF16AScale)
    hdr[1] = m;
  }
}
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) split_weights_kernel(const SplitArgs p) {
  const int t = (int)blockIdx.x / kSplitBlocksPerTensor, blk = (int)blockIdx.x % kSplitBlocksPerTensor;
  if (t >= p.n) return;
  const unsigned cnt = p.count[t];
  const float* __restrict__ src = p.src[t];
  unsigned short* __restrict__ dst = p.dst + p.dst_off[t];
  const float sw = p.f16 ? reinterpret_cast<const float*>(dst - kSplitHeader)[2] : 1.0f;
  const unsigned ci = p.ci[t], co = cnt / ci;
  for (unsigned i = (blk * kThreads + threadIdx.x) * 4; i < cnt; i += kSplitBlocksPerTensor * kThreads * 4) {
    // source element (n, k) of [CO][CI] -> chunk-major position: the 32-channel K chunk of a row is 64 contiguous
    // bytes and the rows of a chunk follow each other, so a workgroup's weight tile is one contiguous, fully used run
    const unsigned n = i / ci, k = i % ci;
    const unsigned o = (k >> 5) * (co * 32u) + n * 32u + (k & 31u);
    if (p.f16) {
      u2v h1, h2;
      split2_f16(ld4(src + i) * sw, h1, h2);
      *reinterpret_cast<u2v*>(dst + o) = h1;
      if (p.f16 == 1) *reinterpret_cast<u2v*>(dst + cnt + o) = h2;
    } else {
      u2v h1, h2, h3;
      split3_bf16(ld4(src + i), h1, h2, h3);
      *reinterpret_cast<u2v*>(dst + o) = h1;
      *reinterpret_cast<u2v*>(dst + cnt + o) = h2;
      *reinterpret_cast<u2v*>(dst + 2 * (size_t)cnt + o) = h3;
    }
  }
}

#endif  // MIGAN_TEMPLATE_KERNELS_ONLY

// ------------------------------------------------------------------------------------------------
// Arbitrary-size forward (reference README.md:87): the [h][w] noise plane of a layer = noise_const [r][r] tiled periodically
// and cropped (what `self.register_buffer('noise_const', ...)`, reference :149, would have to become).
#ifndef MIGAN_TEMPLATE_KERNELS_ONLY
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) noise_plane_kernel(const NoiseArgs p) {
  const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
  if (i >= p.h * p.w) return;
  const int y = i / p.w, x = i % p.w;
  p.dst[i] = p.src[(y % p.r) * p.r + (x % p.r)];
}
#endif

// ------------------------------------------------------------------------------------------------
// Un-fused ToRGB (reference torgb 1x1 conv with bias + Upsample2d of the running image, :308-313)
// for layers whose output channels are split over several workgroups.  16 lanes per pixel, each
// lane strides over the channel float4s, then a 4-step wave-shuffle butterfly.
template <int STV>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) torgb_kernel(const RgbArgs p) {
  typedef Io<STV> IoIn;
  const int sub = threadIdx.x & 15;
  const size_t pixel = ((size_t)blockIdx.x * kThreads + threadIdx.x) >> 4;
  const size_t npix = (size_t)p.B * p.H * p.W;
  const bool ok = pixel < npix;
  float r0 = 0.f, r1 = 0.f, r2 = 0.f;
  if (ok) {
    const char* xp = reinterpret_cast<const char*>(p.x) + pixel * p.C * IoIn::ESZ;
    for (int q = sub; q < (p.C >> 2); q += 16) {
      const f4 v = IoIn::cvt(IoIn::ld(xp, (unsigned)(q * 4) * IoIn::ESZ));
      const f4 w0 = ld4(p.w + q * 4), w1 = ld4(p.w + p.C + q * 4), w2 = ld4(p.w + 2 * p.C + q * 4);
      float d0, d1, d2;
      torgb_partial(v, w0, w1, w2, d0, d1, d2);          // scalar FMA chains, see torgb_partial
      r0 += d0; r1 += d1; r2 += d2;
    }
  }
#pragma unroll
  for (int s = 8; s >= 1; s >>= 1) {
    r0 += __shfl_xor(r0, s);
    r1 += __shfl_xor(r1, s);
    r2 += __shfl_xor(r2, s);
  }
  if (ok && sub == 0) {
    const size_t plane = (size_t)p.H * p.W;
    const int b = (int)(pixel / plane);
    const int rem = (int)(pixel % plane);
    const int oy = rem / p.W, ox = rem % p.W;
    const float rgb[3] = {r0 + p.b[0], r1 + p.b[1], r2 + p.b[2]};
    float o3[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float up = 0.0f;
      if (p.img_prev) up = up_prev3(p.img_prev + ((size_t)b * 3 + ch) * (plane >> 2), p.H >> 1, p.W >> 1, oy, ox);
      o3[ch] = up + rgb[ch];
    }
    if (p.u8_out) {
      compose_pixel(p.u8_img, p.u8_mask, p.u8_out, pixel, o3[0], o3[1], o3[2]);
    } else {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) p.img_out[((size_t)b * 3 + ch) * plane + rem] = o3[ch];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// SeparableConv2d with fewer than 64 channels on either side (reference :154-170 at resolutions above 512: channels(1024) = 32,
// channels(2048) = 16, ...: `min(32768 // res, 512)`, :222-223).  The tiled kernels need 64-wide MFMA column tiles and 32-channel K chunks; these
// layers exist only in generators above the largest checkpoint the reference publishes (512), so they get a plain restatement of the layer
// instead of a tuned kernel: one thread per OUTPUT pixel, fp32 storage, straight fp32 sums -- depthwise 3x3 + bias -> lrelu_agc -> [FIR-down] ->
// 1x1 -> [FIR-up] -> noise -> lrelu_agc -> skip, with EncoderBlock.fromrgb in front (FROMRGB) and ToRGB + the upsampled previous image behind
// (trgb_w != null).  It makes `Generator(1024)` ... `Generator(4096)` run and agree with the reference; it is not a fast path.
// MODE_UP: one workgroup per 16 x 16 low-resolution pixels -- the 1x1 output of the 18 x 18 pixels under the tile goes to LDS once
// ([324][CO] floats, p.tiles_x x p.tiles_y tiles per image), then every thread finishes the 2 x 2 output pixels of its low-resolution pixel
// (thread per output pixel evaluated the 1x1 four times per output: 11.7 ms for synthesis.b1024.conv1; this form: see DESIGN section 9).
constexpr int kNarrowUpTile = 16;
template <int MODE, bool FROMRGB>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) narrow_sepconv_kernel(const SepArgs p) {
  constexpr int CMAX = 64;
  constexpr int T = kNarrowUpTile, TH = T + 2;
  const int CI = p.CI, CO = p.CO, H = p.H, W = p.W;
  const size_t plane_o = (size_t)p.HO * p.WO;
  size_t pix = 0;
  int b = 0, rem = 0, oy = 0, ox = 0, ty0 = 0, tx0 = 0;
  if constexpr (MODE == MODE_UP) {
    int t = (int)blockIdx.x;
    tx0 = (t % p.tiles_x) * T; t /= p.tiles_x;
    ty0 = (t % p.tiles_y) * T;
    b = t / p.tiles_y;
  } else {
    pix = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (pix >= (size_t)p.B * plane_o) return;
    b = (int)(pix / plane_o);
    rem = (int)(pix % plane_o);
    oy = rem / p.WO; ox = rem % p.WO;
  }
  const float* xin = reinterpret_cast<const float*>(p.x);
  // input of the depthwise conv at (yy, xx), channel ci: the stored activation, or act(fromrgb(raw pixel)) (reference :194-195); zero outside
  // the image (the conv's padding, :126)
  auto in_at = [&](int yy, int xx, int ci) -> float {
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) return 0.0f;
    if constexpr (FROMRGB) {
      const float* rp = xin + ((size_t)b * 4) * H * W + (size_t)yy * W + xx;
      float s = p.frgb_b[ci];
#pragma unroll
      for (int k = 0; k < 4; ++k) s += p.frgb_w[ci * 4 + k] * rp[(size_t)k * H * W];
      return act1(s);
    } else {
      return xin[(((size_t)b * H + yy) * W + xx) * CI + ci];
    }
  };
  // conv1 + lrelu_agc at GEMM-resolution pixel (yy, xx), all CI channels (:155-156)
  auto dw_act = [&](int yy, int xx, float (&a)[CMAX]) {
    for (int ci = 0; ci < CI; ++ci) {
      float s = p.bdw[ci];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) s += p.wdw[ci * 9 + ky * 3 + kx] * in_at(yy + ky - 1, xx + kx - 1, ci);
      a[ci] = act1(s);
    }
  };
  auto gemm = [&](const float (&a)[CMAX], float (&g)[CMAX]) {        // conv2, 1x1 without bias (:161)
    for (int co = 0; co < CO; ++co) {
      float s = 0.0f;
      for (int ci = 0; ci < CI; ++ci) s += p.wpw[co * CI + ci] * a[ci];
      g[co] = s;
    }
  };
  float a[CMAX], g[CMAX];
  if constexpr (MODE == MODE_DOWN) {
    // Downsample2d (:58-76): 4x4 FIR [1,3,3,1]^2 / 64, stride 2, zero padding 1, on the activated depthwise output
    float d[CMAX];
    for (int ci = 0; ci < CI; ++ci) d[ci] = 0.0f;
    for (int ky = 0; ky < 4; ++ky)
      for (int kx = 0; kx < 4; ++kx) {
        const int yy = 2 * oy + ky - 1, xx = 2 * ox + kx - 1;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const float f = ((ky == 0 || ky == 3) ? 0.125f : 0.375f) * ((kx == 0 || kx == 3) ? 0.125f : 0.375f);
        dw_act(yy, xx, a);
        for (int ci = 0; ci < CI; ++ci) d[ci] += f * a[ci];
      }
    gemm(d, g);
  } else if constexpr (MODE == MODE_UP) {
    // phase 1: g = conv2(act(conv1(x))) at the 18 x 18 low-resolution pixels under the tile (zero outside the image: Upsample2d's padding)
    MIGAN_DYN_SMEM(g_s);
    for (int i = (int)threadIdx.x; i < TH * TH; i += kThreads) {
      const int yy = ty0 - 1 + i / TH, xx = tx0 - 1 + i % TH;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        dw_act(yy, xx, a);
        gemm(a, g);
        for (int co = 0; co < CO; ++co) g_s[i * CO + co] = g[co];
      } else {
        for (int co = 0; co < CO; ++co) g_s[i * CO + co] = 0.0f;
      }
    }
    __syncthreads();
    // phase 2: Upsample2d (:79-103) in closed form: out[2i] = g[i-1]/4 + 3 g[i]/4, out[2i+1] = 3 g[i]/4 + g[i+1]/4 per axis; thread (ly, lx) of
    // the tile finishes the 2 x 2 output pixels of its low-resolution pixel: noise (:165-167), lrelu_agc (:168-169), skip (:305), store
    const int ly = (int)threadIdx.x / T, lx = (int)threadIdx.x % T;
    const int iy = ty0 + ly, ix = tx0 + lx;
    if (iy >= H || ix >= W) return;
    for (int dyo = 0; dyo < 2; ++dyo)
      for (int dxo = 0; dxo < 2; ++dxo) {
        const int oyy = 2 * iy + dyo, oxx = 2 * ix + dxo;
        const int y0 = dyo ? ly + 1 : ly, x0 = dxo ? lx + 1 : lx;                  // first tap in tile coordinates (+1: the halo ring)
        const float wy0 = dyo ? 0.75f : 0.25f, wx0 = dxo ? 0.75f : 0.25f;
        const size_t opix = ((size_t)b * p.HO + oyy) * p.WO + oxx;
        const float nz = p.noise ? MIGAN_FMUL_RN(p.noise[(size_t)oyy * p.WO + oxx], p.noise_strength[0]) : 0.0f;
        float* yo = reinterpret_cast<float*>(p.y) + opix * CO;
        const float* sk = p.skip ? reinterpret_cast<const float*>(p.skip) + opix * CO : nullptr;
        for (int co = 0; co < CO; ++co) {
          float u = 0.0f;
          for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx)
              u += ((dy ? 1.0f - wy0 : wy0) * (dx ? 1.0f - wx0 : wx0)) * g_s[((y0 + dy) * TH + (x0 + dx)) * CO + co];
          float v = act1(p.noise ? u + nz : u);
          if (sk) v += sk[co];
          yo[co] = v;
        }
      }
    return;
  } else {
    dw_act(oy, ox, a);
    gemm(a, g);
  }
  // noise (:165-167, the product rounded first), lrelu_agc (:168-169), skip (:272 / :305), store
  const float nz = p.noise ? MIGAN_FMUL_RN(p.noise[(size_t)oy * p.WO + ox], p.noise_strength[0]) : 0.0f;
  float* yo = reinterpret_cast<float*>(p.y) + pix * CO;
  const float* sk = p.skip ? reinterpret_cast<const float*>(p.skip) + pix * CO : nullptr;
  float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
  for (int co = 0; co < CO; ++co) {
    float v = act1(p.noise ? g[co] + nz : g[co]);
    if (p.trgb_w) { r0 += p.trgb_w[co] * v; r1 += p.trgb_w[CO + co] * v; r2 += p.trgb_w[2 * CO + co] * v; }
    if (sk) v += sk[co];
    yo[co] = v;
  }
  if (p.trgb_w) {      // torgb (:277 / :312) + Upsample2d of the running image (:308-313)
    const float rgb[3] = {r0 + p.trgb_b[0], r1 + p.trgb_b[1], r2 + p.trgb_b[2]};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float up = 0.0f;
      if (p.img_prev) up = up_prev3(p.img_prev + ((size_t)b * 3 + ch) * (plane_o >> 2), p.HO >> 1, p.WO >> 1, oy, ox);
      p.img_out[((size_t)b * 3 + ch) * plane_o + rem] = up + rgb[ch];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The steps either side of Generator.forward in the reference's scripts/demo.py, at network resolution
// (SURVEY section 8f row N2): uint8 image + mask -> network input, network output -> composited uint8.
// Pure HBM streaming kernels: one thread per 4 horizontally adjacent pixels (12 + 4 bytes in as four
// 32-bit words, one float4 store per plane; resp. three float4 loads and 12 bytes out).
struct PrePostArgs {
  const unsigned char* img;    // [N][R][R][3] uint8, HWC (np.array(PIL RGB image))
  const unsigned char* mask;   // [N][R][R] uint8, 255 = keep the pixel, anything else = hole (demo.py:44,60)
  const float* y;              // compose: network output [N][3][R][R]
  float* x;                    // pack: network input [N][4][R][R] = cat([mask - 0.5, img * mask]) (demo.py:65)
  unsigned char* out;          // compose: [N][R][R][3] uint8
  unsigned nquads;             // N * R * R / 4
  unsigned plane;              // R * R
};
#ifndef MIGAN_TEMPLATE_KERNELS_ONLY
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) pack_input_kernel(const PrePostArgs p) {
  const unsigned q = blockIdx.x * kThreads + threadIdx.x;
  if (q >= p.nquads) return;
  const unsigned pix = q * 4u;                       // first pixel of the quad (flat over N*R*R)
  const unsigned n = pix / p.plane, r = pix % p.plane;
  const unsigned* ip = reinterpret_cast<const unsigned*>(p.img + (size_t)pix * 3);
  const unsigned w0 = ip[0], w1 = ip[1], w2 = ip[2];
  const unsigned mw = *reinterpret_cast<const unsigned*>(p.mask + pix);
  unsigned char rgb[12];
#pragma unroll
  for (int i = 0; i < 4; ++i) { rgb[i] = (unsigned char)(w0 >> (8 * i)); rgb[4 + i] = (unsigned char)(w1 >> (8 * i)); rgb[8 + i] = (unsigned char)(w2 >> (8 * i)); }
  f4 m, c0, c1, c2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float mk = (((mw >> (8 * i)) & 0xffu) == 255u) ? 1.0f : 0.0f;     // demo.py:60: np.array(mask) // 255
    m[i] = mk;
    c0[i] = unit_image(rgb[3 * i + 0]) * mk;                                 // demo.py:65: img * mask
    c1[i] = unit_image(rgb[3 * i + 1]) * mk;
    c2[i] = unit_image(rgb[3 * i + 2]) * mk;
  }
  float* xb = p.x + (size_t)n * 4 * p.plane + r;
  st4(xb, m - 0.5f);                                                         // demo.py:65: mask - 0.5
  st4(xb + p.plane, c0);
  st4(xb + 2 * (size_t)p.plane, c1);
  st4(xb + 3 * (size_t)p.plane, c2);
}
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) compose_output_kernel(const PrePostArgs p) {
  const unsigned q = blockIdx.x * kThreads + threadIdx.x;
  if (q >= p.nquads) return;
  const unsigned pix = q * 4u;
  const unsigned n = pix / p.plane, r = pix % p.plane;
  const float* yb = p.y + (size_t)n * 3 * p.plane + r;
  const f4 y0 = ld4(yb), y1 = ld4(yb + p.plane), y2 = ld4(yb + 2 * (size_t)p.plane);
  const unsigned* ip = reinterpret_cast<const unsigned*>(p.img + (size_t)pix * 3);
  const unsigned w0 = ip[0], w1 = ip[1], w2 = ip[2];
  const unsigned mw = *reinterpret_cast<const unsigned*>(p.mask + pix);
  unsigned char rgb[12], o[12];
#pragma unroll
  for (int i = 0; i < 4; ++i) { rgb[i] = (unsigned char)(w0 >> (8 * i)); rgb[4 + i] = (unsigned char)(w1 >> (8 * i)); rgb[8 + i] = (unsigned char)(w2 >> (8 * i)); }
  auto to_u8 = [](float v) { return unit_to_u8(v); };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool keep = ((mw >> (8 * i)) & 0xffu) == 255u;
    // demo.py:139-140: img * mask + result * (1 - mask) with mask in {0, 1}
    o[3 * i + 0] = keep ? rgb[3 * i + 0] : to_u8(y0[i]);
    o[3 * i + 1] = keep ? rgb[3 * i + 1] : to_u8(y1[i]);
    o[3 * i + 2] = keep ? rgb[3 * i + 2] : to_u8(y2[i]);
  }
  unsigned* op = reinterpret_cast<unsigned*>(p.out + (size_t)pix * 3);
#pragma unroll
  for (int wd = 0; wd < 3; ++wd)
    op[wd] = (unsigned)o[4 * wd] | ((unsigned)o[4 * wd + 1] << 8) | ((unsigned)o[4 * wd + 2] << 16) | ((unsigned)o[4 * wd + 3] << 24);
}
#endif  // MIGAN_TEMPLATE_KERNELS_ONLY


}  // namespace migan

#include "migan_pipeline.hpp"
