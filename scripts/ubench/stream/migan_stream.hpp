// MI-GAN generator forward: the 64-channel full-resolution SeparableConv2d layers as a register-streaming kernel.
//
// Reference: lib/model_zoo/migan_inference.py SeparableConv2d.forward :154-170 (down = up = 1), EncoderBlock.fromrgb :186,:194-195,
// SynthesisBlock torgb + Upsample2d of the running image :308-313.
//
// Why a second kernel family: sepconv_kernel moves every value through LDS four times (input tile, A operand planes, result
// tile, ToRGB partials; ~520 KB of LDS traffic and ~8 barriers per 128-pixel tile) and its phases run back to back.  Here a
// wave owns a 32-pixel-wide column strip of one image and walks down its rows with NO LDS traffic for activations and NO
// barriers:
//
//   lane l = (pixel p = l & 31 of the strip, channel half h = l >> 5); a lane keeps 32 of the 64 input channels of its pixel
//   for three consecutive rows in registers (window, 96 VGPRs), loaded straight from HBM (buffer_load_dwordx4, the row after
//   next always in flight);
//   depthwise 3x3: the x +- 1 neighbours live in the adjacent lanes -> v_fmac_f32 with a DPP wave_shr:1 / wave_shl:1 source,
//   no data movement instruction at all; taps from LDS (broadcast reads);
//   lrelu * sqrt2 * 2^7, clamp, fp16 hi/lo split: the 8 channels a lane holds for one 16-channel K step ARE its B operand of
//   v_mfma_f32_32x32x16_f16 (B[k = 8h + j][n = p]); the 1x1 weights are the A operand (read from LDS in fragment order), so
//   D[m = output channel][n = pixel] leaves every lane with 4 consecutive output channels of its pixel per register quad:
//   the epilogue (scale + noise, lrelu, clamp, ToRGB partial sums) runs on the accumulators and stores 16 B per lane
//   straight to HBM.
//
// The two edge pixels of a strip only feed their neighbours' taps (their own left / right neighbour is in another strip, and
// the DPP shift hands lane 32 the wrong channel half), so strips overlap by two pixels: 30 outputs per 32 lanes.
//
// K-slot <-> channel map of this kernel (any bijection works as long as A and B agree): K step ks, lane half h, element j
// <-> input channel 16 ks + 8 (j >> 2) + 4 h + (j & 3), i.e. a lane's float4 number i = 0..7 holds channels 8 i + 4 h .. + 3.
// stream_prep_kernel writes the 1x1 weights in that order as fp16 planes.
#pragma once

namespace migan {

struct StreamArgs {
  const void* x;                 // NHWC [B][H][W][64] (Io<STV>); FROMRGB: network input fp32 NCHW [B][4][H][W]
  void* y;                       // NHWC [B][H][W][64]
  const float* wdw;              // conv1.weight [64][1][3][3]
  const float* bdw;              // conv1.bias [64]
  const unsigned short* wstream; // conv2.weight as A-operand fragments [mt][ks][plane][lane][8] fp16 (stream_prep_kernel)
  const float* acc_scale;        // 1 / (weight scale * 2^7), written by weight_absmax_kernel (header of the layer's split planes)
  const float* noise;            // noise_const [H][W] or null
  const float* noise_strength;
  const float* frgb_w;           // FROMRGB: fromrgb.weight [64][4], fromrgb.bias [64]
  const float* frgb_b;
  const float* trgb_w;           // TORGB: torgb.weight [3][64], torgb.bias [3]
  const float* trgb_b;
  const float* img_prev;         // planar [B][3][H/2][W/2] or null
  float* img_out;                // planar [B][3][H][W]
  const unsigned char* u8_img;   // uint8 network I/O (see SepArgs)
  const unsigned char* u8_mask;
  unsigned char* u8_out;
  int B, H, W;
  int nstrips;                   // ceil(W / 30)
  int rows_per_wave;             // ceil(B * nstrips * H / (4 * gridDim.x))
};

constexpr int kStreamValid = 30;      // output pixels per 32-lane strip

// LDS carve (floats)
constexpr int kStreamA = 0;                              // [2 mt][4 ks][2 planes][64 lanes][4 dwords]
constexpr int kStreamASz = 2 * 2 * 4 * 64 * 4;
constexpr int kStreamTaps = kStreamA + kStreamASz;       // [9][64] taps, [64] bias
constexpr int kStreamRgbW = kStreamTaps + 640;           // [3][64] ToRGB weights / [4][64] + [64] FromRGB weights
constexpr int kStreamStage = kStreamRgbW + 320;          // per wave: one row of the strip, [32 pixels][16 slots of 4 channels], 8 KB
constexpr int kStreamStageSz = 32 * 64;
constexpr int kStreamLds = kStreamStage + 4 * kStreamStageSz;

#ifndef MIGAN_STREAM_ABL
#define MIGAN_STREAM_ABL 0
#endif
#ifndef MIGAN_STREAM_WAVES
#define MIGAN_STREAM_WAVES 2
#endif
#ifndef MIGAN_DW8
#error "migan_stream.hpp needs MIGAN_DW8 (migan_rt_hip.h / tests/emu/hip_emu.h)"
#endif

// Global memory is only ever touched in whole 1 KB runs (a wave instruction = 4 pixels x 256 B, lane l -> 16 B number l): the
// per-lane fragments the matrix instruction wants (one pixel per lane, 32 B of it per access) would touch 32 cache lines per
// instruction and write quarter lines -- measured 2.3x slower than this kernel's arithmetic (profiles/r03_stream_ablation.txt).
// The transposition between the two layouts goes through a wave-private 8 KB staging row in LDS, 16-byte slots XOR-swizzled
// by the pixel so that both the 16-lanes-one-pixel and the 16-lanes-16-pixels access patterns are conflict free.  No barrier:
// a wave's LDS instructions execute in order.
MIGAN_DEVICE MIGAN_INLINE int stream_slot(int pixel, int slot) { return pixel * 64 + ((slot ^ (pixel & 15)) << 2); }   // float index

// FROMRGB: 0 = NHWC input tensor, 1 = fused FromRGB of the fp32 network input planes, 2 = the same from the uint8 image + mask
// TORGB: 0 = none, 1 = fused ToRGB + upsampled previous image -> fp32 planes, 2 = the same composed with the uint8 image -> uint8
template <int FROMRGB, int TORGB, int STV>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, MIGAN_STREAM_WAVES) sepconv_stream_kernel(const StreamArgs p) {
  static_assert(STV == 0, "16-bit activation storage: not built yet");
  MIGAN_DYN_SMEM(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int px = lane & 31, h = lane >> 5;
  const int wave = MIGAN_UNIFORM(tid >> 6);

  // ---- weights -> LDS (once per workgroup) ------------------------------------------------------------------------------
  {
    const f4* src = reinterpret_cast<const f4*>(p.wstream);
    f4* dst = reinterpret_cast<f4*>(smem + kStreamA);
#pragma unroll
    for (int j = 0; j < kStreamASz / 4 / kThreads; ++j) dst[tid + j * kThreads] = src[tid + j * kThreads];
    if (tid < 144) {
      const f4 v = ld4(p.wdw + tid * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = tid * 4 + e;                     // flat [64][9] -> tap-major [9][64]
        smem[kStreamTaps + (f % 9) * 64 + f / 9] = v[e];
      }
    } else if (tid < 160) {
      st4(smem + kStreamTaps + 576 + (tid - 144) * 4, ld4(p.bdw + (tid - 144) * 4));
    }
    if constexpr (TORGB != 0) {
      if (tid >= 192 && tid < 240) st4(smem + kStreamRgbW + (tid - 192) * 4, ld4(p.trgb_w + (tid - 192) * 4));
    }
    if constexpr (FROMRGB != 0) {
      if (tid >= 160 && tid < 224) {
        const f4 v = ld4(p.frgb_w + (tid - 160) * 4);  // row of channel c = tid - 160 -> input-major [4][64]
#pragma unroll
        for (int e = 0; e < 4; ++e) smem[kStreamRgbW + e * 64 + (tid - 160)] = v[e];
      } else if (tid >= 224 && tid < 240) {
        st4(smem + kStreamRgbW + 256 + (tid - 224) * 4, ld4(p.frgb_b + (tid - 224) * 4));
      }
    }
  }
  __syncthreads();

  const float acc_scale = p.acc_scale[0];
  const bool has_noise = p.noise != nullptr;
  const float ns = has_noise ? p.noise_strength[0] : 0.0f;
  float trgb_b0 = 0.f, trgb_b1 = 0.f, trgb_b2 = 0.f;
  if constexpr (TORGB != 0) { trgb_b0 = p.trgb_b[0]; trgb_b1 = p.trgb_b[1]; trgb_b2 = p.trgb_b[2]; }
  const float* tap_l = smem + kStreamTaps + 4 * h;       // + 64 t + 8 i
  const float* rgbw_l = smem + kStreamRgbW + 4 * h;
  const f4* a_l = reinterpret_cast<const f4*>(smem + kStreamA) + lane;
  float* stg = smem + kStreamStage + wave * kStreamStageSz;
  // staging addresses of this lane: as pixel owner (fragment side) and as 16-byte-run owner (global memory side)
  auto frag_idx = [&](int q) { return stream_slot(px, 2 * q + h); };     // slot 2 gi + h (input side), 8 m + 2 g + h (output side)
  const int run_px = lane >> 4, run_slot = lane & 15;    // instruction k moves pixel 4 k + run_px, slot run_slot

  // ---- my share of the (image, strip, row) sequence -------------------------------------------------------------------------
  const int total = p.B * p.nstrips * p.H;
  int u = ((int)blockIdx.x * 4 + wave) * p.rows_per_wave;
  const int uend = u + p.rows_per_wave < total ? u + p.rows_per_wave : total;
  constexpr unsigned ROWB = 64 * 4;                      // bytes per pixel
  const unsigned rowbytes = (unsigned)p.W * ROWB;

  float win[3][32];                                      // [slot][4 i + e]: channels 8 i + 4 h + e of rows y-1, y, y+1 (rotating)
  f4 nxt[8];                                             // the row after those, as it comes from memory (in flight)
  f16v acc[2];

  while (u < uend) {
    const int col = u / p.H;
    const int y0 = u - col * p.H;
    const int b = col / p.nstrips, s = col - b * p.nstrips;
    int nrows = p.H - y0;
    if (nrows > uend - u) nrows = uend - u;
    u += nrows;
    const int x0 = kStreamValid * s - 1;
    const int x = x0 + px;
    const bool xin = x >= 0 && x < p.W;
    const bool st_ok = px >= 1 && px <= kStreamValid && x < p.W;
    // byte offset of this lane's 16-byte run inside a row (instruction k adds 1 KB); pixels left of the image wrap far out of
    // range and pixels right of it exceed the row: the buffer range check turns both into zeros (loads) / nothing (stores)
    // (the range check is on the lane offset alone, so no lane offset may rely on wrapping back into range: instructions 1..7
    // use a base one instruction in, which is never negative, and instruction 0 has its own offset)
    const unsigned voff_run = (unsigned)(x0 + run_px + 4) * ROWB + 16u * (unsigned)run_slot;
    const unsigned voff_run0 = x0 + run_px >= 0 ? voff_run - 1024u : 0xfffff000u;
    const char* xb = reinterpret_cast<const char*>(p.x) + (size_t)b * p.H * rowbytes;
    char* yb = reinterpret_cast<char*>(p.y) + (size_t)b * p.H * rowbytes;
    const int ylast = y0 + nrows;                                         // last input row this run needs

    // row yy of the input as a buffer (empty outside the image and beyond what this run reads: those loads return zeros,
    // which is the conv zero padding of reference :126)
    auto in_row = [&](int yy) -> MIGAN_BUF {
      const bool rowin = yy >= 0 && yy < p.H && yy <= ylast;
      return MIGAN_MAKE_BUF(xb + (size_t)(rowin ? yy : 0) * rowbytes, rowin ? rowbytes : 0u);
    };
    auto load_run = [&](f4 (&r)[8], int yy) {
      const MIGAN_BUF rb = in_row(yy);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if constexpr ((MIGAN_STREAM_ABL & 16) != 0) r[k] = f4{(float)k, 1.f, 2.f, (float)yy};
        else r[k] = k == 0 ? MIGAN_BUF_LOAD4(rb, voff_run0, 0u) : MIGAN_BUF_LOAD4(rb, voff_run, 1024u * (unsigned)(k - 1));
      }
    };
    auto stage_run = [&](const f4 (&r)[8]) {
      MIGAN_WAVE_SYNC();                               // earlier reads of the staging row (other lanes') are done
#pragma unroll
      for (int k = 0; k < 8; ++k) st4(stg + stream_slot(4 * k + run_px, run_slot), r[k]);
      MIGAN_WAVE_SYNC();
    };
    auto unstage_group = [&](float (&slot)[32], int gi) {
      const f4 v = ld4(stg + frag_idx(gi));
      slot[4 * gi + 0] = v.x; slot[4 * gi + 1] = v.y; slot[4 * gi + 2] = v.z; slot[4 * gi + 3] = v.w;
    };
    // FROMRGB: the window rows are x = act(fromrgb(img)) (reference :194-195), computed from the 4 input planes
    auto load_raw = [&](int yy) -> f4 {
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (xin && yy >= 0 && yy < p.H) {
        if constexpr (FROMRGB == 2) {
          v = pack_pixel(p.u8_img, p.u8_mask, ((size_t)b * p.H + yy) * p.W + x);
        } else {
          const float* src = reinterpret_cast<const float*>(p.x) + ((size_t)b * 4 * p.H + yy) * p.W + x;
          const size_t plane = (size_t)p.H * p.W;
          v = f4{src[0], src[plane], src[2 * plane], src[3 * plane]};
        }
      }
      return v;
    };
    auto fromrgb_row = [&](float (&slot)[32], int yy, f4 raw) {
      const bool rowok = xin && yy >= 0 && yy < p.H;    // zero outside the image (conv padding), not act(bias)
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) {
        const f4 w0 = ld4(rgbw_l + 8 * gi), w1 = ld4(rgbw_l + 64 + 8 * gi), w2 = ld4(rgbw_l + 128 + 8 * gi), w3 = ld4(rgbw_l + 192 + 8 * gi);
        const f4 bb = ld4(rgbw_l + 256 + 8 * gi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = w0[e] * raw.x;
          t = fmaf(w1[e], raw.y, t);
          t = fmaf(w2[e], raw.z, t);
          t = fmaf(w3[e], raw.w, t);
          t = act1(bb[e] + t);
          slot[4 * gi + e] = rowok ? t : 0.0f;
        }
        MIGAN_SCHED_FENCE();
      }
    };
    const MIGAN_BUF nzb = MIGAN_MAKE_BUF(p.noise, has_noise ? (unsigned)(p.H * p.W) * 4u : 0u);      // no noise: every load returns 0
    const unsigned nz_off = xin ? (unsigned)x * 4u : 0u;

    // one output row: TOP / MID / BOT = window slots of rows y - 1, y, y + 1; at entry `nxt` holds (in flight) row y + 2 as it
    // comes from memory; TOP is refilled with it group by group as its channels die
    auto row = [&](float (&T)[32], float (&M)[32], float (&Bt)[32], int y) {
      // small loads of this row first: loads complete in order, and the row fetched below must stay in flight until the next row
      const float nzraw = MIGAN_BUF_LOAD1(nzb, nz_off, (unsigned)(y * p.W) * 4u);
      // ToRGB: Upsample2d of the previous image (reference :308-311) needs its 2 x 2 taps under this pixel; the lane of channel
      // half h fetches tap row h (2 taps x 3 channels) and adds its share to its ToRGB partial sums
      float pv[3][2];
      float pwx0 = 0.f, pwx1 = 0.f;                      // tap weights (0 where the tap lies outside the image), times the row weight
      unsigned char keep_px[4] = {0, 0, 0, 0};           // TORGB 2: mask byte and image bytes of my pixel
      if constexpr (TORGB != 0) {
        const int hp = p.H >> 1, wp = p.W >> 1;
        const int iy = y >> 1, ix = x >> 1;
        const int ty = ((y & 1) ? iy : iy - 1) + h, tx = (x & 1) ? ix : ix - 1;      // my tap row, first tap column
        const float wy = ((y & 1) != 0) == (h == 0) ? 0.75f : 0.25f;
        const float wx = (x & 1) ? 0.75f : 0.25f;
        const bool vy = st_ok && ty >= 0 && ty < hp, v0 = tx >= 0, v1 = tx + 1 < wp;
        const int cy = vy ? ty : 0, c0 = v0 ? tx : 0, c1 = v1 ? tx + 1 : wp - 1;
        pwx0 = (vy && v0) ? wy * wx : 0.0f;
        pwx1 = (vy && v1) ? wy * (1.0f - wx) : 0.0f;
        const unsigned plane4b = (unsigned)(hp * wp) * 4u;
        // the three half-resolution planes of image b as one buffer (empty when there is no previous image: loads return 0)
        const MIGAN_BUF pb = MIGAN_MAKE_BUF(reinterpret_cast<const char*>(p.img_prev) + (size_t)b * 3 * plane4b, p.img_prev ? 3u * plane4b : 0u);
        const unsigned o0 = (unsigned)(cy * wp + c0) * 4u, o1 = (unsigned)(cy * wp + c1) * 4u;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          pv[ch][0] = MIGAN_BUF_LOAD1(pb, o0, (unsigned)ch * plane4b);
          pv[ch][1] = MIGAN_BUF_LOAD1(pb, o1, (unsigned)ch * plane4b);
        }
        if constexpr (TORGB == 2) {
          if (st_ok && h == 0) {
            const size_t pix = ((size_t)b * p.H + y) * p.W + x;
            keep_px[0] = p.u8_mask[pix];
            keep_px[1] = p.u8_img[pix * 3]; keep_px[2] = p.u8_img[pix * 3 + 1]; keep_px[3] = p.u8_img[pix * 3 + 2];
          }
        }
      }
      if constexpr (FROMRGB != 0) {
        // row y + 1 goes into the slot row y - 2 has left, before the taps are read
        fromrgb_row(Bt, y + 1, load_raw(y + 1));
      } else {
        stage_run(nxt);                                  // row y + 2 -> staging (waits for it)
        load_run(nxt, y + 3);                            // in flight during this whole row
        MIGAN_SCHED_FENCE();
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        float d[8];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int gi = 2 * ks + gg;
          // taps of two channels at a time (ds_read_b64): 20 live registers instead of 40
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            f2v w[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const f2v*>(tap_l + 64 * t + 8 * gi + 2 * hf);
            const f2v bias = *reinterpret_cast<const f2v*>(tap_l + 576 + 8 * gi + 2 * hf);
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
              const int e = 2 * hf + e2, j = 4 * gi + e;
              float a;
              MIGAN_DW8(a, bias[e2], T[j], M[j], Bt[j], w[0][e2], w[1][e2], w[2][e2], w[3][e2], w[5][e2], w[6][e2], w[7][e2], w[8][e2]);
              a = fmaf(w[4][e2], M[j], a);
              // lrelu_agc (reference :20-28) with the fp16 operand scale 2^7 folded into gain and clamp (exact)
              float t = fmaxf(a, a * 0.2f);
              t = t * (1.41421356237309515f * kF16AScale);
              d[4 * gg + e] = MIGAN_CLAMP(t, -256.0f * kF16AScale, 256.0f * kF16AScale);
            }
          }
          if constexpr (FROMRGB == 0) unstage_group(T, gi);  // row y - 1 of these channels is dead: row y + 2 goes there
        }
        // B operand of this K step: 8 channels of my pixel, as fp16 hi / lo pieces
        u2v h01, l01, h23, l23;
        split2_f16(f4{d[0], d[1], d[2], d[3]}, h01, l01);
        split2_f16(f4{d[4], d[5], d[6], d[7]}, h23, l23);
        const f4 bhi = __builtin_bit_cast(f4, u4v{h01.x, h01.y, h23.x, h23.y});
        const f4 blo = __builtin_bit_cast(f4, u4v{l01.x, l01.y, l23.x, l23.y});
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const f4 ahi = a_l[((m * 4 + ks) * 2 + 0) * 64], alo = a_l[((m * 4 + ks) * 2 + 1) * 64];
          if constexpr ((MIGAN_STREAM_ABL & 8) != 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][4 * ks + r] += alo[r] * bhi[r] + ahi[r] * blo[r];
          } else {
            acc[m] = MIGAN_MFMA_F16_32X32X16(alo, bhi, acc[m]);
            acc[m] = MIGAN_MFMA_F16_32X32X16(ahi, blo, acc[m]);
            acc[m] = MIGAN_MFMA_F16_32X32X16(ahi, bhi, acc[m]);
          }
        }
      }
      // ---- epilogue on the accumulators: lane holds channels 32 m + 8 g + 4 h + e of its pixel in acc[m][4 g + e] ------------
      MIGAN_SCHED_FENCE();
      MIGAN_WAVE_SYNC();                                 // the window refills above have read the staging row
      const float nz = MIGAN_FMUL_RN(nzraw, ns);                          // product rounded first, reference :166
      float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = fmaf(acc[m][4 * g + e], acc_scale, nz);
            t = fmaxf(t, t * 0.2f);
            t = t * 1.41421356237309515f;
            v[e] = MIGAN_CLAMP(t, -256.0f, 256.0f);
          }
          st4(stg + frag_idx(4 * m + g), v);
          if constexpr (TORGB != 0) {
            const f4 t0 = ld4(rgbw_l + 32 * m + 8 * g), t1 = ld4(rgbw_l + 64 + 32 * m + 8 * g), t2 = ld4(rgbw_l + 128 + 32 * m + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              r0 = fmaf(v[e], t0[e], r0);
              r1 = fmaf(v[e], t1[e], r1);
              r2 = fmaf(v[e], t2[e], r2);
            }
          }
        }
      MIGAN_WAVE_SYNC();
      {
        // staging -> memory in whole 1 KB runs; the two overlap pixels of the strip belong to the neighbouring strips
        const MIGAN_BUF ob = MIGAN_MAKE_BUF(yb + (size_t)y * rowbytes, rowbytes);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const f4 o = ld4(stg + stream_slot(4 * k + run_px, run_slot));
          unsigned vo = voff_run;
          if (k == 0) vo = run_px == 0 ? 0xfffff000u : voff_run0;
          if (k == 7) vo = run_px == 3 ? 0xfffff000u : vo;
          if (!(MIGAN_STREAM_ABL & 4) || nz == 12345.0f) MIGAN_BUF_STORE4(ob, vo, k == 0 ? 0u : 1024u * (unsigned)(k - 1), o);
        }
      }
      if constexpr (TORGB != 0) {
        // + my tap row of the upsampled previous image, then the two channel halves of a pixel (lanes p and p + 32) are summed
        r0 += pwx0 * pv[0][0] + pwx1 * pv[0][1];
        r1 += pwx0 * pv[1][0] + pwx1 * pv[1][1];
        r2 += pwx0 * pv[2][0] + pwx1 * pv[2][1];
        r0 += __shfl_xor(r0, 32);
        r1 += __shfl_xor(r1, 32);
        r2 += __shfl_xor(r2, 32);
        const float o3[3] = {r0 + trgb_b0, r1 + trgb_b1, r2 + trgb_b2};
        if constexpr (TORGB == 2) {
          if (st_ok && h == 0) {
            // composed = img * mask + result * (1 - mask), mask in {0, 1} (demo.py:139-140)
            unsigned char* o = p.u8_out + (((size_t)b * p.H + y) * p.W + x) * 3;
            const bool keep = keep_px[0] == 255;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) o[ch] = keep ? keep_px[1 + ch] : unit_to_u8(o3[ch]);
          }
        } else {
          const unsigned planeb = (unsigned)(p.H * p.W) * 4u;
          const MIGAN_BUF ib = MIGAN_MAKE_BUF(reinterpret_cast<char*>(p.img_out) + (size_t)b * 3 * planeb, 3u * planeb);
          const unsigned vo = (st_ok && h == 0) ? (unsigned)(y * p.W + x) * 4u : 0xfffff000u;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) MIGAN_BUF_STORE1(ib, vo, (unsigned)ch * planeb, o3[ch]);
        }
      }
    };

    // prime the window: rows y0 - 1, y0 (and y0 + 1; y0 + 2 in flight)
    if constexpr (FROMRGB != 0) {
      fromrgb_row(win[0], y0 - 1, load_raw(y0 - 1));
      fromrgb_row(win[1], y0, load_raw(y0));
    } else {
      f4 r0[8], r1[8], r2[8];
      load_run(r0, y0 - 1);
      load_run(r1, y0);
      load_run(r2, y0 + 1);
      load_run(nxt, y0 + 2);
      stage_run(r0);
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) unstage_group(win[0], gi);
      stage_run(r1);
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) unstage_group(win[1], gi);
      stage_run(r2);
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) unstage_group(win[2], gi);
    }
    int y = y0;
    for (;;) {
      row(win[0], win[1], win[2], y);
      if (++y >= ylast) break;
      row(win[1], win[2], win[0], y);
      if (++y >= ylast) break;
      row(win[2], win[0], win[1], y);
      if (++y >= ylast) break;
    }
  }
}

// conv2.weight [64][64] fp32 -> the A-operand fragments of sepconv_stream_kernel: for M tile mt (32 output channels), K step ks and
// plane pl, lane l = (m = l & 31, hh = l >> 5) holds W[32 mt + m][16 ks + 8 (j >> 2) + 4 hh + (j & 3)], j = 0..7, scaled by the
// tensor's power-of-two weight scale (weight_absmax_kernel) and split into fp16 hi / lo.
struct StreamPrepArgs {
  const float* w;              // [64][64]
  const float* scale_hdr;      // header of the layer's split planes: [0] accumulator scale, [2] weight scale
  unsigned short* dst;         // [2 mt][4 ks][2 planes][64 lanes][8]
};
#ifndef MIGAN_TEMPLATE_KERNELS_ONLY
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) stream_prep_kernel(const StreamPrepArgs p) {
  const float sw = p.scale_hdr[2];
  for (int it = threadIdx.x; it < 2 * 4 * 64; it += kThreads) {
    const int l = it & 63, ks = (it >> 6) & 3, mt = it >> 8;
    const int m = l & 31, hh = l >> 5;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p.w[(32 * mt + m) * 64 + 16 * ks + 8 * (j >> 2) + 4 * hh + (j & 3)] * sw;
    u2v h0, l0, h1, l1;
    split2_f16(f4{v[0], v[1], v[2], v[3]}, h0, l0);
    split2_f16(f4{v[4], v[5], v[6], v[7]}, h1, l1);
    unsigned* d = reinterpret_cast<unsigned*>(p.dst) + (((mt * 4 + ks) * 2) * 64 + l) * 4;
    d[0] = h0.x; d[1] = h0.y; d[2] = h1.x; d[3] = h1.y;
    d[256 + 0] = l0.x; d[256 + 1] = l0.y; d[256 + 2] = l1.x; d[256 + 3] = l1.y;
  }
}
#endif

}  // namespace migan
