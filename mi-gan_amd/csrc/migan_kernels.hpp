// MI-GAN generator forward: gfx950 (MI355X / CDNA4) device kernels.
//
// One fused kernel per SeparableConv2d (reference lib/model_zoo/migan_inference.py:154-170):
//
//   HBM (NHWC fp32) --coalesced float4--> LDS input tile (+halo, zero padded)
//     -> depthwise 3x3 + bias            (VALU, LDS-resident, column strips with rotating accumulators)
//     -> lrelu*sqrt2, clamp              (reference :20-28)
//     -> [4x4 FIR, stride 2]             (reference Downsample2d :58-76; separable, vertical pass fused in the strip)
//     -> 1x1 conv as an fp32 MFMA GEMM   (v_mfma_f32_32x32x2_f32, exact f32; M = tile pixels, K = Cin chunk, N = Cout tile)
//     -> LDS result tile
//     -> [polyphase 2x FIR upsample]     (reference Upsample2d :79-103; zero-insertion folded into 2x2 taps)
//     -> [+ noise_const*noise_strength]  (reference :165-167)
//     -> lrelu*sqrt2, clamp -> [+ skip]  (reference :272 / :305)
//     -> [ToRGB 1x1 + bias + upsampled previous image, wave-shuffle channel reduction] (reference :308-313)
//     -> HBM (NHWC fp32), float4 stores, 256..512 B contiguous per pixel
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
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;          // 4 wave64 per workgroup, one per SIMD
constexpr int kInItemsMax = 9;         // float4 input-tile items per thread per K chunk (host checks)

enum : int { MODE_NORMAL = 0, MODE_DOWN = 1, MODE_UP = 2 };

struct SepArgs {
  // activations
  const float* x;              // NHWC [B][H][W][CI]  (FROMRGB: network input, NCHW [B][4][H][W])
  float* y;                    // NHWC [B][HO][WO][CO]
  const float* skip;           // NHWC like y, added after the activation, or null
  // SeparableConv2d parameters in the reference's native layouts
  const float* wdw;            // conv1.weight [CI][1][3][3]
  const float* bdw;            // conv1.bias   [CI]
  const float* wpw;            // conv2.weight [CO][CI][1][1]
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
  int B, H, W, CI, CO, HO, WO;
  // GEMM pixel grid of one workgroup: IMGS images x GH x GW (all powers of two), MT = IMGS*GH*GW rows
  int lgGH, lgGW, lgIMGS;
  int tiles_x, tiles_y, nchunks;   // workgroup grid: tiles_x * tiles_y * ceil(B/IMGS) * nchunks
  int sy, sx, off;                 // tile pitch in GEMM-resolution pixels; off = 1 for MODE_UP (recomputed halo)
  int lgRS;                        // log2(row segments per column) of the depthwise stage (NORMAL/UP)
  int off_a, off_b, off_v, off_rgb;// LDS carve, in floats
};

struct RgbArgs {
  const float* x;        // NHWC [B][H][W][C]
  const float* w;        // [3][C]
  const float* b;        // [3]
  const float* img_prev; // planar [B][3][H/2][W/2] or null
  float* img_out;        // planar [B][3][H][W]
  int B, H, W, C;
};

// ------------------------------------------------------------------------------------------------
// small device helpers

// lrelu_agc(alpha=0.2, gain=sqrt(2), clamp=256), reference :20-28: leaky_relu, fp32 multiply by
// float(np.sqrt(2)), clamp.  Same operation order as the reference so results round identically.
MIGAN_DEVICE MIGAN_INLINE float act1(float v) {
  float t = v > 0.0f ? v : v * 0.2f;
  t = t * 1.41421356237309515f;
  return fminf(fmaxf(t, -256.0f), 256.0f);
}
MIGAN_DEVICE MIGAN_INLINE f4 act4(f4 v) { return f4{act1(v.x), act1(v.y), act1(v.z), act1(v.w)}; }

MIGAN_DEVICE MIGAN_INLINE f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
MIGAN_DEVICE MIGAN_INLINE void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }

// XCD-aware workgroup order (MI355X: block b runs on XCD b%8, each XCD has a private 4 MiB L2):
// give every XCD one contiguous range of logical tiles so halo rows shared by neighbouring tiles
// and the Cout chunks of one tile hit the same L2.  Bijective for any grid size.
MIGAN_DEVICE MIGAN_INLINE int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// 2x polyphase FIR weights of Upsample2d per axis (reference :95-103, taps [1,3,3,1]/4 on a
// zero-inserted signal): out[2i] = g[i-1]/4 + 3g[i]/4 ; out[2i+1] = 3g[i]/4 + g[i+1]/4.
MIGAN_DEVICE MIGAN_INLINE float up_prev3(const float* plane, int hp, int wp, int oy, int ox) {
  const int iy = oy >> 1, ix = ox >> 1;
  const int y0 = (oy & 1) ? iy : iy - 1, x0 = (ox & 1) ? ix : ix - 1;   // first of the two taps
  const float wy0 = (oy & 1) ? 0.75f : 0.25f, wx0 = (ox & 1) ? 0.75f : 0.25f;
  float acc = 0.0f;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int yy = y0 + a;
    const float wy = a == 0 ? wy0 : 1.0f - wy0;
    if (yy < 0 || yy >= hp) continue;
    float row = 0.0f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int xx = x0 + c;
      const float wx = c == 0 ? wx0 : 1.0f - wx0;
      if (xx >= 0 && xx < wp) row += wx * plane[(size_t)yy * wp + xx];
    }
    acc += wy * row;
  }
  return acc;
}

// ------------------------------------------------------------------------------------------------
// fused SeparableConv2d
//
//   MODE   : NORMAL (down=1, up=1) | DOWN (FIR stride 2 before the 1x1) | UP (FIR x2 after the 1x1)
//   MT     : GEMM rows (pixels) per workgroup, 128 or 64
//   NT     : GEMM columns (output channels) per workgroup, 128 or 64
//   KC     : input-channel chunk staged per K step, 32 or 16
//   FROMRGB: input tile is act(fromrgb(network input)) computed on the fly (encoder first block)
//
// Waves are laid out 2x2 over the MT x NT tile; each wave owns (MT/2)x(NT/2) as 32x32 MFMA tiles.
template <int MODE, int MT, int NT, int KC, bool FROMRGB>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) sepconv_kernel(const SepArgs p) {
  MIGAN_DYN_SMEM(smem);

  constexpr int QC = KC / 4;                       // float4 groups per pixel in a K chunk
  constexpr int LG_QC = (QC == 8) ? 3 : 2;
  static_assert(QC == 8 || QC == 4, "KC must be 32 or 16");
  constexpr int AS = KC + 4;                       // A/B row pitch (floats): odd number of 16-B slots -> conflict-free b128
  constexpr int GS = NT + 4;                       // result tile row pitch
  constexpr int QN = NT / 4;
  constexpr int LG_QN = (QN == 32) ? 5 : 4;
  static_assert(QN == 32 || QN == 16, "NT must be 128 or 64");
  constexpr int WROWS = MT / 2, WCOLS = NT / 2;    // per-wave tile
  constexpr int MTI = WROWS / 32, NTI = WCOLS / 32;
  static_assert(MTI >= 1 && NTI >= 1, "wave tile must be at least 32x32");

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;

  const int GH = 1 << p.lgGH, GW = 1 << p.lgGW, IMGS = 1 << p.lgIMGS;

  // ---- which tile am I -------------------------------------------------------------------
  int t = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int nch = t % p.nchunks; t /= p.nchunks;
  const int tx = t % p.tiles_x;  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int bgrp = t / p.tiles_y;
  const int n0 = nch * NT;
  const int b0 = bgrp << p.lgIMGS;
  const int gy0 = ty * p.sy - p.off, gx0 = tx * p.sx - p.off;   // GEMM grid origin (GEMM-resolution image coords)

  // ---- LDS carve --------------------------------------------------------------------------
  float* in_s = smem;                       // [npix_in][KC]
  float* a_s = smem + p.off_a;              // [MT][AS]
  float* b_s = smem + p.off_b;              // [NT][AS]
  float* v_s = smem + p.off_v;              // DOWN: [IMGS][GH][2GW+2][KC]
  float* rgb_s = smem + p.off_rgb;          // FROMRGB: [npix_in][4]
  float* g_s = smem;                        // after the K loop: [MT][GS], aliases the buffers above

  // input-tile geometry (input-resolution coordinates)
  const int IGH = (MODE == MODE_DOWN) ? 2 * GH + 4 : GH + 2;
  const int IGW = (MODE == MODE_DOWN) ? 2 * GW + 4 : GW + 2;
  const int iy0 = (MODE == MODE_DOWN) ? 2 * gy0 - 2 : gy0 - 1;
  const int ix0 = (MODE == MODE_DOWN) ? 2 * gx0 - 2 : gx0 - 1;
  const int npix_in = IMGS * IGH * IGW;
  const int nitems_in = npix_in * QC;

  // per-thread global offsets of its input items (constant across K chunks):
  //   >= 0 : element offset of the float4 inside image group b0 (before adding the chunk's k0)
  //   -1   : outside the image / batch -> zero fill (conv zero padding, reference :126)
  //   -2   : no such item
  int goff[kInItemsMax];
#pragma unroll
  for (int j = 0; j < kInItemsMax; ++j) {
    const int i = tid + j * kThreads;
    int g = -2;
    if (i < nitems_in) {
      const int c4 = i & (QC - 1);
      const int pix = i >> LG_QC;
      const int ix = pix % IGW;
      const int r = pix / IGW;
      const int iy = r % IGH, img = r / IGH;
      const int yy = iy0 + iy, xx = ix0 + ix;
      g = -1;
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && (b0 + img) < p.B)
        g = FROMRGB ? 0 : ((img * p.H + yy) * p.W + xx) * p.CI + c4 * 4;
    }
    goff[j] = g;
  }
  const float* xb = p.x + (size_t)b0 * p.H * p.W * (FROMRGB ? 4 : p.CI);

  if constexpr (FROMRGB) {
    // raw 4-channel network input (NCHW) of the halo tile, loaded once
    for (int pix = tid; pix < npix_in; pix += kThreads) {
      const int ix = pix % IGW;
      const int r = pix / IGW;
      const int iy = r % IGH, img = r / IGH;
      const int yy = iy0 + iy, xx = ix0 + ix;
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && (b0 + img) < p.B) {
        const float* src = xb + ((size_t)img * 4 * p.H + yy) * p.W + xx;
        const size_t plane = (size_t)p.H * p.W;
        v = f4{src[0], src[plane], src[2 * plane], src[3 * plane]};
      }
      st4(rgb_s + pix * 4, v);
    }
  }

  f16v acc[MTI][NTI];
#pragma unroll
  for (int i = 0; i < MTI; ++i)
#pragma unroll
    for (int j = 0; j < NTI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // ======================================= K loop ==========================================
  const int nkc = p.CI / KC;
  for (int c = 0; c < nkc; ++c) {
    const int k0 = c * KC;
    __syncthreads();   // previous chunk's MFMA reads of a_s/b_s (and rgb_s writes) are complete

    // ---- S1: input tile chunk -> LDS, pointwise-weight chunk -> LDS ------------------------
#pragma unroll
    for (int j = 0; j < kInItemsMax; ++j) {
      if (goff[j] != -2) {
        const int i = tid + j * kThreads;
        const int c4 = i & (QC - 1);
        const int pix = i >> LG_QC;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (goff[j] >= 0) {
          if constexpr (FROMRGB) {
            // x = act(fromrgb(img)) (reference :194-195), 4 -> CI pointwise with bias
            const f4 raw = ld4(rgb_s + pix * 4);
            const float* wr = p.frgb_w + (size_t)(k0 + c4 * 4) * 4;
            const f4 w0 = ld4(wr), w1 = ld4(wr + 4), w2 = ld4(wr + 8), w3 = ld4(wr + 12);
            const f4 bb = ld4(p.frgb_b + k0 + c4 * 4);
            v.x = bb.x + (w0.x * raw.x + w0.y * raw.y + w0.z * raw.z + w0.w * raw.w);
            v.y = bb.y + (w1.x * raw.x + w1.y * raw.y + w1.z * raw.z + w1.w * raw.w);
            v.z = bb.z + (w2.x * raw.x + w2.y * raw.y + w2.z * raw.z + w2.w * raw.w);
            v.w = bb.w + (w3.x * raw.x + w3.y * raw.y + w3.z * raw.z + w3.w * raw.w);
            v = act4(v);
          } else {
            v = ld4(xb + goff[j] + k0);
          }
        }
        st4(in_s + pix * KC + c4 * 4, v);
      }
    }
    for (int i = tid; i < NT * QC; i += kThreads) {
      const int c4 = i & (QC - 1), n = i >> LG_QC;
      st4(b_s + n * AS + c4 * 4, ld4(p.wpw + (size_t)(n0 + n) * p.CI + k0 + c4 * 4));
    }
    __syncthreads();

    // ---- S2: depthwise 3x3 + bias + act (+ FIR down) -> A operand in LDS --------------------
    // One thread walks a column of the tile for 4 channels.  Every input row it reads (3 float4
    // from LDS) is scattered into three running sums (the outputs it is the bottom / middle / top
    // tap row of), so each LDS value is read once per column and no register window is kept.
    if constexpr (MODE != MODE_DOWN) {
      const int RS = 1 << p.lgRS;
      const int SEGH = GH >> p.lgRS;
      const int ncols = (IMGS * GW * QC) << p.lgRS;
      for (int it = tid; it < ncols; it += kThreads) {
        const int c4 = it & (QC - 1);
        int r = it >> LG_QC;
        const int gx = r & (GW - 1); r >>= p.lgGW;
        const int seg = r & (RS - 1);
        const int img = r >> p.lgRS;
        const int r0 = seg * SEGH;
        // 4 channels x 9 taps are 36 contiguous floats in conv1.weight [C][1][3][3]
        float wf[36];
        {
          const float* wp = p.wdw + (size_t)(k0 + c4 * 4) * 9;
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            const f4 tq = ld4(wp + 4 * q);
            wf[4 * q + 0] = tq.x; wf[4 * q + 1] = tq.y; wf[4 * q + 2] = tq.z; wf[4 * q + 3] = tq.w;
          }
        }
        f4 w[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) w[tap] = f4{wf[tap], wf[9 + tap], wf[18 + tap], wf[27 + tap]};
        const f4 bias = ld4(p.bdw + k0 + c4 * 4);
        f4 s2 = bias, s1 = bias, s0 = bias;   // outputs (row-2, row-1, row) of the current input row
        const float* ip = in_s + ((img * IGH + r0) * IGW + gx) * KC + c4 * 4;
        for (int rr = 0; rr < SEGH + 2; ++rr) {
          const f4 L = ld4(ip), M = ld4(ip + KC), R = ld4(ip + 2 * KC);
          ip += IGW * KC;
          s2 += w[6] * L + w[7] * M + w[8] * R;
          s1 += w[3] * L + w[4] * M + w[5] * R;
          s0 += w[0] * L + w[1] * M + w[2] * R;
          if (rr >= 2) {
            const int g = r0 + rr - 2;
            const int m = (img << (p.lgGH + p.lgGW)) + (g << p.lgGW) + gx;
            st4(a_s + m * AS + c4 * 4, act4(s2));
          }
          s2 = s1; s1 = s0; s0 = bias;
        }
      }
    } else {
      // DOWN: column of the (2GH+2)x(2GW+2) high-resolution depthwise grid; the vertical half of the
      // separable 4x4 FIR ([1,3,3,1]/8 per axis, stride 2, zero pad 1; reference :58-76) is folded
      // into the walk, the horizontal half runs as a second pass over v_s.
      const int DW = 2 * GW + 2;
      const int ncols = IMGS * DW * QC;
      for (int it = tid; it < ncols; it += kThreads) {
        const int c4 = it & (QC - 1);
        const int r = it >> LG_QC;
        const int dx = r % DW, img = r / DW;
        const int xim = 2 * gx0 - 1 + dx;
        const bool colin = xim >= 0 && xim < p.W;
        float wf[36];
        {
          const float* wp = p.wdw + (size_t)(k0 + c4 * 4) * 9;
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            const f4 tq = ld4(wp + 4 * q);
            wf[4 * q + 0] = tq.x; wf[4 * q + 1] = tq.y; wf[4 * q + 2] = tq.z; wf[4 * q + 3] = tq.w;
          }
        }
        f4 w[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) w[tap] = f4{wf[tap], wf[9 + tap], wf[18 + tap], wf[27 + tap]};
        const f4 bias = ld4(p.bdw + k0 + c4 * 4);
        f4 s2 = bias, s1 = bias, s0 = bias;
        f4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        const float* ip = in_s + ((img * IGH) * IGW + dx) * KC + c4 * 4;
        const int yim0 = 2 * gy0 - 1;
        for (int I = 0; I < 2 * GH + 4; ++I) {
          const f4 L = ld4(ip), M = ld4(ip + KC), R = ld4(ip + 2 * KC);
          ip += IGW * KC;
          s2 += w[6] * L + w[7] * M + w[8] * R;
          s1 += w[3] * L + w[4] * M + w[5] * R;
          s0 += w[0] * L + w[1] * M + w[2] * R;
          if (I >= 2) {
            const int dy = I - 2;
            const int yim = yim0 + dy;
            f4 d = {0.f, 0.f, 0.f, 0.f};                        // FIR zero padding outside the image
            if (colin && yim >= 0 && yim < p.H) d = act4(s2);
            if ((dy & 1) == 0) {          // d-row 2j: tap 0 of output j, tap 2 of output j-1
              v1 = 0.125f * d;
              v0 += 0.375f * d;
            } else {                      // d-row 2j+1: tap 1 of output j, tap 3 of output j-1 (completes it)
              v1 += 0.375f * d;
              v0 += 0.125f * d;
              const int oy = (dy >> 1) - 1;
              if (oy >= 0) st4(v_s + (((img << p.lgGH) + oy) * DW + dx) * KC + c4 * 4, v0);
              v0 = v1;
            }
          }
          s2 = s1; s1 = s0; s0 = bias;
        }
      }
      __syncthreads();
      for (int it = tid; it < MT * QC; it += kThreads) {
        const int c4 = it & (QC - 1);
        const int m = it >> LG_QC;
        const int ox = m & (GW - 1);
        const int rr = m >> p.lgGW;                 // img*GH + oy
        const float* vp = v_s + (rr * DW + 2 * ox) * KC + c4 * 4;
        const f4 a = 0.125f * ld4(vp) + 0.375f * ld4(vp + KC) + 0.375f * ld4(vp + 2 * KC) + 0.125f * ld4(vp + 3 * KC);
        st4(a_s + m * AS + c4 * 4, a);
      }
    }
    __syncthreads();

    // ---- S3: acc += A[MT x KC] * W^T[KC x NT] on the matrix cores ---------------------------
    // v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31].  Each lane
    // reads 4 consecutive k with one ds_read_b128; the two lane halves take k = 8kk+4*half+t, the
    // same for A and B, so any assignment of k to (half,t) sums the full K.
    {
      const float* ap = a_s + (wm * WROWS + l31) * AS + 4 * half;
      const float* bp = b_s + (wn * WCOLS + l31) * AS + 4 * half;
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
  }

  // ======================================= epilogue ========================================
  __syncthreads();                       // all waves done with a_s/b_s before g_s overwrites them
  // accumulator fragment -> LDS result tile.  C/D layout of the 32x32 MFMA: lane holds column
  // l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
#pragma unroll
  for (int i = 0; i < MTI; ++i)
#pragma unroll
    for (int j = 0; j < NTI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int col = wn * WCOLS + j * 32 + l31;
        float v = acc[i][j][r];
        if constexpr (MODE == MODE_UP) {
          // halo pixels outside the low-resolution image contribute zeros to the upsampling FIR
          // (reference pads with zeros :101), not the conv of a zero-padded input
          const int gx = row & (GW - 1);
          const int gy = (row >> p.lgGW) & (GH - 1);
          const int ly = gy0 + gy, lx = gx0 + gx;
          if (ly < 0 || ly >= p.H || lx < 0 || lx >= p.W) v = 0.0f;
        }
        g_s[row * GS + col] = v;
      }
  __syncthreads();

  const bool has_noise = p.noise != nullptr;
  const float ns = has_noise ? p.noise_strength[0] : 0.0f;

  if constexpr (MODE != MODE_UP) {
    const bool do_rgb = p.trgb_w != nullptr;
    for (int it = tid; it < MT * QN; it += kThreads) {
      const int c4 = it & (QN - 1);
      const int m = it >> LG_QN;
      const int gx = m & (GW - 1);
      const int gy = (m >> p.lgGW) & (GH - 1);
      const int img = m >> (p.lgGW + p.lgGH);
      const int oy = gy0 + gy, ox = gx0 + gx, b = b0 + img;
      const bool ok = b < p.B;
      f4 v = ld4(g_s + m * GS + c4 * 4);
      if (has_noise) {
        const float nz = MIGAN_FMUL_RN(p.noise[(size_t)oy * p.WO + ox], ns);   // product rounded first, reference :166
        v += nz;
      }
      v = act4(v);
      const size_t o = (((size_t)b * p.HO + oy) * p.WO + ox) * p.CO + n0 + c4 * 4;
      if (ok) {
        f4 out = v;
        if (p.skip) out += ld4(p.skip + o);
        st4(p.y + o, out);
      }
      if (do_rgb) {
        // ToRGB: 3 dot products over the CO channels of this pixel; the QN lanes holding one pixel
        // are contiguous in the wave -> butterfly reduction with wave shuffles.
        const f4 w0 = ld4(p.trgb_w + n0 + c4 * 4);
        const f4 w1 = ld4(p.trgb_w + p.CO + n0 + c4 * 4);
        const f4 w2 = ld4(p.trgb_w + 2 * p.CO + n0 + c4 * 4);
        float r0 = v.x * w0.x + v.y * w0.y + v.z * w0.z + v.w * w0.w;
        float r1 = v.x * w1.x + v.y * w1.y + v.z * w1.z + v.w * w1.w;
        float r2 = v.x * w2.x + v.y * w2.y + v.z * w2.z + v.w * w2.w;
#pragma unroll
        for (int s = QN / 2; s >= 1; s >>= 1) {
          r0 += __shfl_xor(r0, s);
          r1 += __shfl_xor(r1, s);
          r2 += __shfl_xor(r2, s);
        }
        if (c4 == 0 && ok) {
          const float rgb[3] = {r0 + p.trgb_b[0], r1 + p.trgb_b[1], r2 + p.trgb_b[2]};
          const size_t plane = (size_t)p.HO * p.WO;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            float up = 0.0f;
            if (p.img_prev)
              up = up_prev3(p.img_prev + ((size_t)b * 3 + ch) * (plane >> 2), p.HO >> 1, p.WO >> 1, oy, ox);
            p.img_out[((size_t)b * 3 + ch) * plane + (size_t)oy * p.WO + ox] = up + rgb[ch];
          }
        }
      }
    }
  } else {
    // UP: each item owns one interior low-resolution pixel x 4 channels and produces its 2x2
    // output pixels from the 3x3 neighbourhood in g_s (separable polyphase taps 1/4, 3/4).
    for (int it = tid; it < MT * QN; it += kThreads) {
      const int c4 = it & (QN - 1);
      const int m = it >> LG_QN;
      const int gx = m & (GW - 1);
      const int gy = (m >> p.lgGW) & (GH - 1);
      const int img = m >> (p.lgGW + p.lgGH);
      const int ly = gy0 + gy, lx = gx0 + gx, b = b0 + img;
      if (gy < 1 || gy > GH - 2 || gx < 1 || gx > GW - 2) continue;     // recomputed halo rows
      if (ly >= p.H || lx >= p.W || b >= p.B) continue;                  // ragged tile edge
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
          const int oy = 2 * ly + a, ox = 2 * lx + bb;
          f4 v = out[a][bb];
          if (has_noise) {
            const float nz = MIGAN_FMUL_RN(p.noise[(size_t)oy * p.WO + ox], ns);
            v += nz;
          }
          v = act4(v);
          const size_t off = (((size_t)b * p.HO + oy) * p.WO + ox) * p.CO + n0 + c4 * 4;
          if (p.skip) v += ld4(p.skip + off);
          st4(p.y + off, v);
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Un-fused ToRGB (reference torgb 1x1 conv with bias + Upsample2d of the running image, :308-313)
// for layers whose output channels are split over several workgroups.  16 lanes per pixel, each
// lane strides over the channel float4s, then a 4-step wave-shuffle butterfly.
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) torgb_kernel(const RgbArgs p) {
  const int sub = threadIdx.x & 15;
  const size_t pixel = ((size_t)blockIdx.x * kThreads + threadIdx.x) >> 4;
  const size_t npix = (size_t)p.B * p.H * p.W;
  const bool ok = pixel < npix;
  float r0 = 0.f, r1 = 0.f, r2 = 0.f;
  if (ok) {
    const float* xp = p.x + pixel * p.C;
    for (int q = sub; q < (p.C >> 2); q += 16) {
      const f4 v = ld4(xp + q * 4);
      const f4 w0 = ld4(p.w + q * 4), w1 = ld4(p.w + p.C + q * 4), w2 = ld4(p.w + 2 * p.C + q * 4);
      r0 += v.x * w0.x + v.y * w0.y + v.z * w0.z + v.w * w0.w;
      r1 += v.x * w1.x + v.y * w1.y + v.z * w1.z + v.w * w1.w;
      r2 += v.x * w2.x + v.y * w2.y + v.z * w2.z + v.w * w2.w;
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
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float up = 0.0f;
      if (p.img_prev) up = up_prev3(p.img_prev + ((size_t)b * 3 + ch) * (plane >> 2), p.H >> 1, p.W >> 1, oy, ox);
      p.img_out[((size_t)b * 3 + ch) * plane + rem] = up + rgb[ch];
    }
  }
}

}  // namespace migan
