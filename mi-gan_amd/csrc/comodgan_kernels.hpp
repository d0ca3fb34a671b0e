// Co-Mod-GAN generator forward: gfx950 (MI355X / CDNA4) device kernels.
//
// Reference being replaced: lib/model_zoo/comodgan.py::Generator.forward (:435-455) = Mapping (stylegan.py:396-439),
// Encoder (comodgan.py:192-204), Synthesis (comodgan.py:395-420); layers stylegan.py: dense :64-99,
// modulated_conv2d :102-195, conv2d_layer :197-244, synthesis_layer :247-309, torgb_layer :312-344;
// resampling torch_utils/ops/conv2d_resample.py and upfirdn2d.py.
//
// Everything that is a dense contraction (the 3x3 convolutions, 99 % of the 240 GFLOP per 512x512 image) runs
// on the matrix cores as an implicit GEMM (cm_conv_kernel); the rest are streaming kernels:
//
//   cm_conv_kernel      3x3 convolution, NHWC, M = 8x16 output-grid pixels, N = 64/128 output channels, K = taps x Cin.
//                       A operand: the input halo tile of a 32-channel chunk is staged ONCE in LDS (scaled by the
//                       per-sample style = the "scale activations" form of weight modulation, stylegan.py:171-182,
//                       split into two fp16 planes) and re-read at 1..9 shifted positions, one per filter tap.
//                       B operand: per (tap, chunk) weight tile, pre-split fp16 planes, double-buffered in LDS.
//                       v_mfma_f32_32x32x16_f16 x 3 per fp32 product (error-compensated, fp32 accumulate), same
//                       scheme as the MI-GAN 1x1 GEMM (migan_kernels.hpp, GEMMV 2).
//                       Epilogue: per-(sample, channel) demodulation coefficient, noise, bias, lrelu*sqrt2, clamp, skip.
//                       The tap list is data: plain 3x3 (9 taps, stride 1), stride-2 on the FIR-filtered input
//                       (encoder down path), and the four output phases of the stride-2 transposed convolution
//                       (synthesis up path: 4 + 2 + 2 + 1 taps, no multiplications by inserted zeros) -- one launch per
//                       phase, or all four in one launch with one accumulator set per phase.
//                       Tiles: 8x16 pixels x 64/128 channels (two waves per SIMD) or 16x16 pixels x 256 channels (one
//                       wave per SIMD, 128x128 wave tiles, accumulators in AGPRs), chosen per launch by the host.
//   cm_fir_kernel<0>    upfirdn2d [1,3,3,1] FIR with pad 2 in front of the strided convolution (conv2d_resample down path)
//   cm_fir_kernel<1>    upfirdn2d FIR (gain 4) behind the transposed convolution + noise/bias/activation/skip epilogue
//   cm_fromrgb_kernel   1x1 conv 4 -> C with bias and activation, NCHW planes -> NHWC
//   cm_torgb_kernel     modulated 1x1 conv C -> 3 (no demodulation) + bias + 2x FIR upsample of the running image
//   cm_dense_kernel     fully connected layers (mapping, affine, encoder fc, synthesis fc): weight streaming, fp32 FMA
//   cm_wprep_kernel     per-tensor statistics: max |w| per output channel (fp16 range scale), demodulation sums of w^2
//   cm_style_multi_kernel  styles -> normalised input scales (with the per-sample power-of-two fp16 range scale) and
//                       demodulation coefficients, or (ToRGB) the per-sample modulated 1x1 weights
//   cm_split_conv_kernel  3x3 weights -> two fp16 planes, [plane][tap][Cin/32][Cout][32]
//
// Compiled twice like migan_kernels.hpp: by hipcc for gfx950 and by the host compiler against tests/emu/hip_emu.h.
#pragma once

namespace migan {

constexpr int kCmMaxTaps = 9;
constexpr float kCmInBound = 512.0f;        // |conv input| <= 256 (lrelu_agc clamp) + 256 (skip added after the activation)
constexpr float kCmF16Top = 32768.0f;       // scaled A operands stay below 2^15 (fp16 max 65504)

struct CmConvArgs {
  const float* x;               // NHWC [B][H][W][CI]
  float* y;                     // NHWC [B][HO][WO][CO]
  const float* skip;            // NHWC like y, added after the activation, or null
  const unsigned short* wsplit; // fp16 planes [2][9][CI/32][CO][32] behind a 16-byte header (cm_split_conv_kernel)
  const float* sa;              // [B][CI] per-sample input scale (normalised style x 2^e) or null -> a_scale
  const float* coef;            // [B][CO] per-sample output coefficient or null -> cgain
  const float* bias;            // [CO]
  const float* noise;           // [HO][WO] (+ noise_bstride floats per image) or null
  const float* noise_strength;  // scalar
  long long noise_bstride;
  float a_scale, cgain;
  int B, H, W, CI, CO, HO, WO;
  int stride;                   // input pixels per output-grid pixel (1, or 2 for the strided convolution)
  int ntaps;
  int dy[kCmMaxTaps], dx[kCmMaxTaps], wtap[kCmMaxTaps];   // input offset of each tap (relative to grid*stride), weight tap plane
  int dymin, dxmin, IH, IW;     // input tile origin offset and extent for an 8x16 grid tile
  int oy_mul, oy_add, ox_mul, ox_add;   // output pixel of grid pixel (gy, gx)
  int GHn, GWn;                 // grid extent; grid pixels beyond are not stored
  int tiles_x, tiles_y, nchunks;
  int raw;                      // 1: store acc * coef only (transposed-convolution phases; cm_fir_kernel<1> finishes the layer)
  int off_b;                    // LDS carve in bytes: start of the B tile buffers (the result tile aliases everything)
  unsigned long long* prof;     // phase-cycle accumulators (MIGAN_PHASE_PROF builds only), else null
};

// LDS layout of the A (input tile) and B (weight tile) operands: one row per pixel / output channel holding both fp16
// planes of the KC channels back to back plus 16 bytes of padding: pitch = 2 * KC * 2 + 16 bytes (144 for KC = 32, 80 for
// KC = 16).  16 consecutive rows of one 16-byte slot then fall into 16 different bank quads (pitch / 4 mod 64 = 36 resp. 20
// generates all multiples of 4), and -- unlike an XOR swizzle -- the address of a shifted row is the address of the row plus
// a constant, so one filter tap costs two VALU adds of address arithmetic and every other offset is an immediate.
// (Every VALU instruction in the tap loop costs matrix-core time on gfx950: MFMA and VALU issue cycles add up.)
template <int V>
struct IntT { static constexpr int value = V; };

// GEMM row r (0..127 of the workgroup tile; MFMA row = r % 32 of its 32-row tile) -> grid pixel gy * 16 + gx of the
// 8 x 16 tile.  ds_read_b128 is served in lane groups {0-3,12-15,20-27} and {4-11,16-19,28-31} (and the same + 32): with
// the identity map and the 18-pixel tile rows of the plain 3x3 mode, lanes 12,13 and 26,27 of a group read pixels 16
// rows apart = the same banks.  Swapping which pixels of the second tile row the lanes 16-31 take (16-19 -> columns
// 0,1,10,11; 20-27 -> 2..9; 28-31 -> 12..15) gives every group 16 pixels that are distinct modulo 16.
MIGAN_DEVICE MIGAN_INLINE int cm_pixel_of_row(int r) {
  const int k = r & 15;
  int col = k;
  if (r & 16) col = k < 2 ? k : (k < 4 ? k + 8 : (k < 12 ? k - 2 : k));
  return (r & ~15) | col;
}

//   NT  : output channels per workgroup (64 / 128 / 256)
//   KC  : input channels per K chunk (32; 16 for the strided mode, whose 17x33-pixel input tile would otherwise
//         leave room for one workgroup per CU only)
//   NIA : float4 input-tile items per thread per chunk = ceil(tile pixels * KC/4 / 256) (prefetch registers)
//   NINE: the tap list has exactly nine entries and CI / KC is even (plain and strided 3x3): K loop unrolled over two
//         chunks, two weight tiles in flight
//   MTI : 32-row MFMA tiles per wave along M: the workgroup owns MT = 64 * MTI grid pixels ((4 * MTI) x 16).  MTI = 4 (16 x 16
//         pixels): every weight tile fetched from L2 feeds twice the MFMAs -- the weight-tile stream (MT-independent bytes per
//         workgroup and tap) is what saturates the per-CU vector-memory path with 128-pixel tiles
//   Register budget: accumulators MTI * NT/64 * 16; more than 64 of them -> one workgroup per CU, one wave per SIMD (512 registers)
//   UP4 : all four output phases of the stride-2 transposed convolution in one launch (synthesis conv0): the nine taps of the 3x3
//         kernel, each feeding the accumulator set of its phase (tap (ky, kx) -> output parity (ky == 1, kx == 1), input shift
//         (-(ky == 2), -(kx == 2))).  The low-resolution input tile is staged once per chunk for all phases instead of once per
//         phase launch, and the K loop has nine taps per chunk (the unrolled NINE path) instead of 1 / 2 / 2 / 4.
template <int NT, int KC, int NIA, bool NINE, int MTI, bool UP4 = false>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, ((MTI * NT * (UP4 ? 4 : 1) > 512) ? 1 : 2)) cm_conv_kernel(const CmConvArgs p) {
  MIGAN_DYN_SMEM(smem);
  constexpr int MT = 64 * MTI, GW = 16, GH = MT / GW, WROWS = 32 * MTI;
  constexpr int WCOLS = NT / 2, NTI = WCOLS / 32;
  constexpr int GS = NT + 4;
  constexpr int RB = KC * 2;                                  // bytes of one plane of one row
  constexpr int PB = 2 * RB + 16;                             // LDS row pitch in bytes (both planes + pad)
  constexpr int NSLOT = KC / 8, QK = KC / 4;                  // 16-byte slots / float4 quads per row and plane
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;

  // logical tile: Cout chunk fastest, then x, y, image (XCD-contiguous ranges share halo rows and weight tiles in L2)
  int t = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int nc = t % p.nchunks; t /= p.nchunks;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y; t /= p.tiles_y;
  const int b = t;
  const int co0 = nc * NT;
  const int gy0 = ty * GH, gx0 = tx * GW;
  const int iy0 = gy0 * p.stride + p.dymin, ix0 = gx0 * p.stride + p.dxmin;
  const int npix = p.IH * p.IW;
  const int nck = p.CI / KC;

  char* a_s = reinterpret_cast<char*>(smem);                  // [npix][PB]
  char* b_s = reinterpret_cast<char*>(smem) + p.off_b;        // [2 buffers][NT][PB]
  float* g_s = smem;                                          // [64][GS] per epilogue pass, after the K loop
  constexpr int b_buf = NT * PB;

  const float* __restrict__ xb = p.x + (size_t)b * p.H * p.W * p.CI;
  const float* __restrict__ sab = p.sa ? p.sa + (size_t)b * p.CI : nullptr;
  const unsigned w_plane_bytes = (unsigned)(9 * p.CI * p.CO) * 2u;     // bytes of one weight plane (< 2^23)
  [[maybe_unused]] const int total = nck * p.ntaps;            // generic tap list only

  constexpr int NPH = UP4 ? 4 : 1;                            // accumulator sets (output phases)
  static_assert(!UP4 || NINE, "the four-phase mode runs the nine-tap K loop");
  f16v acc[NPH][MTI][NTI];
#pragma unroll
  for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
    for (int i = 0; i < MTI; ++i)
#pragma unroll
      for (int j = 0; j < NTI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ph][i][j][r] = 0.0f;
  // phase profile (MIGAN_PHASE_PROF builds): 0 prologue, 1 load issue, 2 LDS reads + MFMAs, 3 weight tile -> LDS (incl. the wait for
  // its loads), 4 barrier, 5 input tile -> LDS + barrier, 6 epilogue
  PROF_BEGIN();

  // ---- input-tile items of this thread: item = tid + k*256 -> pixel q = item / QK, channel quad c4 = item % QK
  // (c4 is the same for every k).  goff = element offset of the pixel's channel 0 in the image, -1 = zero padding / no item.
  const int c4 = tid & (QK - 1);
  int goff[NIA];
#pragma unroll
  for (int k = 0; k < NIA; ++k) {
    const int q = (tid + k * 256) / QK;
    const int py = q / p.IW, px = q - py * p.IW;
    const int iy = iy0 + py, ix = ix0 + px;
    goff[k] = (q < npix && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? (iy * p.W + ix) * p.CI + c4 * 4 : -1;
  }
  f4 areg[NIA];
  auto load_a = [&](int c) {
#pragma unroll
    for (int k = 0; k < NIA; ++k) {
      areg[k] = f4{0.f, 0.f, 0.f, 0.f};
      if (goff[k] >= 0) areg[k] = ld4(xb + (size_t)(unsigned)goff[k] + c * KC);
    }
  };
  auto store_a = [&](int c) {
    f4 sc = f4{p.a_scale, p.a_scale, p.a_scale, p.a_scale};
    if (sab) sc = ld4(sab + c * KC + c4 * 4);
#pragma unroll
    for (int k = 0; k < NIA; ++k) {
      const int q = (tid + k * 256) / QK;
      if (q < npix) {
        u2v h1, h2;
        split2_f16(areg[k] * sc, h1, h2);
        char* dst = a_s + q * PB + c4 * 8;
        *reinterpret_cast<u2v*>(dst) = h1;
        *reinterpret_cast<u2v*>(dst + RB) = h2;
      }
    }
  };

  // ---- weight tile of (tap plane wt, channel chunk c): rows co0..co0+NT-1 of [tap][CI/32][CO][32], both planes.
  // Per-thread pieces (16 bytes each): the lane byte offset inside the tile and the LDS destination are constants of the
  // thread; the tile's base address is wave-uniform (SGPR pair + 32-bit lane offset: no 64-bit VALU address adds).
  constexpr int BPIECES = 2 * NT * NSLOT / 256;               // 16-byte pieces per thread
  unsigned bsrc[BPIECES];
  int bdst[BPIECES];
#pragma unroll
  for (int k = 0; k < BPIECES; ++k) {
    const int piece = tid + k * 256;                          // [plane][row][slot]
    const int pl = piece / (NT * NSLOT), rs = piece % (NT * NSLOT);
    const int row = rs / NSLOT, slot = rs % NSLOT;
    bsrc[k] = (unsigned)pl * w_plane_bytes + (unsigned)(row * 64 + slot * 16);
    bdst[k] = row * PB + pl * RB + slot * 16;
  }
  f4 breg[2][BPIECES];                                        // two register sets: the NINE path keeps two tiles in flight
  auto load_b = [&](int c, int tp, auto set) {
    constexpr int S = decltype(set)::value;
    const int c32 = (c * KC) >> 5, hc = ((c * KC) & 31) >> 3;  // 32-channel chunk of the planes, first 16-byte slot inside it
    const float* src = reinterpret_cast<const float*>(p.wsplit + ((size_t)(p.wtap[tp] * (p.CI >> 5) + c32) * p.CO + co0) * 32 + hc * 8);
#pragma unroll
    for (int k = 0; k < BPIECES; ++k) breg[S][k] = ld4(at_bytes(src, bsrc[k]));
  };
  auto store_b = [&](int buf, auto set) {
    constexpr int S = decltype(set)::value;
#pragma unroll
    for (int k = 0; k < BPIECES; ++k) st4(reinterpret_cast<float*>(b_s + buf * b_buf + bdst[k]), breg[S][k]);
  };
  // ---- MFMA operand addresses: lane (l31, half) supplies row l31 of a 32-row tile, k = 8 * half .. + 7 of a 16-k step
  int a_off[MTI], b_off[NTI];
#pragma unroll
  for (int i = 0; i < MTI; ++i) {
    const int m = cm_pixel_of_row(wm * WROWS + i * 32 + l31);
    a_off[i] = (((m >> 4) * p.stride) * p.IW + (m & 15) * p.stride) * PB + half * 16;
  }
#pragma unroll
  for (int j = 0; j < NTI; ++j) b_off[j] = (wn * WCOLS + j * 32 + l31) * PB + half * 16;
  // MFMAs of one filter tap: A rows = the staged input tile shifted by the tap's offset, B = weight tile in LDS buffer `buf`
  auto mfma_tap = [&](int tp, int buf, auto phc) {
    constexpr int PH = decltype(phc)::value;                   // accumulator set of this tap
    const int delta = ((p.dy[tp] - p.dymin) * p.IW + (p.dx[tp] - p.dxmin)) * PB;       // wave-uniform
    const char* aa[MTI];
    const char* bb[NTI];
#pragma unroll
    for (int i = 0; i < MTI; ++i) aa[i] = a_s + (a_off[i] + delta);
#pragma unroll
    for (int j = 0; j < NTI; ++j) bb[j] = b_s + (b_off[j] + buf * b_buf);
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      f4 av[MTI][2], bv[NTI][2];
#pragma unroll
      for (int i = 0; i < MTI; ++i) {
        av[i][0] = ld4(reinterpret_cast<const float*>(aa[i] + ks * 32));
        av[i][1] = ld4(reinterpret_cast<const float*>(aa[i] + ks * 32 + RB));
      }
#pragma unroll
      for (int j = 0; j < NTI; ++j) {
        bv[j][0] = ld4(reinterpret_cast<const float*>(bb[j] + ks * 32));
        bv[j][1] = ld4(reinterpret_cast<const float*>(bb[j] + ks * 32 + RB));
      }
      // product-major order: consecutive MFMAs write different accumulators (a dependent MFMA issued straight after
      // its producer waits out the 16-pass latency), smallest products first
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int i = 0; i < MTI; ++i)
#pragma unroll
          for (int j = 0; j < NTI; ++j)
            acc[PH][i][j] = MIGAN_MFMA_F16_32X32X16(av[i][pr == 0 ? 1 : 0], bv[j][pr == 1 ? 1 : 0], acc[PH][i][j]);
    }
  };

  if constexpr (!NINE) {
    // ---- generic tap list (the 1/2/4-tap phases of the transposed convolution): weight tile prefetched one tap ahead.
    // One filter tap of channel chunk c: issue the next weight tile's loads (and, on the last tap of the chunk, the next
    // chunk's input-tile loads: they fly under this tap's MFMAs and are consumed by store_a right after the barrier),
    // MFMAs of this tap from LDS, next weight tile -> LDS, barrier.  The prefetches are unconditional (the last ones
    // re-load the last tile) so that no branch sits between a load and its use.
    auto tap_body = [&](int c, int tp, auto prefetch_a) {
      const int it = c * p.ntaps + tp;
      int cn = c, tn = tp + 1;
      if (tn == p.ntaps) { tn = 0; ++cn; }
      if (cn == nck) { cn = c; tn = tp; }
      load_b(cn, tn, IntT<0>{});
      if constexpr (decltype(prefetch_a)::value) load_a(c + 1 < nck ? c + 1 : c);
      if constexpr (MTI == 2 && !decltype(prefetch_a)::value) {
        // same pipeline hints as the nine-tap path: operand reads, weight loads spread over the first step's MFMAs
        mfma_tap(tp, it & 1, IntT<0>{});
        constexpr int M = MTI * NTI * 3, RD = (MTI + NTI) * 2;
        constexpr int A1 = M / BPIECES > 0 ? M / BPIECES : 1;
        MIGAN_SCHED_GROUP(0x100, RD);
#pragma unroll
        for (int k = 0; k < BPIECES; ++k) {
          MIGAN_SCHED_GROUP(0x008, A1);
          MIGAN_SCHED_GROUP(0x020, 1);
        }
        if constexpr (M - A1 * BPIECES > 0) MIGAN_SCHED_GROUP(0x008, M - A1 * BPIECES);
        MIGAN_SCHED_GROUP(0x100, RD);
        MIGAN_SCHED_GROUP(0x008, M);
      } else {
        MIGAN_SCHED_FENCE();        // keep the loads ahead of the MFMAs (the scheduler otherwise sinks them to their use)
        mfma_tap(tp, it & 1, IntT<0>{});
      }
      MIGAN_SCHED_FENCE();
      store_b((it + 1) & 1, IntT<0>{});
      __syncthreads();
    };
    load_a(0);
    load_b(0, 0, IntT<0>{});
    store_b(0, IntT<0>{});
    for (int c = 0; c < nck; ++c) {
      // input halo tile of channel chunk c: registers -> x style scale -> two fp16 planes in LDS
      store_a(c);
      __syncthreads();
      for (int tp = 0; tp + 1 < p.ntaps; ++tp) tap_body(c, tp, FalseT{});
      tap_body(c, p.ntaps - 1, TrueT{});
    }
  } else {
    // ---- nine taps (plain and strided 3x3): two weight tiles are kept in flight in two register sets.  The K loop is
    // straight-line code over two channel chunks (18 taps: the register set of a tile is its global tap index mod 2, a
    // compile-time constant), which also lets the compiler count outstanding loads exactly (no s_waitcnt vmcnt(0) at
    // control-flow joins).
    //   tap `it`:  issue loads of tile it+2 -> set it%2 | MFMAs of tap it from LDS buffer it%2 |
    //              tile it+1 (set (it+1)%2, in flight since tap it-1) -> LDS buffer (it+1)%2 | barrier
    // The next chunk's input tile is loaded two taps before the chunk ends, ahead of that tap's weight loads, so waiting
    // for it at the chunk boundary leaves the newer weight loads in flight.
    auto tap9 = [&](int c, auto tpc, auto parc) {
      constexpr int TP = decltype(tpc)::value, PAR = decltype(parc)::value;
      constexpr int TN = (TP + 2) % 9, CN = (TP + 2) / 9;      // tile two taps ahead
      if constexpr (TP == 7) load_a(c + 1 < nck ? c + 1 : c);
      const bool inside = c + CN < nck;                        // beyond the end: re-load the last tile (never used)
      load_b(inside ? c + CN : nck - 1, inside ? TN : 8, IntT<PAR>{});
      if constexpr (MTI == 2) {
        // Two-waves-per-SIMD tiles: instruction-class pipeline hints instead of hard fences (phase profile: 30 % of a tap went
        // to issuing the weight loads, storing the previous tile to LDS and the barrier, serialised around the MFMAs): first
        // 16-k step's operand reads, the weight-tile loads spread over its MFMAs, then the second step.  Measured +8..12 % on
        // the 64- and 128-column kernels; the 512-register 16 x 16 tiles lose 5 % with it and keep the fences.
        mfma_tap(TP, PAR, IntT<(UP4 ? (((TP / 3) == 1) * 2 + ((TP % 3) == 1)) : 0)>{});
        store_b(PAR ^ 1, IntT<PAR ^ 1>{});
        constexpr int M = MTI * NTI * 3, RD = (MTI + NTI) * 2, NKS = KC / 16;
        constexpr int A1 = M / BPIECES > 0 ? M / BPIECES : 1;
        MIGAN_SCHED_GROUP(0x100, RD);
#pragma unroll
        for (int k = 0; k < BPIECES; ++k) {
          MIGAN_SCHED_GROUP(0x008, A1);
          MIGAN_SCHED_GROUP(0x020, 1);
        }
        if constexpr (M - A1 * BPIECES > 0) MIGAN_SCHED_GROUP(0x008, M - A1 * BPIECES);
        if constexpr (NKS == 2) {
          MIGAN_SCHED_GROUP(0x100, RD);
#pragma unroll
          for (int k = 0; k < BPIECES; ++k) {
            MIGAN_SCHED_GROUP(0x008, A1);
            MIGAN_SCHED_GROUP(0x200, 1);
          }
          if constexpr (M - A1 * BPIECES > 0) MIGAN_SCHED_GROUP(0x008, M - A1 * BPIECES);
        } else {
          MIGAN_SCHED_GROUP(0x200, BPIECES);
        }
        __syncthreads();
      } else {
        MIGAN_SCHED_FENCE();
        PROF_MARK(1);
        mfma_tap(TP, PAR, IntT<(UP4 ? (((TP / 3) == 1) * 2 + ((TP % 3) == 1)) : 0)>{});
        MIGAN_SCHED_FENCE();
        PROF_MARK(2);
        store_b(PAR ^ 1, IntT<PAR ^ 1>{});
        PROF_MARK(3);
        __syncthreads();
        PROF_MARK(4);
      }
    };
    auto chunk9 = [&](int c, auto cpar) {
      constexpr int CP = decltype(cpar)::value;         // parity of the chunk's first global tap index
      store_a(c);
      __syncthreads();
      PROF_MARK(5);
      tap9(c, IntT<0>{}, IntT<CP>{});     tap9(c, IntT<1>{}, IntT<CP ^ 1>{}); tap9(c, IntT<2>{}, IntT<CP>{});
      tap9(c, IntT<3>{}, IntT<CP ^ 1>{}); tap9(c, IntT<4>{}, IntT<CP>{});     tap9(c, IntT<5>{}, IntT<CP ^ 1>{});
      tap9(c, IntT<6>{}, IntT<CP>{});     tap9(c, IntT<7>{}, IntT<CP ^ 1>{}); tap9(c, IntT<8>{}, IntT<CP>{});
    };
    load_a(0);
    load_b(0, 0, IntT<0>{});
    load_b(0, 1, IntT<1>{});
    store_b(0, IntT<0>{});
    PROF_MARK(0);
    for (int c = 0; c < nck; c += 2) {      // nck is even (host check)
      chunk9(c, IntT<0>{});
      chunk9(c + 1, IntT<1>{});
    }
  }

  // ---- epilogue, one pass per MFMA row tile i (64 GEMM rows: rows i*32..i*32+31 of both wave rows): accumulators -> LDS result
  // tile -> per float4: coefficient, noise, bias, activation, skip
  const float inv_wscale = 1.0f / reinterpret_cast<const float*>(p.wsplit)[-2];      // power of two (cm_split_conv_kernel)
  const float ns = p.noise ? p.noise_strength[0] : 0.0f;
  constexpr int QN = NT / 4;
  static_assert(256 % QN == 0 && (64 * QN) % 256 == 0, "epilogue items: one channel quad per thread");
  const int eq_q4 = tid % QN, eq_row = tid / QN, eq_co = co0 + eq_q4 * 4;
  const f4 eq_cf = (p.coef ? ld4(p.coef + (size_t)b * p.CO + eq_co) : f4{p.cgain, p.cgain, p.cgain, p.cgain}) * inv_wscale;
  const bool raw_out = UP4 || p.raw != 0;                      // (the four-phase launch always writes the raw tensor: cm_fir_kernel<1> finishes the layer)
  const f4 eq_bias = raw_out ? f4{0.f, 0.f, 0.f, 0.f} : ld4(p.bias + eq_co);
#pragma unroll
  for (int ph = 0; ph < NPH; ++ph) {
    // four-phase mode: phase (ey, ex) writes raw[2 g + e]; its grid extent is H + (ey == 0) by W + (ex == 0)
    const int ey = ph >> 1, ex = ph & 1;
    const int ghn = UP4 ? p.H + (ey == 0) : p.GHn, gwn = UP4 ? p.W + (ex == 0) : p.GWn;
    const int oym = UP4 ? 2 : p.oy_mul, oya = UP4 ? ey : p.oy_add, oxm = UP4 ? 2 : p.ox_mul, oxa = UP4 ? ex : p.ox_add;
#pragma unroll
    for (int i = 0; i < MTI; ++i) {
      if (ph + i > 0) __syncthreads();       // the previous pass has been read (first pass: the K loop ended with a barrier)
#pragma unroll
      for (int j = 0; j < NTI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lrow = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const int col = wn * WCOLS + j * 32 + l31;
          g_s[lrow * GS + col] = acc[ph][i][j][r];
        }
      __syncthreads();
      // a thread's items of a pass share the channel quad (256 % QN == 0): coefficient and bias are read once per workgroup (above);
      // the per-pixel operands (noise, skip) of FOUR items are requested before the first of them is finished and stored -- a load
      // consumed right behind the previous item's store made every item wait for that store (vmcnt retires in order)
      constexpr int IPT = 64 * QN / 256, GRP = UP4 ? 2 : (IPT < 4 ? IPT : 4);       // (the four-phase form sits at its 256-register cap: two items ahead there)
#pragma unroll
      for (int k0 = 0; k0 < IPT; k0 += GRP) {
        size_t o[GRP];
        bool ok[GRP];
        float nz[GRP];
        f4 sk[GRP];
#pragma unroll
        for (int g = 0; g < GRP; ++g) {
          const int lrow = eq_row + (k0 + g) * (256 / QN);
          const int m = cm_pixel_of_row((lrow >> 5) * WROWS + i * 32 + (lrow & 31));
          const int gy = gy0 + (m >> 4), gx = gx0 + (m & 15);
          ok[g] = gy < ghn && gx < gwn;
          const int oy = gy * oym + oya, ox = gx * oxm + oxa;
          o[g] = ok[g] ? (((size_t)b * p.HO + oy) * p.WO + ox) * p.CO + eq_co : 0;
          nz[g] = 0.0f;
          sk[g] = f4{0.f, 0.f, 0.f, 0.f};
          if (!raw_out && ok[g]) {
            if (p.noise) nz[g] = p.noise[(size_t)b * p.noise_bstride + (size_t)oy * p.WO + ox];
            if (p.skip) sk[g] = ld4once(p.skip + o[g]);
          }
        }
#pragma unroll
        for (int g = 0; g < GRP; ++g) {
          if (!ok[g]) continue;
          const int lrow = eq_row + (k0 + g) * (256 / QN);
          f4 v = ld4(g_s + lrow * GS + eq_q4 * 4) * eq_cf;
          if (!raw_out) {
            if (p.noise) v = v + MIGAN_FMUL_RN(nz[g], ns);
            v = act4(v + eq_bias);
            if (p.skip) v = v + sk[g];
          }
          st4o(p.y + o[g], v);
        }
      }
    }
  }
  PROF_MARK(6);
  PROF_END();
}

// ------------------------------------------------------------------------------------------------
// Per-tensor statistics of one 3x3 weight tensor, one workgroup per output channel:
//   amax[co]     = max |w[co]|                       (fp16 range scale of the split GEMM operands)
//   wsq[co][ci]  = sum_k w^2, wn2[co] = 1 / mean_{ci,k} w^2    (demodulation, stylegan.py:138,147; modulated layers only)
struct CmWprepArgs {
  const float* w;      // [CO][CI][3][3]
  float* amax;         // [CO]
  float* wsq;          // [CO][CI] or null
  float* wn2;          // [CO] or null
  int CO, CI;
};
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) cm_wprep_kernel(const CmWprepArgs p) {
  MIGAN_DYN_SMEM(red);
  const int co = (int)blockIdx.x;
  float tot = 0.0f, mx = 0.0f;
  for (int ci = threadIdx.x; ci < p.CI; ci += 256) {
    const float* s = p.w + ((size_t)co * p.CI + ci) * 9;
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      a += s[k] * s[k];
      mx = fmaxf(mx, fabsf(s[k]));
    }
    if (p.wsq) p.wsq[(size_t)co * p.CI + ci] = a;
    tot += a;
  }
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) {
    tot += __shfl_xor(tot, sft);
    mx = fmaxf(mx, __shfl_xor(mx, sft));
  }
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = tot;
    red[4 + (threadIdx.x >> 6)] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (p.wn2) p.wn2[co] = (float)(p.CI * 9) / (red[0] + red[1] + red[2] + red[3]);
    p.amax[co] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  }
}

// 3x3 weights [CO][CI][3][3] fp32 -> two fp16 planes [plane][tap][CI/32][CO][32] of w * wscale, wscale = the power of
// two that maps max|w| into [2^13, 2^14) (every workgroup derives it from amax[]; workgroup 0 also writes it to the
// 16-byte header in front of plane 0, float [2], where cm_conv_kernel reads it).
struct CmSplitArgs {
  const float* src;
  const float* amax;          // [CO]
  unsigned short* dst;        // plane 0; a 16-byte header precedes it
  int CO, CI;
};
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) cm_split_conv_kernel(const CmSplitArgs p) {
  MIGAN_DYN_SMEM(red);
  float m = 0.0f;
  for (int i = threadIdx.x; i < p.CO; i += 256) m = fmaxf(m, p.amax[i]);
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu) - 127;       // floor(log2(max|w|))
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  const float sw = __builtin_bit_cast(float, (unsigned)(127 + 13 - e) << 23);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float* hdr = reinterpret_cast<float*>(p.dst) - 4;
    hdr[0] = 1.0f / sw; hdr[1] = m; hdr[2] = sw; hdr[3] = 0.0f;
  }
  const size_t plane = (size_t)9 * p.CI * p.CO;
  const size_t n = (size_t)p.CO * p.CI;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int co = (int)(i / p.CI), ci = (int)(i % p.CI);
    const float* s = p.src + i * 9;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const float v = s[tp] * sw;
      const unsigned pk = MIGAN_PACK_F16(v, 0.0f);
      const float r = v - MIGAN_F16LO_F32(pk);
      const unsigned pk2 = MIGAN_PACK_F16(r, 0.0f);
      const size_t o = ((size_t)(tp * (p.CI / 32) + (ci >> 5)) * p.CO + co) * 32 + (ci & 31);
      p.dst[o] = (unsigned short)(pk & 0xffffu);
      p.dst[plane + o] = (unsigned short)(pk2 & 0xffffu);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// styles [B][CI] (output of the affine dense layer) ->
//   demod = 1 (synthesis_layer): s~ = styles * rsqrt(mean over batch and channels of styles^2)   (stylegan.py:139)
//             sa[b][ci]   = s~ * 2^e_b, e_b the largest power of two keeping |x * sa| < 2^15 for |x| <= kCmInBound
//             coef[b][co] = wn[co] * rsqrt(wn2[co] * sum_ci s~^2 wsq[co][ci] + 1e-8) / 2^e_b           (stylegan.py:138-147,161)
//   demod = 0 (torgb_layer): wm[b][3][ci] = w[c][ci] * styles[b][ci] * wgain                             (stylegan.py:337-338)
// One workgroup per sample.
struct CmStyleArgs {
  const float* styles;  // [B][CI]
  const float* wsq;     // [CO][CI]   (demod)
  const float* wn2;     // [CO]       (demod)
  const float* w;       // [3][CI]    (torgb)
  float* sa;            // [B][CI]    (demod)
  float* coef;          // [B][CO]    (demod)
  float* wm;            // [B][3][CI] (torgb)
  float wgain;
  int B, CI, CO, demod;
};
constexpr int kCmStyleSlice = 16;       // output channels per workgroup (demod); grid = B * ceil(CO / 16)
MIGAN_DEVICE MIGAN_INLINE void cm_style_block(const CmStyleArgs& p, int block) {
  MIGAN_DYN_SMEM(sm);            // [CI] s~^2 of this sample, then 8 floats of reduction scratch
  const int tid = threadIdx.x;
  if (!p.demod) {
    const int b = block;
    const float* st = p.styles + (size_t)b * p.CI;
    for (int i = tid; i < 3 * p.CI; i += 256) {
      const int ci = i % p.CI;
      p.wm[(size_t)b * 3 * p.CI + i] = p.w[i] * (st[ci] * p.wgain);
    }
    return;
  }
  const int nsl = (p.CO + kCmStyleSlice - 1) / kCmStyleSlice;
  const int b = block / nsl, sl = block % nsl;
  const float* st = p.styles + (size_t)b * p.CI;
  float* red = sm + p.CI;
  float ss = 0.0f;
  for (int i = tid; i < p.B * p.CI; i += 256) ss += p.styles[i] * p.styles[i];
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) ss += __shfl_xor(ss, sft);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float g = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)(p.B * p.CI));
  float mx = 0.0f;
  for (int ci = tid; ci < p.CI; ci += 256) {
    const float s = st[ci] * g;
    sm[ci] = s * s;
    mx = fmaxf(mx, fabsf(s));
  }
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sft));
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
  // 2^e <= kCmF16Top / (kCmInBound * mx): e = 6 - ceil(log2 mx); exponent arithmetic on the float bits
  int ex = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 0xffu) - 127;          // floor(log2 mx)
  ex = ex < -60 ? -60 : (ex > 60 ? 60 : ex);
  const int e = 6 - (ex + 1);                                                      // mx < 2^(ex+1)
  const float up = __builtin_bit_cast(float, (unsigned)(127 + e) << 23);
  const float dn = __builtin_bit_cast(float, (unsigned)(127 - e) << 23);
  if (sl == 0)
    for (int ci = tid; ci < p.CI; ci += 256) p.sa[(size_t)b * p.CI + ci] = st[ci] * g * up;
  // one wave per output channel of the slice: sum_ci s~^2 wsq[co][ci]
  const int lane = tid & 63, wave = tid >> 6;
  for (int k = wave; k < kCmStyleSlice; k += 4) {
    const int co = sl * kCmStyleSlice + k;
    if (co >= p.CO) break;
    const float* wq = p.wsq + (size_t)co * p.CI;
    float a = 0.0f;
    for (int ci = lane; ci < p.CI; ci += 64) a += sm[ci] * wq[ci];
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) a += __shfl_xor(a, sft);
    if (lane == 0) {
      const float n2 = p.wn2[co];
      p.coef[(size_t)b * p.CO + co] = sqrtf(n2) * (1.0f / sqrtf(n2 * a + 1e-8f)) * dn;
    }
  }
}

// Every style computation of the synthesis network in one launch: they depend only on the affine outputs (one launch, above)
// and on the weight statistics, not on activations, so the 23 small launches that used to sit between the convolutions
// (0.4 ms of a 19 ms forward at comodgan-512) become one.
constexpr int kCmMaxStyle = 32;
struct CmStyleMultiArgs {
  CmStyleArgs job[kCmMaxStyle];
  int blk0[kCmMaxStyle + 1];     // first workgroup of each job
  int njobs;
};
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) cm_style_multi_kernel(const CmStyleMultiArgs p) {
  int j = 0;
  while (j + 1 < p.njobs && (int)blockIdx.x >= p.blk0[j + 1]) ++j;
  cm_style_block(p.job[j], (int)blockIdx.x - p.blk0[j]);
}

// ------------------------------------------------------------------------------------------------
// dense (stylegan.py:64-99): y[n][o] = act((sum_k x[n][k] W[o][k]) * wgain + b[o] * bgain)
// Optional: per-row input normalisation x * rsqrt(mean(x^2) + 1e-8) (normalize_2nd_moment, stylegan.py:351-352),
// NHWC<->NCHW index permutations of the 4x4 bottleneck (in_c / out_c = channel count, 0 = none), `add` (same layout
// as the output, after the activation: comodgan.py:243), truncation lerp towards w_avg (stylegan.py:432-437),
// row-concatenated input (x2 supplies columns >= K1: torch.cat([w, w0]) of comodgan.py:247).
// One wave per 2 output features, 8 batch rows per pass.
struct CmDenseArgs {
  const float* x;      // [N][K1]
  const float* x2;     // [N][K-K1] or null
  const float* w;      // [O][K]
  const float* b;      // [O]
  const float* add;    // or null
  const float* lerp0;  // w_avg [O] or null
  float* y;            // [N][O]
  float* y_raw;        // with lerp0: also the value before the truncation lerp (the ws rows beyond truncation_cutoff), or null
  float wgain, bgain, psi;
  int N, K, K1, O;
  int act, norm, in_c, out_c;
};
MIGAN_DEVICE MIGAN_INLINE void cm_dense_block(const CmDenseArgs& p, int block) {
  constexpr int RP = 8;                      // batch rows per pass over the weights (16 spills)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o0 = (block * 4 + wave) * 2;
  if (o0 >= p.O) return;
  const int no = (o0 + 1 < p.O) ? 2 : 1;
  const int K2 = p.K - p.K1;
  const float* wr0 = p.w + (size_t)o0 * p.K;
  const float* wr1 = p.w + (size_t)(o0 + (no > 1 ? 1 : 0)) * p.K;
  for (int n0 = 0; n0 < p.N; n0 += RP) {
    float acc[2][RP], nrm[RP];
#pragma unroll
    for (int r = 0; r < RP; ++r) { acc[0][r] = 0.0f; acc[1][r] = 0.0f; nrm[r] = 0.0f; }
    if (p.in_c) {
      // The input is the NHWC 4x4 bottleneck [N][16][C] and feature k of the reference's NCHW flatten is c * 16 + pos: a lane
      // owns channel c, reads its 16 weights per output row as four float4 (64 contiguous bytes per lane) and the 16 inputs
      // x[n][pos][c] with consecutive lanes on consecutive channels -- every access coalesced (walking k in weight order
      // would gather the inputs at a 2 KB stride)
      for (int c = lane; c < p.in_c; c += 64) {
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {          // (kept rolled: unrolling all 16 positions x 16 rows spills)
          const f4 a = ld4(wr0 + (size_t)c * 16 + q * 4), bq = ld4(wr1 + (size_t)c * 16 + q * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int pos = q * 4 + e;
#pragma unroll
            for (int r = 0; r < RP; ++r) {
              const int n = n0 + r;
              const float xv = n < p.N ? p.x[(size_t)n * p.K1 + (size_t)pos * p.in_c + c] : 0.0f;
              acc[0][r] += xv * a[e];
              acc[1][r] += xv * bq[e];
              nrm[r] += xv * xv;
            }
          }
        }
      }
    } else {
      for (int k = lane; k < p.K; k += 64) {
        const float w0 = wr0[k];
        const float w1 = no > 1 ? wr1[k] : 0.0f;
#pragma unroll
        for (int r = 0; r < RP; ++r) {
          const int n = n0 + r;
          float xv = 0.0f;
          if (n < p.N) xv = (k < p.K1) ? p.x[(size_t)n * p.K1 + k] : p.x2[(size_t)n * K2 + (k - p.K1)];
          acc[0][r] += xv * w0;
          acc[1][r] += xv * w1;
          nrm[r] += xv * xv;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RP; ++r) {
#pragma unroll
      for (int sft = 32; sft >= 1; sft >>= 1) {
        acc[0][r] += __shfl_xor(acc[0][r], sft);
        acc[1][r] += __shfl_xor(acc[1][r], sft);
        if (p.norm) nrm[r] += __shfl_xor(nrm[r], sft);
      }
    }
    if (lane < 2 * RP) {
      const int r = lane & (RP - 1), j = lane / RP;
      const int n = n0 + r, o = o0 + j;
      if (n < p.N && j < no) {
        // select without dynamic register indexing
        float a = 0.0f, q = 0.0f;
#pragma unroll
        for (int rr = 0; rr < RP; ++rr)
          if (rr == r) { a = j ? acc[1][rr] : acc[0][rr]; q = nrm[rr]; }
        if (p.norm) a = a * (1.0f / sqrtf(q / (float)p.K + 1e-8f));
        float v = a * p.wgain + p.b[o] * p.bgain;
        if (p.act) v = act1(v);
        const int oi = p.out_c ? (o & 15) * p.out_c + (o >> 4) : o;
        if (p.add) v += p.add[(size_t)n * p.O + oi];
        if (p.lerp0) {
          if (p.y_raw) p.y_raw[(size_t)n * p.O + oi] = v;
          const float s = p.lerp0[o];
          v = (p.psi < 0.5f) ? s + p.psi * (v - s) : v - (v - s) * (1.0f - p.psi);     // torch.lerp
        }
        p.y[(size_t)n * p.O + oi] = v;
      }
    }
  }
}

MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) cm_dense_kernel(const CmDenseArgs p) { cm_dense_block(p, (int)blockIdx.x); }

// All affine layers of the synthesis network in one launch (they depend only on the latent w and the global code w0,
// stylegan.py:282): a table of (weight, bias, output, out_features); same input cat([w, w0]), no activation.
constexpr int kCmMaxAffine = 48;
struct CmDenseMultiArgs {
  const float* w[kCmMaxAffine];
  const float* b[kCmMaxAffine];
  float* y[kCmMaxAffine];
  int O[kCmMaxAffine];
  int blk0[kCmMaxAffine + 1];     // first workgroup of each job
  const float* x;
  const float* x2;
  const float* x_alt;             // truncation_cutoff: the un-truncated w, read by the jobs whose bit is set in alt_mask (stylegan.py:436-437)
  unsigned long long alt_mask;
  float wgain;
  int N, K, K1, njobs;
};
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) cm_dense_multi_kernel(const CmDenseMultiArgs p) {
  int j = 0;
  while (j + 1 < p.njobs && (int)blockIdx.x >= p.blk0[j + 1]) ++j;
  CmDenseArgs a{};
  a.x = ((p.alt_mask >> j) & 1ull) ? p.x_alt : p.x; a.x2 = p.x2; a.w = p.w[j]; a.b = p.b[j]; a.y = p.y[j];
  a.wgain = p.wgain; a.bgain = 1.0f; a.psi = 1.0f; a.N = p.N; a.K = p.K; a.K1 = p.K1; a.O = p.O[j];
  cm_dense_block(a, (int)blockIdx.x - p.blk0[j]);
}

// ------------------------------------------------------------------------------------------------
// conv2d_layer(4 -> C, kernel 1, bias, activation) of the first encoder block (comodgan.py:46-48, stylegan.py:231-244):
// NCHW network input -> NHWC features.  One thread per pixel and channel quad.
struct CmFromRgbArgs {
  const float* x;      // [B][4][R][R]
  const float* w;      // [C][4]
  const float* b;      // [C]
  float* y;            // [B][R][R][C]
  float wgain;
  int B, R, C;
};
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) cm_fromrgb_kernel(const CmFromRgbArgs p) {
  // a thread keeps one channel quad (its 4 x 4 weights and bias stay in registers) and walks pixels; 256 / (C/4) pixels per
  // workgroup step, consecutive lanes = consecutive channel quads of a pixel (1 KiB contiguous store per wave)
  const int qn = p.C >> 2;                      // 16 ... 256, divides 256
  const int c4 = (int)threadIdx.x % qn;
  const int ppb = 256 / qn;
  const size_t plane = (size_t)p.R * p.R;
  const size_t npix = (size_t)p.B * plane;
  f4 w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = ld4(p.w + (c4 * 4 + j) * 4) * p.wgain;
  const f4 bias = ld4(p.b + c4 * 4);
  for (size_t pix = (size_t)blockIdx.x * ppb + threadIdx.x / qn; pix < npix; pix += (size_t)gridDim.x * ppb) {
    const size_t bi = pix / plane, rem = pix % plane;
    const float* xp = p.x + bi * 4 * plane + rem;
    const float x0 = xp[0], x1 = xp[plane], x2 = xp[2 * plane], x3 = xp[3 * plane];
    f4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (x0 * w[j].x + x1 * w[j].y + x2 * w[j].z + x3 * w[j].w) + bias[j];
    st4(p.y + pix * p.C + c4 * 4, act4(v));
  }
}

// ------------------------------------------------------------------------------------------------
// upfirdn2d with the separable [1,3,3,1] filter (taps [1,3,3,1] * fs per axis), up = down = 1, zero padding `pad` on
// every side, NHWC: out[y][x] = sum_{a,b} f[a] f[b] in[y + a - pad][x + b - pad].  Each thread produces a 2 x 4 block of
// output pixels for one channel quad from a 5 x 7 input window (4.4 loads per output instead of 16), one input row at
// a time: horizontal taps into 4 row sums, which feed the two output rows.
//   EPI = 0: FIR in front of the strided convolution (conv2d_resample.py down path: pad = 2, gain 1, fs = 1/8),
//                      [B][H][W][C] -> [B][H+1][W+1][C]
//   EPI = 1: second half of an up=2 synthesis_layer: FIR (gain 4: fs = 1/4, pad 1) over the (2H+1)^2 output
//                      of the transposed convolution (conv2d_resample.py up path), then noise, bias, activation, skip
//                      (stylegan.py:300-309, comodgan.py:331-332); the demodulation coefficient is already applied
//                      (cm_conv_kernel raw mode)
struct CmFirArgs {
  const float* x;              // NHWC [B][H][W][C]
  float* y;                    // NHWC [B][HO][WO][C]
  const float* skip;           // EPI 1: NHWC like y or null
  const float* bias;           // EPI 1: [C]
  const float* noise;          // EPI 1: [HO][WO] (+ noise_bstride per image) or null
  const float* noise_strength;
  long long noise_bstride;
  float fs;
  int B, H, W, C, HO, WO, pad;
};
template <int EPI>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) cm_fir_kernel(const CmFirArgs p) {
  const int qn = p.C >> 2;
  const int nbx = (p.WO + 3) >> 2, nby = (p.HO + 1) >> 1;
  const size_t total = (size_t)p.B * nby * nbx * qn;
  const float f0 = p.fs, f1 = 3.0f * p.fs;
  const float ns = (EPI == 1 && p.noise) ? p.noise_strength[0] : 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % qn);
    size_t blk = i / qn;
    const int bx = (int)(blk % nbx); blk /= nbx;
    const int by = (int)(blk % nby);
    const int b = (int)(blk / nby);
    const int x0 = bx * 4, y0 = by * 2;
    const float* xb = p.x + (size_t)b * p.H * p.W * p.C + c4 * 4;
    // EPI 1: the skip tensor and the noise plane of the 2 x 4 output block are requested first, so that they travel with
    // the 35 window loads instead of after the arithmetic that needs them last
    f4 sk[2][4];
    float nz[2][4];
    f4 bias4 = {0.f, 0.f, 0.f, 0.f};                             // (read here, not between the stores below: a load behind a store waits for that store)
    if constexpr (EPI == 1) {
      bias4 = ld4(p.bias + c4 * 4);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int oy = y0 + r, ox = x0 + c;
          const bool ok = oy < p.HO && ox < p.WO;
          sk[r][c] = f4{0.f, 0.f, 0.f, 0.f};
          nz[r][c] = 0.0f;
          if (ok && p.skip) sk[r][c] = ld4once(p.skip + (((size_t)b * p.HO + oy) * p.WO + ox) * p.C + c4 * 4);
          if (ok && p.noise) nz[r][c] = p.noise[(size_t)b * p.noise_bstride + (size_t)oy * p.WO + ox];
        }
    }
    f4 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int iy = y0 - p.pad + r;
      const bool yok = iy >= 0 && iy < p.H;
      f4 v[7];
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        const int ix = x0 - p.pad + c;
        v[c] = f4{0.f, 0.f, 0.f, 0.f};
        if (yok && ix >= 0 && ix < p.W) v[c] = ld4(xb + ((size_t)iy * p.W + ix) * p.C);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f4 h = (v[c] + v[c + 3]) * f0 + (v[c + 1] + v[c + 2]) * f1;
        if (r < 4) acc[0][c] = acc[0][c] + h * ((r == 0 || r == 3) ? f0 : f1);
        if (r > 0) acc[1][c] = acc[1][c] + h * ((r == 1 || r == 4) ? f0 : f1);
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int oy = y0 + r;
      if (oy >= p.HO) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ox = x0 + c;
        if (ox >= p.WO) continue;
        const size_t o = (((size_t)b * p.HO + oy) * p.WO + ox) * p.C + c4 * 4;
        f4 v = acc[r][c];
        if constexpr (EPI == 1) {
          if (p.noise) v = v + MIGAN_FMUL_RN(nz[r][c], ns);
          v = act4(v + bias4);
          if (p.skip) v = v + sk[r][c];
        }
        st4o(p.y + o, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// torgb_layer (stylegan.py:330-344) with per-sample modulated weights wm [B][3][C] (cm_style_multi_kernel) + bias +
// upsample2d of the running image (comodgan.py:334-343).  16 lanes per pixel, wave-shuffle butterfly.
struct CmRgbArgs {
  const float* x;        // NHWC [B][H][W][C]
  const float* wm;       // [B][3][C]
  const float* bias;     // [3]
  const float* img_prev; // planar [B][3][H/2][W/2] or null
  float* img_out;        // planar [B][3][H][W]
  int B, H, W, C;
};
// LPP lanes per pixel (4 for 64 channels ... 16 for >= 256): each lane reads C / (4 LPP) float4 of the pixel, a butterfly over
// the LPP lanes forms the three sums, lane ch of the group finishes channel ch (bias, upsampled previous image) -- a wave
// writes 64 / LPP consecutive pixels of each plane.
template <int LPP>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) cm_torgb_kernel(const CmRgbArgs p) {
  static_assert(LPP == 4 || LPP == 8 || LPP == 16, "lanes per pixel");
  const int sub = threadIdx.x & (LPP - 1);
  const size_t pixel = ((size_t)blockIdx.x * 256 + threadIdx.x) / LPP;
  const size_t plane = (size_t)p.H * p.W;
  const size_t npix = (size_t)p.B * plane;
  const bool ok = pixel < npix;
  float r0 = 0.f, r1 = 0.f, r2 = 0.f;
  const int b = ok ? (int)(pixel / plane) : 0;
  if (ok) {
    const float* xp = p.x + pixel * p.C;
    const float* w = p.wm + (size_t)b * 3 * p.C;
    for (int q = sub; q < (p.C >> 2); q += LPP) {
      const f4 v = ld4(xp + q * 4);
      const f4 w0 = ld4(w + q * 4), w1 = ld4(w + p.C + q * 4), w2 = ld4(w + 2 * p.C + q * 4);
      // scalar FMA chains, not SLP-vectorised packed-fp32 dot products (profiles/r02_torgb_packed_f32_hazard.md)
      float a0, a1, a2;
      torgb_partial(v, w0, w1, w2, a0, a1, a2);
      r0 += a0;
      r1 += a1;
      r2 += a2;
    }
  }
#pragma unroll
  for (int s = LPP / 2; s >= 1; s >>= 1) {
    r0 += __shfl_xor(r0, s);
    r1 += __shfl_xor(r1, s);
    r2 += __shfl_xor(r2, s);
  }
  if (ok && sub < 3) {
    const int rem = (int)(pixel % plane);
    const int oy = rem / p.W, ox = rem % p.W;
    const float sum = sub == 0 ? r0 : (sub == 1 ? r1 : r2);
    float up = 0.0f;
    if (p.img_prev) up = up_prev3(p.img_prev + ((size_t)b * 3 + sub) * (plane >> 2), p.H >> 1, p.W >> 1, oy, ox);
    p.img_out[((size_t)b * 3 + sub) * plane + rem] = up + (sum + p.bias[sub]);
  }
}

}  // namespace migan
