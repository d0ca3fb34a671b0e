// MI-GAN generator forward: software-pipelined SeparableConv2d for the full-resolution layers (round 4).
//
// Same arithmetic as sepconv_kernel (reference lib/model_zoo/migan_inference.py:154-170: depthwise 3x3 + bias -> lrelu_agc -> 1x1 conv
// -> [2x FIR upsample] -> noise -> lrelu_agc -> [+ skip] -> [ToRGB]), other schedule.  The stage ablation of the round-3 kernels
// (profiles/r03_ablation_batch32.txt) shows their phases ADD UP: a workgroup loads, computes, then stores, and the 15 GB of output
// stores of a forward are fully exposed (27 % of the time).  Here ONE persistent workgroup per CU (12 or 16 waves) is a four-stage
// pipeline over its tiles, every stage on its own hardware queue:
//
//   DMA    buffer_load ... lds: input tile of K-chunk s+R-1 (and 1x1 weight planes) HBM/L2 -> LDS ring, no registers, issued by group A,
//          R-1 chunks (24-48 KB per CU) in flight across the barriers (counted s_waitcnt vmcnt, raw s_barrier)
//   A      NA waves: depthwise 3x3 + bias + act + fp16 hi/lo split of chunk s+1 -> A-operand planes (VALU + LDS)
//   B      8 waves (4 row blocks x 2 column halves of the 128 x NT tile): v_mfma_f32_32x32x16_f16 x 3 of chunk s (matrix pipe) ...
//   store  ... and, between their MFMA groups, the EPILOGUE OF THE PREVIOUS TILE: its accumulators wait in a second register set
//          (plain layers: transposed through a wave-private LDS patch, no barrier) or in a dedicated LDS result tile (FIR-up layers),
//          and a slice of its noise / activation / skip / ToRGB work and of its global stores is issued in every K step of the next
//          tile.  The stores of tile t therefore drain while tile t+1 is loaded and computed; B never waits for them (it issues no
//          DMA, and everything it loads is requested ahead of the stores that precede its use: vmcnt retires in issue order).
//
// One workgroup barrier per K chunk.  The 1x1 weight planes and the depthwise taps of ALL chunks stay in LDS for the life of the
// workgroup where they fit (Cin x Cout <= 128 x 64: the 512x512 layers), otherwise the planes stream through a two-slot ring.
// Why 12-16 waves: each stage is a dependent instruction stream (one wave issues a VALU instruction every ~5 cycles, an LDS round
// trip is > 100); the first form of this kernel (4 + 4 waves) was bound by the lone epilogue wave per SIMD
// (profiles/r04_pipe_phase_profile.txt), so the stages are spread over three to four waves per SIMD.
// fp32 activation storage, f16x2 GEMM (the default of that storage format); everything else keeps sepconv_kernel.
#pragma once

namespace migan {

constexpr int kPipeBWaves = 8;
constexpr int pipe_threads(int na) { return (na + kPipeBWaves) * 64; }

// LDS carve of one instantiation (bytes), shared with the host plan
template <int MODE, int NT, int CIN, bool FROMRGB, int R>
struct PipeLds {
  static constexpr int NKC = CIN / 32;
  static constexpr int IN_SLOT = 1536 * 16;                         // ring slot: [180 pixels][32 channels] fp32 + padding (1440 of 1536 16-byte units)
  static constexpr int A_BUF = 2 * 128 * 64;                        // hi + lo plane of the A operand, [128 rows][32 k] fp16 each
  static constexpr int B_CHUNK = 2 * NT * 64;                       // hi + lo plane of one K chunk of the weights, [NT rows][32 k] fp16
  static constexpr bool WRES = NKC * B_CHUNK <= 32 * 1024;          // all chunks resident
  static constexpr int NBUF_B = WRES ? NKC : 2;
  static constexpr int OFF_IN = 0;
  static constexpr int OFF_A = OFF_IN + R * IN_SLOT;
  static constexpr int OFF_B = OFF_A + 2 * A_BUF;
  static constexpr int OFF_W = OFF_B + NBUF_B * B_CHUNK;            // depthwise taps, per chunk tap-major [9][32] + bias [32]
  static constexpr bool TAPS_RES = NKC <= 8;                        // all chunks resident; otherwise a two-chunk ring refilled through registers
  // FROMRGB (the 1x1 conv 4 -> Cin as a bf16x3-split MFMA, K = 4 inputs + the bias against a "pixel is inside the image" flag):
  // K of one MFMA = the three bf16 pieces of a pixel side by side: k 0..4 = h1 [x0 x1 x2 x3 flag], k 5..9 = h2 [.. 0], k 10..14 = h3 [.. 0], k 15 = 0
  static constexpr int OFF_F = OFF_W + (TAPS_RES ? NKC : 2) * 1280; // its B operands: per chunk three arrangements [32 columns][16 bf16]: (b1 b1 b1), (b2 b2 0), (b3 0 0)
  static constexpr int F_SZ = FROMRGB ? NKC * 3 * 32 * 32 : 0;
  static constexpr int OFF_RGB = OFF_F + F_SZ;                      // its A operand, two tiles: [192 rows][16 bf16]
  static constexpr int RGB_BUF = 192 * 32;
  static constexpr int OFF_T = OFF_RGB + (FROMRGB ? 2 * RGB_BUF : 0);
  static constexpr int T_SZ = MODE == MODE_UP ? 128 * (64 + 4) * 4 : kPipeBWaves * 32 * 32 * 4;   // FIR-up: shared result tile of 64 columns (a 128-column
                                                                    // layer passes its two halves through it one after the other); plain: one transpose patch per B wave
  static constexpr int OFF_P = OFF_T + T_SZ;                       // plain + ToRGB: per-pixel partial sums of the waves of column half 1, [128 pixels][4]
  static constexpr int OFF_PW = OFF_P + (MODE == MODE_UP ? 0 : 128 * 16);      // ToRGB: the window of the previous (half-resolution) image under a tile,
  static constexpr int PW_SZ = 3 * 64 * 4;                                       // two tiles x [3 colours][6 rows x 10 columns, padded to 64] fp32
  static constexpr int TOTAL = OFF_PW + (MODE == MODE_UP ? 0 : 2 * PW_SZ);
  static_assert(TOTAL <= 160 * 1024, "LDS budget");
};

// phase profile of this kernel (-DMIGAN_PHASE_PROF builds): 16 accumulators; group A -> slots 0..3 [DMA issue + offsets / tile build,
// depthwise, vmcnt wait, barrier], group B -> 4 MFMAs, 5 first use of the prefetched noise / taps, 6 barrier, 7 hand-over + rest, 9 block
// epilogue (transpose, activation, stores), 10 ToRGB partial sums, 11 ToRGB tail, 12 prefetch requests; slot 8 counts workgroups
#ifdef MIGAN_PHASE_PROF
#define PPROF_BEGIN() long long pprof_t = (long long)MIGAN_CLOCK(); long long pprof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PPROF_MARK(i) do { const long long n_ = (long long)MIGAN_CLOCK(); pprof_acc[i] += n_ - pprof_t; pprof_t = n_; } while (0)
#define PPROF_END(AT_) do { if (p.prof && (tid == 0 || tid == (AT_))) { for (int i_ = 0; i_ < 16; ++i_) if (i_ != 8) MIGAN_ATOMIC_ADD_U64(p.prof + i_, (unsigned long long)pprof_acc[i_]); if (tid == 0) MIGAN_ATOMIC_ADD_U64(p.prof + 8, 1ull); } } while (0)
#else
#define PPROF_BEGIN() do {} while (0)
#define PPROF_MARK(i) do {} while (0)
#define PPROF_END(AT_) do {} while (0)
#endif

template <int MODE, int NT, int CIN, bool FROMRGB, bool TORGB, int R, int NA>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(pipe_threads(NA), (NA + kPipeBWaves) / 4) sepconv_pipe_kernel(const SepArgs p) {
  static_assert(MODE == MODE_NORMAL || MODE == MODE_UP, "plain and FIR-up layers");
  static_assert(!FROMRGB || MODE == MODE_NORMAL, "FromRGB is fused into the first plain layer");
  static_assert(!TORGB || MODE == MODE_NORMAL, "ToRGB is fused into plain layers");
  static_assert(R == 2 || R == 3, "ring depth");
  static_assert(NA == 4 || NA == 8, "4 or 8 waves in group A");
  typedef PipeLds<MODE, NT, CIN, FROMRGB, R> L;
  constexpr int MT = 128, KC = 32, QC = 8, LG_QC = 3, GH = 8, GW = 16, lgGW = 4, IGW = GW + 2, NPIX = (GH + 2) * IGW, NITEMS = NPIX * QC;
  constexpr int NKC = L::NKC, PB = 64, NSLOT = 4, NPL = 2;
  constexpr int AT = NA * 64;                                       // threads of group A
  constexpr int DNI = 1536 / AT;                                    // input-tile DMAs (or built items) per group-A thread and chunk
  constexpr bool WRES = L::WRES;
  constexpr int DNB = NPL * NT * NSLOT / AT;                        // weight-plane DMAs per group-A thread and chunk
  static_assert(DNB >= 1, "weight planes must split over group A");
  constexpr int NTIW = NT / 64;                                     // 32-column blocks of a B wave (it owns 32 rows x NT/2 columns)
  constexpr int SEGH = MT * QC / AT;                                // rows of a depthwise strip: 4 (NA 4) or 2 (NA 8)
  static_assert(NKC >= 2 && NKC % 2 == 0, "an even number of K chunks (the A-operand buffer of a step is then a compile-time choice)");
  static_assert(NTIW + (TORGB ? 1 : 0) <= NKC, "the epilogue slices of a tile must fit the K steps of the next one");
  MIGAN_DYN_SMEM(smem);
  char* const lds = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = MIGAN_UNIFORM(tid >> 6);
  const bool groupA = tid < AT;

  // ---- tile schedule: the XCD-contiguous ranges of sepconv_kernel, walked by the persistent workgroups of each XCD ----------------
  const int ntiles = p.tiles_x * p.tiles_y * p.nchunks * p.B;
  const int xcd = (int)blockIdx.x & 7;
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int tcnt = tq + (xcd < tr ? 1 : 0);
  const int tbase = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int tstep = ((int)gridDim.x + 7 - xcd) >> 3;
  const int tl0 = (int)blockIdx.x >> 3;
  const int T = tl0 < tcnt ? (tcnt - tl0 + tstep - 1) / tstep : 0;   // my tiles
  if (T == 0) return;                                                 // (uniform: the whole workgroup leaves)
  const int G = T * NKC;                                              // my K steps
  // phase profile (-DMIGAN_PHASE_PROF builds): group A -> slots 0..3 [DMA issue + offsets / tile build, depthwise, vmcnt wait, barrier],
  // group B -> slots 4..7 [MFMAs, epilogue slice, barrier, rest]
  PPROF_BEGIN();
  // A workgroup's tiles are tstep apart in the logical order (column chunk fastest, then x, y, image): every cursor below walks them with
  // a mixed-radix add of tstep (computed once; scalar ALU only) instead of dividing a tile number by run-time extents once per tile.
  struct TileCur {
    int n, x, y, b;
  };
  const int st_n = tstep % p.nchunks, st_r1 = tstep / p.nchunks;
  const int st_x = st_r1 % p.tiles_x, st_r2 = st_r1 / p.tiles_x;
  const int st_y = st_r2 % p.tiles_y, st_b = st_r2 / p.tiles_y;
  TileCur tile0;
  {
    int t = tbase + tl0;
    tile0.n = t % p.nchunks; t /= p.nchunks;
    tile0.x = t % p.tiles_x; t /= p.tiles_x;
    tile0.y = t % p.tiles_y;
    tile0.b = t / p.tiles_y;
  }
  auto tile_next = [&](TileCur& c) {
    int carry = 0;
    c.n += st_n;
    if (c.n >= p.nchunks) { c.n -= p.nchunks; carry = 1; }
    c.x += st_x + carry; carry = 0;
    if (c.x >= p.tiles_x) { c.x -= p.tiles_x; carry = 1; }
    c.y += st_y + carry; carry = 0;
    if (c.y >= p.tiles_y) { c.y -= p.tiles_y; carry = 1; }
    c.b += st_b + carry;
  };
  auto tile_coords = [&](const TileCur& c, int& n0_, int& b0_, int& gy0_, int& gx0_) {
    n0_ = c.n * NT;
    b0_ = c.b;
    gy0_ = c.y * p.sy - p.off;
    gx0_ = c.x * p.sx - p.off;
  };

  // ---- FROMRGB: the input tile is act(fromrgb(network input)) (reference :194-195), BUILT instead of copied.  fromrgb is a 1x1
  // convolution (4 -> Cin, with bias): it runs on the matrix cores like the other 1x1s, as v_mfma_f32_32x32x16_bf16 on bf16x3-split operands
  // (x = x1 + x2 + x3, the six products of order <= 2^-16: fp32-grade, and bf16 has fp32's exponent range, so the unbounded network input
  // needs no scaling).  Only 5 of an MFMA's 16 K slots would be used by one piece (the four input planes and a flag that is 1 inside the
  // image and 0 on the conv's zero padding, against which the bias is multiplied -- a padding pixel comes out as exactly 0 = act(0), as the
  // reference's F.pad of the activated tensor), so the three pieces of a pixel sit side by side along K and three MFMAs against the
  // weight arrangements (w1 w1 w1), (w2 w2 0), (w3 0 0) produce the six products (first form: one piece per MFMA, six MFMAs and three
  // operand reads per unit: +5 %).  Group A turns the raw pixels of the next tile into A-operand rows; group B (waves 0..5: one 32-pixel
  // row block each) multiplies, activates and writes chunk s+2 of the tile while group A runs the depthwise stage of chunk s+1.  (Rounds
  // 1-3 and the first form of this kernel did the 4 -> Cin products as scalar FMA chains on the VALU.)
  struct BuildCursor {
    int is, ic, ik, slot;
  };
  auto build_begin = [&](BuildCursor& bc) { bc.is = bc.ic = bc.ik = bc.slot = 0; };
  // move the cursor one K step on; returns true when that was the last chunk of a tile
  auto build_advance = [&](BuildCursor& bc) -> bool {
    if (bc.is >= G) return false;
    ++bc.is;
    bc.slot = bc.slot + 1 == R ? 0 : bc.slot + 1;
    if (++bc.ic < NKC) return false;
    bc.ic = 0;
    ++bc.ik;
    return true;
  };

  // unit u of a step = row block u (32 halo-tile pixels) x the chunk's 32 channels: A pieces from the pixel buffer of the step's tile
  // (lanes 32..63 = k 8..15 read the zero slot), B pieces of the chunk (gain folded in), six bf16 MFMAs (smallest products first),
  // leaky-relu + clamp on the accumulators, one ds_write_b32 per value (lane = channel).  Rows 180..191 of a slot are never read.
  // (with 8 + 8 waves the six units of a step are split: group B's waves 0..2 take units 0..2, group A's waves 3..5 units 3..5 -- the
  // depthwise group has the slack for them; with 4 + 8 waves group A is the busy one and group B's waves 0..5 take them all)
  auto build_units = [&](const BuildCursor& bc, int wave, int nwaves) {
    if (bc.is >= G) return;
    const int bl31 = tid & 31, bhalf = (tid >> 5) & 1;
    const char* r_s = lds + L::OFF_RGB + (bc.ik & 1) * L::RGB_BUF + bhalf * 16;
    const char* f_s = lds + L::OFF_F + bc.ic * (3 * 32 * 32) + bl31 * 32 + bhalf * 16;
    float* in_s = reinterpret_cast<float*>(lds + L::OFF_IN + bc.slot * L::IN_SLOT);
    for (int u = wave; u >= 0 && u < 6; u += nwaves) {
      const f4 a = ld4(reinterpret_cast<const float*>(r_s + (u * 32 + bl31) * 32));
      f4 b[3];                                                   // (held in registers across the tiles they measured 15 % slower: indexed by chunk)
#pragma unroll
      for (int q = 0; q < 3; ++q) b[q] = ld4(reinterpret_cast<const float*>(f_s + q * (32 * 32)));
      // (x1 + x2 + x3) b1 + (x1 + x2) b2 + x1 b3: the six products of order <= 2^-16, smallest first
      f16v c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      c = MIGAN_MFMA_BF16_32X32X16(a, b[2], c);
      c = MIGAN_MFMA_BF16_32X32X16(a, b[1], c);
      c = MIGAN_MFMA_BF16_32X32X16(a, b[0], c);
      float* o = in_s + (u * 32 + 4 * bhalf) * KC + bl31;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f4 v = f4{c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};      // = gain * fromrgb(x)
        const f4 a4 = clamp4(__builtin_elementwise_max(v, v * 0.2f), -256.0f, 256.0f);
        o[(8 * q) * KC] = a4.x; o[(8 * q + 1) * KC] = a4.y; o[(8 * q + 2) * KC] = a4.z; o[(8 * q + 3) * KC] = a4.w;
      }
    }
  };

  if (groupA) {
    if (FROMRGB && NA == 4) MIGAN_SETPRIO(2);
    // =============================================== group A: DMA issue + depthwise stage ===========================================
    const int lt = tid;
    float* const w_s = reinterpret_cast<float*>(lds + L::OFF_W);
    // depthwise taps + bias: conv1.weight [CIN][9] -> per chunk tap-major [9][32], then bias [32].  All chunks once per workgroup where they
    // fit (TAPS_RES), otherwise chunk s+2 is fetched into registers during interval s and stored into the ring half the depthwise stage of
    // step s has finished with
    constexpr bool TAPS_RES = L::TAPS_RES;
    if constexpr (TAPS_RES) {
      for (int i = lt; i < CIN * 9 / 4; i += AT) {
        const f4 v = ld4(p.wdw + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = i * 4 + e, ch = f / 9, tap = f - ch * 9;
          w_s[(ch >> 5) * 320 + tap * 32 + (ch & 31)] = v[e];
        }
      }
      for (int i = lt; i < CIN / 4; i += AT) st4(w_s + ((i * 4) >> 5) * 320 + 288 + ((i * 4) & 31), ld4(p.bdw + i * 4));
    }
    f4 rtap = {0.f, 0.f, 0.f, 0.f};
    auto load_taps = [&](int chunk) {                    // (72 threads: 4 consecutive floats of the chunk's [32][9] taps; 8 more: its bias)
      if (lt < 72) rtap = ld4(p.wdw + chunk * 288 + lt * 4);
      else if (lt < 80) rtap = ld4(p.bdw + chunk * 32 + (lt - 72) * 4);
    };
    auto store_taps = [&](int buf) {
      float* wc = w_s + buf * 320;
      if (lt < 72) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = lt * 4 + e;
          wc[(f % 9) * 32 + f / 9] = rtap[e];
        }
      } else if (lt < 80) {
        st4(wc + 288 + (lt - 72) * 4, rtap);
      }
    };
    if constexpr (FROMRGB) {
      // fromrgb.weight [CIN][4] + bias [CIN] (reference :186), x gain -> the three B operand arrangements, per chunk [3][32 columns][16 bf16]
      char* const f_s = lds + L::OFF_F;
      for (int i = lt; i < CIN; i += AT) {
        const f4 w = ld4(p.frgb_w + i * 4) * 1.41421356237309515f;             // lrelu_agc's gain (positive) commutes with the leaky relu
        u2v w1, w2, w3, b1, b2, b3;
        split3_bf16(w, w1, w2, w3);
        split3_bf16(f4{p.frgb_b[i] * 1.41421356237309515f, 0.f, 0.f, 0.f}, b1, b2, b3);
        // one 5-slot group [w0 w1 w2 w3 bias] starting at k = 0, the same shifted to k = 5 (its bias slot meets the zero flag slot of x2 / x3:
        // left 0) and to k = 10
        auto g0 = [](u2v wq, u2v bq, unsigned (&e)[8]) { e[0] = wq.x; e[1] = wq.y; e[2] |= bq.x & 0xffffu; };
        auto g1 = [](u2v wq, unsigned (&e)[8]) { e[2] |= wq.x << 16; e[3] = (wq.x >> 16) | (wq.y << 16); e[4] |= wq.y >> 16; };
        auto g2 = [](u2v wq, unsigned (&e)[8]) { e[5] = wq.x; e[6] = wq.y; };
        unsigned e1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, e2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, e3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        g0(w1, b1, e1); g1(w1, e1); g2(w1, e1);
        g0(w2, b2, e2); g1(w2, e2);
        g0(w3, b3, e3);
        char* d = f_s + (i >> 5) * (3 * 32 * 32) + (i & 31) * 32;
        *reinterpret_cast<u4v*>(d) = u4v{e1[0], e1[1], e1[2], e1[3]};
        *reinterpret_cast<u4v*>(d + 16) = u4v{e1[4], e1[5], e1[6], e1[7]};
        *reinterpret_cast<u4v*>(d + 32 * 32) = u4v{e2[0], e2[1], e2[2], e2[3]};
        *reinterpret_cast<u4v*>(d + 32 * 32 + 16) = u4v{e2[4], e2[5], e2[6], e2[7]};
        *reinterpret_cast<u4v*>(d + 2 * 32 * 32) = u4v{e3[0], e3[1], e3[2], e3[3]};
        *reinterpret_cast<u4v*>(d + 2 * 32 * 32 + 16) = u4v{e3[4], e3[5], e3[6], e3[7]};
      }
      // rows 180..191 of both A-operand buffers (the sixth row block's padding): written once
      char* const r_s = lds + L::OFF_RGB;
      for (int i = lt; i < 2 * 12 * 2; i += AT)
        *reinterpret_cast<u4v*>(r_s + (i / 24) * L::RGB_BUF + 180 * 32 + (i % 24) * 16) = u4v{0u, 0u, 0u, 0u};
    }

    // ---- 1x1 weight planes (split_weights_kernel: chunk-major [plane][CIN/32][CO][32] fp16) -> LDS, XOR swizzle on the SOURCE side ----
    const MIGAN_BUF wbuf = MIGAN_MAKE_BUF(p.wsplit, (unsigned)(NPL * p.CO * CIN) * 2u);
    unsigned dboff[DNB];
#pragma unroll
    for (int j = 0; j < DNB; ++j) {
      const int i = lt + j * AT;                         // 16-byte unit of the LDS image [plane][NT rows][4 slots]
      const int plane = i / (NT * NSLOT), rem = i % (NT * NSLOT);
      const int n = rem / NSLOT, sp = rem % NSLOT;       // LDS row n, stored slot sp holds source slot sp ^ swizzle(n)
      dboff[j] = (unsigned)(plane * p.CO * CIN + n * KC + ((sp ^ ((n >> 2) & (NSLOT - 1))) * 8)) * 2u;
    }
    auto dma_b = [&](int n0_, int chunk, int buf) {
      float* bb = reinterpret_cast<float*>(lds + L::OFF_B + buf * L::B_CHUNK);
      const unsigned soff = (unsigned)(chunk * KC * p.CO + n0_ * KC) * 2u;
#pragma unroll
      for (int j = 0; j < DNB; ++j) MIGAN_LDS_DMA16(wbuf, dboff[j], soff, bb + (j * AT + wave_u * 64) * 4);
    };

    // ---- input tile of one K chunk -> ring slot.  The image is a buffer descriptor: a halo pixel outside it (the conv's zero padding,
    // reference :126) is a lane offset beyond its range and arrives as zeros; so do the padding units of the slot ----
    // Interior tiles (the whole 10 x 18 halo window inside the image: 9 of 10 tiles at 512 x 512) use offsets relative to the window's
    // first pixel, computed once per workgroup; the window's position rides in the scalar offset of the DMA.  Border tiles compute
    // absolute offsets with the out-of-range marker on padding pixels.
    unsigned dgoff[DNI], drel[DNI], tile_soff = 0;
#pragma unroll
    for (int j = 0; j < DNI; ++j) {
      const int i = lt + j * AT;
      drel[j] = 0xfffff000u;
      if (i < NITEMS) {
        const int c4 = i & (QC - 1), pix = i >> LG_QC;
        drel[j] = (unsigned)(((pix / IGW) * p.W + (pix % IGW)) * CIN + c4 * 4) * 4u;
      }
    }
    auto make_dgoff = [&](int gy0_, int gx0_) {
      if (gy0_ >= 1 && gy0_ + GH + 1 <= p.H && gx0_ >= 1 && gx0_ + GW + 1 <= p.W) {
#pragma unroll
        for (int j = 0; j < DNI; ++j) dgoff[j] = drel[j];
        tile_soff = (unsigned)(((gy0_ - 1) * p.W + (gx0_ - 1)) * CIN) * 4u;
        return;
      }
      tile_soff = 0;
#pragma unroll
      for (int j = 0; j < DNI; ++j) {
        const int i = lt + j * AT;
        unsigned g = 0xfffff000u;
        if (i < NITEMS) {
          const int c4 = i & (QC - 1), pix = i >> LG_QC;
          const int ix = pix % IGW, iy = pix / IGW;
          const int yy = gy0_ - 1 + iy, xx = gx0_ - 1 + ix;
          if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) g = (unsigned)((yy * p.W + xx) * CIN + c4 * 4) * 4u;
        }
        dgoff[j] = g;
      }
    };
    const unsigned img_bytes = (unsigned)(p.H * p.W * CIN) * 4u;
    auto dma_in = [&](int b0_, int chunk, int slot) {
      float* in_s = reinterpret_cast<float*>(lds + L::OFF_IN + slot * L::IN_SLOT);
      const MIGAN_BUF xbuf = MIGAN_MAKE_BUF(reinterpret_cast<const char*>(p.x) + (size_t)b0_ * img_bytes, img_bytes);
#pragma unroll
      for (int j = 0; j < DNI; ++j) MIGAN_LDS_DMA16(xbuf, dgoff[j], tile_soff + (unsigned)(chunk * KC) * 4u, in_s + (j * AT + wave_u * 64) * 4);
    };

    // ---- FROMRGB: the raw 4-channel network input of the halo tile, one tile ahead, through registers into LDS (this group) ----------
    f4 rraw = {0.f, 0.f, 0.f, 0.f};
    u4v rbytes = {0u, 0u, 0u, 0u};                         // uint8 input: the pixel's bytes as loaded (packed into x by store_raw, a tile later)
    bool rvalid = false;
    auto load_raw = [&](int b0_, int gy0_, int gx0_) {
      f4 v = {0.f, 0.f, 0.f, 0.f};
      rvalid = false;
      if (lt < NPIX) {
        const int ix = lt % IGW, iy = lt / IGW;
        const int yy = gy0_ - 1 + iy, xx = gx0_ - 1 + ix;
        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
          rvalid = true;
          if (p.u8_img) {
            rbytes = fetch_pixel_bytes(p.u8_img, p.u8_mask, ((size_t)b0_ * p.H + yy) * p.W + xx);
          } else {
            const float* src = reinterpret_cast<const float*>(p.x) + ((size_t)b0_ * 4 * p.H + yy) * p.W + xx;
            const size_t plane = (size_t)p.H * p.W;
            v = f4{src[0], src[plane], src[2 * plane], src[3 * plane]};
          }
        }
      }
      rraw = v;
    };
    // raw pixel -> its A-operand row: the three bf16 pieces [x0 x1 x2 x3 flag] (flag = 1.0 inside the image in the first piece: exact in
    // bf16; 0 in the others) side by side along K
    auto store_raw = [&](int buf) {
      if (lt < NPIX) {
        if (p.u8_img) rraw = rvalid ? pack_pixel_bytes(rbytes) : f4{0.f, 0.f, 0.f, 0.f};
        u2v h1, h2, h3;
        split3_bf16(rraw, h1, h2, h3);
        char* d = lds + L::OFF_RGB + buf * L::RGB_BUF + lt * 32;
        *reinterpret_cast<u4v*>(d) = u4v{h1.x, h1.y, (rvalid ? 0x3f80u : 0u) | (h2.x << 16), (h2.x >> 16) | (h2.y << 16)};
        *reinterpret_cast<u4v*>(d + 16) = u4v{h2.y >> 16, h3.x, h3.y, 0u};
      }
    };

    // ---- depthwise 3x3 + bias + act (x 2^7) + fp16 hi/lo split of one chunk: one SEGH-row strip x 4 channels per thread --------------
    auto depthwise = [&](int slot, int chunk, int abuf) {
      const float* in_s = reinterpret_cast<const float*>(lds + L::OFF_IN + slot * L::IN_SLOT);
      const float* wc = w_s + (TAPS_RES ? chunk : (chunk & 1)) * 320;
      char* a_b = lds + L::OFF_A + abuf * L::A_BUF;
      const int c4 = lt & (QC - 1);
      const int gx = (lt >> LG_QC) & (GW - 1);
      const int r0 = (lt >> (LG_QC + lgGW)) * SEGH;
      f4 w[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) w[tap] = ld4(wc + tap * KC + c4 * 4);
      const f4 bias = ld4(wc + KC * 9 + c4 * 4);
      const float* ip = in_s + (r0 * IGW + gx) * KC + c4 * 4;
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
        u2v h1, h2;
        split2_f16(act4_scaled<7>(sacc), h1, h2);
        *reinterpret_cast<u2v*>(d) = h1;
        *reinterpret_cast<u2v*>(d + MT * PB) = h2;
      }
    };

    // ---- cursors: the step whose input is issued / built next (is), the step whose weights are issued next (bs: streamed planes only) --
    int is = 0, ic = 0, ik = 0, islot = 0, ib0 = 0, in0 = 0, igy0 = 0, igx0 = 0;
    TileCur itc = tile0;
    tile_coords(itc, in0, ib0, igy0, igx0);
    if constexpr (!FROMRGB) make_dgoff(igy0, igx0);
    auto advance_issue = [&]() {
      ++is;
      islot = islot + 1 == R ? 0 : islot + 1;
      if (++ic == NKC) {
        ic = 0;
        if (++ik < T) {
          tile_next(itc);
          tile_coords(itc, in0, ib0, igy0, igx0);
          if constexpr (!FROMRGB) make_dgoff(igy0, igx0);
        }
      }
    };
    auto issue_in = [&]() {
      if (is < G) {
        dma_in(ib0, ic, islot);
        advance_issue();
      }
    };
    int bs = 0, bc = 0, bk = 0, bn0 = in0;
    TileCur btc = tile0;
    auto issue_b = [&]() {                               // streamed weight planes of step bs -> slot bs & 1
      if (bs < G) {
        dma_b(bn0, bc, bs & 1);
        ++bs;
        if (++bc == NKC) {
          bc = 0;
          if (++bk < T) { tile_next(btc); bn0 = btc.n * NT; }
        }
      }
    };
    int dslot = 0;                                       // ring slot of the step the depthwise stage works on next

    if constexpr (FROMRGB) {
      // ---- two-slot ring filled by this group itself: interval g runs the depthwise stage of step g+1 (slot (g+1) & 1) and builds the
      // input tile of step g+2 into slot g & 1, which the depthwise stage of step g finished reading before the last barrier ----
      static_assert(!FROMRGB || (NKC == 2 && R == 2 && WRES), "the fused-FromRGB form is built for Cin = 64 (two chunks, resident planes) and a two-slot ring");
#pragma unroll
      for (int c = 0; c < NKC; ++c) dma_b(in0, c, c);          // the resident weight planes: the only DMAs of this form
      load_raw(ib0, igy0, igx0);
      store_raw(0);
      MIGAN_WAIT_VMCNT(0);
      int rk = 1, rn0 = 0, rb0 = 0, rgy0 = 0, rgx0 = 0;         // tile whose raw pixels are in registers
      TileCur rtc = tile0;
      if (rk < T) { tile_next(rtc); tile_coords(rtc, rn0, rb0, rgy0, rgx0); load_raw(rb0, rgy0, rgx0); }
      BuildCursor bcur;
      build_begin(bcur);
      MIGAN_BARRIER_LDS();                                     // P1: taps, fromrgb operands, weight planes, raw(0) visible (B waits here too)
      // produce(step s): group B builds the input tile of that step; here: after the last chunk of a tile, hand the next tile's pixels over
      auto produce = [&]() {
        if constexpr (NA == 8) build_units(bcur, (tid >> 6) >= 3 && (tid >> 6) < 6 ? (tid >> 6) : -1, 8);
        if (build_advance(bcur) && bcur.ik < T) {
          store_raw(bcur.ik & 1);                               // raw(tile ik): that buffer was last read two tiles ago
          ++rk;
          if (rk < T) { tile_next(rtc); tile_coords(rtc, rn0, rb0, rgy0, rgx0); load_raw(rb0, rgy0, rgx0); }
        }
      };
      produce();                                               // step 0 -> slot 0
      produce();                                               // step 1 -> slot 1 (+ raw(1) -> LDS)
      MIGAN_BARRIER_LDS();                                     // P2
      depthwise(0, 0, 0);
      MIGAN_BARRIER_LDS();                                     // barrier 0
      dslot = 1;
      int dc = 1;
      for (int g = 0; g < G; ++g) {
        if (g + 1 < G) depthwise(dslot, dc, (g + 1) & 1);
        PPROF_MARK(1);
        produce();                                             // step g+2 -> slot g & 1
        PPROF_MARK(0);
        dslot ^= 1;
        dc = dc + 1 == NKC ? 0 : dc + 1;
        MIGAN_BARRIER_LDS();
        PPROF_MARK(3);
      }
    } else {
      // ---- prologue: resident weight planes, the first R input chunks in flight -----------------------------------------------------
      if constexpr (!TAPS_RES) {
        load_taps(0); store_taps(0);
        load_taps(1); store_taps(1);
      }
      if constexpr (WRES) {
#pragma unroll
        for (int c = 0; c < NKC; ++c) dma_b(in0, c, c);
      } else {
        issue_b();                                             // step 0
      }
#pragma unroll
      for (int s = 0; s < R; ++s) issue_in();
      // weights + input of step 0 landed; the inputs of steps 1..R-1 stay in flight across the barrier
      if (G >= R) MIGAN_WAIT_VMCNT((R - 1) * DNI); else MIGAN_WAIT_VMCNT(DNI);      // (G is even: G < R means G == 2, R == 3)
      MIGAN_BARRIER_LDS();                                     // P1
      MIGAN_BARRIER_LDS();                                     // P2 (the FROMRGB form needs two: same count in group B)
      depthwise(0, 0, 0);
      if constexpr (!WRES) issue_b();                          // step 1 -> slot 1
      // input of step 1 landed (for the depthwise stage of interval 0); later inputs and the step-1 weights may stay in flight
      if constexpr (R == 3) { if (G >= 3) MIGAN_WAIT_VMCNT(DNI + (WRES ? 0 : DNB)); else MIGAN_WAIT_VMCNT(WRES ? 0 : DNB); }
      else MIGAN_WAIT_VMCNT(WRES ? 0 : DNB);
      MIGAN_BARRIER_LDS();                                     // barrier 0
      dslot = 1 % R;
      int dc = 1;
      // ToRGB: the 6 x 10 window of the previous image under the tile whose first K step this is -> LDS (one 4-byte DMA instruction per
      // colour, issued by wave 0; positions outside the image are lane offsets beyond the buffer = the upsampling FIR's zero padding).
      // Group B reads it when it finishes that tile's ToRGB tail, NKC + NTIW steps later; the buffer is rewritten two tiles on.
      TileCur wtc = tile0;
      int wk = 0, wstep = 0;
      const MIGAN_BUF pbuf = MIGAN_MAKE_BUF(p.img_prev, p.img_prev ? (unsigned)((size_t)p.B * 3 * (p.HO >> 1) * (p.WO >> 1)) * 4u : 0u);
      auto issue_prev_window = [&]() {
        if constexpr (TORGB) {
          if (wstep == 0) {
            if (p.img_prev && wave_u == 0) {                                    // (one wave: spread over three it measured 7 % slower)
              const int hp = p.HO >> 1, wp = p.WO >> 1;
              const int row = lane / 10, col = lane - row * 10;                 // (lanes 60..63: padding)
              const int yy = ((wtc.y * p.sy - p.off) >> 1) - 1 + row, xx = ((wtc.x * p.sx - p.off) >> 1) - 1 + col;
              const bool ok = lane < 60 && yy >= 0 && yy < hp && xx >= 0 && xx < wp;
              const unsigned voff = ok ? (unsigned)(yy * wp + xx) * 4u : 0xfffff000u;
              float* pw = reinterpret_cast<float*>(lds + L::OFF_PW + (wk & 1) * L::PW_SZ);
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) MIGAN_LDS_DMA4(pbuf, voff, (unsigned)((wtc.b * 3 + ch) * hp * wp) * 4u, pw + ch * 64);
            }
            ++wk;
            if (wk < T) tile_next(wtc);
          }
          wstep = wstep + 1 == NKC ? 0 : wstep + 1;
        }
      };
      for (int g = 0; g < G; ++g) {
        // interval g: B runs the MFMAs of step g.  Slot g % R (read by the depthwise stage of step g) and weight slot (g+1) & 1 (read by the
        // MFMAs of step g-1) are free: refill them first, then the depthwise stage of step g+1
        if constexpr (!WRES) { if (g >= 1) issue_b(); }        // step g+1 (steps 0 and 1 were issued by the prologue)
        issue_prev_window();                                   // (older than the input chunk issued next: covered by the counted wait below)
        issue_in();                                            // step g+R
        if constexpr (!TAPS_RES) { if (g + 2 < G) load_taps((dc + 1) % NKC); }      // taps of step g+2 (dc is the chunk of step g+1)
        PPROF_MARK(0);
        if (g + 1 < G) depthwise(dslot, dc, (g + 1) & 1);
        PPROF_MARK(1);
        if constexpr (!TAPS_RES) { if (g + 2 < G) store_taps(g & 1); }              // ring half of step g: its depthwise stage ran an interval ago
        dslot = dslot + 1 == R ? 0 : dslot + 1;
        dc = dc + 1 == NKC ? 0 : dc + 1;
        // before the barrier that starts interval g+1: input of step g+2 and weights of step g+1 landed.  Everything issued before the
        // newest input chunk is then complete, and that chunk (step g+R, R = 3) stays in flight
        if (R == 3 && g + 3 < G) MIGAN_WAIT_VMCNT(DNI); else MIGAN_WAIT_VMCNT(0);
        PPROF_MARK(2);
        MIGAN_BARRIER_LDS();
        PPROF_MARK(3);
      }
    }
    if constexpr (TORGB) MIGAN_BARRIER_LDS();                   // the last tile: group B hands its ToRGB partial sums over
    if constexpr (MODE == MODE_UP) {
#pragma unroll
      for (int i = 0; i < 2 * (NT / 64) - 1; ++i) MIGAN_BARRIER_LDS();      // ... publishes its result tile, one 64-column half at a time
    }
    PPROF_END(AT);
    return;
  }

  // ================================================= group B: MFMAs + the previous tile's epilogue ====================================
  const int tb = tid - AT, wb = wave_u - NA;                    // wave wb = 4 cbk + rb: GEMM rows 32 rb .. 32 rb + 31, columns cbk NT/2 .. (cbk + 1) NT/2 - 1
  // (waves go to SIMDs round-robin: rb = wb & 3 puts one wave of each column half on every SIMD, so the ToRGB tail -- run by half 0 -- and
  // the hand-off writes of half 1 are spread over all four)
  const int rb = wb & 3, cbk = wb >> 2;
  const int l31 = lane & 31, half = lane >> 5;
  (void)tb;
  f16v acc[NTIW], accp[NTIW];
#pragma unroll
  for (int j = 0; j < NTIW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[j][r] = 0.0f; accp[j][r] = 0.0f; }

  // first: the first product of a tile reads the constant 0 as C (no accumulator clearing at the hand-over)
  auto mfma_chunk = [&](int abuf, int bbuf, bool first) {
    const char* ab = lds + L::OFF_A + abuf * L::A_BUF;
    const char* bb = lds + L::OFF_B + bbuf * L::B_CHUNK;
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      f4 av[NPL], bv[NTIW][NPL];
      {
        const int row = rb * 32 + l31;
        const char* q = ab + row * PB + (((2 * ks + half) ^ ((row >> 2) & (NSLOT - 1))) << 4);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) av[pl] = ld4(reinterpret_cast<const float*>(q + pl * MT * PB));
      }
#pragma unroll
      for (int j = 0; j < NTIW; ++j) {
        const int row = cbk * (NT / 2) + j * 32 + l31;
        const char* q = bb + row * PB + (((2 * ks + half) ^ ((row >> 2) & (NSLOT - 1))) << 4);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) bv[j][pl] = ld4(reinterpret_cast<const float*>(q + pl * NT * PB));
      }
      // smallest products first
#pragma unroll
      for (int j = 0; j < NTIW; ++j) {
        if (first && ks == 0) acc[j] = f16v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[j] = MIGAN_MFMA_F16_32X32X16(av[1], bv[j][0], acc[j]);
      }
#pragma unroll
      for (int j = 0; j < NTIW; ++j) acc[j] = MIGAN_MFMA_F16_32X32X16(av[0], bv[j][1], acc[j]);
#pragma unroll
      for (int j = 0; j < NTIW; ++j) acc[j] = MIGAN_MFMA_F16_32X32X16(av[0], bv[j][0], acc[j]);
    }
  };

  const float acc_scale = reinterpret_cast<const float*>(p.wsplit)[-4];     // 1 / (activation scale x weight scale), a power of two
  const bool has_noise = p.noise != nullptr;
  const float ns = has_noise ? p.noise_strength[0] : 0.0f;
  const size_t img_out_bytes = (size_t)p.HO * p.WO * p.CO * 4;

  // coordinates of the tile whose accumulators are in `accp` (pn0 etc.) and of the tile being accumulated (cn0 etc.)
  int ck = 0, cn0 = 0, cb0 = 0, cgy0 = 0, cgx0 = 0, pn0 = 0, pb0 = 0, pgy0 = 0, pgx0 = 0;
  int ppar = 0;                                                   // parity of the tile being finished = its previous-image window buffer
  TileCur ctc = tile0;
  tile_coords(ctc, cn0, cb0, cgy0, cgx0);
  auto hand_over = [&]() {
#pragma unroll
    for (int j = (MODE == MODE_NORMAL ? 1 : 0); j < NTIW; ++j) accp[j] = acc[j];      // (plain layers stage block 0 straight from `acc`)
    pn0 = cn0; pb0 = cb0; pgy0 = cgy0; pgx0 = cgx0; ppar = ck & 1;
    if (++ck < T) { tile_next(ctc); tile_coords(ctc, cn0, cb0, cgy0, cgx0); }
  };

  BuildCursor bcur;
  if constexpr (FROMRGB) build_begin(bcur);
  auto build_step = [&](BuildCursor& bc) {
    build_units(bc, NA == 8 && wb >= 3 ? -1 : wb, 8);
    build_advance(bc);
  };
  MIGAN_BARRIER_LDS();                                          // P1
  if constexpr (FROMRGB) {                                      // the input tiles of steps 0 and 1
    build_step(bcur);
    build_step(bcur);
  }
  MIGAN_BARRIER_LDS();                                          // P2
  MIGAN_BARRIER_LDS();                                          // barrier 0: A planes of step 0 (and the weight planes) are in LDS

  if constexpr (MODE == MODE_NORMAL) {
    // ---- plain layers: the epilogue runs on the wave's own 32 x NT/2 accumulators, one 32 x 32 block at a time through a wave-private
    // LDS patch (C layout: lane = column, 16 rows per lane -> rows of 32 channels = one 128-byte line per 8 lanes) ----
    // The patch is [32 pixel rows][32 channels] fp32 with its 16-byte slots XOR-swizzled by (row >> 1) & 7: the column writes
    // (ds_write_b32, one row per instruction) and the row reads (ds_read_b128) are both conflict-free without padding.
    float* const t_s = reinterpret_cast<float*>(lds + L::OFF_T) + wb * (32 * 32);
    const int q4 = lane & 7, prow = lane >> 3;                 // this lane's channel quad of a block, its pixel row inside a group of 8
    // write side: element r of the fragment is row (r & 3) + 8 (r >> 2) + 4 half, column l31 -> slot (l31 >> 2) ^ ((row >> 1) & 7);
    // (row >> 1) & 7 = 2 half + s_r with s_r = ((r >> 1) & 1) + 4 ((r >> 2) & 1) in {0, 1, 4, 5}: four lane addresses + constant row offsets
    int twr[4];
    {
      const int u = (l31 >> 2) ^ (2 * half);
#pragma unroll
      for (int v = 0; v < 4; ++v) twr[v] = (4 * half) * 32 + ((u ^ ((v & 1) + 4 * (v >> 1))) << 2) + (l31 & 3);
    }
    // read side: row 8 q + prow, slot q4 ^ ((prow >> 1) + 4 (q & 1))
    const int trd0 = prow * 32 + ((q4 ^ (prow >> 1)) << 2), trd1 = prow * 32 + ((q4 ^ (prow >> 1) ^ 4) << 2);
    float nz[4] = {0.f, 0.f, 0.f, 0.f};                         // noise_const of this lane's 4 pixels (rows 8q + prow) of the tile being finished
    float nzn[4] = {0.f, 0.f, 0.f, 0.f};                        // ... of the tile being accumulated (requested one K step before the hand-over)
    float rs[4][3];                                             // ToRGB partial sums of those pixels over this wave's columns
    unsigned pix0 = 0;                                          // output pixel index of row prow of the wave's first image row
    f4 tw[NTIW][3];
    float tbias[3] = {0.f, 0.f, 0.f};                           // (read once: a load inside the tile loop would wait for every store before it)
    if constexpr (TORGB) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) tbias[ch] = p.trgb_b[ch];
#pragma unroll
      for (int j = 0; j < NTIW; ++j)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) tw[j][ch] = ld4(p.trgb_w + ch * p.CO + cbk * (NT / 2) + j * 32 + q4 * 4);
    }
    // ToRGB tail (reference :308-313), run by the waves of column half 0: lane q4 < 4 of each 8-lane pixel group finishes pixel q = q4 of
    // the group: own sums + the other half's (through LDS) + bias + the 2x-upsampled previous image (reference Upsample2d :79-104), whose
    // 2x2 taps per colour it reads from the window group A has put into LDS (zeros outside the image: no border cases here)
    float* const part_s = reinterpret_cast<float*>(lds + L::OFF_P) + rb * (32 * 4);
    // pixel of (q, lane): GEMM row m = 32 rb + 8 q + prow -> tile row 2 rb + (q >> 1), column 8 (q & 1) + prow.
    // Everything the next epilogue loads (noise values, previous-image taps) is requested before the last store slice of the previous
    // tile is issued, and only used a K step later: the wait in front of the first use then leaves those stores in flight.
    // (uniform base pointer + 32-bit lane byte offset: the saddr + voffset form, no 64-bit address arithmetic)
    u4v u8c = {0u, 0u, 0u, 0u}, u8n = {0u, 0u, 0u, 0u};          // uint8 output: image / mask bytes of the pixel this lane composes (tile being finished, next)
    auto request_next = [&]() {
      if (has_noise) {
        const unsigned px = (unsigned)((cgy0 + 2 * rb) * p.WO + cgx0 + prow) * 4u;
#pragma unroll
        for (int q = 0; q < 4; ++q) nzn[q] = *at_bytes(p.noise + ((q >> 1) * p.WO + (q & 1) * 8), px);
      }
      if constexpr (TORGB) {
        if (p.u8_out && cbk == 0 && q4 < 4)
          u8n = fetch_pixel_bytes(p.u8_img, p.u8_mask, (size_t)cb0 * p.HO * p.WO + (size_t)(cgy0 + 2 * rb + (q4 >> 1)) * p.WO + cgx0 + (q4 & 1) * 8 + prow);
      }
    };
    auto begin_tile_epilogue = [&]() {
      pix0 = (unsigned)((pgy0 + 2 * rb) * p.WO + pgx0 + prow);
#pragma unroll
      for (int q = 0; q < 4; ++q) nz[q] = nzn[q];
      if constexpr (TORGB) {
        u8c = u8n;
#pragma unroll
        for (int q = 0; q < 4; ++q) rs[q][0] = rs[q][1] = rs[q][2] = 0.0f;
      }
    };
    // A block of accumulators goes through the patch in three moves that sit in DIFFERENT places of the K step, so that no LDS round
    // trip is waited for: stage (16 ds_write_b32, at the hand-over or right after the previous block's rows were consumed), fetch
    // (4 ds_read_b128, issued before the step's MFMAs) and finish (noise, activation, store, ToRGB share -- after the MFMAs).
    auto stage_block = [&](const f16v& a) {
      MIGAN_WAVE_SYNC();                                        // (rows of the previous block were read by other lanes of this wave)
#pragma unroll
      for (int r = 0; r < 16; ++r) t_s[twr[((r >> 1) & 1) + 2 * ((r >> 2) & 1)] + ((r & 3) + 8 * (r >> 2)) * 32] = a[r];
      MIGAN_WAVE_SYNC();
    };
    f4 tv[4];
    auto fetch_block = [&]() {
#pragma unroll
      for (int q = 0; q < 4; ++q) tv[q] = ld4(t_s + ((q & 1) ? trd1 : trd0) + q * 8 * 32);
    };
    auto finish_block = [&](int j) {
      char* yb = reinterpret_cast<char*>(p.y) + (size_t)pb0 * img_out_bytes;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f4 v = tv[q] * acc_scale + MIGAN_FMUL_RN(nz[q], ns);        // product rounded first, reference :166
        v = act4(v);
        const unsigned pix = pix0 + (unsigned)((q >> 1) * p.WO + (q & 1) * 8);
        Io<0>::st(yb, (pix * (unsigned)p.CO + (unsigned)(pn0 + cbk * (NT / 2) + j * 32 + q4 * 4)) * 4u, v);
        if constexpr (TORGB) {
          float r0, r1, r2;
          torgb_partial(v, tw[j][0], tw[j][1], tw[j][2], r0, r1, r2);
          rs[q][0] += r0; rs[q][1] += r1; rs[q][2] += r2;
        }
      }
    };
    // after a wave's last block: its per-pixel partial sums (8 lanes per pixel); column half 1 hands them to the wave of half 0 that
    // owns the same rows
    float mine[3] = {0.f, 0.f, 0.f};
    auto rgb_partials = [&]() {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float s0 = MIGAN_SUM8(rs[q][0]), s1 = MIGAN_SUM8(rs[q][1]), s2 = MIGAN_SUM8(rs[q][2]);
        if (q4 == q) { mine[0] = s0; mine[1] = s1; mine[2] = s2; }
      }
      if (cbk == 1 && q4 < 4) st4(part_s + (8 * q4 + prow) * 4, f4{mine[0], mine[1], mine[2], 0.0f});
    };
    auto rgb_finish = [&]() {                                   // (one barrier after rgb_partials)
      if (cbk == 0 && q4 < 4) {
        const f4 other = ld4(part_s + (8 * q4 + prow) * 4);
        const size_t plane = (size_t)p.HO * p.WO;
        const int oy = pgy0 + 2 * rb + (q4 >> 1), ox = pgx0 + (q4 & 1) * 8 + prow;
        // (scalar adds kept apart: the packed form of these sums is the op_sel hazard of DESIGN 5.7, refused by the ISA lint)
        float o3[3] = {mine[0] + other.x, mine[1] + other.y, mine[2] + other.z};
        MIGAN_OPAQUE_F(o3[0]); MIGAN_OPAQUE_F(o3[1]); MIGAN_OPAQUE_F(o3[2]);
        o3[0] += tbias[0]; MIGAN_OPAQUE_F(o3[0]);
        o3[1] += tbias[1]; MIGAN_OPAQUE_F(o3[1]);
        o3[2] += tbias[2]; MIGAN_OPAQUE_F(o3[2]);
        if (p.img_prev) {
          const int y0 = (oy & 1) ? (oy >> 1) : (oy >> 1) - 1, x0 = (ox & 1) ? (ox >> 1) : (ox >> 1) - 1;     // first of the two taps per axis
          const float wy0 = (oy & 1) ? 0.75f : 0.25f, wx0 = (ox & 1) ? 0.75f : 0.25f;
          const float* pw = reinterpret_cast<const float*>(lds + L::OFF_PW + ppar * L::PW_SZ) + (y0 - (pgy0 >> 1) + 1) * 10 + (x0 - (pgx0 >> 1) + 1);
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            const float r0 = wx0 * pw[ch * 64] + (1.0f - wx0) * pw[ch * 64 + 1];
            const float r1 = wx0 * pw[ch * 64 + 10] + (1.0f - wx0) * pw[ch * 64 + 11];
            o3[ch] = (wy0 * r0 + (1.0f - wy0) * r1) + o3[ch];
          }
        }
        if (p.u8_out) {
          compose_pixel_bytes(u8c, p.u8_out, (size_t)pb0 * plane + (size_t)oy * p.WO + ox, o3[0], o3[1], o3[2]);      // (bytes requested a K step before the hand-over)
        } else {
          const unsigned po = (unsigned)(oy * p.WO + ox) * 4u;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) *at_bytes(p.img_out + ((size_t)pb0 * 3 + ch) * plane, po) = o3[ch];
        }
      }
    };
    // slice c of a tile's epilogue: block c (c < NTIW); the partial ToRGB sums after the last block; the ToRGB tail one step later.
    // The first tile is peeled off (nothing to finish under it): inside the steady-state loop every load -> use pair then sits on ONE
    // control path, so the compiler's waitcnt pass can count the stores issued in between instead of falling back to vmcnt(0) (a
    // "previous tile exists" branch around the uses made it drain every store of the wave once per tile: 2.9k cycles, measured).
#pragma unroll
    for (int c = 0; c < NKC; ++c) {
      mfma_chunk(c & 1, WRES ? c : (c & 1), c == 0);
      if (c == NKC - 1) {
        request_next();
        stage_block(acc[0]);
        hand_over();
      }
      if constexpr (FROMRGB) build_step(bcur);                  // this group's share of the input tile two steps ahead
      MIGAN_BARRIER_LDS();
    }
    for (int t = 1; t < T; ++t) {
#pragma unroll
      for (int c = 0; c < NKC; ++c) {
        PPROF_MARK(7);
        if (c < NTIW) fetch_block();                            // rows of block c: on their way while the MFMAs run
        mfma_chunk(c & 1, WRES ? c : (c & 1), c == 0);
        PPROF_MARK(4);
        if (c == NKC - 1) request_next();
        PPROF_MARK(12);
        if (c == 0) begin_tile_epilogue();                      // (first use of what request_next asked for a K step ago)
        PPROF_MARK(5);
        if (c < NTIW) {
          finish_block(c);
          if (c + 1 < NTIW) stage_block(accp[c + 1 < NTIW ? c + 1 : 0]);
        }
        PPROF_MARK(9);
        if constexpr (TORGB) {
          if (c == NTIW - 1) rgb_partials();
          PPROF_MARK(10);
          if (c == NTIW) rgb_finish();
          PPROF_MARK(11);
        }
        if (c == NKC - 1) {
          stage_block(acc[0]);
          hand_over();
        }
        PPROF_MARK(7);
        if constexpr (FROMRGB) build_step(bcur);
        PPROF_MARK(13);
        MIGAN_BARRIER_LDS();
        PPROF_MARK(6);
      }
    }
    // the last tile: nothing left to hide it under
    begin_tile_epilogue();
#pragma unroll
    for (int j = 0; j < NTIW; ++j) {
      fetch_block();
      finish_block(j);
      if (j + 1 < NTIW) stage_block(accp[j + 1 < NTIW ? j + 1 : 0]);
    }
    if constexpr (TORGB) {
      rgb_partials();
      MIGAN_BARRIER_LDS();                                      // (group A joins this one)
      rgb_finish();
    }
  } else {
    // ---- FIR-up layers (reference Upsample2d :79-103 after the 1x1): the 2x polyphase FIR needs the 3x3 neighbourhood of the GEMM
    // result, so the accumulators of a finished tile go to a dedicated LDS result tile during the first K step of the next tile
    // (published by that step's barrier) and the 6 x 14 interior pixels x NT/4 channel quads are worked off in the steps after it ----
    // A 128-column layer passes its two 64-column halves through the result tile one after the other (phase h: the waves of column half h
    // write, then all eight waves run that half's FIR items), half the K steps of the next tile each.
    constexpr int GS = 64 + 4, QN = 16, LG_QN = 4, NH = NT / 64;
    static_assert(NT == 64 || NT == 128, "FIR-up tiles: 64 or 128 output channels");
    constexpr int SP = NKC / NH;                                // K steps per phase: one to write the half, SP - 1 for its items
    static_assert(NKC % NH == 0 && SP >= 2, "too few K steps to hide the FIR epilogue under");
    constexpr int BT = kPipeBWaves * 64;
    constexpr int STEP = BT >> LG_QN;                           // GEMM rows between the items of a thread
    constexpr int ITEMS = MT * QN / BT;
    float* const g_s = reinterpret_cast<float*>(lds + L::OFF_T);
    auto acc_to_lds = [&](int h) {
      if (NH == 2 && cbk != h) return;                          // (wave-uniform)
#pragma unroll
      for (int j = 0; j < NTIW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          // halo pixels outside the low-resolution image contribute zeros to the FIR (reference pads with zeros :101)
          const int ly = pgy0 + (row >> lgGW), lx = pgx0 + (row & (GW - 1));
          float v = accp[j][r];
          if (ly < 0 || ly >= p.H || lx < 0 || lx >= p.W) v = 0.0f;
          g_s[row * GS + (NH == 2 ? j * 32 : cbk * 32) + l31] = v;
        }
    };
    const int c4 = tb & (QN - 1), m0 = tb >> LG_QN;
    // Item k of this thread: GEMM row m0 + k STEP (one interior low-resolution pixel x 4 channels -> its 2x2 output pixels).  An item is
    // worked off in two halves so that the global loads of item k+1 (noise, skip) are requested BEFORE the stores of item k are issued:
    // the wait in front of their use then leaves those stores in flight (vmcnt retires in issue order).
    struct ItemIo {
      f4 sk[2][2];
      float nzv[2][2];
    };
    auto item_geo = [&](int k, int h, int& m, unsigned& lpix, unsigned& loff) -> bool {
      m = m0 + k * STEP;
      const int gy = m >> lgGW, gx = m & (GW - 1);
      if (gy < 1 || gy > GH - 2 || gx < 1 || gx > GW - 2) return false;
      const int ly = pgy0 + gy, lx = pgx0 + gx;
      if (ly >= p.H || lx >= p.W) return false;                  // ragged right / bottom edge of the tile grid
      lpix = (unsigned)((2 * ly) * p.WO + 2 * lx);
      loff = (lpix * (unsigned)p.CO + (unsigned)(pn0 + h * 64 + c4 * 4)) * 4u;
      return true;
    };
    auto item_load = [&](int k, int h, ItemIo& io) {
      int m;
      unsigned lpix, loff;
      if (!item_geo(k, h, m, lpix, loff)) return;
      const char* sb = p.skip ? reinterpret_cast<const char*>(p.skip) + (size_t)pb0 * img_out_bytes : nullptr;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const unsigned dp = (unsigned)(a * p.WO + bb);
          io.nzv[a][bb] = has_noise ? p.noise[lpix + dp] : 0.0f;
          io.sk[a][bb] = sb ? Io<0>::ld_once(sb, loff + dp * (unsigned)p.CO * 4u) : f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto item_finish = [&](int k, int h, const ItemIo& io) {
      int m;
      unsigned lpix, loff;
      if (!item_geo(k, h, m, lpix, loff)) return;
      char* yb = reinterpret_cast<char*>(p.y) + (size_t)pb0 * img_out_bytes;
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
          f4 v = out[a][bb] * acc_scale + MIGAN_FMUL_RN(io.nzv[a][bb], ns);    // product rounded first, reference :166
          v = act4(v);
          v += io.sk[a][bb];
          Io<0>::st(yb, loff + (unsigned)(a * p.WO + bb) * (unsigned)p.CO * 4u, v);
        }
    };
    // step c of a tile: phase h = c / SP; its first step writes the half, step 1 + i of the phase runs the items k with slice_of(k) == i
    constexpr int SL = SP - 1;
    auto slice_of = [](int k) { return k * SL / ITEMS; };
    ItemIo io[2];
    // (first tile peeled off, for the reason given in the plain form)
#pragma unroll
    for (int c = 0; c < NKC; ++c) {
      mfma_chunk(c & 1, WRES ? c : (c & 1), c == 0);
      if (c == NKC - 1) hand_over();
      MIGAN_BARRIER_LDS();
    }
    for (int t = 1; t < T; ++t) {
#pragma unroll
      for (int c = 0; c < NKC; ++c) {
        constexpr int dummy = 0; (void)dummy;
        const int h = c / SP, cs = c % SP;                      // (compile-time after unrolling)
        // (the first item of this step's slice asks for its noise / skip values before the MFMAs)
        if (cs >= 1) {
#pragma unroll
          for (int k = 0; k < ITEMS; ++k)
            if (slice_of(k) == cs - 1 && (k == 0 || slice_of(k - 1) != cs - 1)) item_load(k, h, io[k & 1]);
        }
        PPROF_MARK(7);
        mfma_chunk(c & 1, WRES ? c : (c & 1), c == 0);
        PPROF_MARK(4);
        if (cs == 0) {
          acc_to_lds(h);                                        // (the previous half's items finished before the last barrier)
        } else {
#pragma unroll
          for (int k = 0; k < ITEMS; ++k)
            if (slice_of(k) == cs - 1) {
              if (k + 1 < ITEMS && slice_of(k + 1) == cs - 1) item_load(k + 1, h, io[(k + 1) & 1]);
              item_finish(k, h, io[k & 1]);
            }
        }
        if (c == NKC - 1) hand_over();
        PPROF_MARK(5);
        MIGAN_BARRIER_LDS();
        PPROF_MARK(6);
      }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (h > 0) MIGAN_BARRIER_LDS();                           // every wave is done with the previous half (group A joins these barriers)
      acc_to_lds(h);
      MIGAN_BARRIER_LDS();
      item_load(0, h, io[0]);
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        if (k + 1 < ITEMS) item_load(k + 1, h, io[(k + 1) & 1]);
        item_finish(k, h, io[k & 1]);
      }
    }
  }
  PPROF_MARK(5);
  PPROF_END(AT);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// down=2 SeparableConv2d as ONE pipelined kernel (reference :154-163: depthwise 3x3 + bias -> lrelu_agc -> Downsample2d (4x4 FIR,
// stride 2, :58-76) -> 1x1 conv -> lrelu_agc).  Rounds 1-3 ran it as dwfir_kernel + a pointwise GEMM with the half-resolution
// Cin-channel tensor round-tripping through HBM (1.35x the algorithmic traffic of these layers, profiles/r03_pmc_traffic_*).  Here the
// FIR result never leaves the CU: it is written as the fp16 hi/lo A operand of the 1x1 straight into LDS.
//
//   tile    4 x 16 low-resolution output pixels (64 GEMM rows) x NT output channels, K walked in chunks of 16 input channels
//   DMA     the 12 x 36-pixel input window of chunk s+R-1 (27 KB) -> LDS ring, and the chunk's weight planes
//   A       stage 1: depthwise 3x3 + bias + act on the 10 x 34 grid the FIR window touches (zero outside the image: the FIR's padding)
//           -> d_s; [barrier]; stage 2: 16-tap FIR, x 2^7, fp16 split -> A planes of chunk s+1        (two barriers per chunk)
//   B       NB = 8 or 4 waves = 2 row blocks x 4 or 2 column groups: the MFMAs of chunk s in the first half of a step (beside stage 1), a slice
//           of the previous tile's epilogue (wave-private transpose, activation, stores) in the second (beside stage 2)
//   The kernel is bound by group A (stage 1 above all): NA = 12, NB = 4 is the default split (profiles/r04_pipe_layers.txt).
//
// The depthwise + FIR stage runs once per pixel only while one workgroup owns all of Cout (Cout <= 256); wider layers keep the two-kernel form.
template <int NT, int CIN, int R>
struct DownLds {
  static constexpr int KC = 16, NKC = CIN / KC;
  static constexpr int IN_UNITS = 12 * 36 * 4;                      // 16-byte units of one input window chunk ([432 pixels][16 channels] fp32)
  static constexpr int A_BUF = 2 * 64 * 32;                         // hi + lo plane, [64 rows][16 k] fp16
  static constexpr int B_CHUNK = 2 * NT * 32;                       // hi + lo plane of one chunk, [NT rows][16 k] fp16
  static constexpr bool WRES = NKC * B_CHUNK <= 32 * 1024;
  static constexpr int NBUF_B = WRES ? NKC : 2;
  static constexpr int DWP = 17;                                    // depthwise grid [10 rows][column parity][17][16 channels] fp32: the stride-2 FIR reads
  static constexpr int D_SZ = 10 * 2 * DWP * KC * 4;                // become contiguous; odd columns 64 bytes mod 128 after the even ones (stage-1 stores)
  static constexpr int IN_SLOT = IN_UNITS * 16;                     // (1728 units = 27 waves' worth of DMA lanes exactly)
  static constexpr int OFF_D = R * IN_SLOT;
  static constexpr int OFF_A = OFF_D + D_SZ;
  static constexpr int OFF_B = OFF_A + 2 * A_BUF;
  static constexpr int OFF_W = OFF_B + NBUF_B * B_CHUNK;
  static constexpr int OFF_T = OFF_W + NKC * 640;
  template <int NB> static constexpr int total() { return OFF_T + NB * 32 * 32 * 4; }
};

template <int NT, int CIN, int R, int NA, int NB>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS((NA + NB) * 64, (NA + NB) / 4) sepconv_pipedown_kernel(const SepArgs p) {
  typedef DownLds<NT, CIN, R> L;
  constexpr int AT = NA * 64, KC = 16, QC = 4, LG_QC = 2, NKC = L::NKC, MT = 64, GH = 4, GW = 16, lgGW = 4;
  constexpr int IGH = 12, IGW = 36, NPIXW = IGH * IGW, NITEMS = NPIXW * QC, DH = 10, DW = 34, DH2 = 5, DWP = L::DWP;
  constexpr int DNI = (NITEMS + AT - 1) / AT;                       // (the waves past unit 1728 skip their last one: `dshort`)
  constexpr bool WRES = L::WRES;
  constexpr int DNB = (4 * NT + AT - 1) / AT;                       // weight-plane DMAs per group-A thread and chunk (4 NT units)
  static_assert(NB == 4 || NB == 8, "2 row blocks x 2 or 4 column groups");
  constexpr int NTIW = NT * 2 / (32 * NB);                          // 32-column blocks of a B wave
  // two blocks per wave at most ride under the next tile (block 0 in the transpose patch, block 1 in a second register set); with more,
  // the whole epilogue runs at the tile's end -- group B has the slack (this kernel is bound by the depthwise + FIR stage of group A)
  constexpr bool DEFER = NTIW <= 2;
  static_assert(NTIW <= NKC, "one epilogue block per K step");
  static_assert(NT == 128 || NT == 256, "one workgroup owns 128 or 256 output channels");
  static_assert(R == 2, "two-slot ring");
  static_assert(L::template total<NB>() <= 160 * 1024, "LDS budget");
  MIGAN_DYN_SMEM(smem);
  char* const lds = reinterpret_cast<char*>(smem);
  constexpr int IN_SLOT = L::IN_SLOT, OFF_D = L::OFF_D, OFF_A = L::OFF_A, OFF_B = L::OFF_B, OFF_W = L::OFF_W, OFF_T = L::OFF_T;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = MIGAN_UNIFORM(tid >> 6);
  const bool groupA = tid < AT;
  const int HO = p.H >> 1, WO = p.W >> 1;                            // p.H, p.W: the full-resolution input; the GEMM runs at HO x WO

  // ---- tile schedule (as sepconv_pipe_kernel) ------------------------------------------------------------------------------------
  const int ntiles = p.tiles_x * p.tiles_y * p.nchunks * p.B;
  const int xcd = (int)blockIdx.x & 7;
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int tcnt = tq + (xcd < tr ? 1 : 0);
  const int tbase = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int tstep = ((int)gridDim.x + 7 - xcd) >> 3;
  const int tl0 = (int)blockIdx.x >> 3;
  const int T = tl0 < tcnt ? (tcnt - tl0 + tstep - 1) / tstep : 0;
  if (T == 0) return;
  const int G = T * NKC;
  PPROF_BEGIN();
  struct TileCur {
    int n, x, y, b;
  };
  const int st_n = tstep % p.nchunks, st_r1 = tstep / p.nchunks;
  const int st_x = st_r1 % p.tiles_x, st_r2 = st_r1 / p.tiles_x;
  const int st_y = st_r2 % p.tiles_y, st_b = st_r2 / p.tiles_y;
  TileCur tile0;
  {
    int t = tbase + tl0;
    tile0.n = t % p.nchunks; t /= p.nchunks;
    tile0.x = t % p.tiles_x; t /= p.tiles_x;
    tile0.y = t % p.tiles_y;
    tile0.b = t / p.tiles_y;
  }
  auto tile_next = [&](TileCur& c) {
    int carry = 0;
    c.n += st_n;
    if (c.n >= p.nchunks) { c.n -= p.nchunks; carry = 1; }
    c.x += st_x + carry; carry = 0;
    if (c.x >= p.tiles_x) { c.x -= p.tiles_x; carry = 1; }
    c.y += st_y + carry; carry = 0;
    if (c.y >= p.tiles_y) { c.y -= p.tiles_y; carry = 1; }
    c.b += st_b + carry;
  };

  if (groupA) {
    MIGAN_SETPRIO(2);                                    // (the critical group of this kernel)
    const int lt = tid;
    float* const w_s = reinterpret_cast<float*>(lds + OFF_W);
    float* const d_s = reinterpret_cast<float*>(lds + OFF_D);
    // depthwise taps + bias of every 16-channel chunk: per chunk tap-major [9][16], then bias [16]
    for (int i = lt; i < CIN * 9 / 4; i += AT) {
      const f4 v = ld4(p.wdw + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = i * 4 + e, ch = f / 9, tap = f - ch * 9;
        w_s[(ch >> 4) * 160 + tap * 16 + (ch & 15)] = v[e];
      }
    }
    for (int i = lt; i < CIN / 4; i += AT) st4(w_s + ((i * 4) >> 4) * 160 + 144 + ((i * 4) & 15), ld4(p.bdw + i * 4));

    // weight planes (chunk-major [plane][CIN/32][CO][32] fp16): the 16-channel half (c & 1) of 32-channel block c >> 1 -> [plane][NT][16]
    const MIGAN_BUF wbuf = MIGAN_MAKE_BUF(p.wsplit, (unsigned)(2 * p.CO * CIN) * 2u);
    const bool bshort = MIGAN_UNIFORM((DNB - 1) * AT + wave_u * 64) >= 4 * NT;     // this wave's last unit lies past the planes (12 waves)
    unsigned dboff[DNB];
#pragma unroll
    for (int j = 0; j < DNB; ++j) {
      const int u = lt + j * AT;                         // unit (plane, row n, 16-byte slot)
      const int plane = u / (2 * NT), rem = u % (2 * NT);
      dboff[j] = (unsigned)(plane * p.CO * CIN + (rem >> 1) * 32 + (rem & 1) * 8) * 2u;
    }
    auto dma_b = [&](int n0_, int chunk, int buf) {
      float* bb = reinterpret_cast<float*>(lds + OFF_B + buf * L::B_CHUNK);
      const unsigned soff = (unsigned)((chunk >> 1) * 32 * p.CO + n0_ * 32 + (chunk & 1) * 16) * 2u;
#pragma unroll
      for (int j = 0; j < DNB; ++j)
        if (j + 1 < DNB || !bshort) MIGAN_LDS_DMA16(wbuf, dboff[j], soff, bb + (j * AT + wave_u * 64) * 4);
    };
    // input window: interior windows use offsets relative to their first pixel (computed once) + a scalar origin
    const bool dshort = MIGAN_UNIFORM((DNI - 1) * AT + wave_u * 64) >= NITEMS;     // this wave's last window unit lies past the slot
    unsigned dgoff[DNI], drel[DNI], tile_soff = 0;
#pragma unroll
    for (int j = 0; j < DNI; ++j) {
      const int i = lt + j * AT;
      drel[j] = 0xfffff000u;
      if (i < NITEMS) {
        const int c4 = i & (QC - 1), pix = i >> LG_QC;
        drel[j] = (unsigned)(((pix / IGW) * p.W + (pix % IGW)) * CIN + c4 * 4) * 4u;
      }
    }
    auto make_dgoff = [&](int gy0_, int gx0_) {            // (gy0_, gx0_): low-resolution tile origin
      const int iy0 = 2 * gy0_ - 2, ix0 = 2 * gx0_ - 2;
      if (iy0 >= 0 && iy0 + IGH <= p.H && ix0 >= 0 && ix0 + IGW <= p.W) {
#pragma unroll
        for (int j = 0; j < DNI; ++j) dgoff[j] = drel[j];
        tile_soff = (unsigned)((iy0 * p.W + ix0) * CIN) * 4u;
        return;
      }
      tile_soff = 0;
#pragma unroll
      for (int j = 0; j < DNI; ++j) {
        const int i = lt + j * AT;
        unsigned g = 0xfffff000u;
        if (i < NITEMS) {
          const int c4 = i & (QC - 1), pix = i >> LG_QC;
          const int yy = iy0 + pix / IGW, xx = ix0 + pix % IGW;
          if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) g = (unsigned)((yy * p.W + xx) * CIN + c4 * 4) * 4u;
        }
        dgoff[j] = g;
      }
    };
    const unsigned img_bytes = (unsigned)(p.H * p.W * CIN) * 4u;
    auto dma_in = [&](int b0_, int chunk, int slot) {
      float* in_s = reinterpret_cast<float*>(lds + slot * IN_SLOT);
      const MIGAN_BUF xbuf = MIGAN_MAKE_BUF(reinterpret_cast<const char*>(p.x) + (size_t)b0_ * img_bytes, img_bytes);
#pragma unroll
      for (int j = 0; j < DNI; ++j)
        if (j + 1 < DNI || !dshort) MIGAN_LDS_DMA16(xbuf, dgoff[j], tile_soff + (unsigned)(chunk * KC) * 4u, in_s + (j * AT + wave_u * 64) * 4);
    };
    // stage 1: depthwise 3x3 + bias + act on the 10 x 34 grid (rows 2 gy0 - 1 .., columns 2 gx0 - 1 ..) -> d_s; one item = 2 vertically
    // adjacent grid pixels x 4 channels; positions outside the image are the FIR's zero padding (reference :67)
    auto stage1 = [&](int slot, int chunk, int gy0_, int gx0_) {
      const float* in_s = reinterpret_cast<const float*>(lds + slot * IN_SLOT);
      const float* wc = w_s + chunk * 160;
      const int c4 = lt & (QC - 1);
      f4 w[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) w[tap] = ld4(wc + tap * KC + c4 * 4);
      const f4 bias = ld4(wc + KC * 9 + c4 * 4);
      const int yim0 = 2 * gy0_ - 1, xim0 = 2 * gx0_ - 1;
      for (int it = lt; it < DH2 * DW * QC; it += AT) {
        const int r = it >> LG_QC;
        const int dx = r % DW, dy0 = 2 * (r / DW);
        const float* ip = in_s + (dy0 * IGW + dx) * KC + c4 * 4;
        f4 win[4][3];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          win[rr][0] = ld4(ip); win[rr][1] = ld4(ip + KC); win[rr][2] = ld4(ip + 2 * KC);
          ip += IGW * KC;
        }
        const int xim = xim0 + dx;
        const bool colin = xim >= 0 && xim < p.W;
        float* dp = d_s + ((dy0 * 2 + (dx & 1)) * DWP + (dx >> 1)) * KC + c4 * 4;
        f4 dv[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          f4 sacc = bias;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) sacc += w[ky * 3 + kx] * win[o + ky][kx];
          const int yim = yim0 + dy0 + o;
          dv[o] = f4{0.f, 0.f, 0.f, 0.f};
          if (colin && yim >= 0 && yim < p.H) dv[o] = act4(sacc);
        }
        // The FIR is separable ([1,3,3,1]/8 per axis) and its stride-2 window of output row oy is the row pairs oy and oy + 1: the pair's two
        // vertical partial sums go to LDS instead of its two rows -- taps (1, 3)/8 for the window that STARTS with this pair (row slot 2k),
        // (3, 1)/8 for the window that ends with it (row slot 2k + 1) -- and stage 2 reads 8 values per output instead of 16 (round 5;
        // dwfir_kernel sums in the same order)
        st4(dp, 0.125f * dv[0] + 0.375f * dv[1]);
        st4(dp + 2 * DWP * KC, 0.375f * dv[0] + 0.125f * dv[1]);
      }
    };
    // stage 2: the horizontal half of the 4x4 FIR, stride 2, taps outer([1,3,3,1])/64 (reference Downsample2d :58-76), x 2^7, fp16 hi/lo -> A planes.  (Splitting an
    // item's 16 taps over two lanes -- 512 items, eight waves' worth -- measured 4 % slower: profiles/r04_pipe_layers.txt)
    auto stage2 = [&](int abuf) {
      char* a_b = lds + OFF_A + abuf * L::A_BUF;
      for (int it = lt; it < MT * QC; it += AT) {
        const int c4 = it & (QC - 1), m = it >> LG_QC;
        const int ox = m & (GW - 1), oy = m >> lgGW;
        const float* dp = d_s + ((2 * oy) * 2 * DWP + ox) * KC + c4 * 4;
        f4 a = {0.f, 0.f, 0.f, 0.f};
        // rows 2 oy (the partial sum of the pair that opens this output's window) and 2 oy + 3 (of the pair that closes it), four columns
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const float fx = (kx == 0 || kx == 3) ? 0.125f : 0.375f;
          const float* q = dp + ((kx & 1) * DWP + (kx >> 1)) * KC;
          a += fx * (ld4(q) + ld4(q + 3 * 2 * DWP * KC));
        }
        u2v h1, h2;
        split2_f16(a * kF16AScale, h1, h2);
        char* d = a_b + m * 32 + c4 * 8;
        *reinterpret_cast<u2v*>(d) = h1;
        *reinterpret_cast<u2v*>(d + MT * 32) = h2;
      }
    };

    // cursors
    int is = 0, ic = 0, ik = 0, ib0 = 0, in0 = 0, igy0 = 0, igx0 = 0;
    TileCur itc = tile0;
    auto icoords = [&]() { in0 = itc.n * NT; ib0 = itc.b; igy0 = itc.y * GH; igx0 = itc.x * GW; };
    icoords();
    make_dgoff(igy0, igx0);
    auto issue = [&]() {                                  // input window of step `is` -> slot is & 1
      if (is < G) {
        dma_in(ib0, ic, is & 1);
        ++is;
        if (++ic == NKC) {
          ic = 0;
          if (++ik < T) { tile_next(itc); icoords(); make_dgoff(igy0, igx0); }
        }
      }
    };
    // streamed weight planes of step `bs` -> slot bs & 1 (that slot is read by the MFMAs of step bs - 2: issued after they are done)
    int bs = 0, bc = 0, bk = 0, bn0 = in0;
    TileCur btc = tile0;
    auto issue_b = [&]() -> bool {
      if (bs >= G) return false;
      dma_b(bn0, bc, bs & 1);
      ++bs;
      if (++bc == NKC) {
        bc = 0;
        if (++bk < T) { tile_next(btc); bn0 = btc.n * NT; }
      }
      return true;
    };
    // the tile the depthwise stage is working on
    int dk = 0, dc = 0, dgy0 = igy0, dgx0 = igx0;
    TileCur dtc = tile0;
    auto dadvance = [&]() {
      if (++dc == NKC) {
        dc = 0;
        if (++dk < T) { tile_next(dtc); dgy0 = dtc.y * GH; dgx0 = dtc.x * GW; }
      }
    };
    if constexpr (WRES) {
#pragma unroll
      for (int c = 0; c < NKC; ++c) dma_b(in0, c, c);
    } else {
      issue_b();                                          // steps 0 and 1
      issue_b();
    }
    issue();                                              // step 0
    issue();                                              // step 1
    if (dshort) MIGAN_WAIT_VMCNT(DNI - 1); else MIGAN_WAIT_VMCNT(DNI);       // everything but the window of step 1 landed (G >= NKC >= 4 steps)
    MIGAN_BARRIER_LDS();                                  // P1
    stage1(0, 0, dgy0, dgx0);
    MIGAN_BARRIER_LDS();                                  // P2
    stage2(0);
    dadvance();
    MIGAN_WAIT_VMCNT(0);                                  // step 1 landed
    MIGAN_BARRIER_LDS();                                  // barrier 0
    for (int g = 0; g < G; ++g) {
      // step g, first half: B multiplies chunk g; here: refill slot g & 1 (read by stage 1 of step g), stage 1 of step g+1
      issue();                                            // step g+2
      PPROF_MARK(0);
      if (g + 1 < G) stage1((g + 1) & 1, dc, dgy0, dgx0);
      PPROF_MARK(1);
      MIGAN_BARRIER_LDS();
      PPROF_MARK(3);
      // second half: stage 2 of step g+1 -> A planes (g+1) & 1, last read by the MFMAs of step g-1
      if (g + 1 < G) { stage2((g + 1) & 1); dadvance(); }
      PPROF_MARK(14);
      // window of step g+2 (issued a half-step ago) and weight planes of step g+1 (issued a step ago) landed; the planes of step g+2,
      // issued now into the slot the MFMAs of step g have just finished with, stay in flight
      bool newer = false;
      if constexpr (!WRES) newer = issue_b();
      if (newer && !bshort) MIGAN_WAIT_VMCNT(DNB); else if (newer && DNB > 1) MIGAN_WAIT_VMCNT(DNB > 1 ? DNB - 1 : 0); else MIGAN_WAIT_VMCNT(0);
      PPROF_MARK(2);
      MIGAN_BARRIER_LDS();
      PPROF_MARK(3);
    }
    PPROF_END(AT);
    return;
  }

  // ================================================= group B ============================================================================
  const int wb = wave_u - NA;
  const int rb = wb & 1, cb = wb >> 1;                           // row block (32 GEMM rows), column group (NT / (NB / 2) columns)
  const int l31 = lane & 31, half = lane >> 5;
  f16v acc[NTIW], accp[DEFER ? NTIW : 1];
#pragma unroll
  for (int j = 0; j < NTIW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[j][r] = 0.0f; if (DEFER) accp[j][r] = 0.0f; }
  auto mfma_chunk = [&](int abuf, int bbuf, bool first) {
    const char* ab = lds + OFF_A + abuf * L::A_BUF + (rb * 32 + l31) * 32 + half * 16;
    const char* bb = lds + OFF_B + bbuf * L::B_CHUNK + half * 16;
    const f4 a_hi = ld4(reinterpret_cast<const float*>(ab)), a_lo = ld4(reinterpret_cast<const float*>(ab + MT * 32));
    f4 b_hi[NTIW], b_lo[NTIW];
#pragma unroll
    for (int j = 0; j < NTIW; ++j) {
      const char* q = bb + ((cb * NTIW + j) * 32 + l31) * 32;
      b_hi[j] = ld4(reinterpret_cast<const float*>(q));
      b_lo[j] = ld4(reinterpret_cast<const float*>(q + NT * 32));
      if (first) acc[j] = f16v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    }
    // product by product across the wave's column blocks: consecutive MFMAs go to different accumulators
#pragma unroll
    for (int j = 0; j < NTIW; ++j) acc[j] = MIGAN_MFMA_F16_32X32X16(a_lo, b_hi[j], acc[j]);
#pragma unroll
    for (int j = 0; j < NTIW; ++j) acc[j] = MIGAN_MFMA_F16_32X32X16(a_hi, b_lo[j], acc[j]);
#pragma unroll
    for (int j = 0; j < NTIW; ++j) acc[j] = MIGAN_MFMA_F16_32X32X16(a_hi, b_hi[j], acc[j]);
  };
  const float acc_scale = reinterpret_cast<const float*>(p.wsplit)[-4];
  const float gain_s = 1.41421356237309515f * acc_scale;         // (no noise on these layers: the scale folds into the activation gain, exactly)
  const size_t img_out_bytes = (size_t)HO * WO * p.CO * 4;
  int ck = 0, cn0 = 0, cb0 = 0, cgy0 = 0, cgx0 = 0, pn0 = 0, pb0 = 0, pgy0 = 0, pgx0 = 0;
  TileCur ctc = tile0;
  auto ccoords = [&]() { cn0 = ctc.n * NT; cb0 = ctc.b; cgy0 = ctc.y * GH; cgx0 = ctc.x * GW; };
  ccoords();
  // transpose patch (as sepconv_pipe_kernel: [32 rows][32 channels] fp32, 16-byte slots XOR-swizzled by (row >> 1) & 7)
  float* const t_s = reinterpret_cast<float*>(lds + OFF_T) + wb * (32 * 32);
  const int q4 = lane & 7, prow = lane >> 3;
  int twr[4];
  {
    const int u = (l31 >> 2) ^ (2 * half);
#pragma unroll
    for (int v = 0; v < 4; ++v) twr[v] = (4 * half) * 32 + ((u ^ ((v & 1) + 4 * (v >> 1))) << 2) + (l31 & 3);
  }
  const int trd0 = prow * 32 + ((q4 ^ (prow >> 1)) << 2), trd1 = prow * 32 + ((q4 ^ (prow >> 1) ^ 4) << 2);
  auto stage_block = [&](const f16v& a) {
    MIGAN_WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 16; ++r) t_s[twr[((r >> 1) & 1) + 2 * ((r >> 2) & 1)] + ((r & 3) + 8 * (r >> 2)) * 32] = a[r];
    MIGAN_WAVE_SYNC();
  };
  f4 tv[4];
  auto fetch_block = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) tv[q] = ld4(t_s + ((q & 1) ? trd1 : trd0) + q * 8 * 32);
  };
  // pixel of (q, lane): GEMM row 32 rb + 8 q + prow -> tile row 2 rb + (q >> 1), column 8 (q & 1) + prow
  auto finish_block = [&](int j) {
    char* yb = reinterpret_cast<char*>(p.y) + (size_t)pb0 * img_out_bytes;
    const unsigned pix0 = (unsigned)((pgy0 + 2 * rb) * WO + pgx0 + prow);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f4 v = act4g(tv[q], gain_s);
      const unsigned pix = pix0 + (unsigned)((q >> 1) * WO + (q & 1) * 8);
      Io<0>::st(yb, (pix * (unsigned)p.CO + (unsigned)(pn0 + (cb * NTIW + j) * 32 + q4 * 4)) * 4u, v);
    }
  };
  auto hand_over = [&]() {
    pn0 = cn0; pb0 = cb0; pgy0 = cgy0; pgx0 = cgx0;
    if constexpr (DEFER) {
      stage_block(acc[0]);
#pragma unroll
      for (int j = 1; j < NTIW; ++j) accp[j] = acc[j];
    } else {
#pragma unroll
      for (int j = 0; j < NTIW; ++j) {
        stage_block(acc[j]);
        fetch_block();
        finish_block(j);
      }
    }
    if (++ck < T) { tile_next(ctc); ccoords(); }
  };
  MIGAN_BARRIER_LDS();                                          // P1
  MIGAN_BARRIER_LDS();                                          // P2
  MIGAN_BARRIER_LDS();                                          // barrier 0
  if constexpr (!DEFER) {
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int c = 0; c < NKC; ++c) {
        PPROF_MARK(7);
        mfma_chunk(c & 1, WRES ? c : (c & 1), c == 0);
        PPROF_MARK(4);
        MIGAN_BARRIER_LDS();
        PPROF_MARK(6);
        if (c == NKC - 1) hand_over();                          // (second half of the step: group A runs the FIR stage)
        PPROF_MARK(9);
        MIGAN_BARRIER_LDS();
        PPROF_MARK(6);
      }
    }
    PPROF_END(AT);
    return;
  } else {
  // first tile: nothing to finish under it
#pragma unroll
  for (int c = 0; c < NKC; ++c) {
    mfma_chunk(c & 1, WRES ? c : (c & 1), c == 0);
    MIGAN_BARRIER_LDS();
    if (c == NKC - 1) hand_over();
    MIGAN_BARRIER_LDS();
  }
  for (int t = 1; t < T; ++t) {
#pragma unroll
    for (int c = 0; c < NKC; ++c) {
      PPROF_MARK(7);
      mfma_chunk(c & 1, WRES ? c : (c & 1), c == 0);
      PPROF_MARK(4);
      MIGAN_BARRIER_LDS();
      PPROF_MARK(6);
      // second half of the step (group A runs the FIR stage): a slice of the previous tile's epilogue
      if (c < NTIW) {
        fetch_block();
        finish_block(c);
        if (c + 1 < NTIW) stage_block(accp[c + 1 < NTIW ? c + 1 : 0]);
      }
      PPROF_MARK(9);
      if (c == NKC - 1) hand_over();
      PPROF_MARK(7);
      MIGAN_BARRIER_LDS();
      PPROF_MARK(6);
    }
  }
#pragma unroll
  for (int j = 0; j < NTIW; ++j) {
    fetch_block();
    finish_block(j);
    if (j + 1 < NTIW) stage_block(accp[j + 1 < NTIW ? j + 1 : 0]);
  }
  PPROF_MARK(9);
  PPROF_END(AT);
  }
}

}  // namespace migan
