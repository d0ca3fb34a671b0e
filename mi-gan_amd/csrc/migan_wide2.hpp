// MI-GAN generator forward: the 256- / 512-channel plain SeparableConv2d layers (reference lib/model_zoo/migan_inference.py:154-170,
// down = up = 1: depthwise 3x3 + bias -> lrelu_agc -> 1x1 conv -> noise -> lrelu_agc) as a software-pipelined PERSISTENT kernel on
// 256-pixel x 256-channel tiles (round 5).  Same arithmetic, operand split and summation order as sepconv_wide_kernel (128 x 256
// tiles, one tile per workgroup), which it replaces wherever a launch has enough 16 x 16-pixel tiles to fill the chip.
//
// Why another tile.  The 128 x 256 kernel is bound by what one CU can pull through its vector-memory pipeline (one in-order queue of about
// 48 - 64 KB of requests that drains at the CU's share of HBM while input tiles stream, L2-resident weight planes queueing behind the misses:
// profiles/r05_wide2.md): per 32-channel K chunk a workgroup DMAs 23 KB of input tile and 32 KB of weight planes, every byte with ONE chunk
// period of flight (the double-buffered 145 KB of LDS leave no room for a third slot), and each tile pays its own prologue and a
// compute-then-store epilogue (profiles/r04_pipe_layers.txt (d), (k): 3.6k cycles per chunk against 1.5k of MFMA time, 8.5 us per tile
// outside the K loop).  Here:
//   * 16 x 16 pixels per tile: the weight planes -- 58 % of the bytes above -- are streamed once per 256 pixels instead of once per
//     128, and the halo shrinks from 1.41x to 1.27x: 37 KB per (256 pixels x 16 channels) instead of 55 KB for the same MACs;
//   * K in sub-chunks of 16 channels: input ring of three slots, 64 KB of weight-plane ring (two whole 32-channel chunks: whole 64-byte rows
//     = full cache lines; or four 16-channel halves), taps ring of three -- every DMA has TWO to THREE barrier intervals of flight under a
//     counted s_waitcnt vmcnt, with the same MFMA work between barriers as before (eight MFMA waves x 24 v_mfma_f32_32x32x16_f16 = 1536
//     cycles per SIMD);
//   * one persistent workgroup per CU, 4 depthwise waves (group A, which also issues every DMA) + 8 MFMA waves (group B: 4 row
//     blocks x 2 column halves, 64 x 128 accumulators = 128 registers each): the ring runs on across tile boundaries (group A is
//     three sub-chunks ahead with the loads and one ahead with the depthwise stage), so a tile has no prologue of its own;
//   * the epilogue runs straight from the accumulator registers of group B (C layout: a lane owns one output channel of 16
//     pixels; per accumulator register the two half-waves store one 128-byte line each) -- no result tile in LDS, no barrier, and
//     group A's next depthwise stage runs beside it.
// LDS: 3 x 20.25 KB input + 64 KB weight planes + 2 x 16 KB A planes + 3 x 640 B taps = 159.1 KB.  fp32 storage, f16x2 GEMM.
// Measured: -22 .. -24 % per layer against the 128 x 256 tile; what bounds it now and the variants that were built, measured and dropped
// (loader wave, DMAs on the MFMA waves, 16-byte stores, activation + split on the MFMA waves, lookahead 2, stagger): profiles/r05_wide2.md.
#pragma once

namespace migan {

constexpr int kW2AWaves = 4, kW2BWaves = 8, kW2Threads = (kW2AWaves + kW2BWaves) * 64;

struct W2Lds {
  static constexpr int R_IN = 3, R_B = 4, R_T = 3;
  static constexpr int IN_SLOT = 18 * 18 * 16 * 4;       // [324 halo pixels][16 channels] fp32
  static constexpr int B_SLOT = 2 * 256 * 32;            // hi + lo plane of one sub-chunk of the weights, [256 rows][16 k] fp16
  static constexpr int A_BUF = 2 * 256 * 32;             // hi + lo plane of the A operand, [256 rows][16 k] fp16
  static constexpr int TAP_SLOT = 160 * 4;               // tap-major [9][16] + bias [16]
  static constexpr int OFF_IN = 0;
  static constexpr int OFF_B = OFF_IN + R_IN * IN_SLOT;
  static constexpr int OFF_A = OFF_B + R_B * B_SLOT;
  static constexpr int OFF_W = OFF_A + 2 * A_BUF;
  static constexpr int TOTAL = OFF_W + R_T * TAP_SLOT;
  static_assert(TOTAL <= 160 * 1024, "LDS budget");
};

template <int N> struct W2Int { static constexpr int value = N; };


// V bit 1: the pointwise GEMM of a down=2 layer (reference :155-163 after Downsample2d: the 1x1 on dwfir_kernel's half-resolution output) -- the
// same ring and MFMA / epilogue code, a 16 x 16 input tile without halo, and x 2^7 + fp16 split in place of the depthwise stage.
// V bit 0: the weight planes arrive as whole 32-channel chunks (two slots of 32 KB, eight DMA instructions per wave every other sub-step), V = 0:
// as 16-channel halves (four slots of 16 KB, four instructions per sub-step).  The planes are stored chunk-major [plane][CI/32][CO][32], a
// row = 64 bytes: a DMA of half rows touches 32 rows x 32 bytes = sixteen half-used cache lines per instruction and measured 195 cycles at
// issue (profiles/r05_wide2.md), one of whole rows sixteen rows x 64 bytes = eight full lines.  (A template also so that translation units
// that only need W2Lds do not emit the kernel.)
template <int V>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(kW2Threads, 3) sepconv_wide2_kernel(const SepArgs p) {
  typedef W2Lds L;
  constexpr int KS = 16, GW = 16, IGW = 18, NPIX = 18 * 18, NITEMS = NPIX * 4;     // 1296 16-byte units per input slot
  constexpr int AT = kW2AWaves * 64;
  MIGAN_DYN_SMEM(smem);
  char* const lds = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = MIGAN_UNIFORM(tid >> 6);
  const bool groupA = tid < AT;
  const int CI = p.CI;
  const int nks = CI / KS;                                           // sub-chunks per tile (a multiple of 4: the host checks CI % 64 == 0)

  // ---- tile schedule: XCD-contiguous ranges walked by the persistent workgroups of each XCD (as sepconv_pipe_kernel) ----------------
  const int ntiles = p.tiles_x * p.tiles_y * p.nchunks * p.B;
  const int xcd = (int)blockIdx.x & 7;
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int tcnt = tq + (xcd < tr ? 1 : 0);
  const int tbase = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int tstep = ((int)gridDim.x + 7 - xcd) >> 3;
  const int tl0 = (int)blockIdx.x >> 3;
  const int T = tl0 < tcnt ? (tcnt - tl0 + tstep - 1) / tstep : 0;    // my tiles
  if (T == 0) return;                                                  // (uniform: the whole workgroup leaves)
  const int G = T * nks;                                               // my sub-steps
  // phase profile (-DMIGAN_PHASE_PROF builds): group A -> slots 0..3 [DMA issue, depthwise, vmcnt wait, barrier], group B -> 4 MFMAs (with their
  // fragment reads), 5 epilogue, 6 barrier; slot 8 counts workgroups
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
    // =============================================== group A: every DMA + the depthwise stage ========================================
    const int lt = tid;
    constexpr int LA = 3;                                  // lookahead of the DMA requests in sub-steps = input / tap ring slots
    constexpr bool B32 = (V & 1) != 0;
    constexpr bool PW = (V & 2) != 0;                      // pointwise GEMM: no halo, no depthwise stage (second half of a down=2 layer)
    static_assert(!PW || B32, "the pointwise form is built on the 32-channel-chunk weight ring");
    // ---- input tile of one sub-chunk -> ring slot: 1296 units of 16 bytes = 5 per thread + 4 lanes of every wave (so that each wave
    // issues the same six instructions and one vmcnt count holds for all of them).  The image is a buffer descriptor: a halo pixel
    // outside it (the conv's zero padding, reference :126) is a lane offset beyond its range and arrives as zeros.  Interior tiles use
    // offsets relative to the window's first pixel, computed once; the window's position rides in the scalar offset.
    unsigned drel[6], dgoff[6], tile_soff = 0;
    auto unit_of = [&](int j) { return j < 5 ? lt + j * AT : 1280 + wave_u * 4 + lane; };
    const bool tail_lane = lane < 4;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int i = unit_of(j);
      drel[j] = 0xfffff000u;
      if constexpr (PW) {                                  // 1024 units: the tile's 256 pixels x 4 channel quads, four per thread, no padding anywhere
        if (j < 4) drel[j] = (unsigned)((((i >> 2) >> 4) * p.W + ((i >> 2) & 15)) * CI + (i & 3) * 4) * 4u;
      } else if (i < NITEMS) {
        const int c4 = i & 3, pix = i >> 2;
        drel[j] = (unsigned)(((pix / IGW) * p.W + (pix % IGW)) * CI + c4 * 4) * 4u;
      }
    }
    auto make_dgoff = [&](int gy0_, int gx0_) {
      if constexpr (PW) {
#pragma unroll
        for (int j = 0; j < 6; ++j) dgoff[j] = drel[j];
        tile_soff = (unsigned)((gy0_ * p.W + gx0_) * CI) * 4u;
        return;
      }
      if (gy0_ >= 1 && gy0_ + 17 <= p.H && gx0_ >= 1 && gx0_ + 17 <= p.W) {
#pragma unroll
        for (int j = 0; j < 6; ++j) dgoff[j] = drel[j];
        tile_soff = (unsigned)(((gy0_ - 1) * p.W + (gx0_ - 1)) * CI) * 4u;
        return;
      }
      tile_soff = 0;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int i = unit_of(j);
        unsigned g = 0xfffff000u;
        if (i < NITEMS) {
          const int c4 = i & 3, pix = i >> 2;
          const int yy = gy0_ - 1 + pix / IGW, xx = gx0_ - 1 + pix % IGW;
          if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) g = (unsigned)((yy * p.W + xx) * CI + c4 * 4) * 4u;
        }
        dgoff[j] = g;
      }
    };
    const unsigned img_bytes = (unsigned)(p.H * p.W * CI) * 4u;
    auto dma_in = [&](int b0_, int ks, int slot) {
      float* in_s = reinterpret_cast<float*>(lds + L::OFF_IN + slot * L::IN_SLOT);
      const MIGAN_BUF xbuf = MIGAN_MAKE_BUF(reinterpret_cast<const char*>(p.x) + (size_t)b0_ * img_bytes, img_bytes);
      const unsigned soff = tile_soff + (unsigned)(ks * KS) * 4u;
#pragma unroll
      for (int j = 0; j < (PW ? 4 : 5); ++j) MIGAN_LDS_DMA16(xbuf, dgoff[j], soff, in_s + (j * AT + wave_u * 64) * 4);
      if constexpr (!PW) MIGAN_LDS_DMA16_IF(tail_lane, xbuf, dgoff[5], soff, in_s + (1280 + wave_u * 4) * 4);
    };
    // ---- depthwise taps + bias of one sub-chunk: conv1.weight [CI][9], conv1.bias [CI] -> tap-major [9][16] + [16]: 4-byte DMAs (a
    // gather through the lane offsets), 36 + 4 lanes of every wave ----
    const MIGAN_BUF tbuf = MIGAN_MAKE_BUF(p.wdw, (unsigned)(CI * 9) * 4u);
    const MIGAN_BUF bbuf = MIGAN_MAKE_BUF(p.bdw, (unsigned)CI * 4u);
    const bool tap_lane = lane < 36;
    const unsigned tap_voff = (unsigned)((((wave_u * 36 + lane) & 15) * 9) + ((wave_u * 36 + lane) >> 4)) * 4u;     // unit u = tap * 16 + channel
    const unsigned bias_voff = (unsigned)(wave_u * 4 + lane) * 4u;
    auto dma_taps = [&](int ks, int slot) {
      if constexpr (PW) return;
      float* w_s = reinterpret_cast<float*>(lds + L::OFF_W + slot * L::TAP_SLOT);
      MIGAN_LDS_DMA4_IF(tap_lane, tbuf, tap_voff, (unsigned)(ks * KS * 9) * 4u, w_s + wave_u * 36);
      MIGAN_LDS_DMA4_IF(tail_lane, bbuf, bias_voff, (unsigned)(ks * KS) * 4u, w_s + 144 + wave_u * 4);
    };
    // ---- 1x1 weight planes (split_weights_kernel: chunk-major [plane][CI/32][CO][32] fp16): the 16 k values of a sub-chunk are 32
    // contiguous bytes of a row; LDS image [plane][256 rows][2 slots of 16 bytes], the slots of a row swapped where (row >> 3) & 1 (on
    // the SOURCE side: the LDS destination of a DMA is linear in the lane) so that the fragment reads are conflict-free ----
    const MIGAN_BUF wbuf = MIGAN_MAKE_BUF(p.wsplit, (unsigned)(2 * p.CO * CI) * 2u);
    // B32: LDS image of a chunk [plane][256 rows][4 slots of 16 bytes], slot s of row n holding source slot s ^ ((n >> 2) & 3) (the swizzle of
    // the other kernels; on the SOURCE side, the LDS destination of a DMA is linear in the lane): 8 instructions of 16 rows per wave
    unsigned dboff[B32 ? 8 : 4];
#pragma unroll
    for (int j = 0; j < (B32 ? 8 : 4); ++j) {
      const int i = lt + j * AT;                          // 16-byte unit of the LDS image
      if constexpr (B32) {
        const int plane = i >> 10, n = (i >> 2) & 255, sp = i & 3;
        dboff[j] = (unsigned)(plane * p.CO * CI + n * 32 + ((sp ^ ((n >> 2) & 3)) * 8)) * 2u;
      } else {
        const int plane = i >> 9, n = (i >> 1) & 255, sp = i & 1;
        dboff[j] = (unsigned)(plane * p.CO * CI + n * 32 + ((sp ^ ((n >> 3) & 1)) * 8)) * 2u;
      }
    }
    auto dma_b = [&](int n0_, int ks, int slot) {
      if constexpr (B32) {                                 // the whole 32-channel chunk ks >> 1
        float* bb = reinterpret_cast<float*>(lds + L::OFF_B + slot * (2 * L::B_SLOT));
        const unsigned soff = (unsigned)(((ks >> 1) * p.CO + n0_) * 32) * 2u;
#pragma unroll
        for (int j = 0; j < 8; ++j) MIGAN_LDS_DMA16(wbuf, dboff[j], soff, bb + (j * AT + wave_u * 64) * 4);
      } else {
        float* bb = reinterpret_cast<float*>(lds + L::OFF_B + slot * L::B_SLOT);
        const unsigned soff = (unsigned)(((ks >> 1) * p.CO + n0_) * 32 + (ks & 1) * 16) * 2u;
#pragma unroll
        for (int j = 0; j < 4; ++j) MIGAN_LDS_DMA16(wbuf, dboff[j], soff, bb + (j * AT + wave_u * 64) * 4);
      }
    };

    // ---- depthwise 3x3 + bias + act (x 2^7) + fp16 hi/lo split of one sub-chunk: one 4-row strip x 4 channels per thread ------------
    auto depthwise = [&](int slot, int tslot, int abuf) {
      if constexpr (PW) {
        // the input pixels ARE the A operand rows (dwfir_kernel's output: activated, FIR-filtered, fp32): x 2^7, fp16 hi / lo split
        const float* in_s = reinterpret_cast<const float*>(lds + L::OFF_IN + slot * L::IN_SLOT);
        char* a_b = lds + L::OFF_A + abuf * L::A_BUF;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = lt + j * AT, m = i >> 2, c4 = i & 3;
          char* d = a_b + m * 32 + (((c4 >> 1) ^ ((m >> 3) & 1)) << 4) + ((c4 & 1) << 3);
          u2v h1, h2;
          split2_f16(ld4(in_s + i * 4) * kF16AScale, h1, h2);
          *reinterpret_cast<u2v*>(d) = h1;
          *reinterpret_cast<u2v*>(d + 256 * 32) = h2;
        }
        return;
      }
      const float* in_s = reinterpret_cast<const float*>(lds + L::OFF_IN + slot * L::IN_SLOT);
      const float* wc = reinterpret_cast<const float*>(lds + L::OFF_W + tslot * L::TAP_SLOT);
      char* a_b = lds + L::OFF_A + abuf * L::A_BUF;
      const int c4 = lt & 3;
      const int gx = (lt >> 2) & (GW - 1);
      const int r0 = (lt >> 6) * 4;
      f4 w[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) w[tap] = ld4(wc + tap * KS + c4 * 4);
      const f4 bias = ld4(wc + 144 + c4 * 4);
      const float* ip = in_s + (r0 * IGW + gx) * KS + c4 * 4;
      f4 win[3][3], nxt[3];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        win[rr][0] = ld4(ip); win[rr][1] = ld4(ip + KS); win[rr][2] = ld4(ip + 2 * KS);
        ip += IGW * KS;
      }
      nxt[0] = ld4(ip); nxt[1] = ld4(ip + KS); nxt[2] = ld4(ip + 2 * KS);
      ip += IGW * KS;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int nr = (o + 2) % 3;
        win[nr][0] = nxt[0]; win[nr][1] = nxt[1]; win[nr][2] = nxt[2];
        if (o + 1 < 4) {
          nxt[0] = ld4(ip); nxt[1] = ld4(ip + KS); nxt[2] = ld4(ip + 2 * KS);
          ip += IGW * KS;
        }
        MIGAN_SCHED_FENCE();
        f4 sacc = bias;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) sacc += w[ky * 3 + kx] * win[(o + ky) % 3][kx];
        const int m = ((r0 + o) << 4) + gx;
        char* d = a_b + m * 32 + (((c4 >> 1) ^ ((m >> 3) & 1)) << 4) + ((c4 & 1) << 3);
        u2v h1, h2;
        split2_f16(act4_scaled<7>(sacc), h1, h2);
        *reinterpret_cast<u2v*>(d) = h1;
        *reinterpret_cast<u2v*>(d + 256 * 32) = h2;
      }
    };

    // ---- issue cursor: the sub-step whose input tile, taps and weight planes are requested next -------------------------------------
    int is = 0, ic = 0, ik = 0, islot = 0, tslot = 0;
    TileCur itc = tile0;
    int in0 = itc.n * 256, ib0 = itc.b;
    make_dgoff(itc.y * 16, itc.x * 16);
    auto issue = [&]() {
      if (is >= G) return;
      PPROF_MARK(14);
      dma_in(ib0, ic, islot);
      PPROF_MARK(0);
      dma_taps(ic, tslot);
      PPROF_MARK(9);
      // B32: the chunk that holds this sub-step rides with its ODD sub-step -- requested at interval is - 3, first read by the MFMAs of the even
      // sub-step is - 1, i.e. two intervals later; its slot (chunk parity) was last read by the MFMAs of sub-step is - 4, an interval before
      if constexpr (B32) { if (ic & 1) dma_b(in0, ic, (is >> 1) & 1); }
      else dma_b(in0, ic, is & 3);
      PPROF_MARK(10);
      ++is;
      islot = islot + 1 == LA ? 0 : islot + 1;
      tslot = tslot + 1 == LA ? 0 : tslot + 1;
      if (++ic == nks) {
        ic = 0;
        if (++ik < T) {
          tile_next(itc);
          in0 = itc.n * 256; ib0 = itc.b;
          make_dgoff(itc.y * 16, itc.x * 16);
        }
      }
    };
    // One sub-step = 6 input + 2 tap + 4 weight-plane instructions per wave, in that order.  Before the barrier that ends interval g the
    // input and taps of sub-step g+2 (for the depthwise stage of interval g+1) and the weight planes of sub-step g+1 (for its MFMAs) must
    // have landed; the weight planes of g+2 and everything of g+3 may stay in flight: 4 + 12 = 16 operations.
    // B32: a sub-step is 8 instructions (even) or 8 + 8 (odd, with its chunk behind the input and the taps).  An even interval g issued an
    // odd sub-step: its 16 operations may fly, everything older has landed (the previous interval's input and taps); an odd interval issued 8,
    // and the chunk issued the interval before is needed next: only those 8 may fly.
    constexpr int NEVEN = PW ? 4 : 8, NODD = NEVEN + 8;    // B32: operations of an even / odd sub-step (input [+ taps] [+ the chunk])
    issue(); issue();
    issue();
    MIGAN_WAIT_VMCNT(B32 ? NODD + NEVEN : 28);             // input + taps of sub-step 0 (everything issued after them may fly)
    MIGAN_BARRIER_LDS();                                   // P1
    depthwise(0, 0, 0);
    MIGAN_WAIT_VMCNT(B32 ? NEVEN : 16);                    // planes of sub-step 0 (of chunk 0), input + taps of sub-step 1
    MIGAN_BARRIER_LDS();                                   // barrier 0: A planes + weight planes of sub-step 0, input + taps of sub-step 1
    int dslot = 1, dtap = 1;
    for (int g = 0; g < G; ++g) {
      // interval g: group B runs the MFMAs of sub-step g.  Input slot g % 3 and tap slot g % 3 (read by the depthwise stage of sub-step g,
      // an interval ago) and weight slot (g + 3) & 3 (read by the MFMAs of sub-step g - 1) are free: refill them, then run the
      // depthwise stage of sub-step g + 1
      const bool more = g + LA < G;
      issue();                                             // sub-step g + LA
      PPROF_MARK(14);
      if (g + 1 < G) depthwise(dslot, dtap, (g + 1) & 1);
      PPROF_MARK(1);
      dslot = dslot + 1 == LA ? 0 : dslot + 1;
      dtap = dtap + 1 == LA ? 0 : dtap + 1;
      if (!more) MIGAN_WAIT_VMCNT(0);
      else if (B32 && (g & 1)) MIGAN_WAIT_VMCNT(NEVEN);    // (the chunk issued an interval ago, behind that interval's input tile, is needed next)
      else MIGAN_WAIT_VMCNT(B32 ? NODD : 16);
      PPROF_MARK(2);
      MIGAN_BARRIER_LDS();
      PPROF_MARK(3);
    }
    PPROF_END(AT);
    return;
  }

  // ================================================= group B: MFMAs + the epilogue from registers =====================================
  const int wb = wave_u - kW2AWaves;
  const int wm = wb & 3, wn = wb >> 2;                          // GEMM rows 64 wm .. 64 wm + 63, columns 128 wn .. 128 wn + 127
  const int l31 = lane & 31, half = lane >> 5;
  // fragment reads: row (or column) 32 i + l31 of a block, k half `half`, slots swapped where (row >> 3) & 1 -- the same lane offset for
  // the A and the B planes
  const int foff = l31 * 32 + ((half ^ ((l31 >> 3) & 1)) << 4);
  f16v acc[2][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  };
  // (one code path for every sub-step: a "first product of the tile reads C = 0" variant made the register allocator copy whole
  // accumulator blocks around the join and spill; the 128 v_mov_b32 per tile are noise beside 384+ MFMAs per wave)
  constexpr bool B32 = (V & 1) != 0;
  // B32: weight-plane rows of 64 bytes, 16-byte slot (2 ks + half) ^ ((row >> 2) & 3) (ks = sub-step parity inside the chunk)
  const int foffb = l31 * 64 + ((half ^ ((l31 >> 2) & 3)) << 4);
  auto mfma_step = [&](int abuf, int bidx) {               // bidx = sub-step & 3: the weight-plane slot (B32: chunk slot bidx >> 1, half bidx & 1)
    const char* ab = lds + L::OFF_A + abuf * L::A_BUF + (wm * 64) * 32 + foff;
    const char* bb = B32 ? lds + L::OFF_B + (bidx >> 1) * (2 * L::B_SLOT) + (wn * 128) * 64 + (foffb ^ ((bidx & 1) << 5))
                         : lds + L::OFF_B + bidx * L::B_SLOT + (wn * 128) * 32 + foff;
    constexpr int BROW = B32 ? 64 : 32;
    f4 av[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      av[i][0] = ld4(reinterpret_cast<const float*>(ab + i * 32 * 32));
      av[i][1] = ld4(reinterpret_cast<const float*>(ab + i * 32 * 32 + 256 * 32));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f4 b0v = ld4(reinterpret_cast<const float*>(bb + j * 32 * BROW));
      const f4 b1v = ld4(reinterpret_cast<const float*>(bb + j * 32 * BROW + 256 * BROW));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // smallest products first (the order of every f16x2 kernel of the library: a layer's K chunks are summed identically)
        acc[i][j] = MIGAN_MFMA_F16_32X32X16(av[i][1], b0v, acc[i][j]);
        acc[i][j] = MIGAN_MFMA_F16_32X32X16(av[i][0], b1v, acc[i][j]);
        acc[i][j] = MIGAN_MFMA_F16_32X32X16(av[i][0], b0v, acc[i][j]);
      }
    }
  };

  const float acc_scale = reinterpret_cast<const float*>(p.wsplit)[-4];     // 1 / (activation scale x weight scale), a power of two
  const float gain_s = 1.41421356237309515f * acc_scale;
  const bool has_noise = p.noise != nullptr;
  const float ns = has_noise ? p.noise_strength[0] : 0.0f;
  const size_t img_out_bytes = (size_t)p.H * p.W * p.CO * 4;
  // element r of a 32 x 32 accumulator block: row (r & 3) + 8 (r >> 2) + 4 half, column l31
  const unsigned lane_off = (unsigned)((4 * half) * p.CO + wn * 128 + l31) * 4u;      // this lane's share of an output address (bytes)
  float nzl = 0.0f;                                              // noise_const of pixel `lane` of this wave's 64 rows (tile rows 4 wm .. 4 wm + 3)
  TileCur ctc = tile0;
  auto request_noise = [&]() {
    if (has_noise) {
      int ln = lane;
      MIGAN_OPAQUE(ln);                                          // (recomputed per tile: hoisted out of the persistent loop these lane terms cost registers the MFMA steps need)
      nzl = p.noise[(unsigned)((ctc.y * 16 + wm * 4 + (ln >> 4)) * p.W + ctc.x * 16 + (ln & 15))];
    }
  };
  auto epilogue = [&](auto hn_) {
    constexpr bool HN = decltype(hn_)::value;
    // (opaque copies: the 32 row addresses below are functions of W and CO only -- left visible, the compiler hoists all of them out of the
    // persistent loop and spills them)
    int W_ = p.W, CO_ = p.CO;
    MIGAN_OPAQUE_S(W_); MIGAN_OPAQUE_S(CO_);
    unsigned lo = lane_off;
    MIGAN_OPAQUE(lo);                                            // (a 32-bit value across the K loop, not its zero-extended pair)
    const size_t px_bytes = (size_t)CO_ * 4;
    const char* yt = reinterpret_cast<char*>(p.y) + (size_t)ctc.b * img_out_bytes +
                     ((size_t)((ctc.y * 16 + wm * 4) * W_ + ctc.x * 16) * (size_t)CO_ + (size_t)(ctc.n * 256)) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2);                    // (+ 4 half) row inside the 32-row block: tile row 2 i + (rr >> 4), column rr & 15
        float nsn = 0.0f;
        if constexpr (HN) {
          const float nlo = MIGAN_READLANE(nzl, i * 32 + rr), nhi = MIGAN_READLANE(nzl, i * 32 + rr + 4);
          nsn = MIGAN_FMUL_RN(half ? nhi : nlo, ns);               // product rounded first, reference :166
        }
        char* yr = const_cast<char*>(yt) + (size_t)((2 * i + (rr >> 4)) * W_ + (rr & 15)) * px_bytes;
        // (the four column blocks of a row as one vector: the clamp's NaN test covers two values per compare, clamp4)
        f4 v = f4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
        if constexpr (HN) v = act4(v * acc_scale + nsn);
        else v = act4g(v, gain_s);
        MIGAN_STORE_NT(at_bytes(reinterpret_cast<float*>(yr), lo), v.x);
        MIGAN_STORE_NT(at_bytes(reinterpret_cast<float*>(yr + 128), lo), v.y);
        MIGAN_STORE_NT(at_bytes(reinterpret_cast<float*>(yr + 256), lo), v.z);
        MIGAN_STORE_NT(at_bytes(reinterpret_cast<float*>(yr + 384), lo), v.w);
      }
  };

  MIGAN_BARRIER_LDS();                                          // P1
  MIGAN_BARRIER_LDS();                                          // barrier 0
  zero_acc();
  for (int t = 0; t < T; ++t) {
    for (int c = 0; c < nks; c += 4) {
      mfma_step(0, 0);
      PPROF_MARK(4);
      MIGAN_BARRIER_LDS();
      PPROF_MARK(6);
      mfma_step(1, 1);
      PPROF_MARK(4);
      MIGAN_BARRIER_LDS();
      PPROF_MARK(6);
      if (c + 4 == nks) request_noise();                        // (one interval ahead of its use)
      mfma_step(0, 2);
      PPROF_MARK(4);
      MIGAN_BARRIER_LDS();
      PPROF_MARK(6);
      mfma_step(1, 3);
      PPROF_MARK(4);
      if (c + 4 == nks) {
        if (has_noise) epilogue(TrueT{}); else epilogue(FalseT{});
        zero_acc();
        if (t + 1 < T) tile_next(ctc);
        PPROF_MARK(5);
      }
      MIGAN_BARRIER_LDS();
      PPROF_MARK(6);
    }
  }
  PPROF_END(AT);
}

}  // namespace migan
