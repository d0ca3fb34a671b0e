// MI-GAN generator forward: software-pipelined SeparableConv2d for the full-resolution layers (round 4).
//
// Same arithmetic as sepconv_kernel (reference lib/model_zoo/migan_inference.py:154-170: depthwise 3x3 + bias -> lrelu_agc -> 1x1 conv
// -> [2x FIR upsample] -> noise -> lrelu_agc -> [+ skip] -> [ToRGB]), other schedule.  The stage ablation of the round-3 kernels
// (profiles/r03_ablation_batch32.txt) shows their phases ADD UP: a workgroup loads, computes, then stores, and the 15 GB of output
// stores of a forward are fully exposed (27 % of the time).  Here one persistent 8-wave workgroup per CU is a four-stage pipeline over
// its tiles, every stage on its own hardware queue:
//
//   DMA    buffer_load ... lds: input tile of K-chunk s+R-1 (and 1x1 weight planes) HBM/L2 -> LDS ring, no registers, issued by group A,
//          R-1 chunks (24-48 KB per CU) in flight across the barriers (counted s_waitcnt vmcnt, raw s_barrier)
//   A      waves 0-3: depthwise 3x3 + bias + act + fp16 hi/lo split of chunk s+1 -> A-operand planes (VALU + LDS)
//   B      waves 4-7: v_mfma_f32_32x32x16_f16 x 3 of chunk s (matrix pipe) ...
//   store  ... and, between their MFMA groups, the EPILOGUE OF THE PREVIOUS TILE: its accumulators wait in a second register set
//          (plain layers: transposed through a wave-private LDS patch, no barrier) or in a dedicated LDS result tile (FIR-up layers),
//          and a slice of its noise / activation / skip / ToRGB work and of its global stores is issued in every K step of the next
//          tile.  The stores of tile t therefore drain while tile t+1 is loaded and computed; B never waits for them (it issues no
//          DMA, and its own loads are requested one slice ahead of the stores that precede their use).
//
// One workgroup barrier per K chunk.  The 1x1 weight planes and the depthwise taps of ALL chunks stay in LDS for the life of the
// workgroup where they fit (Cin x Cout <= 128 x 64: the 512x512 layers), otherwise the planes stream through a two-slot ring.
// fp32 activation storage, f16x2 GEMM (the default of that storage format); everything else keeps sepconv_kernel.
#pragma once

namespace migan {

constexpr int kPipeThreads = 512;

// LDS carve of one instantiation (bytes), shared with the host plan (pipe_lds_bytes)
template <int MODE, int NT, int CIN, bool FROMRGB, int R>
struct PipeLds {
  static constexpr int NKC = CIN / 32;
  static constexpr int DNI = 6;                                     // input-tile DMAs per group-A thread and chunk (1440 of 1536 units used)
  static constexpr int IN_SLOT = DNI * 256 * 16;                    // ring slot: [180 pixels][32 channels] fp32 + padding
  static constexpr int A_BUF = 2 * 128 * 64;                        // hi + lo plane of the A operand, [128 rows][32 k] fp16 each
  static constexpr int B_CHUNK = 2 * NT * 64;                       // hi + lo plane of one K chunk of the weights, [NT rows][32 k] fp16
  static constexpr bool WRES = NKC * B_CHUNK <= 32 * 1024;          // all chunks resident
  static constexpr int NBUF_B = WRES ? NKC : 2;
  static constexpr int OFF_IN = 0;
  static constexpr int OFF_A = OFF_IN + R * IN_SLOT;
  static constexpr int OFF_B = OFF_A + 2 * A_BUF;
  static constexpr int OFF_W = OFF_B + NBUF_B * B_CHUNK;            // depthwise taps, per chunk tap-major [9][32] + bias [32]
  static constexpr int OFF_F = OFF_W + NKC * 1280;                  // FROMRGB: fromrgb weights per chunk input-major [4][32] + bias [32]
  static constexpr int OFF_RGB = OFF_F + (FROMRGB ? NKC * 640 : 0); // FROMRGB: raw network input of the halo tile, two tiles
  static constexpr int OFF_T = OFF_RGB + (FROMRGB ? 2 * 180 * 16 : 0);
  static constexpr int T_SZ = MODE == MODE_UP ? 128 * (NT + 4) * 4 : 4 * 32 * 36 * 4;   // FIR-up: shared result tile; plain: one transpose patch per B wave
  static constexpr int TOTAL = OFF_T + T_SZ;
  static_assert(TOTAL <= 160 * 1024, "LDS budget");
};

template <int MODE, int NT, int CIN, bool FROMRGB, bool TORGB, int R>
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(kPipeThreads, 2) sepconv_pipe_kernel(const SepArgs p) {
  static_assert(MODE == MODE_NORMAL || MODE == MODE_UP, "plain and FIR-up layers");
  static_assert(!FROMRGB || MODE == MODE_NORMAL, "FromRGB is fused into the first plain layer");
  static_assert(!TORGB || MODE == MODE_NORMAL, "ToRGB is fused into plain layers");
  static_assert(R == 2 || R == 3, "ring depth");
  typedef PipeLds<MODE, NT, CIN, FROMRGB, R> L;
  constexpr int MT = 128, KC = 32, QC = 8, LG_QC = 3, GH = 8, GW = 16, lgGW = 4, IGW = GW + 2, NPIX = (GH + 2) * IGW, NITEMS = NPIX * QC;
  constexpr int NKC = L::NKC, DNI = L::DNI, PB = 64, NSLOT = 4, NPL = 2;
  constexpr bool WRES = L::WRES;
  constexpr int DNB = NPL * NT * NSLOT / 256;                       // weight-plane DMAs per group-A thread and chunk
  constexpr int NTI = NT / 32;                                      // 32-column blocks of a B wave (it owns 32 rows x NT)
  static_assert(NKC >= 2 && NKC % 2 == 0, "an even number of K chunks (the A-operand buffer of a step is then a compile-time choice)");
  MIGAN_DYN_SMEM(smem);
  char* const lds = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = MIGAN_UNIFORM(tid >> 6);
  const bool groupA = tid < 256;

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
  auto decode = [&](int k, int& n0_, int& b0_, int& gy0_, int& gx0_) {
    int t = tbase + tl0 + k * tstep;
    const int nch = t % p.nchunks; t /= p.nchunks;
    const int tx = t % p.tiles_x;  t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    n0_ = nch * NT;
    b0_ = t / p.tiles_y;
    gy0_ = ty * p.sy - p.off;
    gx0_ = tx * p.sx - p.off;
  };

  if (groupA) {
    // =============================================== group A: DMA issue + depthwise stage ===========================================
    const int lt = tid;
    float* const w_s = reinterpret_cast<float*>(lds + L::OFF_W);
    // depthwise taps + bias of every chunk, once per workgroup: conv1.weight [CIN][9] -> per chunk tap-major [9][32], then bias [32]
    for (int i = lt; i < CIN * 9 / 4; i += 256) {
      const f4 v = ld4(p.wdw + i * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = i * 4 + e, ch = f / 9, tap = f - ch * 9;
        w_s[(ch >> 5) * 320 + tap * 32 + (ch & 31)] = v[e];
      }
    }
    for (int i = lt; i < CIN / 4; i += 256) st4(w_s + ((i * 4) >> 5) * 320 + 288 + ((i * 4) & 31), ld4(p.bdw + i * 4));
    if constexpr (FROMRGB) {
      // fromrgb.weight [CIN][4] -> per chunk input-major [4][32], then bias [32] (reference :186)
      float* const f_s = reinterpret_cast<float*>(lds + L::OFF_F);
      for (int i = lt; i < CIN; i += 256) {
        const f4 v = ld4(p.frgb_w + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) f_s[(i >> 5) * 160 + e * 32 + (i & 31)] = v[e];
      }
      for (int i = lt; i < CIN / 4; i += 256) st4(f_s + ((i * 4) >> 5) * 160 + 128 + ((i * 4) & 31), ld4(p.frgb_b + i * 4));
    }

    // ---- 1x1 weight planes (split_weights_kernel: chunk-major [plane][CIN/32][CO][32] fp16) -> LDS, XOR swizzle on the SOURCE side ----
    const MIGAN_BUF wbuf = MIGAN_MAKE_BUF(p.wsplit, (unsigned)(NPL * p.CO * CIN) * 2u);
    unsigned dboff[DNB];
#pragma unroll
    for (int j = 0; j < DNB; ++j) {
      const int i = lt + j * 256;                        // 16-byte unit of the LDS image [plane][NT rows][4 slots]
      const int plane = i / (NT * NSLOT), rem = i % (NT * NSLOT);
      const int n = rem / NSLOT, sp = rem % NSLOT;       // LDS row n, stored slot sp holds source slot sp ^ swizzle(n)
      dboff[j] = (unsigned)(plane * p.CO * CIN + n * KC + ((sp ^ ((n >> 2) & (NSLOT - 1))) * 8)) * 2u;
    }
    auto dma_b = [&](int n0_, int chunk, int buf) {
      float* bb = reinterpret_cast<float*>(lds + L::OFF_B + buf * L::B_CHUNK);
      const unsigned soff = (unsigned)(chunk * KC * p.CO + n0_ * KC) * 2u;
#pragma unroll
      for (int j = 0; j < DNB; ++j) MIGAN_LDS_DMA16(wbuf, dboff[j], soff, bb + (j * 256 + wave_u * 64) * 4);
    };

    // ---- input tile of one K chunk -> ring slot.  The image is a buffer descriptor: a halo pixel outside it (the conv's zero padding,
    // reference :126) is a lane offset beyond its range and arrives as zeros; so do the padding units of the slot ----
    unsigned dgoff[DNI];
    auto make_dgoff = [&](int gy0_, int gx0_) {
#pragma unroll
      for (int j = 0; j < DNI; ++j) {
        const int i = lt + j * 256;
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
      for (int j = 0; j < DNI; ++j) MIGAN_LDS_DMA16(xbuf, dgoff[j], (unsigned)(chunk * KC) * 4u, in_s + (j * 256 + wave_u * 64) * 4);
    };

    // ---- FROMRGB: the input tile is act(fromrgb(network input)) (reference :194-195), built by this group instead of copied ----------
    f4 rraw = {0.f, 0.f, 0.f, 0.f};
    auto load_raw = [&](int b0_, int gy0_, int gx0_) {
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (lt < NPIX) {
        const int ix = lt % IGW, iy = lt / IGW;
        const int yy = gy0_ - 1 + iy, xx = gx0_ - 1 + ix;
        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
          if (p.u8_img) {
            v = pack_pixel(p.u8_img, p.u8_mask, ((size_t)b0_ * p.H + yy) * p.W + xx);
          } else {
            const float* src = reinterpret_cast<const float*>(p.x) + ((size_t)b0_ * 4 * p.H + yy) * p.W + xx;
            const size_t plane = (size_t)p.H * p.W;
            v = f4{src[0], src[plane], src[2 * plane], src[3 * plane]};
          }
        }
      }
      rraw = v;
    };
    auto store_raw = [&](int buf) {
      if (lt < NPIX) st4(reinterpret_cast<float*>(lds + L::OFF_RGB + buf * NPIX * 16) + lt * 4, rraw);
    };
    auto build_in = [&](int gy0_, int gx0_, int chunk, int slot, int rbuf) {
      float* in_s = reinterpret_cast<float*>(lds + L::OFF_IN + slot * L::IN_SLOT);
      const float* rgb_s = reinterpret_cast<const float*>(lds + L::OFF_RGB + rbuf * NPIX * 16);
      const float* f_s = reinterpret_cast<const float*>(lds + L::OFF_F) + chunk * 160;
#pragma unroll
      for (int j = 0; j < DNI; ++j) {
        const int i = lt + j * 256;
        if (i < NITEMS) {
          const int c4 = i & (QC - 1), pix = i >> LG_QC;
          const int ix = pix % IGW, iy = pix / IGW;
          const int yy = gy0_ - 1 + iy, xx = gx0_ - 1 + ix;
          f4 v = {0.f, 0.f, 0.f, 0.f};
          if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
            const f4 raw = ld4(rgb_s + pix * 4);
            const float* wr = f_s + c4 * 4;
            v = act4(fromrgb_quad(raw, ld4(wr), ld4(wr + 32), ld4(wr + 64), ld4(wr + 96), ld4(f_s + 128 + c4 * 4)));
          }
          st4(in_s + i * 4, v);
        }
      }
    };

    // ---- depthwise 3x3 + bias + act (x 2^7) + fp16 hi/lo split of one chunk: one 4-row strip x 4 channels per thread ------------------
    auto depthwise = [&](int slot, int chunk, int abuf) {
      const float* in_s = reinterpret_cast<const float*>(lds + L::OFF_IN + slot * L::IN_SLOT);
      const float* wc = w_s + chunk * 320;
      char* a_b = lds + L::OFF_A + abuf * L::A_BUF;
      const int c4 = lt & (QC - 1);
      const int gx = (lt >> LG_QC) & (GW - 1);
      const int r0 = (lt >> (LG_QC + lgGW)) * 4;
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
      for (int o = 0; o < 4; ++o) {
        const int nr = (o + 2) % 3;
        win[nr][0] = nxt[0]; win[nr][1] = nxt[1]; win[nr][2] = nxt[2];
        if (o + 1 < 4) {
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

    // ---- cursors: the step whose input is issued next (is), the step whose weights are issued next (bs: streamed planes only) ------
    int is = 0, ic = 0, ik = 0, islot = 0, ib0 = 0, in0 = 0, igy0 = 0, igx0 = 0;
    decode(0, in0, ib0, igy0, igx0);
    if constexpr (!FROMRGB) make_dgoff(igy0, igx0);
    auto advance_issue = [&]() {
      ++is;
      islot = islot + 1 == R ? 0 : islot + 1;
      if (++ic == NKC) {
        ic = 0;
        if (++ik < T) {
          decode(ik, in0, ib0, igy0, igx0);
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
    auto issue_b = [&]() {                               // streamed weight planes of step bs -> slot bs & 1
      if (bs < G) {
        dma_b(bn0, bc, bs & 1);
        ++bs;
        if (++bc == NKC) {
          bc = 0;
          if (++bk < T) { int b_, y_, x_; decode(bk, bn0, b_, y_, x_); }
        }
      }
    };
    // the depthwise cursor: step ds (chunk dc, ring slot dslot, tile coordinates only matter to FROMRGB)
    int dslot = 0;

    if constexpr (FROMRGB) {
      // ---- prologue: raw(tile 0) -> LDS, steps 0 and 1 built (NKC == 2: tile 0 complete), raw(tile 1) in registers -----------------
      static_assert(!FROMRGB || (NKC == 2 && R == 3), "the fused-FromRGB form is built for Cin = 64 (two chunks) and a three-slot ring");
#pragma unroll
      for (int c = 0; c < NKC; ++c) dma_b(in0, c, c);          // the resident weight planes: the only DMAs of this form
      load_raw(ib0, igy0, igx0);
      store_raw(0);
      MIGAN_WAIT_VMCNT(0);
      int rk = 1, rn0 = 0, rb0 = 0, rgy0 = 0, rgx0 = 0;         // tile whose raw pixels are in registers
      if (rk < T) { decode(rk, rn0, rb0, rgy0, rgx0); load_raw(rb0, rgy0, rgx0); }
      MIGAN_BARRIER_LDS();                                     // P1: taps, fromrgb weights, raw(0) visible (B waits here too)
      // produce(step s): build its input tile; after the last chunk of a tile, hand the next tile's raw pixels over
      auto produce = [&]() {
        if (is < G) {
          build_in(igy0, igx0, ic, islot, ik & 1);
          const bool last = ic == NKC - 1;
          advance_issue();
          if (last && ik < T) {
            store_raw(ik & 1);                                  // raw(tile ik), last read (as buffer ik & 1) two tiles ago
            ++rk;
            if (rk < T) { decode(rk, rn0, rb0, rgy0, rgx0); load_raw(rb0, rgy0, rgx0); }
          }
        }
      };
      produce();                                               // step 0 -> slot 0
      produce();                                               // step 1 -> slot 1 (+ raw(1) -> LDS)
      MIGAN_BARRIER_LDS();                                     // P2
      depthwise(0, 0, 0);
      produce();                                               // step 2 -> slot 2
      MIGAN_BARRIER_LDS();                                     // barrier 0
      dslot = 1;
      int dc = 1;
      for (int g = 0; g < G; ++g) {
        // interval g: B runs the MFMAs of step g; here: depthwise of step g+1, input tile of step g+3
        if (g + 1 < G) depthwise(dslot, dc, (g + 1) & 1);
        produce();                                             // step g+3 -> slot g % 3 (read by the depthwise stage of step g, one interval ago)
        dslot = dslot + 1 == R ? 0 : dslot + 1;
        dc = dc + 1 == NKC ? 0 : dc + 1;
        MIGAN_BARRIER_LDS();
      }
    } else {
      // ---- prologue: resident weight planes, the first R input chunks in flight -----------------------------------------------------
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
      for (int g = 0; g < G; ++g) {
        // interval g: B runs the MFMAs of step g.  Slot g % R (read by the depthwise stage of step g) and weight slot (g+1) & 1 (read by the
        // MFMAs of step g-1) are free: refill them first, then the depthwise stage of step g+1
        if constexpr (!WRES) { if (g >= 1) issue_b(); }        // step g+1 (steps 0 and 1 were issued by the prologue)
        issue_in();                                            // step g+R
        if (g + 1 < G) depthwise(dslot, dc, (g + 1) & 1);
        dslot = dslot + 1 == R ? 0 : dslot + 1;
        dc = dc + 1 == NKC ? 0 : dc + 1;
        // before the barrier that starts interval g+1: input of step g+2 and weights of step g+1 landed.  Everything issued before the
        // newest input chunk is then complete, and that chunk (step g+R, R = 3) stays in flight
        if (R == 3 && g + 3 < G) MIGAN_WAIT_VMCNT(DNI); else MIGAN_WAIT_VMCNT(0);
        MIGAN_BARRIER_LDS();
      }
    }
    if constexpr (MODE == MODE_UP) {
      MIGAN_BARRIER_LDS();                                     // the last tile's result tile is published by group B
    }
    return;
  }

  // ================================================= group B: MFMAs + the previous tile's epilogue ====================================
  const int tb = tid - 256, wb = wave_u - 4;                    // wave wb owns GEMM rows 32 wb .. 32 wb + 31 (image rows 2 wb, 2 wb + 1 of the tile)
  const int l31 = lane & 31, half = lane >> 5;
  f16v acc[NTI], accp[NTI];
#pragma unroll
  for (int j = 0; j < NTI; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[j][r] = 0.0f; accp[j][r] = 0.0f; }

  auto mfma_chunk = [&](int abuf, int bbuf) {
    const char* ab = lds + L::OFF_A + abuf * L::A_BUF;
    const char* bb = lds + L::OFF_B + bbuf * L::B_CHUNK;
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      f4 av[NPL], bv[NTI][NPL];
      {
        const int row = wb * 32 + l31;
        const char* q = ab + row * PB + (((2 * ks + half) ^ ((row >> 2) & (NSLOT - 1))) << 4);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) av[pl] = ld4(reinterpret_cast<const float*>(q + pl * MT * PB));
      }
#pragma unroll
      for (int j = 0; j < NTI; ++j) {
        const int row = j * 32 + l31;
        const char* q = bb + row * PB + (((2 * ks + half) ^ ((row >> 2) & (NSLOT - 1))) << 4);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) bv[j][pl] = ld4(reinterpret_cast<const float*>(q + pl * NT * PB));
      }
      // smallest products first; consecutive MFMAs go to different accumulators
#pragma unroll
      for (int j = 0; j < NTI; ++j) acc[j] = MIGAN_MFMA_F16_32X32X16(av[1], bv[j][0], acc[j]);
#pragma unroll
      for (int j = 0; j < NTI; ++j) acc[j] = MIGAN_MFMA_F16_32X32X16(av[0], bv[j][1], acc[j]);
#pragma unroll
      for (int j = 0; j < NTI; ++j) acc[j] = MIGAN_MFMA_F16_32X32X16(av[0], bv[j][0], acc[j]);
    }
  };

  const float acc_scale = reinterpret_cast<const float*>(p.wsplit)[-4];     // 1 / (activation scale x weight scale), a power of two
  const bool has_noise = p.noise != nullptr;
  const float ns = has_noise ? p.noise_strength[0] : 0.0f;
  const size_t img_out_bytes = (size_t)p.HO * p.WO * p.CO * 4;

  // coordinates of the tile whose accumulators are in `accp` (pn0 etc.) and of the tile being accumulated (cn0 etc.)
  int ck = 0, cn0 = 0, cb0 = 0, cgy0 = 0, cgx0 = 0, pn0 = 0, pb0 = 0, pgy0 = 0, pgx0 = 0;
  decode(0, cn0, cb0, cgy0, cgx0);
  bool have_prev = false;

  MIGAN_BARRIER_LDS();                                          // P1
  MIGAN_BARRIER_LDS();                                          // P2
  MIGAN_BARRIER_LDS();                                          // barrier 0: A planes of step 0 (and the weight planes) are in LDS

  if constexpr (MODE == MODE_NORMAL) {
    // ---- plain layers: the epilogue runs on the wave's own 32 x NT accumulators, one 32 x 32 block at a time through a wave-private
    // LDS patch (C layout: lane = column, 16 rows per lane -> rows of 32 channels = one 128-byte line per 8 lanes) ----
    float* const t_s = reinterpret_cast<float*>(lds + L::OFF_T) + wb * (32 * 36);
    const int q4 = lane & 7, prow = lane >> 3;                 // this lane's channel quad of a block, its pixel row inside a group of 8
    float nz[4] = {0.f, 0.f, 0.f, 0.f};                         // noise_const of this lane's 4 pixels (rows 8q + prow) of the tile in `accp`
    float nzn[4] = {0.f, 0.f, 0.f, 0.f};                        // ... of the tile being accumulated (requested one K step before the hand-over)
    float rs[4][3];                                             // ToRGB partial sums of those pixels
    unsigned pix0 = 0;                                          // output pixel index of row prow of the wave's first image row
    f4 tw[NTI][3];
    if constexpr (TORGB) {
#pragma unroll
      for (int j = 0; j < NTI; ++j)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) tw[j][ch] = ld4(p.trgb_w + ch * p.CO + j * 32 + q4 * 4);
    }
    // pixel of (q, lane): GEMM row m = 32 wb + 8 q + prow -> tile row 2 wb + (q >> 1), column 8 (q & 1) + prow.
    // The noise values of the NEXT tile are loaded before the last store slice of the previous one is issued, and only used a K step
    // later: the wait in front of their first use then leaves those stores in flight (vmcnt retires in issue order).
    auto request_noise = [&]() {
      if (has_noise) {
        const unsigned px = (unsigned)((cgy0 + 2 * wb) * p.WO + cgx0 + prow);
#pragma unroll
        for (int q = 0; q < 4; ++q) nzn[q] = p.noise[px + (unsigned)((q >> 1) * p.WO + (q & 1) * 8)];
      }
    };
    auto begin_tile_epilogue = [&]() {
      pix0 = (unsigned)((pgy0 + 2 * wb) * p.WO + pgx0 + prow);
#pragma unroll
      for (int q = 0; q < 4; ++q) nz[q] = nzn[q];
      if constexpr (TORGB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) rs[q][0] = rs[q][1] = rs[q][2] = 0.0f;
      }
    };
    auto epi_block = [&](const f16v& a, int j) {
      // accumulator fragment -> patch: lane holds column l31, rows (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
      for (int r = 0; r < 16; ++r) t_s[((r & 3) + 8 * (r >> 2) + 4 * half) * 36 + l31] = a[r];
      MIGAN_WAVE_SYNC();
      char* yb = reinterpret_cast<char*>(p.y) + (size_t)pb0 * img_out_bytes;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f4 v = ld4(t_s + (8 * q + prow) * 36 + q4 * 4);
        v = v * acc_scale + MIGAN_FMUL_RN(nz[q], ns);               // product rounded first, reference :166
        v = act4(v);
        const unsigned pix = pix0 + (unsigned)((q >> 1) * p.WO + (q & 1) * 8);
        Io<0>::st(yb, (pix * (unsigned)p.CO + (unsigned)(pn0 + j * 32 + q4 * 4)) * 4u, v);
        if constexpr (TORGB) {
          float r0, r1, r2;
          torgb_partial(v, tw[j][0], tw[j][1], tw[j][2], r0, r1, r2);
          rs[q][0] += r0; rs[q][1] += r1; rs[q][2] += r2;
        }
      }
      MIGAN_WAVE_SYNC();                                        // the patch is rewritten by the next block
    };
    auto end_tile_epilogue = [&]() {
      if constexpr (TORGB) {
        // sum the 8 lanes of a pixel (butterfly inside groups of 8), then lane q4 == 0 adds bias + the 2x-upsampled previous image
        // (reference :308-313) and writes the three planes
        const size_t plane = (size_t)p.HO * p.WO;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float s0 = rs[q][0], s1 = rs[q][1], s2 = rs[q][2];
          s0 += MIGAN_SWIZZLE_XOR(s0, 1); s1 += MIGAN_SWIZZLE_XOR(s1, 1); s2 += MIGAN_SWIZZLE_XOR(s2, 1);
          s0 += MIGAN_SWIZZLE_XOR(s0, 2); s1 += MIGAN_SWIZZLE_XOR(s1, 2); s2 += MIGAN_SWIZZLE_XOR(s2, 2);
          s0 += MIGAN_SWIZZLE_XOR(s0, 4); s1 += MIGAN_SWIZZLE_XOR(s1, 4); s2 += MIGAN_SWIZZLE_XOR(s2, 4);
          if (q4 == 0) {
            const int oy = pgy0 + 2 * wb + (q >> 1), ox = pgx0 + (q & 1) * 8 + prow;
            float o3[3] = {s0 + p.trgb_b[0], s1 + p.trgb_b[1], s2 + p.trgb_b[2]};
            if (p.img_prev) {
              const size_t plane4 = plane >> 2;
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) {
                float pv[4];
                up_taps(p.img_prev + ((size_t)pb0 * 3 + ch) * plane4, p.HO >> 1, p.WO >> 1, oy, ox, pv);
                o3[ch] = up_combine(pv, oy, ox, p.HO >> 1, p.WO >> 1) + o3[ch];
              }
            }
            if (p.u8_out) {
              compose_pixel(p.u8_img, p.u8_mask, p.u8_out, (size_t)pb0 * plane + (size_t)oy * p.WO + ox, o3[0], o3[1], o3[2]);
            } else {
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) p.img_out[((size_t)pb0 * 3 + ch) * plane + (size_t)oy * p.WO + ox] = o3[ch];
            }
          }
        }
      }
    };
    // slice c of a tile's epilogue = the blocks j with j * NKC / NTI == c (NTI <= NKC: at most one block per step)
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int c = 0; c < NKC; ++c) {
        mfma_chunk(c & 1, WRES ? c : (c & 1));
        if (c == NKC - 1) request_noise();
        if (have_prev) {
#pragma unroll
          for (int j = 0; j < NTI; ++j)
            if ((NTI >= NKC ? j / (NTI / NKC) : j * (NKC / NTI)) == c) epi_block(accp[j], j);
          if (c == NKC - 1) end_tile_epilogue();
        }
        if (c == NKC - 1) {
#pragma unroll
          for (int j = 0; j < NTI; ++j) {
            accp[j] = acc[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
          }
          pn0 = cn0; pb0 = cb0; pgy0 = cgy0; pgx0 = cgx0;
          have_prev = true;
          if (++ck < T) decode(ck, cn0, cb0, cgy0, cgx0);
          begin_tile_epilogue();                               // its noise values are requested a whole K step before their first use
        }
        MIGAN_BARRIER_LDS();
      }
    }
    // the last tile: nothing left to hide it under
#pragma unroll
    for (int j = 0; j < NTI; ++j) epi_block(accp[j], j);
    end_tile_epilogue();
  } else {
    // ---- FIR-up layers (reference Upsample2d :79-103 after the 1x1): the 2x polyphase FIR needs the 3x3 neighbourhood of the GEMM
    // result, so the accumulators of a finished tile go to a dedicated LDS result tile during the first K step of the next tile
    // (published by that step's barrier) and the 6 x 14 interior pixels x NT/4 channel quads are worked off in the steps after it ----
    constexpr int GS = NT + 4, QN = NT / 4, LG_QN = (QN == 16) ? 4 : 5;
    static_assert(QN == 16 || QN == 32, "FIR-up tiles: 64 or 128 output channels");
    constexpr int STEP = 256 >> LG_QN;                          // GEMM rows between the items of a thread
    constexpr int ITEMS = MT * QN / 256;
    float* const g_s = reinterpret_cast<float*>(lds + L::OFF_T);
    auto acc_to_lds = [&]() {
#pragma unroll
      for (int j = 0; j < NTI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          // halo pixels outside the low-resolution image contribute zeros to the FIR (reference pads with zeros :101)
          const int ly = pgy0 + (row >> lgGW), lx = pgx0 + (row & (GW - 1));
          float v = accp[j][r];
          if (ly < 0 || ly >= p.H || lx < 0 || lx >= p.W) v = 0.0f;
          g_s[row * GS + j * 32 + l31] = v;
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
    auto item_geo = [&](int k, int& m, unsigned& lpix, unsigned& loff) -> bool {
      m = m0 + k * STEP;
      const int gy = m >> lgGW, gx = m & (GW - 1);
      if (gy < 1 || gy > GH - 2 || gx < 1 || gx > GW - 2) return false;
      const int ly = pgy0 + gy, lx = pgx0 + gx;
      if (ly >= p.H || lx >= p.W) return false;                  // ragged right / bottom edge of the tile grid
      lpix = (unsigned)((2 * ly) * p.WO + 2 * lx);
      loff = (lpix * (unsigned)p.CO + (unsigned)(pn0 + c4 * 4)) * 4u;
      return true;
    };
    auto item_load = [&](int k, ItemIo& io) {
      int m;
      unsigned lpix, loff;
      if (!item_geo(k, m, lpix, loff)) return;
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
    auto item_finish = [&](int k, const ItemIo& io) {
      int m;
      unsigned lpix, loff;
      if (!item_geo(k, m, lpix, loff)) return;
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
    // items of slice c (c = 1 .. NKC-1): an even share of the items that can be interior rows of the 8 x 16 grid (tile rows 1..6)
    constexpr int SL = NKC - 1, K0 = GW / STEP, K1 = ITEMS - K0;
    auto slice_of = [](int k) { return 1 + (k - K0) * SL / (K1 - K0); };
    ItemIo io[2];
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int c = 0; c < NKC; ++c) {
        // (the first item of this step's slice asks for its noise / skip values before the MFMAs)
        if (have_prev && c >= 1) {
#pragma unroll
          for (int k = K0; k < K1; ++k)
            if (slice_of(k) == c && (k == K0 || slice_of(k - 1) != c)) item_load(k, io[k & 1]);
        }
        mfma_chunk(c & 1, WRES ? c : (c & 1));
        if (have_prev) {
          if (c == 0) {
            acc_to_lds();                                       // (the previous result tile was consumed before the last barrier)
          } else {
#pragma unroll
            for (int k = K0; k < K1; ++k)
              if (slice_of(k) == c) {
                if (k + 1 < K1 && slice_of(k + 1) == c) item_load(k + 1, io[(k + 1) & 1]);
                item_finish(k, io[k & 1]);
              }
          }
        }
        if (c == NKC - 1) {
#pragma unroll
          for (int j = 0; j < NTI; ++j) {
            accp[j] = acc[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
          }
          pn0 = cn0; pb0 = cb0; pgy0 = cgy0; pgx0 = cgx0;
          have_prev = true;
          if (++ck < T) decode(ck, cn0, cb0, cgy0, cgx0);
        }
        MIGAN_BARRIER_LDS();
      }
    }
    acc_to_lds();
    MIGAN_BARRIER_LDS();                                        // (group A joins this one)
    item_load(K0, io[K0 & 1]);
#pragma unroll
    for (int k = K0; k < K1; ++k) {
      if (k + 1 < K1) item_load(k + 1, io[(k + 1) & 1]);
      item_finish(k, io[k & 1]);
    }
  }
}

}  // namespace migan
