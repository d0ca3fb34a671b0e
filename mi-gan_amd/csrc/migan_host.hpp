// Host side of libmigan_hip.so: network plan, tile geometry, launch sequence and the C ABI of
// include/migan_hip.h.  Needs an `rt` namespace (migan_rt_hip.h for the product, tests/emu/hip_emu.h
// for the CPU test harness) and migan_kernels.hpp included before it.
//
// Reference being replaced: lib/model_zoo/migan_inference.py (Generator :355-369, Encoder :203-246,
// Synthesis :318-352, SeparableConv2d :106-170).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/migan_hip.h"
#include "migan_table.hpp"

namespace migan {

// ------------------------------------------------------------------------------------------------
// errors

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline std::string& last_error_ref() {
  static thread_local std::string s;
  return s;
}
#define MIGAN_CHECK(cond, code, msg)                        \
  do {                                                      \
    if (!(cond)) throw ::migan::Error((code), (msg));       \
  } while (0)

inline void rt_check(int rc, const char* what) {
  if (rc != 0) throw Error(MIGAN_ERUNTIME, std::string(what) + ": " + rt::error_string(rc));
}

// ------------------------------------------------------------------------------------------------
// tile geometry of one fused SeparableConv2d launch

struct Geo {
  int mode = MODE_NORMAL, MT = 128, NT = 128, KC = 32;
  bool fromrgb = false;
  int NI = 6, MINW = 2;            // prefetch items per thread, workgroups per CU the kernel is built for
  bool maing = true;               // compile-time tile geometry (8x16 pixels, one image)
  bool persist = false;            // kernel variant whose workgroups walk several tiles
  int gemmv = 0;                   // 0 exact fp32 MFMA, 1 bf16x3-split MFMA, 2 f16x2-split MFMA
  int stv = 0;                     // activation storage format: 0 fp32, 1 bf16, 2 fp16
  bool torgb = false;              // kernel variant with the ToRGB tail in its epilogue
  bool wide = false;               // sepconv_wide_kernel: 128 pixels x 256 channels, 512 threads, specialised wave groups
  int a_stride = 0;
  int lgGH = 3, lgGW = 4, lgIMGS = 0;
  int sy = 8, sx = 16, off = 0, lgRS = 1;
  int tiles_x = 1, tiles_y = 1, nchunks = 1;
  int off_a = 0, off_b = 0, off_v = 0, off_rgb = 0, off_w = 0, b_stride = 0;
  size_t lds_bytes = 0;
  int npix_in = 0;
};

inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Tuning knobs read once from the environment (experiments only; defaults are the shipped choice).
struct Tuning {
  int force_single_b = 0;      // MIGAN_SINGLE_B=1: never double-buffer the 1x1 weight tile
  int gemm = 2;                // MIGAN_GEMM=f32|bf16x3|f16x2: 0 exact fp32 MFMA; 1 error-compensated bf16 MFMA (6 products of
                               // 3-way bf16 splits); 2 (default) error-compensated fp16 MFMA (3 products of scaled 2-way
                               // fp16 splits); all accumulate in fp32 and have the same end-to-end error
  int wide = 3;                // MIGAN_WIDE=0..3: 8-wave 128 x 256 tiles for plain layers with Cout % 256 == 0 (f16x2 / f16 GEMM): 1 = round-1 form
                               // (waves 0-3 depthwise + half the MFMAs), 2 = waves 4-7 run all the MFMAs (-3..6 % per layer), 3 (default) = 2 +
                               // LDS-DMA staging of the input tile and the weight planes where the storage is fp32 (another -5..10 %):
                               // profiles/r03_wide_kernel.md
  int wide_up = 1;             // FIR-up layers with Cout % 256 == 0 on the 256-column tile as well (sepconv_wide_kernel<..., UP>; needs wide == 3)
  int nt256 = 1;               // MIGAN_NT256=0|1: 64-pixel x 256-channel tiles for the 256-channel layer that feeds ToRGB (fuses it)
  int persist_min = 8192;      // MIGAN_PERSIST_MIN: launches with at least this many tiles run persistent workgroups
  int persist_grid = 512;      // MIGAN_PERSIST_GRID: ... that many (2 per CU on MI355X), each walking its share of tiles
  int kc16 = 0;                // MIGAN_KC16 bit mask: 16-channel K chunks for the 64-output-channel main-geometry layers (f16x2 GEMM):
                               // 1 plain / ToRGB layers, 2 fused-FromRGB layer, 4 FIR-up layers
  int kc16_minw = 3;           // MIGAN_KC16_MINW=2|3|4: workgroups per CU those kernels are built for
  int w3 = 3;                  // bit mask like kc16: the same layers on 32-channel chunks at 3 workgroups per CU (one tile per workgroup).
                               // Default: the fused-FromRGB layer (encoder.b512.conv1: 1.19 ms as persistent tiles at 2 per CU, 0.98 ms at 3
                               // per CU, 164 VGPRs) and the plain / ToRGB layers with Cin = 64, whose two K chunks unroll at compile time
                               // (135 VGPRs instead of 184; synthesis.b512.conv2 1.34 -> 1.07 ms).  The FIR-up tile (4 chunks) needs 60 bytes
                               // of scratch at that budget and loses 10 %: profiles/r02_w3_and_persistence_sweep.txt
  int small = 1;               // launches of at most small_max_wgs small tiles use them (default GEMM variant of the storage format)
  int small_max_wgs = 512;     // (MIGAN_GEOMETRIES_SMALL; single-image latency and the <= 16x16 layers)
  int small_kc = 64;           // K chunk of those tiles: 32 or 64 channels (batch 1: 0.85 ms with 32, 0.81 ms with 64)
  int small_up32 = 1;          // FIR-up layers: try the 32-row tile before the 64-row one
  int small_dwfir = 1;         // dwfir_kernel: small launches walk fewer channel chunks per workgroup (more workgroups)
  int small_ksplit = 1;        // single-image forwards: 32 x 32 tiles whose four waves split the K steps (a quarter of the weight panel per workgroup)
  int streams = 2;             // MIGAN_STREAMS=1|2: default of migan_set_streams
  int stagger = -1;            // MIGAN_STAGGER: launch index of the first half after which the second half starts (-1: plan default)
  int debug_split = 0;         // diagnostics: keep the two-sub-batch execution in keep-intermediates mode
  int stagger_pct = 15;        // MIGAN_STAGGER_PCT: the next sub-batch starts after this share of a forward's launches (round 5 sweep, two runs each:
                               // 5 .. 18 % 3680 - 3693 images/s, 22 % (rounds 2 - 4) 3655 - 3663, 30 % 3655, 40 % 3638)
  int pipe = 15;               // MIGAN_PIPE bit mask: software-pipelined persistent kernels (sepconv_pipe_kernel; fp32 storage, f16x2 GEMM) for
                               // 1 plain (+ fused ToRGB) layers, 2 the fused-FromRGB layer, 4 FIR-up layers, 8 down=2 layers as one fused launch
                               // (sepconv_pipedown_kernel) -- wherever an instantiation exists
                               // (the 512 x 512 layers of migan-512: -5 / -10 / -9 % per layer, profiles/r04_pipe_layers.txt)
  int pipe_grid = 256;         // MIGAN_PIPE_GRID: persistent workgroups of those launches (one 12- or 16-wave workgroup per CU on MI355X)
  int pipe_na = 4;             // waves of the depthwise group of those workgroups (4 or 8), for the layers in pipe_na8 the other value
  int pipe_na8 = 9;            // bit mask like `pipe` (bits 1, 2, 4): layers whose depthwise group has 8 waves whatever pipe_na says (default: plain layers
                               // run 8 + 8 waves, fused-FromRGB and FIR-up layers 4 + 8: profiles/r04_pipe_layers.txt)
  int pipe_dna = 12;           // depthwise + FIR waves of the fused down=2 kernel: 4 or 8 (+ 8 GEMM / epilogue waves) or 12 (+ 4)
  int pipe_min_tiles = 256;    // launches with fewer tiles keep the one-tile-per-workgroup kernels
  int w2 = 2;                  // MIGAN_W2=0|1|2: the 256 / 512-channel plain layers on persistent 256-pixel x 256-channel tiles (sepconv_wide2_kernel, round 5)
                               // where a launch has at least w2_min_tiles of them (fp32 storage, f16x2 GEMM, whole 16 x 16 tiles)
  int w2_pw = 1;               // ... and the pointwise GEMM of the un-fused down=2 layers (Cout = 512) on the same tile
  int w2_min_tiles = 256;      // (one tile per CU; below that the 128-pixel one-tile kernel spreads the launch over more CUs)
  int pipe_min_batch = 1;      // smallest batch that takes them (1: a single-image forward runs them on its 512x512 / 256x256 layers too -- 2048 / 512 tiles;
                               // latency_b1 0.70 -> 0.66 ms; its other layers run the latency tiles, so it is not bit-identical to a batched forward either way)
};
inline Tuning& tuning() {
  static Tuning t = [] {
    Tuning v;
    if (const char* e = std::getenv("MIGAN_SINGLE_B")) v.force_single_b = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIGAN_GEMM")) v.gemm = std::string(e) == "f32" ? 0 : (std::string(e) == "bf16x3" ? 1 : 2);
    if (const char* e = std::getenv("MIGAN_WIDE")) v.wide = std::atoi(e);
    if (const char* e = std::getenv("MIGAN_NT256")) v.nt256 = std::atoi(e) != 0;
    if (const char* e = std::getenv("MIGAN_PERSIST_MIN")) v.persist_min = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("MIGAN_PERSIST_GRID")) v.persist_grid = std::max(8, std::atoi(e) / 8 * 8);   // multiple of 8: one share per XCD
    if (const char* e = std::getenv("MIGAN_KC16")) v.kc16 = std::atoi(e);
    if (const char* e = std::getenv("MIGAN_KC16_MINW")) v.kc16_minw = std::min(4, std::max(2, std::atoi(e)));
    if (const char* e = std::getenv("MIGAN_W3")) v.w3 = std::atoi(e);
    if (const char* e = std::getenv("MIGAN_STREAMS")) v.streams = std::min(4, std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("MIGAN_STAGGER")) v.stagger = std::atoi(e);
    if (const char* e = std::getenv("MIGAN_PIPE")) v.pipe = std::atoi(e);
    if (const char* e = std::getenv("MIGAN_W2")) v.w2 = std::min(2, std::max(0, std::atoi(e)));
    if (const char* e = std::getenv("MIGAN_PIPE_GRID")) v.pipe_grid = std::max(8, std::atoi(e) / 8 * 8);
    return v;
  }();
  return t;
}

// Tile geometry for a layer whose GEMM runs on an h_in x w_in pixel grid.  Square power-of-two sizes (the reference's
// fixed resolutions) get the tuned geometries; any other size (migan_forward_hw) gets 8x16 tiles with ragged edges.
// the small-launch tiles are instantiated for the default GEMM variant of each storage format (migan_k_slice.inc)
inline bool has_small_tiles(int gemmv, int stv) { return stv == 0 ? gemmv == 2 : gemmv == 3; }

// small: the 32-row variant of a plain / pointwise layer for launches of few workgroups (MIGAN_GEOMETRIES_SMALL): 4x8-pixel tiles, or two
// 4x4 images per tile; never wide, never with a fused ToRGB tail.
inline Geo choose_geo(int mode, int cin, int cout, int h_in, int w_in, bool fromrgb, bool with_torgb = false, int gemmv = -1, int stv = 0,
                      int small = 0) {   // 0 regular | 1 small | 2 (FIR-up only) the 32-row tile | 3 the 32 x 32 K-split tile
  Geo g;
  g.gemmv = gemmv < 0 ? tuning().gemm : gemmv;
  g.stv = stv;
  MIGAN_CHECK(g.gemmv >= 0 && g.gemmv <= 3, MIGAN_EINVAL, "unknown GEMM variant");
  MIGAN_CHECK(stv == 0 ? g.gemmv <= 2 : g.gemmv >= 2, MIGAN_EINVAL,
              "fp32 activation storage runs the f32 / bf16x3 / f16x2 GEMM variants, 16-bit storage the f16x2 / f16 variants");
  g.mode = mode;
  g.fromrgb = fromrgb;
  MIGAN_CHECK(cin % 32 == 0 && cout % 64 == 0, MIGAN_EINVAL,
              "channel counts must be multiples of 32 (in) / 64 (out)");
  MIGAN_CHECK(h_in >= 1 && w_in >= 1, MIGAN_EINVAL, "empty image");
  const bool sq2 = h_in == w_in && (h_in & (h_in - 1)) == 0;      // square power of two
  const bool full = h_in % 8 == 0 && w_in % 16 == 0;               // whole 8x16 tiles
  g.NT = (cout % 128 == 0) ? 128 : 64;
  int GH, GW, IMGS;
  if (mode == MODE_NORMAL) {
    g.MT = 128; g.KC = 32;
    if (small) {
      g.MT = 32; g.NT = 128; GH = 4;
      if (sq2 && h_in == 4) { GW = 4; IMGS = 2; } else { GW = 8; IMGS = 1; }
    } else if (full && cout % 256 == 0 && !fromrgb && g.gemmv >= 2 && tuning().wide) {
      // wide layers: one 8-wave workgroup owns 256 output channels of 8x16 pixels; half of its waves run the
      // depthwise stage of the next K chunk while the other half keeps the matrix cores busy (sepconv_wide_kernel)
      g.wide = true; g.NT = 256; GH = 8; GW = 16; IMGS = 1;
    } else if (full && cout == 256 && with_torgb && tuning().nt256) {
      // 256-channel layer followed by ToRGB: one workgroup owns all 256 output channels of 4x16 pixels so the
      // ToRGB tail fuses into its epilogue (saves the feature re-read of torgb_kernel).  Measured on its own the
      // 64 x 256 tile is ~7 % slower than 128 x 128 (2-row depthwise strips, twice the weight-tile traffic), so
      // it is used only where it removes a launch.
      g.MT = 64; g.NT = 256; GH = 4; GW = 16; IMGS = 1;
    } else if (sq2 && h_in == 8) { GH = 8; GW = 8; IMGS = 2; }
    else if (sq2 && h_in == 4) { GH = 4; GW = 4; IMGS = 8; }
    else { GH = 8; GW = 16; IMGS = 1; }
    g.sy = GH; g.sx = GW; g.off = 0;
    g.tiles_y = cdiv(h_in, GH); g.tiles_x = cdiv(w_in, GW);
  } else if (mode == MODE_PW) {
    // pointwise GEMM at h_in x w_in (second half of a down=2 layer; its input is dwfir_kernel's output)
    MIGAN_CHECK(!fromrgb, MIGAN_EINVAL, "fromrgb is only fused into plain layers");
    g.MT = 128; g.KC = 32;
    if (small) {
      g.MT = 32; g.NT = 128; GH = 4;
      if (sq2 && h_in == 4) { GW = 4; IMGS = 2; } else { GW = 8; IMGS = 1; }
    }
    else if (sq2 && h_in == 8) { GH = 8; GW = 8; IMGS = 2; }
    else if (sq2 && h_in == 4) { GH = 4; GW = 4; IMGS = 8; }
    else { GH = 8; GW = 16; IMGS = 1; }
    g.sy = GH; g.sx = GW; g.off = 0;
    g.tiles_y = cdiv(h_in, GH); g.tiles_x = cdiv(w_in, GW);
  } else {
    MIGAN_CHECK(!fromrgb, MIGAN_EINVAL, "fromrgb is only fused into plain layers");
    g.MT = 128; g.KC = 32;
    if (small >= 2) { g.MT = 32; g.NT = 128; GH = 4; GW = 8; IMGS = 1; }   // 4x8 grid of GEMM pixels, 2x6 interior
    else if (small) { g.MT = 64; g.NT = 128; GH = 8; GW = 8; IMGS = 1; }   // 8x8 grid, 6x6 interior
    else if (sq2 && h_in < 8) { GH = 8; GW = 8; IMGS = 2; }
    else if (cout % 256 == 0 && g.gemmv == 2 && stv == 0 && tuning().wide == 3 && tuning().wide_up) {
      // FIR-up with 256-column tiles (sepconv_wide_kernel<..., UP>): the depthwise stage and the input tile once per 256 output channels
      g.wide = true; g.NT = 256; GH = 8; GW = 16; IMGS = 1;
    }
    else { GH = 8; GW = 16; IMGS = 1; }
    g.sy = GH - 2; g.sx = GW - 2; g.off = 1;     // 1-pixel halo of GEMM outputs is recomputed per tile
    g.tiles_y = cdiv(h_in, g.sy); g.tiles_x = cdiv(w_in, g.sx);
  }
  g.nchunks = cout / g.NT;
  g.lgGH = ilog2(GH); g.lgGW = ilog2(GW); g.lgIMGS = ilog2(IMGS);
  g.MINW = 2;                                    // 2 workgroups per CU
  // 64-output-channel layers on main tiles (the 512x512 layers): optionally 16-channel K chunks at 3-4 workgroups per CU
  if (g.gemmv == 2 && g.NT == 64 && g.MT == 128 && !g.wide && IMGS == 1 && GW == 16 && GH == 8 && mode != MODE_PW) {
    const int bit = fromrgb ? 2 : (mode == MODE_UP ? 4 : 1);
    if (tuning().kc16 & bit) { g.KC = 16; g.MINW = tuning().kc16_minw; }
  }
  MIGAN_CHECK(IMGS * GH * GW == g.MT, MIGAN_EINVAL, "internal: tile geometry does not fill the GEMM tile");
  if (small) {
    MIGAN_CHECK(has_small_tiles(g.gemmv, stv) && cout % 128 == 0 && !fromrgb && (mode == MODE_UP || (h_in % 4 == 0 && w_in % GW == 0)), MIGAN_EINVAL,
                "internal: no small-launch variant of this layer");
    if (tuning().small_kc == 64 && cin % 64 == 0) g.KC = 64;
    if (small == 3) {                              // 32 x 32 tiles, the waves split the K steps of a 64-channel chunk
      MIGAN_CHECK(g.KC == 64 && g.MT == 32 && cout % 32 == 0, MIGAN_EINVAL, "internal: no K-split tile for this layer");
      g.NT = 32; g.nchunks = cout / 32;
    }
  }
  const int QC = g.KC / 4;
  const int segh = g.MT >= 128 ? 4 : 2;          // output rows per depthwise strip (the kernel's SEGH)
  const int rs = (g.MT == 32 || (g.MT == 64 && small)) ? GH / segh : (GH / 4 > 0 ? GH / 4 : 1);   // (RS * SEGH = GH)
  g.lgRS = ilog2(rs);
  const int halo = (mode == MODE_PW) ? 0 : 1;
  g.npix_in = IMGS * (GH + 2 * halo) * (GW + 2 * halo);
  const int items = cdiv(g.npix_in * QC, kThreads);
  // compile-time tile geometry: plain / pointwise tiles must be whole (no bounds checks in that epilogue), FIR-up tiles
  // are ragged by construction
  g.maing = (IMGS == 1 && GW == 16 && GH == (g.MT == 64 ? 4 : 8)) && (full || mode == MODE_UP);
  if (g.maing && g.gemmv >= 2 && g.NT == 64 && g.MT == 128 && g.KC == 32 && !g.wide && mode != MODE_PW) {
    const int bit = fromrgb ? 2 : (mode == MODE_UP ? 4 : 1);
    // (the plain 3-workgroup tiles are compiled for exactly two K chunks: Cin == 64, the layers they were measured on)
    if ((tuning().w3 & bit) && (bit != 1 || cin == 64)) g.MINW = 3;
  }
  if (g.MT == 64 && g.NT == 128 && mode == MODE_NORMAL) g.MINW = 3;   // 32 accumulator registers per lane: three workgroups per CU
  if (small) {   // prefetch items per thread of the MIGAN_GEOMETRIES_SMALL instantiations: [K chunk 32 | 64][plain, FIR-up, pointwise]
    static const int ni_small[2][4] = {{3, 4, 1, 2}, {5, 7, 2, 4}};      // (last column: FIR-up on the 32-row tile)
    g.NI = ni_small[g.KC == 64][mode == MODE_PW ? 2 : (mode == MODE_UP ? (g.MT == 32 ? 3 : 1) : 0)];
  }
  else if (mode == MODE_PW || g.MT == 64) g.NI = 4;
  else if (g.KC == 16) g.NI = 3;
  else g.NI = g.maing ? 6 : 9;
  MIGAN_CHECK(items <= g.NI, MIGAN_EINVAL, "internal: input tile too large");
  const int AS = g.KC + 4, GS = g.NT + 4;
  // operand tiles in floats: fp32 rows of pitch KC+4, or three unpadded (XOR-swizzled) bf16 planes
  const int npl = g.gemmv == 3 ? 1 : (g.gemmv == 2 ? 2 : 3);
  const int asz = g.gemmv ? npl * g.MT * (g.KC * 2) / 4 : g.MT * AS;
  const int bsz = g.gemmv ? npl * g.NT * (g.KC * 2) / 4 : g.NT * AS;
  const int gs = g.MT * (GS + 4) * (g.NT == 32 ? 4 : 1);   // accumulator tile (row pitch NT+4, NT+8 with the fused ToRGB tail); the ToRGB partial
                                                         // sums reuse its slots; K-split tiles: one partial tile per wave
  const size_t limit = (size_t)(160 * 1024 / g.MINW);
  if (mode == MODE_PW) {
    // A and B operands double buffered: one barrier per K chunk
    g.off_a = 0; g.off_v = g.off_rgb = g.off_w = 0;
    const size_t dbl = (size_t)std::max(2 * asz + 2 * bsz, gs) * sizeof(float);
    if (dbl <= limit && !tuning().force_single_b) { g.a_stride = asz; g.off_b = 2 * asz; g.b_stride = bsz; g.lds_bytes = dbl; }
    else { g.a_stride = 0; g.off_b = asz; g.b_stride = 0; g.lds_bytes = (size_t)std::max(asz + bsz, gs) * sizeof(float); }
  } else {
    int o = g.npix_in * g.KC;
    g.off_a = o; o += asz;
    g.off_v = o;
    g.off_rgb = o; if (fromrgb) o += g.npix_in * 4;
    g.off_w = o; o += g.KC * 10 + (fromrgb ? g.KC * 5 : 0);
    g.off_b = o;
    const size_t dbl = (size_t)std::max(o + 2 * bsz, gs) * sizeof(float);
    const size_t sgl = (size_t)std::max(o + bsz, gs) * sizeof(float);
    if (dbl <= limit && !tuning().force_single_b) { g.b_stride = bsz; g.lds_bytes = dbl; }   // double-buffered 1x1 weights: 2 barriers per K chunk
    else { g.b_stride = 0; g.lds_bytes = sgl; }
  }
  if (g.wide) {
    // sepconv_wide_kernel carves its own LDS: 2 x (input tile + taps + A planes + B planes), aliased by the result tile
    const int in_sz = 10 * 18 * g.KC, w_sz = g.KC * 10, a_sz = npl * g.MT * (g.KC * 2) / 4, b_sz = npl * g.NT * (g.KC * 2) / 4;
    g.NI = 3; g.b_stride = b_sz; g.a_stride = a_sz;
    g.lds_bytes = (size_t)std::max(2 * (in_sz + w_sz + a_sz + b_sz), g.MT * (g.NT + 4)) * sizeof(float);
  }
  MIGAN_CHECK(g.lds_bytes <= 160 * 1024, MIGAN_EINVAL, "internal: LDS tile exceeds 160 KiB");
  return g;
}

// one slice per (GEMM variant, storage format): migan_k_slice.inc
KernelSlice slice_g0s0();
KernelSlice slice_g1s0();
KernelSlice slice_g2s0();
KernelSlice slice_g2s1();
KernelSlice slice_g2s2();
KernelSlice slice_g3s1();
KernelSlice slice_g3s2();
inline const std::vector<KernelEntry>& kernel_table() {
  static const std::vector<KernelEntry> t = [] {
    std::vector<KernelEntry> v;
    for (const KernelSlice& sl : {slice_g0s0(), slice_g1s0(), slice_g2s0(), slice_g2s1(), slice_g2s2(), slice_g3s1(), slice_g3s2()})
      v.insert(v.end(), sl.entries, sl.entries + sl.n);
    return v;
  }();
  return t;
}

// sepconv_wide_kernel<TORGB, STV, X1, BALL, DMA>: x1 = GEMM variant "f16" (one fp16 piece per operand; 16-bit storage only);
// ball = waves 4-7 run all the MFMAs (tuning().wide >= 2); dma = LDS-DMA staging (tuning().wide == 3, fp32 storage)
#define MIGAN_WIDE_ROW(T, X, B) {sepconv_wide_kernel<T, 0, X, B>, sepconv_wide_kernel<T, 1, X, B>, sepconv_wide_kernel<T, 2, X, B>}
#define MIGAN_WIDE_NAMES(T, X, B) {"migan::sepconv_wide_kernel<" #T ", 0, " #X ", " #B ", false, false>", "migan::sepconv_wide_kernel<" #T ", 1, " #X ", " #B ", false, false>", \
                                   "migan::sepconv_wide_kernel<" #T ", 2, " #X ", " #B ", false, false>"}
inline bool wide_ball() { return tuning().wide >= 2; }
inline bool wide_dma(int stv, bool x1) { return tuning().wide == 3 && stv == 0 && !x1; }
inline SepKernelFn wide_fn(bool torgb, int stv, bool x1 = false, bool ball = false, bool dma = false) {
  static const SepKernelFn f[2][2][2][3] = {
      {{MIGAN_WIDE_ROW(false, false, false), MIGAN_WIDE_ROW(true, false, false)}, {MIGAN_WIDE_ROW(false, true, false), MIGAN_WIDE_ROW(true, true, false)}},
      {{MIGAN_WIDE_ROW(false, false, true), MIGAN_WIDE_ROW(true, false, true)}, {MIGAN_WIDE_ROW(false, true, true), MIGAN_WIDE_ROW(true, true, true)}}};
  if (x1 && stv == 0) return nullptr;       // the "f16" GEMM variant exists for 16-bit storage only
  if (dma) return torgb ? sepconv_wide_kernel<true, 0, false, true, true> : sepconv_wide_kernel<false, 0, false, true, true>;
  return f[ball ? 1 : 0][x1 ? 1 : 0][torgb ? 1 : 0][stv];
}
inline SepKernelFn wide_up_fn() { return sepconv_wide_kernel<false, 0, false, true, true, true>; }
constexpr const char* kWideUpName = "migan::sepconv_wide_kernel<false, 0, false, true, true, true>";
inline const char* wide_name(const Geo& g) {
  if (g.mode == MODE_UP) return kWideUpName;
  static const char* n[2][2][2][3] = {
      {{MIGAN_WIDE_NAMES(false, false, false), MIGAN_WIDE_NAMES(true, false, false)}, {MIGAN_WIDE_NAMES(false, true, false), MIGAN_WIDE_NAMES(true, true, false)}},
      {{MIGAN_WIDE_NAMES(false, false, true), MIGAN_WIDE_NAMES(true, false, true)}, {MIGAN_WIDE_NAMES(false, true, true), MIGAN_WIDE_NAMES(true, true, true)}}};
  if (wide_dma(g.stv, g.gemmv == 3))
    return g.torgb ? "migan::sepconv_wide_kernel<true, 0, false, true, true, false>" : "migan::sepconv_wide_kernel<false, 0, false, true, true, false>";
  return n[wide_ball() ? 1 : 0][g.gemmv == 3 ? 1 : 0][g.torgb ? 1 : 0][g.stv];
}
// symbol of the fused-SeparableConv2d kernel this thread launched last (migan_last_kernel: tests ask which form ran)
inline const char*& last_kernel_ref() {
  thread_local const char* n = "";
  return n;
}
// sepconv_pipe_kernel (migan_pipe.hpp): the software-pipelined persistent form of a layer, where an instantiation exists.  Chosen per
// launch from the batch: every launch of two or more images of a given layer takes the same decision (tiles >= pipe_min_tiles holds
// from batch 2 on for the layers that have an instantiation), so an image is bit-identical whatever batch >= 2 it is in.
PipeSlice pipe_slice();
inline bool PipeResident(const PipeEntry& e) { return (e.cin / 32) * (2 * e.NT * 64) <= 32 * 1024; }    // all weight planes of a column tile stay in LDS
inline const PipeEntry* pick_pipe(const Geo& g, int cin, int cout, int batch, bool fused_rgb, bool u8, bool has_skip = false) {
  const int bit = g.fromrgb ? 2 : (g.mode == MODE_UP ? 4 : 1);
  // the plain epilogue of the pipelined kernels has no skip add (no layer of the generator needs one there: its only plain + skip layer is the
  // 512-channel synthesis.b4.conv1); a plain SeparableConv2d with a skip tensor through migan_sepconv_forward keeps the one-tile kernels
  if (g.mode == MODE_NORMAL && has_skip) return nullptr;
  if (!(tuning().pipe & bit) || g.stv != 0 || g.gemmv != 2 || !g.maing || g.wide || g.MT != 128 || g.KC != 32 || g.lgIMGS != 0) return nullptr;
  if (g.mode != MODE_NORMAL && g.mode != MODE_UP) return nullptr;
  if (batch < tuning().pipe_min_batch || g.tiles_x * g.tiles_y * g.nchunks * batch < tuning().pipe_min_tiles) return nullptr;
  (void)u8;
  const PipeSlice sl = pipe_slice();
  for (int i = 0; i < sl.n; ++i) {
    const PipeEntry& e = sl.entries[i];
    // FIR-up layers run 64-column tiles here (the shared result tile of 128 columns does not fit beside the ring), and only where that is the
    // whole layer (synthesis.b512.conv1): walking a wider layer as 64-column chunks recomputes the depthwise stage per chunk and measured
    // 10-30 % slower than the 128-column one-tile kernels (rounds 4-5; removed)
    const int nt = g.mode == MODE_UP ? 64 : g.NT;
    int na = (tuning().pipe_na8 & bit) ? 8 : tuning().pipe_na;
    if ((g.mode == MODE_NORMAL && nt == 128) || g.mode == MODE_UP) na = 4;    // (these forms exist with 4 depthwise waves only: migan_pipe_table.inc)
    if (g.mode == MODE_UP && cout != nt) continue;
    if (e.mode == g.mode && e.NT == nt && e.cin == cin && e.fromrgb == g.fromrgb && e.torgb == fused_rgb && e.na == na && cout % nt == 0 &&
        cout == g.NT * g.nchunks && (PipeResident(e) ? cout == nt : true))
      return &e;
  }
  return nullptr;
}
// sepconv_pipedown_kernel: a down=2 layer as ONE launch (depthwise + FIR-down feed the 1x1 through LDS) instead of dwfir_kernel + pointwise
// GEMM with the half-resolution tensor going through HBM; where one workgroup can own all of Cout (an instantiation exists) and the
// launch has enough tiles.  Same decision for every batch >= 2 (see pick_pipe).
DownSlice pipedown_slice();
inline const DownEntry* pick_pipedown(int cin, int cout, int h_in, int w_in, int batch, int stv, int gemmv) {
  if (!(tuning().pipe & 8) || stv != 0 || gemmv != 2 || batch < tuning().pipe_min_batch) return nullptr;
  if (h_in % 8 != 0 || w_in % 32 != 0) return nullptr;                        // whole 4 x 16 low-resolution tiles
  if ((h_in / 8) * (w_in / 32) * batch < tuning().pipe_min_tiles) return nullptr;
  const int na = tuning().pipe_dna;
  const DownSlice sl = pipedown_slice();
  const DownEntry* best = nullptr;
  for (int i = 0; i < sl.n; ++i) {
    const DownEntry& e = sl.entries[i];
    if (e.NT == cout && e.cin == cin && (e.na == na || best == nullptr)) best = &e;
    if (best && best->na == na) break;
  }
  return best;
}
// sepconv_wide2_kernel (migan_wide2.hpp): persistent 16 x 16-pixel x 256-channel tiles for the plain layers the 128 x 256 wide tile serves, where
// the launch fills the chip with them.  The choice depends on the batch (tiles per image: 64 at 128 x 128, 16 at 64 x 64): the two forms sum
// a layer's K chunks in the same order with the same operand split, so an image does not depend on which one ran (tests compare them bit for bit).
SepKernelFn wide2_fn(int variant);
// tuning().w2: 0 off | 1 weight planes DMA'd as 16-channel halves (four 16 KB slots) | 2 (default) as whole 32-channel chunks (two 32 KB slots)
inline int wide2_variant() { return tuning().w2 == 1 ? 0 : 1; }
inline const char* wide2_name() { return wide2_variant() ? "migan::sepconv_wide2_kernel<1>" : "migan::sepconv_wide2_kernel<0>"; }
inline bool use_wide2(const Geo& g, int cin, int cout, int batch, bool fused_rgb, bool has_skip, bool u8) {
  if (!tuning().w2 || !g.wide || g.mode != MODE_NORMAL || g.stv != 0 || g.gemmv != 2 || g.fromrgb || fused_rgb || has_skip || u8) return false;
  const int h = g.tiles_y * 8, w = g.tiles_x * 16;               // (wide tiles are whole 8 x 16 tiles)
  if (h % 16 != 0 || w % 16 != 0 || cin % 64 != 0 || cout % 256 != 0) return false;
  return (h / 16) * (w / 16) * (cout / 256) * batch >= tuning().w2_min_tiles;
}
// ... and its pointwise form for the second half of a down=2 layer that is not fused (Cout = 512: encoder.b128 / b64 .conv2 of migan-512): the
// 128-column pointwise tiles re-read their A operand per column chunk (measured traffic 1.70x algorithmic), this one reads it once per 256
constexpr const char* kWide2PwName = "migan::sepconv_wide2_kernel<3>";
inline bool use_wide2_pw(const Geo& g, int cin, int cout, int h, int w, int batch, bool has_skip) {
  if (!tuning().w2 || !tuning().w2_pw || g.mode != MODE_PW || !g.maing || g.stv != 0 || g.gemmv != 2 || has_skip) return false;
  if (h % 16 != 0 || w % 16 != 0 || cin % 64 != 0 || cout % 256 != 0) return false;
  return (h / 16) * (w / 16) * (cout / 256) * batch >= tuning().w2_min_tiles;
}
inline const char* kernel_name(const Geo& g);
inline const KernelEntry& pick_kernel(const Geo& g) {
  for (const auto& e : kernel_table())
    if (e.mode == g.mode && e.MT == g.MT && e.NT == g.NT && e.KC == g.KC && e.fromrgb == g.fromrgb && e.NI == g.NI &&
        e.MINW == g.MINW && e.maing == g.maing && e.persist == g.persist && e.gemmv == g.gemmv && e.torgb == g.torgb && e.stv == g.stv)
      return e;
  throw Error(MIGAN_EINVAL, "internal: no kernel instantiation for this geometry");
}

inline const char* kernel_name(const Geo& g) { return g.wide ? wide_name(g) : pick_kernel(g).name; }

// persistent variants exist where they fit the register budget without spilling
inline bool has_persistent_variant(const Geo& g) {
  return g.maing && ((g.NT == 64 && g.mode != MODE_PW && (g.KC == 16 || g.MINW == 2)) || (g.mode == MODE_PW && g.NT == 128 && g.MINW == 2));
}

// ---- depthwise + FIR-down kernel (first half of down=2 layers) ----
struct DwGeo {
  int lgGH = 2, lgGW = 4, lgIMGS = 0, tiles_x = 1, tiles_y = 1, nkg = 1, kpw = 1, NI = 7;
  bool maing = true;
  int off_d = 0, off_w = 0;
  size_t lds_bytes = 0;
};
inline DwGeo choose_dwfir_geo(int c, int h_in, int w_in) {
  DwGeo g;
  MIGAN_CHECK(c % 16 == 0 && h_in >= 2 && w_in >= 2 && h_in % 2 == 0 && w_in % 2 == 0, MIGAN_EINVAL, "dwfir: bad shape");
  const int ho = h_in / 2, wo = w_in / 2;
  const bool sq2 = ho == wo && (ho & (ho - 1)) == 0;
  int GH = 4, GW, IMGS;
  if (sq2 && ho == 8) { GW = 8; IMGS = 2; }
  else if (sq2 && ho <= 4) { GW = 4; IMGS = 4; }
  else { GW = 16; IMGS = 1; }
  g.lgGH = 2; g.lgGW = ilog2(GW); g.lgIMGS = ilog2(IMGS);
  g.tiles_y = cdiv(ho, GH); g.tiles_x = cdiv(wo, GW);
  g.kpw = (c / 16) % 4 == 0 ? 4 : ((c / 16) % 2 == 0 ? 2 : 1);   // chunks walked per workgroup (software pipelined)
  g.nkg = (c / 16) / g.kpw;
  g.maing = (IMGS == 1 && GW == 16) && ho % 4 == 0 && wo % 16 == 0;
  const int npix = IMGS * (2 * GH + 4) * (2 * GW + 4);
  const int items = cdiv(npix * 4, kThreads);
  g.NI = g.maing ? 7 : 9;
  MIGAN_CHECK(items <= g.NI, MIGAN_EINVAL, "internal: dwfir input tile too large");
  g.off_d = npix * 16;
  g.off_w = g.off_d + IMGS * (2 * GH + 2) * (2 * GW + 2) * 16;
  g.lds_bytes = (size_t)(g.off_w + 160) * sizeof(float);
  return g;
}
inline unsigned dwfir_grid(const DwGeo& g, int batch) { return (unsigned)(g.tiles_x * g.tiles_y * cdiv(batch, 1 << g.lgIMGS) * g.nkg); }
typedef void (*DwFirKernelFn)(const DwFirArgs);
// stv: activation storage format 0/1/2; + 2 (3/4) when the pointwise GEMM behind it is the "f16" variant and takes its A operand
// ready-made (dwfir_variant)
inline int dwfir_variant(int stv, int gemmv) { return (gemmv == 3 && stv != 0) ? stv + 2 : stv; }
inline const char* dwfir_name(const DwGeo& g, int stv = 0) {
  static const char* n[2][5] = {{"migan::dwfir_kernel<9, false, 0>", "migan::dwfir_kernel<9, false, 1>", "migan::dwfir_kernel<9, false, 2>",
                                 "migan::dwfir_kernel<9, false, 3>", "migan::dwfir_kernel<9, false, 4>"},
                                {"migan::dwfir_kernel<7, true, 0>", "migan::dwfir_kernel<7, true, 1>", "migan::dwfir_kernel<7, true, 2>",
                                 "migan::dwfir_kernel<7, true, 3>", "migan::dwfir_kernel<7, true, 4>"}};
  return n[g.maing ? 1 : 0][stv];
}
inline DwFirKernelFn dwfir_fn(bool maing, int stv) {
  static const DwFirKernelFn f[2][5] = {{dwfir_kernel<9, false, 0>, dwfir_kernel<9, false, 1>, dwfir_kernel<9, false, 2>,
                                         dwfir_kernel<9, false, 3>, dwfir_kernel<9, false, 4>},
                                        {dwfir_kernel<7, true, 0>, dwfir_kernel<7, true, 1>, dwfir_kernel<7, true, 2>,
                                         dwfir_kernel<7, true, 3>, dwfir_kernel<7, true, 4>}};
  return f[maing ? 1 : 0][stv];
}
typedef void (*RgbKernelFn)(const RgbArgs);
inline RgbKernelFn torgb_fn(int stv) {
  static const RgbKernelFn f[3] = {torgb_kernel<0>, torgb_kernel<1>, torgb_kernel<2>};
  return f[stv];
}
inline const char* torgb_name(int stv) {
  static const char* n[3] = {"migan::torgb_kernel<0>", "migan::torgb_kernel<1>", "migan::torgb_kernel<2>"};
  return n[stv];
}

// layers with fewer than 64 channels (resolutions above 512): one plain kernel per (down / up, fromrgb)
// (the tiled kernels take Cin % 32 == 0 and Cout % 64 == 0: choose_geo)
inline bool is_narrow(int cin, int cout) { return cin % 32 != 0 || cout % 64 != 0; }
inline SepKernelFn narrow_fn(int mode, bool fromrgb) {
  if (fromrgb) return narrow_sepconv_kernel<MODE_NORMAL, true>;
  return mode == MODE_DOWN ? narrow_sepconv_kernel<MODE_DOWN, false> : (mode == MODE_UP ? narrow_sepconv_kernel<MODE_UP, false> : narrow_sepconv_kernel<MODE_NORMAL, false>);
}
inline const char* narrow_name(int mode, bool fromrgb) {
  if (fromrgb) return "migan::narrow_sepconv_kernel<0, true>";
  return mode == MODE_DOWN ? "migan::narrow_sepconv_kernel<1, false>" : (mode == MODE_UP ? "migan::narrow_sepconv_kernel<2, false>" : "migan::narrow_sepconv_kernel<0, false>");
}

// (a.B / H / W / HO / WO / CI / CO set by the caller)
inline void launch_narrow(int mode, bool fromrgb, SepArgs a, rt::stream_t stream) {
  unsigned grid;
  size_t lds = 0;
  if (mode == MODE_UP) {           // one workgroup per 16 x 16 low-resolution pixels, their 18 x 18 1x1 outputs in LDS
    a.tiles_x = cdiv(a.W, kNarrowUpTile); a.tiles_y = cdiv(a.H, kNarrowUpTile);
    grid = (unsigned)(a.tiles_x * a.tiles_y * a.B);
    lds = (size_t)(kNarrowUpTile + 2) * (kNarrowUpTile + 2) * a.CO * sizeof(float);
  } else {                         // one thread per output pixel
    grid = (unsigned)(((size_t)a.B * a.HO * a.WO + kThreads - 1) / kThreads);
  }
  rt_check(rt::launch(narrow_fn(mode, fromrgb), a, grid, kThreads, lds, stream), narrow_name(mode, fromrgb));
  last_kernel_ref() = narrow_name(mode, fromrgb);
}

// Raise the dynamic-LDS limit of every instantiation (tiles use up to 145 KiB).  The attribute is per device, so
// this runs once per device ordinal (the caller has made that device current).
inline void prepare_kernels() {
  static std::vector<char> done;
  static std::mutex mu;
  int dev = 0;
  rt_check(rt::get_device(&dev), "hipGetDevice");
  std::lock_guard<std::mutex> lock(mu);
  if (dev >= 0 && dev < (int)done.size() && done[dev]) return;
  for (const auto& e : kernel_table()) rt_check(rt::allow_dynamic_lds((const void*)e.fn, 96 * 1024), "hipFuncSetAttribute");
  {
    const PipeSlice sl = pipe_slice();
    for (int i = 0; i < sl.n; ++i) rt_check(rt::allow_dynamic_lds((const void*)sl.entries[i].fn, 160 * 1024), "hipFuncSetAttribute");
    const DownSlice ds = pipedown_slice();
    for (int i = 0; i < ds.n; ++i) rt_check(rt::allow_dynamic_lds((const void*)ds.entries[i].fn, 160 * 1024), "hipFuncSetAttribute");
  }
  for (int t = 0; t < 2; ++t)
    for (int sv = 0; sv < 3; ++sv) {
      for (int ball = 0; ball < 2; ++ball) {
        rt_check(rt::allow_dynamic_lds((const void*)wide_fn(t != 0, sv, false, ball != 0), 160 * 1024), "hipFuncSetAttribute");
        if (sv) rt_check(rt::allow_dynamic_lds((const void*)wide_fn(t != 0, sv, true, ball != 0), 160 * 1024), "hipFuncSetAttribute");
      }
      if (sv == 0) rt_check(rt::allow_dynamic_lds((const void*)wide_fn(t != 0, 0, false, true, true), 160 * 1024), "hipFuncSetAttribute");
      if (sv == 0 && t == 0) rt_check(rt::allow_dynamic_lds((const void*)wide_up_fn(), 160 * 1024), "hipFuncSetAttribute");
      if (sv == 0 && t == 0)
        for (int v : {0, 1, 3}) rt_check(rt::allow_dynamic_lds((const void*)wide2_fn(v), 160 * 1024), "hipFuncSetAttribute");
      rt_check(rt::allow_dynamic_lds((const void*)dwfir_fn(t != 0, sv), 96 * 1024), "hipFuncSetAttribute");
      if (sv) rt_check(rt::allow_dynamic_lds((const void*)dwfir_fn(t != 0, sv + 2), 96 * 1024), "hipFuncSetAttribute");
    }
  if (dev >= (int)done.size()) done.resize(dev + 1, 0);
  if (dev >= 0) done[dev] = 1;
}

// Every C ABI entry point that touches the device runs on the handle's device and leaves the caller's current
// device as it found it.
struct DeviceGuard {
  int prev = -1;
  bool restore = false;
  explicit DeviceGuard(int device) {
    rt_check(rt::get_device(&prev), "hipGetDevice");
    if (prev != device) {
      rt_check(rt::set_device(device), "hipSetDevice");
      restore = true;
    }
  }
  ~DeviceGuard() {
    if (restore) rt::set_device(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#ifdef MIGAN_PHASE_PROF
inline unsigned long long* prof_buffer() {
  static unsigned long long* buf = nullptr;
  if (!buf) rt_check(rt::prof_alloc(&buf, 16), "prof alloc");
  return buf;
}
#else
inline unsigned long long* prof_buffer() { return nullptr; }
#endif

#ifdef MIGAN_PHASE_PROF
struct ProfRow { unsigned long long v[16]; };
inline std::vector<ProfRow>& prof_layers() {
  static std::vector<ProfRow> rows;
  return rows;
}
#endif

inline void fill_geo(SepArgs& a, const Geo& g) {
  a.lgGH = g.lgGH; a.lgGW = g.lgGW; a.lgIMGS = g.lgIMGS;
  a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y; a.nchunks = g.nchunks;
  a.sy = g.sy; a.sx = g.sx; a.off = g.off; a.lgRS = g.lgRS;
  a.off_a = g.off_a; a.off_b = g.off_b; a.off_v = g.off_v; a.off_rgb = g.off_rgb; a.off_w = g.off_w;
  a.b_stride = g.b_stride;
  a.a_stride = g.a_stride;
  a.prof = prof_buffer();
}

inline unsigned tiles_of(const Geo& g, int batch) {
  return (unsigned)(g.tiles_x * g.tiles_y * cdiv(batch, 1 << g.lgIMGS) * g.nchunks);
}
// Large launches run a fixed grid of persistent workgroups (2 per CU) that walk the tiles and prefetch
// their next tile during the current epilogue; small ones keep one tile per workgroup.
inline bool use_persistent(const Geo& g, int batch, bool fused_rgb) {
  const int total = (int)tiles_of(g, batch);
  // measured on MI355X (profiles/): +2..7 % on the 512x512 layers; with the ToRGB tail fused -7 % (first tail) / -3 % (LDS-reduction tail)
  // 16-bit storage: the one-tile FIR-up kernel needs 164 VGPRs / 41 KB of LDS, i.e. three workgroups per CU, and beats its
  // persistent form (226 VGPRs, two per CU): synthesis.b512.conv1 0.81 -> 0.64 ms (profiles/r02_w3_and_persistence_sweep.txt)
  if (g.stv != 0 && g.mode == MODE_UP) return false;
  return has_persistent_variant(g) && !fused_rgb && total >= tuning().persist_min && total > tuning().persist_grid;
}
inline unsigned grid_of(const Geo& g, int batch, bool fused_rgb = false) {
  return use_persistent(g, batch, fused_rgb) ? (unsigned)tuning().persist_grid : tiles_of(g, batch);
}

// name of the kernel launch_sepconv runs for this geometry and batch
inline const char* launched_kernel_name(Geo g, int cin, int cout, int batch, bool fused_rgb, bool has_skip);

// n_decide: the batch the kernel FORM is chosen for (0 = a.B).  The per-launch timing run launches the whole batch at once but must time the
// forms the production forward -- sub-batches on staggered streams -- launches.
inline void launch_sepconv(Geo g, const SepArgs& a, rt::stream_t stream, int n_decide = 0) {
  prepare_kernels();
  const bool fused_rgb = a.trgb_w != nullptr;
  const int nd = n_decide > 0 ? n_decide : a.B;
  if (const PipeEntry* pe = pick_pipe(g, a.CI, a.CO, nd, fused_rgb, a.u8_img != nullptr, a.skip != nullptr)) {
    MIGAN_CHECK(a.wsplit != nullptr, MIGAN_EINVAL, "internal: the pipelined kernel needs the fp16 weight planes");
    SepArgs ap = a;
    ap.nchunks = a.CO / pe->NT;                          // (FIR-up: 64-column chunks, see pick_pipe)
    const unsigned tiles = (unsigned)(g.tiles_x * g.tiles_y * ap.nchunks * a.B);
    const unsigned grid = std::min(tiles, (unsigned)tuning().pipe_grid);
    rt_check(rt::launch(pe->fn, ap, grid, (unsigned)pipe_threads(pe->na), pe->lds_bytes, stream), pe->name);
    last_kernel_ref() = pe->name;
    return;
  }
  if (use_wide2(g, a.CI, a.CO, nd, fused_rgb, a.skip != nullptr, a.u8_img != nullptr)) {
    MIGAN_CHECK(a.wsplit != nullptr, MIGAN_EINVAL, "internal: the 256 x 256 tile kernel needs the fp16 weight planes");
    SepArgs aw = a;
    aw.tiles_x = a.W / 16; aw.tiles_y = a.H / 16; aw.nchunks = a.CO / 256;
    aw.sy = 16; aw.sx = 16; aw.off = 0;
    const unsigned tiles = (unsigned)(aw.tiles_x * aw.tiles_y * aw.nchunks * a.B);
    rt_check(rt::launch(wide2_fn(wide2_variant()), aw, std::min(tiles, (unsigned)tuning().pipe_grid), (unsigned)kW2Threads, (size_t)W2Lds::TOTAL, stream), wide2_name());
    last_kernel_ref() = wide2_name();
    return;
  }
  if (use_wide2_pw(g, a.CI, a.CO, a.H, a.W, nd, a.skip != nullptr)) {
    MIGAN_CHECK(a.wsplit != nullptr, MIGAN_EINVAL, "internal: the 256 x 256 tile kernel needs the fp16 weight planes");
    SepArgs aw = a;
    aw.tiles_x = a.W / 16; aw.tiles_y = a.H / 16; aw.nchunks = a.CO / 256;
    aw.sy = 16; aw.sx = 16; aw.off = 0;
    const unsigned tiles = (unsigned)(aw.tiles_x * aw.tiles_y * aw.nchunks * a.B);
    rt_check(rt::launch(wide2_fn(3), aw, std::min(tiles, (unsigned)tuning().pipe_grid), (unsigned)kW2Threads, (size_t)W2Lds::TOTAL, stream), kWide2PwName);
    last_kernel_ref() = kWide2PwName;
    return;
  }
  g.persist = use_persistent(g, a.B, fused_rgb);      // (grid size: the batch actually launched)
  g.torgb = fused_rgb;
  MIGAN_CHECK(!fused_rgb || (g.mode == MODE_NORMAL && !g.fromrgb && g.nchunks == 1), MIGAN_EINVAL,
              "ToRGB can only be fused into a plain layer whose output channels fit one column tile");
  if (g.wide) {
    MIGAN_CHECK(a.wsplit != nullptr, MIGAN_EINVAL, "internal: the wide kernel needs the fp16 weight planes");
    const SepKernelFn fn = g.mode == MODE_UP ? wide_up_fn() : wide_fn(fused_rgb, g.stv, g.gemmv == 3, wide_ball(), wide_dma(g.stv, g.gemmv == 3));
    rt_check(rt::launch(fn, a, tiles_of(g, a.B), kWideThreads, g.lds_bytes, stream), wide_name(g));
    last_kernel_ref() = wide_name(g);
    return;
  }
  const KernelEntry& k = pick_kernel(g);
  rt_check(rt::launch(k.fn, a, grid_of(g, a.B, fused_rgb), kThreads, g.lds_bytes, stream), k.name);
  last_kernel_ref() = k.name;
}

// (has_skip: the decisions launch_sepconv takes with a.skip != nullptr -- a plain or pointwise layer with a skip tensor keeps the one-tile kernels)
inline const char* launched_kernel_name(Geo g, int cin, int cout, int batch, bool fused_rgb, bool has_skip) {
  if (const PipeEntry* pe = pick_pipe(g, cin, cout, batch, fused_rgb, false, has_skip)) return pe->name;
  if (use_wide2(g, cin, cout, batch, fused_rgb, has_skip, false)) return wide2_name();
  if (use_wide2_pw(g, cin, cout, g.tiles_y * g.sy, g.tiles_x * g.sx, batch, has_skip)) return kWide2PwName;      // (maing: whole 8 x 16 tiles)
  g.persist = use_persistent(g, batch, fused_rgb);
  g.torgb = fused_rgb;
  return kernel_name(g);
}

inline void launch_pipedown(const DownEntry& e, SepArgs a, rt::stream_t stream) {
  prepare_kernels();
  MIGAN_CHECK(a.wsplit != nullptr, MIGAN_EINVAL, "internal: the fused down=2 kernel needs the fp16 weight planes");
  a.tiles_x = a.W / 32; a.tiles_y = a.H / 8; a.nchunks = 1;
  a.prof = prof_buffer();
  const unsigned tiles = (unsigned)(a.tiles_x * a.tiles_y * a.B);
  rt_check(rt::launch(e.fn, a, std::min(tiles, (unsigned)tuning().pipe_grid), (unsigned)((e.na + e.nb) * 64), e.lds_bytes, stream), e.name);
  last_kernel_ref() = e.name;
}
inline void launch_dwfir(const DwGeo& g, DwFirArgs a, rt::stream_t stream, int stv = 0) {
  prepare_kernels();
  a.lgGH = g.lgGH; a.lgGW = g.lgGW; a.lgIMGS = g.lgIMGS;
  a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y; a.nkg = g.nkg; a.kpw = g.kpw;
  a.off_d = g.off_d; a.off_w = g.off_w;
  // small launches (single-image latency, the <= 16x16 layers): fewer 16-channel chunks per workgroup, more workgroups, as long as
  // the launch stays within the small-launch threshold (same reasoning as MIGAN_GEOMETRIES_SMALL)
  unsigned grid = dwfir_grid(g, a.B);
  if (tuning().small && tuning().small_dwfir)
    while (a.kpw > 1 && grid * 2 <= (unsigned)tuning().small_max_wgs) { a.kpw /= 2; a.nkg *= 2; grid *= 2; }
  rt_check(rt::launch(dwfir_fn(g.maing, stv), a, grid, kThreads, g.lds_bytes, stream), dwfir_name(g, stv));
}

// 16-bit elements one tensor occupies in a weight-split buffer: header + planes, rounded to 16 bytes
inline size_t wsplit_elems_of(int cin, int cout) {
  return (size_t)kSplitHeader + (((size_t)3 * cin * cout + 7) & ~(size_t)7);
}
inline void launch_split(const SplitArgs& a, rt::stream_t stream) {
  if (a.f16) rt_check(rt::launch(weight_absmax_kernel, a, (unsigned)a.n, kThreads, 4 * sizeof(float), stream), "migan::weight_absmax_kernel");
  rt_check(rt::launch(split_weights_kernel, a, (unsigned)(a.n * kSplitBlocksPerTensor), kThreads, 0, stream), "migan::split_weights_kernel");
}

inline void launch_torgb(const RgbArgs& a, rt::stream_t stream, int stv = 0) {
  const size_t npix = (size_t)a.B * a.H * a.W;
  const unsigned grid = (unsigned)((npix * 16 + kThreads - 1) / kThreads);
  rt_check(rt::launch(torgb_fn(stv), a, grid, kThreads, 0, stream), torgb_name(stv));
}

// ------------------------------------------------------------------------------------------------
// state_dict schema (mirror of mi-gan_amd/schema.py; reference registration order)

enum Role { R_DW_W, R_DW_B, R_PW_W, R_RGB_W, R_RGB_B, R_FIR_DOWN, R_FIR_UP, R_FILTER_CONST, R_NOISE_CONST, R_NOISE_STRENGTH };

struct Slot {
  std::string name;
  int64_t shape[4] = {0, 0, 0, 0};
  int ndim = 0;
  bool is_buffer = false;
  Role role = R_DW_W;
  const float* ptr = nullptr;
  size_t numel() const {
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    return n;
  }
};

inline int channels_at(int res) {
  const int c = 32768 / res;
  return c < 512 ? c : 512;   // reference :222-223, :342-343
}

struct Buf {
  std::string name;
  size_t bytes_per_image = 0;
};
enum : int { BUF_NONE = -1, BUF_X = -2, BUF_Y = -3 };

struct Launch {
  std::string layer, kernel;
  mutable std::string kernel_last;   // kernel symbol actually launched by the last forward (persistent variant or not)
  bool is_rgb = false;
  bool is_dwfir = false;
  bool narrow = false;               // fewer than 64 channels on either side (resolutions above 512): narrow_sepconv_kernel<mode, fromrgb>
  int mode = MODE_NORMAL;            // (narrow launches: the SeparableConv2d's down / up)
  bool fromrgb = false;
  Geo g;
  std::vector<Geo> g_small;          // small-tile variants of g, smallest tile first: the first one whose tile count for the launch is at
                                     // most tuning().small_max_wgs runs instead of g
  DwGeo dg;
  int cin = 0, cout = 0, hin = 0, win = 0, hout = 0, wout = 0;
  int in_buf = BUF_NONE, out_buf = BUF_NONE, skip_buf = BUF_NONE, imgprev_buf = BUF_NONE, imgout_buf = BUF_NONE;
  int w_dw = -1, b_dw = -1, w_pw = -1, w_noise = -1, w_ns = -1, w_frgb = -1, b_frgb = -1, w_trgb = -1, b_trgb = -1;
  long long noise_plane_off = -1;    // arbitrary-size plans: byte offset (shared region) of this layer's [hout][wout] noise plane
  double flops = 0, mfma_flops = 0, bytes = 0;
  int wgs_batch1 = 0;
  size_t wsplit_off = 0;             // element offset of this layer's 16-bit weight planes in the shared region
};

// Launch sequence + workspace layout of one forward at H x W (H = W = resolution for the reference's fixed-size forward).
//   workspace = [ shared region: 16-bit weight planes of every 1x1 conv, noise planes of arbitrary-size plans ]
//               [ sub-batch 0: skip tensors, ping-pong activations, dwfir scratch, RGB ping-pong ] [ sub-batch 1: the same ]
struct Plan {
  int H = 0, W = 0;
  std::vector<Buf> bufs;
  std::vector<Launch> launches;
  std::vector<std::pair<std::string, int>> debug_tensors;   // layer name -> buffer id
  size_t shared_bytes = 0;
  size_t wsplit_elems = 0;
  int stagger = 0;                   // launch index of the first sub-batch after which the second one starts
};

inline size_t align256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace migan

struct migan_handle {
  int resolution = 0, device = 0;
  int stv = 0;                        // activation storage format (MIGAN_DTYPE_*)
  int gemm = 2;                       // MIGAN_GEMM_*
  int streams = 2;                    // n > 1: batches of >= 8 n images run as n staggered sub-batches on n streams
  bool static_weights = false;        // caller asserts conv2 weights are unchanged between forwards on the same workspace
  bool committed = false, debug = false;
  std::vector<migan::Slot> slots;
  migan::Plan plan;                   // H = W = resolution
  std::vector<migan::Plan> hw_plans;  // migan_forward_hw sizes seen so far
  std::vector<rt::event_t> events;
  // two-stream execution
  static constexpr int kMaxStreams = 4;
  rt::stream_t aux_stream[kMaxStreams - 1]{};
  rt::event_t ev_mid[kMaxStreams]{}, ev_join[kMaxStreams - 1]{};
  bool aux_ready = false;
  // static weights: where and when the operand planes were last written
  const void* prepared_ws = nullptr;
  rt::stream_t prepared_stream{};
  unsigned long long weight_epoch = 1, prepared_epoch = 0;

  int slot_index(const std::string& n) const {
    for (size_t i = 0; i < slots.size(); ++i)
      if (slots[i].name == n) return (int)i;
    return -1;
  }
  void add_slot(const std::string& n, std::initializer_list<int64_t> shp, bool is_buf, migan::Role role) {
    migan::Slot s;
    s.name = n;
    s.ndim = (int)shp.size();
    int i = 0;
    for (auto v : shp) s.shape[i++] = v;
    s.is_buffer = is_buf;
    s.role = role;
    slots.push_back(s);
  }
  void add_sepconv_slots(const std::string& p, int cin, int cout, int res_out, bool down, bool up, bool noise) {
    using namespace migan;
    if (noise) {
      add_slot(p + ".noise_strength", {}, false, R_NOISE_STRENGTH);
      add_slot(p + ".noise_const", {res_out, res_out}, true, R_NOISE_CONST);
    }
    add_slot(p + ".conv1.weight", {cin, 1, 3, 3}, false, R_DW_W);
    add_slot(p + ".conv1.bias", {cin}, false, R_DW_B);
    add_slot(p + ".conv2.weight", {cout, cin, 1, 1}, false, R_PW_W);
    if (down) add_slot(p + ".downsample.filter.weight", {cin, 1, 4, 4}, false, R_FIR_DOWN);
    if (up) {
      add_slot(p + ".upsample.filter_const", {1, 1, res_out, res_out}, true, R_FILTER_CONST);
      add_slot(p + ".upsample.filter.weight", {cout, 1, 4, 4}, false, R_FIR_UP);
    }
  }
  void build_schema();
  void build_plan(migan::Plan& P, int H, int W) const;
  void rebuild() {
    build_plan(plan, resolution, resolution);
    hw_plans.clear();
    prepared_epoch = 0;
  }
  migan::Plan& plan_for(int H, int W);
  // sub-batch sizes (whole 8-image groups: the 4x4 tiles hold 8 images; the last one takes the remainder)
  int split(int batch, int n[kMaxStreams]) const {
    int parts = 1;
    if (streams >= 2 && (!debug || migan::tuning().debug_split)) parts = std::min(std::min(streams, (int)kMaxStreams), batch / 8);
    if (parts < 2) { n[0] = batch; return 1; }
    const int each = (batch / parts + 7) / 8 * 8;
    int left = batch, k = 0;
    while (left > 0 && k < parts) {
      n[k] = (k == parts - 1) ? left : std::min(each, left);
      left -= n[k];
      ++k;
    }
    return k;
  }
  static size_t sub_offset(const migan::Plan& P, int id, int n) {
    size_t off = 0;
    for (int i = 0; i < id; ++i) off += migan::align256(P.bufs[i].bytes_per_image * (size_t)n);
    return off;
  }
  static size_t sub_bytes(const migan::Plan& P, int n) { return sub_offset(P, (int)P.bufs.size(), n); }
  size_t workspace_bytes(const migan::Plan& P, int batch) const {
    int n[kMaxStreams];
    const int parts = split(batch, n);
    size_t total = migan::align256(P.shared_bytes);
    for (int k = 0; k < parts; ++k) total += sub_bytes(P, n[k]);
    return total;
  }
  void ensure_aux();
  void run_range(const migan::Plan& P, const float* x, float* y, int n, char* sub_ws, char* shared, rt::stream_t stream,
                 bool timed, int mid_after, int part, const migan_io_u8* u8, int n_geo = 0);
  // parts (migan_forward_parts): sub-batch k writes its images to parts->y[k] and, for k >= 1, runs on the CALLER's stream parts->streams[k - 1]
  // instead of the handle's own; the streams are not joined (the caller orders whatever follows behind each of them)
  struct Parts {
    void* const* y = nullptr;
    void* const* streams = nullptr;
    int n_streams = 0;
    int* out_n = nullptr;      // [kMaxStreams] images per sub-batch
    int* out_parts = nullptr;
  };
  void forward(migan::Plan& P, const float* x, float* y, int batch, void* ws, size_t ws_bytes, rt::stream_t stream, float* ms,
               int n_ms, const migan_io_u8* u8 = nullptr, const Parts* parts_io = nullptr);
};

namespace migan {
inline std::string bname(const char* part, int res) { return std::string(part) + ".b" + std::to_string(res); }
}

inline void migan_handle::build_schema() {
  using namespace migan;
  slots.clear();
  const int R = resolution;
  // Generator registers synthesis before encoder (reference :359-360)
  for (int res = 4; res <= R; res *= 2) {
    const int c = channels_at(res);
    const std::string b = bname("synthesis", res);
    if (res == 4) {
      add_sepconv_slots(b + ".conv1", c, c, 4, false, false, false);
      add_sepconv_slots(b + ".conv2", c, c, 4, false, false, false);
    } else {
      add_sepconv_slots(b + ".conv1", channels_at(res / 2), c, res, false, true, true);
      add_sepconv_slots(b + ".conv2", c, c, res, false, false, true);
    }
    add_slot(b + ".torgb.weight", {3, c, 1, 1}, false, R_RGB_W);
    add_slot(b + ".torgb.bias", {3}, false, R_RGB_B);
    if (res > 4) {
      add_slot(b + ".upsample.filter_const", {1, 1, res, res}, true, R_FILTER_CONST);
      add_slot(b + ".upsample.filter.weight", {3, 1, 4, 4}, false, R_FIR_UP);
    }
  }
  for (int res = R; res >= 4; res /= 2) {
    const int c = channels_at(res);
    const std::string b = bname("encoder", res);
    if (res == R) {
      add_slot(b + ".fromrgb.weight", {c, 4, 1, 1}, false, R_RGB_W);
      add_slot(b + ".fromrgb.bias", {c}, false, R_RGB_B);
    }
    add_sepconv_slots(b + ".conv1", c, c, res, false, false, false);
    if (res > 4) add_sepconv_slots(b + ".conv2", c, channels_at(res / 2), res / 2, true, false, false);
    else add_sepconv_slots(b + ".conv2", c, c, 4, false, false, false);
  }
}

// Block "b<res>" of the reference runs at res x res; fed an H x W input (H, W multiples of resolution / 4) the same block
// runs at (H * res / resolution) x (W * res / resolution): the network is fully convolutional except for its two fixed-size
// constants (reference README.md:87; noise_const :149, filter_const :85), which migan_forward_hw crops / tiles.
inline void migan_handle::build_plan(migan::Plan& P, int H, int W) const {
  using namespace migan;
  P = Plan();
  P.H = H; P.W = W;
  const int R = resolution;
  const bool fixed = (H == R && W == R);
  auto hs = [&](int res) { return (int)((long long)H * res / R); };
  auto ws = [&](int res) { return (int)((long long)W * res / R); };
  const size_t esz = stv == 0 ? 4 : 2;
  auto add_buf = [&](const std::string& n, size_t bytes_per_image) {
    P.bufs.push_back({n, bytes_per_image});
    return (int)P.bufs.size() - 1;
  };
  size_t max_act = 0;
  for (int res = 4; res <= R; res *= 2) max_act = std::max(max_act, (size_t)hs(res) * ws(res) * channels_at(res));
  std::vector<int> feat(16, BUF_NONE);
  for (int res = R; res >= 4; res /= 2) feat[ilog2(res)] = add_buf("feat" + std::to_string(res), (size_t)hs(res) * ws(res) * channels_at(res) * esz);
  size_t max_dwt = 16;
  for (int res = R; res > 4; res /= 2) max_dwt = std::max(max_dwt, (size_t)hs(res / 2) * ws(res / 2) * channels_at(res));
  const int DWT = add_buf("dwfir_tmp", max_dwt * sizeof(float));      // intermediate of one SeparableConv2d: always fp32
  int P0 = BUF_NONE, P1 = BUF_NONE, I0 = BUF_NONE, I1 = BUF_NONE;
  if (!debug) {
    P0 = add_buf("act0", max_act * esz);
    P1 = add_buf("act1", max_act * esz);
    I0 = add_buf("img0", (size_t)3 * hs(R / 2) * ws(R / 2) * sizeof(float));
    I1 = add_buf("img1", (size_t)3 * hs(R / 2) * ws(R / 2) * sizeof(float));
  }
  size_t noise_bytes = 0;
  auto out_for = [&](const std::string& layer, int res, int c, int pingpong) -> int {
    if (!debug) return pingpong;
    return add_buf(layer, (size_t)hs(res) * ws(res) * c * esz);
  };
  auto add_sep = [&](const std::string& layer, int mode, int cin, int cout, int res_in, int res_out, bool fromrgb,
                     bool noise, int in_buf, int out_buf, int skip_buf, bool with_torgb = false) -> Launch& {
    Launch L;
    L.layer = layer;
    L.hin = hs(res_in); L.win = ws(res_in); L.hout = hs(res_out); L.wout = ws(res_out);
    L.narrow = is_narrow(cin, cout);
    L.mode = mode; L.fromrgb = fromrgb;
    if (L.narrow) {
      MIGAN_CHECK(stv == 0, MIGAN_EUNSUPPORTED, "resolutions above 512 (layers with fewer than 64 channels) run with fp32 activation storage only");
      MIGAN_CHECK(cin % 4 == 0 && cout % 4 == 0 && cin <= 64 && cout <= 64, MIGAN_EINVAL, "internal: narrow layer with unexpected channel counts");
      L.g.mode = mode; L.g.nchunks = 1;                // (no tile geometry: one thread per output pixel)
      L.kernel = narrow_name(mode, fromrgb);
    } else {
      L.g = choose_geo(mode, cin, cout, L.hin, L.win, fromrgb, with_torgb, gemm, stv);
      L.g.torgb = with_torgb && L.g.nchunks == 1;        // one workgroup owns all channels of its pixels: ToRGB fuses
      L.kernel = kernel_name(L.g);
    }
    if (!L.narrow && tuning().small && has_small_tiles(gemm, stv) && !fromrgb && !L.g.torgb && cout % 128 == 0 &&
        (mode == MODE_UP || (L.hin % 4 == 0 && L.win % (L.hin == 4 && L.win == 4 ? 4 : 8) == 0))) {
      if (tuning().small_ksplit && tuning().small_kc == 64 && cin % 64 == 0)
        L.g_small.push_back(choose_geo(mode, cin, cout, L.hin, L.win, false, false, gemm, stv, 3));
      if (mode == MODE_UP && tuning().small_up32) L.g_small.push_back(choose_geo(mode, cin, cout, L.hin, L.win, false, false, gemm, stv, 2));
      L.g_small.push_back(choose_geo(mode, cin, cout, L.hin, L.win, false, false, gemm, stv, 1));
    }
    L.cin = cin; L.cout = cout;
    L.in_buf = in_buf; L.out_buf = out_buf; L.skip_buf = skip_buf;
    L.w_dw = slot_index(layer + ".conv1.weight");
    L.b_dw = slot_index(layer + ".conv1.bias");
    L.w_pw = slot_index(layer + ".conv2.weight");
    L.wsplit_off = P.wsplit_elems + kSplitHeader;      // planes start after the 16-byte header
    P.wsplit_elems += wsplit_elems_of(cin, cout);
    if (noise) {
      L.w_noise = slot_index(layer + ".noise_const");
      L.w_ns = slot_index(layer + ".noise_strength");
      if (!fixed) {
        L.noise_plane_off = (long long)noise_bytes;
        noise_bytes += align256((size_t)L.hout * L.wout * sizeof(float));
      }
    }
    const double pin = (double)L.hin * L.win, pout = (double)L.hout * L.wout;
    const double pgemm = (mode == MODE_UP) ? pin : pout;
    const double e = (double)esz;
    L.mfma_flops = 2.0 * cin * cout * pgemm;
    L.flops = L.mfma_flops + 2.0 * 9 * cin * pin;
    if (mode == MODE_PW) L.flops = L.mfma_flops;     // depthwise + FIR are accounted to the dwfir launch
    if (mode == MODE_UP) L.flops += 2.0 * 4 * cout * pout;
    if (mode == MODE_DOWN) L.flops += 2.0 * 16 * cin * pout;      // (narrow layers only: the tiled plan splits a down=2 layer into dwfir + pointwise)
    L.bytes = (fromrgb ? 4.0 * 4.0 : e * cin) * pin + e * cout * pout + (skip_buf != BUF_NONE ? e * cout * pout : 0.0);
    if (mode == MODE_PW) L.bytes = e * cout * pout;   // algorithmic input read is accounted to the dwfir launch
    if (fromrgb) L.flops += 2.0 * 4 * cin * pin;
    L.wgs_batch1 = L.narrow ? cdiv(L.hout * L.wout, kThreads) : (int)grid_of(L.g, 1);
    P.launches.push_back(L);
    if (debug) P.debug_tensors.push_back({layer, out_buf});
    return P.launches.back();
  };

  // ---- encoder (reference :235-246, :192-200) ----
  int cur = BUF_X;
  for (int res = R; res >= 4; res /= 2) {
    const int c = channels_at(res);
    const std::string b = bname("encoder", res);
    const bool first = res == R;
    Launch& l1 = add_sep(b + ".conv1", MODE_NORMAL, c, c, res, res, first, false, cur, feat[ilog2(res)], BUF_NONE);
    if (first) {
      l1.w_frgb = slot_index(b + ".fromrgb.weight");
      l1.b_frgb = slot_index(b + ".fromrgb.bias");
    }
    if (debug) P.debug_tensors.back().second = feat[ilog2(res)];
    if (res > 4 && is_narrow(c, channels_at(res / 2))) {
      // (resolutions above 512: one plain launch for the whole down=2 layer)
      const int cn = channels_at(res / 2);
      const int ob = out_for(b + ".conv2", res / 2, cn, P0);
      add_sep(b + ".conv2", MODE_DOWN, c, cn, res, res / 2, false, false, feat[ilog2(res)], ob, BUF_NONE);
      cur = ob;
    } else if (res > 4) {
      const int cn = channels_at(res / 2);
      // down=2 layer = depthwise+FIR kernel (writes the half-resolution cin-channel tensor) + pointwise GEMM
      {
        Launch L;
        L.layer = b + ".conv2.dwfir";
        L.is_dwfir = true;
        L.hin = hs(res); L.win = ws(res); L.hout = hs(res / 2); L.wout = ws(res / 2);
        L.dg = choose_dwfir_geo(c, L.hin, L.win);
        L.kernel = dwfir_name(L.dg, dwfir_variant(stv, gemm));
        L.cin = c; L.cout = c;
        L.in_buf = feat[ilog2(res)]; L.out_buf = DWT;
        L.w_dw = slot_index(b + ".conv2.conv1.weight");
        L.b_dw = slot_index(b + ".conv2.conv1.bias");
        const double pin = (double)L.hin * L.win, pout = (double)L.hout * L.wout;
        L.flops = 2.0 * 9 * c * pin + 2.0 * 16 * c * pout;
        L.bytes = (double)esz * c * pin;
        L.wgs_batch1 = (int)dwfir_grid(L.dg, 1);
        P.launches.push_back(L);
      }
      const int ob = out_for(b + ".conv2", res / 2, cn, P0);
      add_sep(b + ".conv2", MODE_PW, c, cn, res / 2, res / 2, false, false, DWT, ob, BUF_NONE);
      cur = ob;
    } else {
      const int ob = out_for(b + ".conv2", 4, c, P0);
      add_sep(b + ".conv2", MODE_NORMAL, c, c, 4, 4, false, false, feat[ilog2(res)], ob, BUF_NONE);
      cur = ob;
    }
  }
  // ---- synthesis (reference :347-352, :270-279, :303-315) ----
  int img_cur = BUF_NONE;
  for (int res = 4; res <= R; res *= 2) {
    const int c = channels_at(res);
    const std::string b = bname("synthesis", res);
    const int o1 = out_for(b + ".conv1", res, c, P1);
    if (res == 4) add_sep(b + ".conv1", MODE_NORMAL, c, c, 4, 4, false, false, cur, o1, feat[ilog2(4)]);
    else add_sep(b + ".conv1", MODE_UP, channels_at(res / 2), c, res / 2, res, false, true, cur, o1, feat[ilog2(res)]);
    const int o2 = out_for(b + ".conv2", res, c, P0);
    Launch& l2 = add_sep(b + ".conv2", MODE_NORMAL, c, c, res, res, false, res > 4, o1, o2, BUF_NONE, true);
    cur = o2;
    int img_out;
    if (res == R) img_out = BUF_Y;
    else if (debug) img_out = add_buf(b + ".img", (size_t)3 * hs(res) * ws(res) * sizeof(float));
    else img_out = (img_cur == I0) ? I1 : I0;
    const int wt = slot_index(b + ".torgb.weight"), bt = slot_index(b + ".torgb.bias");
    const double pout = (double)hs(res) * ws(res);
    const double rgb_flops = 2.0 * 3 * c * pout + (img_cur != BUF_NONE ? 2.0 * 4 * 3 * pout : 0.0);
    const double rgb_bytes = 4.0 * (3.0 * pout + (img_cur != BUF_NONE ? 3.0 * pout / 4 : 0.0));
    if (l2.narrow) {
      // (resolutions above 512: the plain kernel finishes the pixel's ToRGB itself)
      l2.w_trgb = wt; l2.b_trgb = bt;
      l2.imgprev_buf = img_cur; l2.imgout_buf = img_out;
      l2.flops += rgb_flops; l2.bytes += rgb_bytes;
    } else if (l2.g.nchunks == 1) {
      // one workgroup owns all output channels of its pixels: ToRGB fused into the conv2 epilogue
      l2.w_trgb = wt; l2.b_trgb = bt;
      l2.g.torgb = true;
      l2.kernel = kernel_name(l2.g);
      l2.imgprev_buf = img_cur; l2.imgout_buf = img_out;
      l2.flops += rgb_flops; l2.bytes += rgb_bytes;
    } else {
      Launch L;
      L.layer = b + ".torgb";
      L.kernel = torgb_name(stv);
      L.is_rgb = true;
      L.cin = c; L.cout = 3;
      L.hin = L.hout = hs(res); L.win = L.wout = ws(res);
      L.in_buf = o2; L.imgprev_buf = img_cur; L.imgout_buf = img_out;
      L.w_trgb = wt; L.b_trgb = bt;
      L.flops = rgb_flops; L.bytes = rgb_bytes;
      L.wgs_batch1 = (int)(((size_t)hs(res) * ws(res) * 16 + kThreads - 1) / kThreads);
      P.launches.push_back(L);
    }
    if (debug && res != R) P.debug_tensors.push_back({b + ".img", img_out});
    img_cur = img_out;
  }
  // shared region: [16-bit weight planes][noise planes]
  const size_t wbytes = align256(P.wsplit_elems * sizeof(unsigned short) + 256);
  for (Launch& L : P.launches)
    if (L.noise_plane_off >= 0) L.noise_plane_off += (long long)wbytes;
  P.shared_bytes = wbytes + noise_bytes;
  // the next sub-batch starts when the previous one is about a fifth of its launches in (measured on MI355X, migan-512 and
  // migan-256, batch 32: +6 % at launch 8..12 of 46 / 40, nothing at 4 or 20; profiles/r02_streams_stagger_sweep.txt)
  P.stagger = (int)P.launches.size() * tuning().stagger_pct / 100;
  if (tuning().stagger >= 0) P.stagger = tuning().stagger;
  // always a launch that exists: the event recorded after it is what orders the next sub-batch behind the weight / noise planes
  P.stagger = std::min(std::max(0, (int)P.launches.size() - 1), std::max(0, P.stagger));
  // the kernels address every per-image tensor with 32-bit lane BYTE offsets and int element indices
  for (const Launch& L : P.launches) {
    const unsigned long long ein = (unsigned long long)L.hin * L.win * (unsigned long long)std::max(L.cin, 4);
    const unsigned long long eout = (unsigned long long)L.hout * L.wout * (unsigned long long)std::max(L.cout, 4);
    MIGAN_CHECK(std::max(ein, eout) < (1ull << 31) && std::max(ein, eout) * 4ull < (1ull << 32), MIGAN_EINVAL,
                "image too large: layer " + L.layer + " would exceed the 32-bit per-image offsets of the kernels");
  }
}

inline migan::Plan& migan_handle::plan_for(int H, int W) {
  if (H == resolution && W == resolution) return plan;
  for (auto& P : hw_plans)
    if (P.H == H && P.W == W) return P;
  // built aside: a plan that fails half-way (geometry check, missing kernel instantiation, bad_alloc) must never be found later
  migan::Plan fresh;
  build_plan(fresh, H, W);
  constexpr size_t kMaxCachedPlans = 16;
  if (hw_plans.size() >= kMaxCachedPlans) hw_plans.erase(hw_plans.begin());     // oldest first
  hw_plans.push_back(std::move(fresh));
  return hw_plans.back();
}

inline void migan_handle::ensure_aux() {
  if (aux_ready) return;
  for (int k = 0; k < kMaxStreams - 1; ++k) {
    migan::rt_check(rt::stream_create(&aux_stream[k]), "hipStreamCreate");
    migan::rt_check(rt::event_create_sync(&ev_join[k]), "hipEventCreate");
  }
  for (int k = 0; k < kMaxStreams; ++k) migan::rt_check(rt::event_create_sync(&ev_mid[k]), "hipEventCreate");
  aux_ready = true;
}

// launches of one sub-batch of n images on `stream`
inline void migan_handle::run_range(const migan::Plan& P, const float* x, float* y, int n, char* sub_ws, char* shared,
                                    rt::stream_t stream, bool timed, int mid_after, int part, const migan_io_u8* u8, int n_geo) {
  using namespace migan;
  std::vector<size_t> offs(P.bufs.size());
  for (size_t i = 0; i < P.bufs.size(); ++i) offs[i] = sub_offset(P, (int)i, n);
  auto bptr = [&](int id) -> void* {
    if (id == BUF_NONE) return nullptr;
    if (id == BUF_X) return const_cast<float*>(x);
    if (id == BUF_Y) return y;
    return sub_ws + offs[id];
  };
  auto wptr = [&](int s) -> const float* { return s < 0 ? nullptr : slots[s].ptr; };
  unsigned short* wsplit = reinterpret_cast<unsigned short*>(shared);
  for (size_t li = 0; li < P.launches.size(); ++li) {
    const Launch& L = P.launches[li];
    if (timed) rt_check(rt::event_record(events[2 * li], stream), "hipEventRecord");
    // a down=2 layer is two plan entries (depthwise + FIR-down, pointwise GEMM); where the fused kernel applies, the first entry launches
    // nothing and the second launches sepconv_pipedown_kernel on the first one's input
    // (nd: the sub-batch size the production forward launches with -- the per-launch timing run covers the whole batch in one launch but
    // must time the kernel forms the throughput run uses)
    const int nd = n_geo > 0 ? n_geo : n;
    const bool dw_fused = L.is_dwfir && li + 1 < P.launches.size() &&
                          pick_pipedown(L.cin, P.launches[li + 1].cout, L.hin, L.win, nd, stv, gemm) != nullptr;
    const bool pw_fused = !L.is_dwfir && !L.is_rgb && L.g.mode == MODE_PW && li >= 1 && P.launches[li - 1].is_dwfir &&
                          pick_pipedown(L.cin, L.cout, P.launches[li - 1].hin, P.launches[li - 1].win, nd, stv, gemm) != nullptr;
    if (dw_fused) {
      L.kernel_last = pick_pipedown(L.cin, P.launches[li + 1].cout, L.hin, L.win, nd, stv, gemm)->name;      // (its bytes belong to that launch)
    } else if (pw_fused) {
      const Launch& D = P.launches[li - 1];
      const DownEntry* de = pick_pipedown(L.cin, L.cout, D.hin, D.win, nd, stv, gemm);
      SepArgs a{};
      a.x = bptr(D.in_buf); a.y = bptr(L.out_buf);
      a.wdw = wptr(D.w_dw); a.bdw = wptr(D.b_dw); a.wpw = wptr(L.w_pw);
      a.wsplit = wsplit + L.wsplit_off;
      a.B = n; a.H = D.hin; a.W = D.win; a.CI = L.cin; a.CO = L.cout; a.HO = L.hout; a.WO = L.wout;
      launch_pipedown(*de, a, stream);
      L.kernel_last = de->name;
    } else if (L.is_dwfir) {
      DwFirArgs a{};
      a.x = bptr(L.in_buf); a.y = (float*)bptr(L.out_buf); a.wdw = wptr(L.w_dw); a.bdw = wptr(L.b_dw);
      a.B = n; a.H = L.hin; a.W = L.win; a.C = L.cin;
      launch_dwfir(L.dg, a, stream, dwfir_variant(stv, gemm));
    } else if (L.narrow) {
      MIGAN_CHECK(u8 == nullptr || (L.in_buf != BUF_X && L.imgout_buf != BUF_Y), MIGAN_EUNSUPPORTED,
                  "the uint8-in / uint8-out forward is not available at resolutions above 512");
      SepArgs a{};
      a.x = bptr(L.in_buf); a.y = bptr(L.out_buf); a.skip = bptr(L.skip_buf);
      a.wdw = wptr(L.w_dw); a.bdw = wptr(L.b_dw); a.wpw = wptr(L.w_pw);
      a.noise = L.noise_plane_off >= 0 ? reinterpret_cast<const float*>(shared + L.noise_plane_off) : wptr(L.w_noise);
      a.noise_strength = wptr(L.w_ns);
      a.frgb_w = wptr(L.w_frgb); a.frgb_b = wptr(L.b_frgb);
      a.trgb_w = wptr(L.w_trgb); a.trgb_b = wptr(L.b_trgb);
      a.img_prev = (const float*)bptr(L.imgprev_buf); a.img_out = (float*)bptr(L.imgout_buf);
      a.B = n; a.H = L.hin; a.W = L.win; a.CI = L.cin; a.CO = L.cout; a.HO = L.hout; a.WO = L.wout;
      launch_narrow(L.mode, L.fromrgb, a, stream);
      L.kernel_last = narrow_name(L.mode, L.fromrgb);
    } else if (L.is_rgb) {
      RgbArgs a{};
      a.x = bptr(L.in_buf); a.w = wptr(L.w_trgb); a.b = wptr(L.b_trgb);
      a.img_prev = (const float*)bptr(L.imgprev_buf); a.img_out = (float*)bptr(L.imgout_buf);
      a.B = n; a.H = L.hout; a.W = L.wout; a.C = L.cin;
      if (u8 && L.imgout_buf == BUF_Y) { a.u8_img = (const unsigned char*)u8->img; a.u8_mask = (const unsigned char*)u8->mask; a.u8_out = (unsigned char*)u8->out; }
      launch_torgb(a, stream, stv);
    } else {
      SepArgs a{};
      a.x = bptr(L.in_buf); a.y = bptr(L.out_buf); a.skip = bptr(L.skip_buf);
      a.wdw = wptr(L.w_dw); a.bdw = wptr(L.b_dw); a.wpw = wptr(L.w_pw);
      a.wsplit = L.g.gemmv ? wsplit + L.wsplit_off : nullptr;
      a.noise = L.noise_plane_off >= 0 ? reinterpret_cast<const float*>(shared + L.noise_plane_off) : wptr(L.w_noise);
      a.noise_strength = wptr(L.w_ns);
      a.frgb_w = wptr(L.w_frgb); a.frgb_b = wptr(L.b_frgb);
      a.trgb_w = wptr(L.w_trgb); a.trgb_b = wptr(L.b_trgb);
      a.img_prev = (const float*)bptr(L.imgprev_buf); a.img_out = (float*)bptr(L.imgout_buf);
      if (u8 && L.in_buf == BUF_X) { a.u8_img = (const unsigned char*)u8->img; a.u8_mask = (const unsigned char*)u8->mask; }
      if (u8 && L.imgout_buf == BUF_Y && a.trgb_w) {
        a.u8_img = (const unsigned char*)u8->img; a.u8_mask = (const unsigned char*)u8->mask; a.u8_out = (unsigned char*)u8->out;
      }
      a.B = n; a.H = L.hin; a.W = L.win; a.CI = L.cin; a.CO = L.cout; a.HO = L.hout; a.WO = L.wout;
      // launches that would leave most CUs idle run the 32-row tiles: a quarter of the work per workgroup, four times the workgroups
      const Geo* Gp = &L.g;
      for (const Geo& gs : L.g_small) {
        // K-split tiles sum K in another order than every other tile (four partial sums): single-image forwards only, so that an
        // image of a batch of two or more is bit-identical whatever the batch size and the sub-batch grouping
        if (gs.NT == 32 && nd != 1) continue;
        if ((int)tiles_of(gs, nd) <= tuning().small_max_wgs) { Gp = &gs; break; }
      }
      const Geo& G = *Gp;
      fill_geo(a, G);
      launch_sepconv(G, a, stream, nd);
      L.kernel_last = launched_kernel_name(G, L.cin, L.cout, nd, a.trgb_w != nullptr, a.skip != nullptr);
    }
    if (timed) rt_check(rt::event_record(events[2 * li + 1], stream), "hipEventRecord");
    if (!timed && (int)li == mid_after) rt_check(rt::event_record(ev_mid[part], stream), "hipEventRecord");
#ifdef MIGAN_PHASE_PROF
    if (timed) {
      prof_layers().resize(P.launches.size());
      rt_check(rt::prof_read(prof_buffer(), prof_layers()[li].v, 16, true), "prof read");
    }
#endif
  }
}

inline void migan_handle::forward(migan::Plan& P, const float* x, float* y, int batch, void* ws, size_t ws_bytes,
                                  rt::stream_t stream, float* ms, int n_ms, const migan_io_u8* u8, const Parts* parts_io) {
  using namespace migan;
  MIGAN_CHECK(committed, MIGAN_ESTATE, "migan_forward before migan_commit");
  MIGAN_CHECK((x || u8) && (y || u8 || parts_io) && batch > 0, MIGAN_EINVAL, "null tensor or empty batch");
  MIGAN_CHECK(ws != nullptr, MIGAN_EINVAL, "null workspace");
  MIGAN_CHECK(ws_bytes >= workspace_bytes(P, batch), MIGAN_EINVAL, "workspace too small for this batch");
  DeviceGuard guard(device);
  const bool timed = ms != nullptr;
  if (timed) {
    MIGAN_CHECK(n_ms >= (int)P.launches.size(), MIGAN_EINVAL, "layer_ms array too small");
    while (events.size() < 2 * P.launches.size()) {
      rt::event_t e;
      rt_check(rt::event_create(&e), "hipEventCreate");
      events.push_back(e);
    }
  }
  char* shared = static_cast<char*>(ws);
  char* sub0 = shared + align256(P.shared_bytes);
  // conv2.weight of every layer -> 16-bit operand planes.  Re-done every forward (the weights are read in place, so
  // in-place parameter updates are always picked up) unless the caller asserted static weights and these planes were
  // written, in this workspace and stream order, since the last (re)binding of a weight.
  const bool planes_valid = static_weights && prepared_ws == ws && prepared_epoch == weight_epoch && prepared_stream == stream;
  if (gemm && !planes_valid) {
    SplitArgs sa{};
    sa.dst = reinterpret_cast<unsigned short*>(shared);
    sa.f16 = gemm == 3 ? 2 : (gemm == 2 ? 1 : 0);
    for (const Launch& L : P.launches) {
      if (L.is_rgb || L.is_dwfir || L.narrow) continue;      // (narrow layers read conv2.weight in fp32)
      MIGAN_CHECK(sa.n < 40, MIGAN_EINVAL, "internal: too many layers for the weight-split table");
      sa.src[sa.n] = slots[L.w_pw].ptr;
      sa.dst_off[sa.n] = L.wsplit_off;
      sa.count[sa.n] = (unsigned)(L.cin * L.cout);
      sa.ci[sa.n] = (unsigned)L.cin;
      ++sa.n;
    }
    launch_split(sa, stream);
    prepared_ws = ws; prepared_epoch = weight_epoch; prepared_stream = stream;     // only reached when every launch succeeded
  }
  // arbitrary-size plans: crop / tile every noise_const to its layer's plane
  for (const Launch& L : P.launches) {
    if (L.noise_plane_off < 0) continue;
    NoiseArgs na{};
    na.src = slots[L.w_noise].ptr;
    na.dst = reinterpret_cast<float*>(shared + L.noise_plane_off);
    na.r = (int)slots[L.w_noise].shape[0]; na.h = L.hout; na.w = L.wout;
    rt_check(rt::launch(noise_plane_kernel, na, (unsigned)cdiv(L.hout * L.wout, kThreads), kThreads, 0, stream), "migan::noise_plane_kernel");
  }
  int nsub[kMaxStreams];
  int parts = split(batch, nsub);
  const int n_production = nsub[0];              // what a launch of the throughput path covers
  if (timed) { nsub[0] = batch; parts = 1; }     // per-launch durations: one stream, whole-batch launches (the split workspace always fits them)
  const size_t in_img = (size_t)4 * P.H * P.W, out_img = (size_t)3 * P.H * P.W;
  if (parts_io) {
    MIGAN_CHECK(!timed && !u8 && parts_io->y && parts_io->out_n && parts_io->out_parts, MIGAN_EINVAL, "migan_forward_parts: bad argument");
    MIGAN_CHECK(parts_io->n_streams >= parts - 1 && (parts == 1 || parts_io->streams), MIGAN_EINVAL,
                "migan_forward_parts: one caller stream per sub-batch after the first is needed (migan_forward_split says how many)");
    for (int k = 0; k < parts; ++k) {
      MIGAN_CHECK(parts_io->y[k] != nullptr, MIGAN_EINVAL, "migan_forward_parts: null output of a sub-batch");
      parts_io->out_n[k] = nsub[k];
    }
    *parts_io->out_parts = parts;
  }
  if (parts > 1) {
    // staggered sub-batches on separate streams: sub-batch k+1 starts when sub-batch k is `stagger` launches in, so the
    // small-resolution layers of one (a few dozen workgroups each) and the tail of every launch run beside full-size layers
    // of the others.  The caller's stream carries sub-batch 0 and is joined to all the others at the end (events only).
    ensure_aux();
    int done = 0;
    char* sub = sub0;
    for (int k = 0; k < parts; ++k) {
      rt::stream_t sk = k == 0 ? stream : (parts_io ? (rt::stream_t)parts_io->streams[k - 1] : aux_stream[k - 1]);
      if (k > 0) rt_check(rt::stream_wait_event(sk, ev_mid[k - 1]), "hipStreamWaitEvent");
      migan_io_u8 uk{};
      if (u8) {
        uk.img = (const unsigned char*)u8->img + (size_t)done * P.H * P.W * 3;
        uk.mask = (const unsigned char*)u8->mask + (size_t)done * P.H * P.W;
        uk.out = (unsigned char*)u8->out + (size_t)done * P.H * P.W * 3;
      }
      float* yk = parts_io ? static_cast<float*>(parts_io->y[k]) : (y ? y + (size_t)done * out_img : nullptr);
      run_range(P, x ? x + (size_t)done * in_img : nullptr, yk, nsub[k], sub, shared, sk, false,
                k + 1 < parts ? P.stagger : -1, k, u8 ? &uk : nullptr);
      if (k > 0 && !parts_io) rt_check(rt::event_record(ev_join[k - 1], sk), "hipEventRecord");
      sub += sub_bytes(P, nsub[k]);
      done += nsub[k];
    }
    // (migan_forward_parts: no join -- the caller owns the other streams and orders what follows behind each sub-batch)
    if (!parts_io)
      for (int k = 1; k < parts; ++k) rt_check(rt::stream_wait_event(stream, ev_join[k - 1]), "hipStreamWaitEvent");
  } else {
    run_range(P, x, parts_io ? static_cast<float*>(parts_io->y[0]) : y, nsub[0], sub0, shared, stream, timed, -1, 0, u8, timed ? n_production : 0);
  }
  if (timed) {
    rt_check(rt::stream_sync(stream), "hipStreamSynchronize");
    for (size_t li = 0; li < P.launches.size(); ++li)
      rt_check(rt::event_elapsed(&ms[li], events[2 * li], events[2 * li + 1]), "hipEventElapsedTime");
  }
}

// ------------------------------------------------------------------------------------------------
// C ABI

#define MIGAN_API_BEGIN try {
#define MIGAN_API_END                                       \
  }                                                         \
  catch (const ::migan::Error& e) {                         \
    ::migan::last_error_ref() = e.what();                   \
    return e.code;                                          \
  }                                                         \
  catch (const std::exception& e) {                         \
    ::migan::last_error_ref() = e.what();                   \
    return MIGAN_ERUNTIME;                                  \
  }                                                         \
  return MIGAN_OK;

extern "C" {

int migan_create(int resolution, int dtype, int device, migan_handle** out) {
  MIGAN_API_BEGIN
  using namespace migan;
  MIGAN_CHECK(out != nullptr, MIGAN_EINVAL, "null out pointer");
  *out = nullptr;
  MIGAN_CHECK(resolution > 0 && (resolution & (resolution - 1)) == 0, MIGAN_EINVAL,
              "resolution must be a power of two (reference migan_inference.py:215-216)");
  MIGAN_CHECK(resolution >= 8 && resolution <= 4096, MIGAN_EINVAL, "resolution must be in [8, 4096]");
  MIGAN_CHECK(dtype == MIGAN_DTYPE_F32 || dtype == MIGAN_DTYPE_BF16 || dtype == MIGAN_DTYPE_F16, MIGAN_EINVAL,
              "dtype must be MIGAN_DTYPE_F32, MIGAN_DTYPE_BF16 or MIGAN_DTYPE_F16 (activation storage format)");
  DeviceGuard guard(device);
  prepare_kernels();
  migan_handle* h = new migan_handle();
  h->resolution = resolution;
  h->device = device;
  h->stv = dtype;
  h->gemm = dtype == MIGAN_DTYPE_F32 ? tuning().gemm : MIGAN_GEMM_F16;
  h->streams = tuning().streams;
  h->build_schema();
  try {
    h->rebuild();
  } catch (...) {
    delete h;
    throw;
  }
  *out = h;
  MIGAN_API_END
}

int migan_destroy(migan_handle* h) {
  MIGAN_API_BEGIN
  if (h) {
    migan::DeviceGuard guard(h->device);
    for (auto& e : h->events) rt::event_destroy(e);
    if (h->aux_ready) {
      for (int k = 0; k < migan_handle::kMaxStreams - 1; ++k) {
        rt::event_destroy(h->ev_join[k]);
        rt::stream_destroy(h->aux_stream[k]);
      }
      for (int k = 0; k < migan_handle::kMaxStreams; ++k) rt::event_destroy(h->ev_mid[k]);
    }
    delete h;
  }
  MIGAN_API_END
}

int migan_set_gemm(migan_handle* h, int variant) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  MIGAN_CHECK(variant >= MIGAN_GEMM_F32 && variant <= MIGAN_GEMM_F16, MIGAN_EINVAL, "unknown GEMM variant");
  MIGAN_CHECK(h->stv == MIGAN_DTYPE_F32 ? variant <= MIGAN_GEMM_F16X2 : variant >= MIGAN_GEMM_F16X2, MIGAN_EINVAL,
              "fp32 activation storage runs the f32 / bf16x3 / f16x2 GEMM variants, 16-bit storage the f16x2 / f16 variants");
  h->gemm = variant;
  h->rebuild();
  MIGAN_API_END
}

int migan_get_gemm(const migan_handle* h, int* variant) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && variant, MIGAN_EINVAL, "null argument");
  *variant = h->gemm;
  MIGAN_API_END
}

int migan_assume_static_weights(migan_handle* h, int on) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  h->static_weights = on != 0;
  h->prepared_epoch = 0;          // the next forward prepares the planes once more, then keeps them
  MIGAN_API_END
}

int migan_set_streams(migan_handle* h, int streams) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  MIGAN_CHECK(streams >= 1 && streams <= migan_handle::kMaxStreams, MIGAN_EINVAL, "streams must be 1 .. 4");
  h->streams = streams;
  MIGAN_API_END
}

int migan_num_weights(const migan_handle* h, int* n) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && n, MIGAN_EINVAL, "null argument");
  *n = (int)h->slots.size();
  MIGAN_API_END
}

int migan_weight_info(const migan_handle* h, int index, const char** name, int64_t shape[4], int* ndim, int* is_buffer) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  MIGAN_CHECK(index >= 0 && index < (int)h->slots.size(), MIGAN_EINVAL, "weight index out of range");
  const migan::Slot& s = h->slots[index];
  if (name) *name = s.name.c_str();
  if (shape)
    for (int i = 0; i < 4; ++i) shape[i] = s.shape[i];
  if (ndim) *ndim = s.ndim;
  if (is_buffer) *is_buffer = s.is_buffer ? 1 : 0;
  MIGAN_API_END
}

int migan_set_weight(migan_handle* h, const char* name, const void* dev_ptr, const int64_t* shape, int ndim) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && name, MIGAN_EINVAL, "null argument");
  const int i = h->slot_index(name);
  MIGAN_CHECK(i >= 0, MIGAN_EINVAL, std::string("unexpected key in state_dict: ") + name);
  migan::Slot& s = h->slots[i];
  MIGAN_CHECK(dev_ptr != nullptr, MIGAN_EINVAL, std::string("null pointer for ") + name);
  MIGAN_CHECK(ndim == s.ndim, MIGAN_EINVAL, std::string("size mismatch for ") + name);
  for (int d = 0; d < ndim; ++d) MIGAN_CHECK(shape && shape[d] == s.shape[d], MIGAN_EINVAL, std::string("size mismatch for ") + name);
  MIGAN_CHECK((reinterpret_cast<uintptr_t>(dev_ptr) & 15) == 0 || s.numel() < 4, MIGAN_EINVAL,
              std::string("tensor must be 16-byte aligned: ") + name);
  s.ptr = static_cast<const float*>(dev_ptr);
  h->committed = false;
  ++h->weight_epoch;              // (re)bound weights: operand planes prepared earlier are stale
  MIGAN_API_END
}

int migan_commit(migan_handle* h, void* stream) {
  MIGAN_API_BEGIN
  using namespace migan;
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  DeviceGuard guard(h->device);
  ++h->weight_epoch;
  for (const auto& s : h->slots) MIGAN_CHECK(s.ptr != nullptr, MIGAN_ESTATE, std::string("missing key in state_dict: ") + s.name);
  std::vector<float> host;
  static const double taps[4] = {1.0, 3.0, 3.0, 1.0};
  for (const auto& s : h->slots) {
    if (s.role != R_FIR_DOWN && s.role != R_FIR_UP && s.role != R_FILTER_CONST) continue;
    host.resize(s.numel());
    rt_check(rt::memcpy_d2h(host.data(), s.ptr, host.size() * sizeof(float), (rt::stream_t)stream), "hipMemcpy (FIR check)");
    if (s.role == R_FILTER_CONST) {
      const int r = (int)s.shape[3];
      for (int yy = 0; yy < r; ++yy)
        for (int xx = 0; xx < r; ++xx) {
          const float want = ((yy | xx) & 1) ? 0.0f : 1.0f;   // reference :83-85
          MIGAN_CHECK(host[(size_t)yy * r + xx] == want, MIGAN_EUNSUPPORTED,
                      s.name + " is not the even/even zero-insertion mask the kernels assume");
        }
    } else {
      const double gain = s.role == R_FIR_UP ? 4.0 : 1.0;     // reference :71, :95
      for (int64_t c = 0; c < s.shape[0]; ++c)
        for (int ky = 0; ky < 4; ++ky)
          for (int kx = 0; kx < 4; ++kx) {
            const double want = taps[ky] * taps[kx] / 64.0 * gain;
            MIGAN_CHECK(std::fabs((double)host[(size_t)c * 16 + ky * 4 + kx] - want) <= 1e-6, MIGAN_EUNSUPPORTED,
                        s.name + " differs from setup_filter([1,3,3,1]); only the reference FIR is implemented");
          }
    }
  }
  h->committed = true;
  MIGAN_API_END
}

int migan_workspace_bytes(const migan_handle* h, int batch, size_t* bytes) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && bytes && batch > 0, MIGAN_EINVAL, "bad argument");
  *bytes = h->workspace_bytes(h->plan, batch);
  MIGAN_API_END
}

int migan_forward(migan_handle* h, const void* x, void* y, int batch, void* ws, size_t ws_bytes, void* stream) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  h->forward(h->plan, static_cast<const float*>(x), static_cast<float*>(y), batch, ws, ws_bytes, (rt::stream_t)stream, nullptr, 0);
  MIGAN_API_END
}

int migan_forward_split(const migan_handle* h, int batch, int part_batch[4], int* n_parts) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && part_batch && n_parts && batch > 0, MIGAN_EINVAL, "bad argument");
  int n[migan_handle::kMaxStreams] = {0, 0, 0, 0};
  *n_parts = h->split(batch, n);
  for (int k = 0; k < 4; ++k) part_batch[k] = k < *n_parts ? n[k] : 0;
  MIGAN_API_END
}

int migan_forward_parts(migan_handle* h, const void* x, void* const* y_parts, int batch, void* ws, size_t ws_bytes, void* stream,
                        void* const* part_streams, int n_part_streams, int part_batch[4], int* n_parts) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && y_parts && part_batch && n_parts, MIGAN_EINVAL, "null argument");
  migan_handle::Parts io;
  io.y = y_parts; io.streams = part_streams; io.n_streams = n_part_streams;
  int n[migan_handle::kMaxStreams] = {0, 0, 0, 0};
  io.out_n = n; io.out_parts = n_parts;
  h->forward(h->plan, static_cast<const float*>(x), nullptr, batch, ws, ws_bytes, (rt::stream_t)stream, nullptr, 0, nullptr, &io);
  for (int k = 0; k < 4; ++k) part_batch[k] = n[k];
  MIGAN_API_END
}

int migan_forward_timed(migan_handle* h, const void* x, void* y, int batch, void* ws, size_t ws_bytes, void* stream,
                        float* layer_ms, int n_layer_ms) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && layer_ms, MIGAN_EINVAL, "null argument");
  h->forward(h->plan, static_cast<const float*>(x), static_cast<float*>(y), batch, ws, ws_bytes, (rt::stream_t)stream, layer_ms, n_layer_ms);
  MIGAN_API_END
}

// fully convolutional forward
static void migan_check_hw(const migan_handle* h, int height, int width) {
  const int q = h->resolution / 4;
  MIGAN_CHECK(height >= q && width >= q && height % q == 0 && width % q == 0, MIGAN_EINVAL,
              "height and width must be positive multiples of resolution / 4 (the network halves its input log2(resolution) - 2 times)");
  // per-layer bound on the 32-bit per-image offsets of the kernels: the largest tensor of the plan at this size is
  // h * w * channels_at(resolution) elements at full resolution (build_plan re-checks every layer)
  const unsigned long long top = (unsigned long long)height * width * (unsigned long long)migan::channels_at(h->resolution);
  MIGAN_CHECK(top < (1ull << 31) && top * 4ull < (1ull << 32), MIGAN_EINVAL,
              "image too large: height * width * channels must stay below 2^30 elements per image (32-bit offsets in the kernels)");
}
int migan_workspace_bytes_hw(migan_handle* h, int batch, int height, int width, size_t* bytes) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && bytes && batch > 0, MIGAN_EINVAL, "bad argument");
  migan_check_hw(h, height, width);
  *bytes = h->workspace_bytes(h->plan_for(height, width), batch);
  MIGAN_API_END
}
int migan_forward_hw(migan_handle* h, const void* x, void* y, int batch, int height, int width, void* ws, size_t ws_bytes, void* stream) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  migan_check_hw(h, height, width);
  h->forward(h->plan_for(height, width), static_cast<const float*>(x), static_cast<float*>(y), batch, ws, ws_bytes, (rt::stream_t)stream,
             nullptr, 0);
  MIGAN_API_END
}

// uint8 in, uint8 out: demo.py's preprocess() fused into the first layer's tile builder, its postprocess + compose fused
// into the last ToRGB epilogue
int migan_forward_u8(migan_handle* h, const void* img_hwc_u8, const void* mask_u8, void* out_hwc_u8, int batch, void* ws, size_t ws_bytes,
                     void* stream) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && img_hwc_u8 && mask_u8 && out_hwc_u8, MIGAN_EINVAL, "null argument");
  MIGAN_CHECK(((uintptr_t)img_hwc_u8 % 4) == 0 && ((uintptr_t)mask_u8 % 4) == 0 && ((uintptr_t)out_hwc_u8 % 4) == 0, MIGAN_EINVAL,
              "uint8 tensors must be 4-byte aligned");
  migan_io_u8 io{img_hwc_u8, mask_u8, out_hwc_u8};
  h->forward(h->plan, nullptr, nullptr, batch, ws, ws_bytes, (rt::stream_t)stream, nullptr, 0, &io);
  MIGAN_API_END
}

int migan_num_launches(const migan_handle* h, int* n) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && n, MIGAN_EINVAL, "null argument");
  *n = (int)h->plan.launches.size();
  MIGAN_API_END
}

int migan_launch_info(const migan_handle* h, int index, const char** layer, const char** kernel, double* flops,
                      double* mfma_flops, double* bytes, int* wgs) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  MIGAN_CHECK(index >= 0 && index < (int)h->plan.launches.size(), MIGAN_EINVAL, "launch index out of range");
  const migan::Launch& L = h->plan.launches[index];
  if (layer) *layer = L.layer.c_str();
  if (kernel) *kernel = L.kernel_last.empty() ? L.kernel.c_str() : L.kernel_last.c_str();
  if (flops) *flops = L.flops;
  if (mfma_flops) *mfma_flops = L.mfma_flops;
  if (bytes) *bytes = L.bytes;
  if (wgs) *wgs = L.wgs_batch1;
  MIGAN_API_END
}

int migan_set_debug(migan_handle* h, int keep) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  h->debug = keep != 0;
  h->rebuild();
  MIGAN_API_END
}

int migan_debug_tensor(const migan_handle* h, int batch, const char* layer, size_t* byte_offset, int64_t shape[4]) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && layer && byte_offset && shape, MIGAN_EINVAL, "null argument");
  MIGAN_CHECK(h->debug, MIGAN_ESTATE, "migan_set_debug(h, 1) first");
  const migan::Plan& P = h->plan;
  for (const auto& kv : P.debug_tensors) {
    if (kv.first != layer) continue;
    *byte_offset = migan::align256(P.shared_bytes) + migan_handle::sub_offset(P, kv.second, batch);     // debug plans never split the batch
    const std::string& n = kv.first;
    const bool is_img = n.size() > 4 && n.compare(n.size() - 4, 4, ".img") == 0;
    for (const auto& L : P.launches) {
      if (is_img) {
        if (L.imgout_buf == kv.second) { shape[0] = batch; shape[1] = 3; shape[2] = L.hout; shape[3] = L.wout; return MIGAN_OK; }
      } else if (!L.is_rgb && !L.is_dwfir && L.layer == n) {
        shape[0] = batch; shape[1] = L.hout; shape[2] = L.wout; shape[3] = L.cout;
        return MIGAN_OK;
      }
    }
  }
  throw migan::Error(MIGAN_EINVAL, std::string("no such debug tensor: ") + layer);
  MIGAN_API_END
}

int migan_sepconv_forward(const migan_sepconv_desc* d, void* stream) {
  MIGAN_API_BEGIN
  using namespace migan;
  MIGAN_CHECK(d != nullptr, MIGAN_EINVAL, "null descriptor");
  MIGAN_CHECK(d->x && d->y && d->conv1_weight && d->conv1_bias && d->conv2_weight, MIGAN_EINVAL, "null tensor");
  MIGAN_CHECK((d->down == 1 || d->down == 2) && (d->up == 1 || d->up == 2) && !(d->down == 2 && d->up == 2), MIGAN_EINVAL,
              "down/up must be 1 or 2 and not both 2");
  MIGAN_CHECK(d->batch > 0, MIGAN_EINVAL, "empty batch");
  MIGAN_CHECK(d->dtype >= 0 && d->dtype <= 2, MIGAN_EINVAL, "dtype must be a MIGAN_DTYPE_* value");
  const int stv = d->dtype;
  const int h_in = d->res_in, w_in = d->width_in > 0 ? d->width_in : d->res_in;
  MIGAN_CHECK(d->down == 1 || (h_in % 2 == 0 && w_in % 2 == 0), MIGAN_EINVAL, "down=2 needs even sizes");
  const int h_out = d->down == 2 ? h_in / 2 : (d->up == 2 ? h_in * 2 : h_in);
  const int w_out = d->down == 2 ? w_in / 2 : (d->up == 2 ? w_in * 2 : w_in);
  MIGAN_CHECK(d->noise_const == nullptr || d->noise_strength != nullptr, MIGAN_EINVAL, "noise_const without noise_strength");
  if (is_narrow(d->cin, d->cout) && d->cin <= 64 && d->cout <= 64 && d->cin % 4 == 0 && d->cout % 4 == 0) {
    // fewer than 64 channels on either side (the layers of resolutions above 512): the plain kernel, the whole layer in one launch
    MIGAN_CHECK(stv == 0, MIGAN_EUNSUPPORTED, "layers with fewer than 64 channels run with fp32 activation storage only");
    MIGAN_CHECK(d->torgb_weight == nullptr || (d->up == 1 && d->down == 1 && d->img_out && d->torgb_bias), MIGAN_EINVAL,
                "ToRGB needs up == down == 1, torgb_bias and img_out");
    MIGAN_CHECK(d->fromrgb_weight == nullptr || (d->up == 1 && d->down == 1), MIGAN_EINVAL, "fromrgb is only fused into plain layers");
    const int nmode = d->down == 2 ? MODE_DOWN : (d->up == 2 ? MODE_UP : MODE_NORMAL);
    SepArgs a{};
    a.x = d->x; a.y = d->y; a.skip = d->skip;
    a.wdw = (const float*)d->conv1_weight; a.bdw = (const float*)d->conv1_bias; a.wpw = (const float*)d->conv2_weight;
    a.noise = (const float*)d->noise_const; a.noise_strength = (const float*)d->noise_strength;
    a.frgb_w = (const float*)d->fromrgb_weight; a.frgb_b = (const float*)d->fromrgb_bias;
    a.trgb_w = (const float*)d->torgb_weight; a.trgb_b = (const float*)d->torgb_bias;
    a.img_prev = (const float*)d->img_prev; a.img_out = (float*)d->img_out;
    a.B = d->batch; a.H = h_in; a.W = w_in; a.CI = d->cin; a.CO = d->cout; a.HO = h_out; a.WO = w_out;
    launch_narrow(nmode, d->fromrgb_weight != nullptr, a, (rt::stream_t)stream);
    return MIGAN_OK;
  }
  const void* gemm_in = d->x;
  int mode = d->up == 2 ? MODE_UP : MODE_NORMAL, gemm_h = h_in, gemm_w = w_in;
  // split GEMM variants need room for the 16-bit weight planes; without it the exact fp32 MFMA path runs
  const size_t wsplit_need = wsplit_elems_of(d->cin, d->cout) * sizeof(unsigned short);
  const int want = stv != 0 ? (d->gemm == MIGAN_GEMM_F16X2 ? MIGAN_GEMM_F16X2 : MIGAN_GEMM_F16) : (d->gemm >= 0 ? d->gemm : tuning().gemm);
  MIGAN_CHECK(want >= 0 && want <= 3 && (stv != 0 || want <= 2), MIGAN_EINVAL, "unknown GEMM variant for this storage format");
  const bool have_planes = d->wsplit != nullptr && d->wsplit_bytes >= wsplit_need;
  MIGAN_CHECK(stv == 0 || have_planes, MIGAN_EINVAL, "16-bit activation storage needs the wsplit buffer (fp16 GEMM variants)");
  const int gemmv = have_planes ? want : 0;
  const DownEntry* fused_down = nullptr;
  if (d->down == 2 && have_planes && d->fromrgb_weight == nullptr && d->torgb_weight == nullptr && d->noise_const == nullptr && d->skip == nullptr)
    fused_down = pick_pipedown(d->cin, d->cout, h_in, w_in, d->batch, stv, gemmv);
  if (fused_down) {
    // one launch: depthwise + FIR-down feed the 1x1 through LDS (no scratch tensor)
    SplitArgs sa{};
    sa.dst = (unsigned short*)d->wsplit;
    sa.f16 = 1;
    sa.src[0] = (const float*)d->conv2_weight; sa.dst_off[0] = kSplitHeader; sa.count[0] = (unsigned)(d->cin * d->cout); sa.ci[0] = (unsigned)d->cin; sa.n = 1;
    launch_split(sa, (rt::stream_t)stream);
    SepArgs a{};
    a.x = d->x; a.y = d->y;
    a.wdw = (const float*)d->conv1_weight; a.bdw = (const float*)d->conv1_bias; a.wpw = (const float*)d->conv2_weight;
    a.wsplit = (const unsigned short*)d->wsplit + kSplitHeader;
    a.B = d->batch; a.H = h_in; a.W = w_in; a.CI = d->cin; a.CO = d->cout; a.HO = h_out; a.WO = w_out;
    launch_pipedown(*fused_down, a, (rt::stream_t)stream);
    return MIGAN_OK;
  }
  if (d->down == 2) {
    // reference :155-161: depthwise+act+FIR at res_in (dwfir kernel), then the 1x1 at res_in/2
    MIGAN_CHECK(d->fromrgb_weight == nullptr, MIGAN_EINVAL, "fromrgb is only fused into plain layers");
    const size_t need = (size_t)d->batch * h_out * w_out * d->cin * sizeof(float);
    MIGAN_CHECK(d->scratch != nullptr && d->scratch_bytes >= need, MIGAN_EINVAL,
                "down=2 needs scratch of batch*(res_in/2)^2*cin floats");
    const DwGeo dg = choose_dwfir_geo(d->cin, h_in, w_in);
    DwFirArgs fa{};
    fa.x = d->x; fa.y = (float*)d->scratch; fa.wdw = (const float*)d->conv1_weight; fa.bdw = (const float*)d->conv1_bias;
    fa.B = d->batch; fa.H = h_in; fa.W = w_in; fa.C = d->cin;
    launch_dwfir(dg, fa, (rt::stream_t)stream, dwfir_variant(stv, gemmv));
    gemm_in = d->scratch;
    mode = MODE_PW;
    gemm_h = h_out; gemm_w = w_out;
  }
  const Geo g = choose_geo(mode, d->cin, d->cout, gemm_h, gemm_w, d->fromrgb_weight != nullptr, d->torgb_weight != nullptr, gemmv, stv);
  if (gemmv) {
    SplitArgs sa{};
    sa.dst = (unsigned short*)d->wsplit;
    sa.f16 = gemmv == 3 ? 2 : (gemmv == 2 ? 1 : 0);
    sa.src[0] = (const float*)d->conv2_weight; sa.dst_off[0] = kSplitHeader; sa.count[0] = (unsigned)(d->cin * d->cout); sa.ci[0] = (unsigned)d->cin; sa.n = 1;
    launch_split(sa, (rt::stream_t)stream);
  }
  MIGAN_CHECK(d->torgb_weight == nullptr || (mode != MODE_UP && d->img_out && d->torgb_bias), MIGAN_EINVAL, "ToRGB needs up == 1, torgb_bias and img_out");
  // one workgroup owns all output channels of its pixels (cout <= 128, or 256 on whole 8x16 tiles): ToRGB fuses into the epilogue;
  // otherwise torgb_kernel runs on y afterwards, as in the generator plan
  const bool fuse_rgb = d->torgb_weight != nullptr && g.nchunks == 1;
  SepArgs a{};
  a.x = gemm_in; a.y = d->y; a.skip = d->skip;
  a.wdw = (const float*)d->conv1_weight; a.bdw = (const float*)d->conv1_bias; a.wpw = (const float*)d->conv2_weight;
  a.wsplit = gemmv ? (const unsigned short*)d->wsplit + kSplitHeader : nullptr;
  a.noise = (const float*)d->noise_const; a.noise_strength = (const float*)d->noise_strength;
  a.frgb_w = (const float*)d->fromrgb_weight; a.frgb_b = (const float*)d->fromrgb_bias;
  if (fuse_rgb) {
    a.trgb_w = (const float*)d->torgb_weight; a.trgb_b = (const float*)d->torgb_bias;
    a.img_prev = (const float*)d->img_prev; a.img_out = (float*)d->img_out;
  }
  a.B = d->batch; a.H = gemm_h; a.W = gemm_w; a.CI = d->cin; a.CO = d->cout; a.HO = h_out; a.WO = w_out;
  fill_geo(a, g);
  launch_sepconv(g, a, (rt::stream_t)stream);
  if (d->torgb_weight != nullptr && !fuse_rgb) {
    RgbArgs r{};
    r.x = d->y; r.w = (const float*)d->torgb_weight; r.b = (const float*)d->torgb_bias;
    r.img_prev = (const float*)d->img_prev; r.img_out = (float*)d->img_out;
    r.B = d->batch; r.H = h_out; r.W = w_out; r.C = d->cout;
    launch_torgb(r, (rt::stream_t)stream, stv);
  }
  MIGAN_API_END
}

static int migan_prepost_launch(bool pack, const void* y, const void* img, const void* mask, void* x, void* out, int batch,
                                int resolution, void* stream) {
  MIGAN_API_BEGIN
  using namespace migan;
  MIGAN_CHECK(img && mask && (pack ? x != nullptr : (y != nullptr && out != nullptr)), MIGAN_EINVAL, "null tensor");
  MIGAN_CHECK(batch > 0, MIGAN_EINVAL, "empty batch");
  MIGAN_CHECK(resolution >= 8 && (resolution & (resolution - 1)) == 0, MIGAN_EINVAL, "resolution must be a power of two >= 8");
  MIGAN_CHECK(((uintptr_t)img % 4) == 0 && ((uintptr_t)mask % 4) == 0 && (out == nullptr || ((uintptr_t)out % 4) == 0) &&
              (x == nullptr || ((uintptr_t)x % 16) == 0) && (y == nullptr || ((uintptr_t)y % 16) == 0), MIGAN_EINVAL,
              "pointers must be 4-byte (uint8 tensors) / 16-byte (fp32 tensors) aligned");
  PrePostArgs a{};
  a.img = (const unsigned char*)img; a.mask = (const unsigned char*)mask; a.y = (const float*)y; a.x = (float*)x; a.out = (unsigned char*)out;
  a.plane = (unsigned)(resolution * resolution);
  const size_t quads = (size_t)batch * a.plane / 4;
  MIGAN_CHECK(quads < (1ull << 30), MIGAN_EINVAL, "batch too large for one launch");
  a.nquads = (unsigned)quads;
  const unsigned grid = (unsigned)((quads + kThreads - 1) / kThreads);
  if (pack) rt_check(rt::launch(pack_input_kernel, a, grid, kThreads, 0, (rt::stream_t)stream), "migan::pack_input_kernel");
  else rt_check(rt::launch(compose_output_kernel, a, grid, kThreads, 0, (rt::stream_t)stream), "migan::compose_output_kernel");
  MIGAN_API_END
}
int migan_pack_input(const void* img_hwc_u8, const void* mask_u8, void* x_nchw, int batch, int resolution, void* stream) {
  return migan_prepost_launch(true, nullptr, img_hwc_u8, mask_u8, x_nchw, nullptr, batch, resolution, stream);
}
int migan_compose_output(const void* y_nchw, const void* img_hwc_u8, const void* mask_u8, void* out_hwc_u8, int batch, int resolution,
                         void* stream) {
  return migan_prepost_launch(false, y_nchw, img_hwc_u8, mask_u8, nullptr, out_hwc_u8, batch, resolution, stream);
}

// ---- the deployed pipeline around the generator (reference scripts/create_onnx_pipeline.py::MIGAN_Pipeline :118-264) ----
static void migan_pipeline_check(const void* mask, int height, int width, int resolution) {
  MIGAN_CHECK(mask, MIGAN_EINVAL, "null mask");
  MIGAN_CHECK(height >= 3 && width >= 3 && (unsigned long long)height * width < (1ull << 30), MIGAN_EINVAL,
              "image must be at least 3x3 (reflect padding of the 5x5 blur) and below 2^30 pixels");
  MIGAN_CHECK(resolution >= 8 && (resolution & (resolution - 1)) == 0, MIGAN_EINVAL, "resolution must be a power of two >= 8");
}
static void migan_pipeline_check_bbox(const int bbox[4], int height, int width) {
  MIGAN_CHECK(bbox, MIGAN_EINVAL, "null bbox");
  MIGAN_CHECK(bbox[0] >= 0 && bbox[1] <= width && bbox[2] >= 0 && bbox[3] <= height && bbox[1] - bbox[0] >= 3 && bbox[3] - bbox[2] >= 3,
              MIGAN_EINVAL, "bbox = {x_min, x_max, y_min, y_max} must lie inside the image and be at least 3x3");
}
int migan_pipeline_scratch_bytes(int height, int width, size_t* bytes) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(bytes && height > 0 && width > 0, MIGAN_EINVAL, "bad argument");
  *bytes = migan::align256((size_t)(height + width) * sizeof(int)) + migan::align256((size_t)height * width);
  MIGAN_API_END
}
// tvF.resize(mask, (H, W), NEAREST) (:256): a mask of another size is brought to the image's size first
int migan_pipeline_mask_resize(const void* mask_u8, int mask_height, int mask_width, void* out_u8, int height, int width, void* stream) {
  MIGAN_API_BEGIN
  using namespace migan;
  MIGAN_CHECK(mask_u8 && out_u8 && mask_height > 0 && mask_width > 0 && height > 0 && width > 0 &&
              (unsigned long long)height * width < (1ull << 30), MIGAN_EINVAL, "bad argument");
  PipeArgs a{};
  a.mask = (const unsigned char*)mask_u8; a.pooled = (unsigned char*)out_u8; a.H = height; a.W = width; a.y_max = mask_height; a.x_max = mask_width;
  rt_check(rt::launch(pipe_mask_resize_kernel, a, (unsigned)cdiv(height * width, kThreads), kThreads, 0, (rt::stream_t)stream), "migan::pipe_mask_resize_kernel");
  MIGAN_API_END
}
// get_masked_bbox (:132-231).  The per-row / per-column "contains a pixel below 255" flags are computed on the device, copied to
// the host (this call synchronises `stream`), and the box arithmetic -- a dozen integer min/max, the reference's own order -- runs here.
int migan_pipeline_bbox(const void* mask_u8, int height, int width, int resolution, int padding, void* scratch, int bbox[4], void* stream) {
  MIGAN_API_BEGIN
  using namespace migan;
  migan_pipeline_check(mask_u8, height, width, resolution);
  MIGAN_CHECK(scratch && bbox && padding >= 0, MIGAN_EINVAL, "bad argument");
  PipeArgs a{};
  a.mask = (const unsigned char*)mask_u8; a.flags = (int*)scratch; a.H = height; a.W = width; a.R = resolution;
  rt_check(rt::launch(pipe_flags_clear_kernel, a, (unsigned)cdiv(height + width, kThreads), kThreads, 0, (rt::stream_t)stream), "migan::pipe_flags_clear_kernel");
  rt_check(rt::launch(pipe_flags_kernel, a, (unsigned)cdiv(height * width, kThreads), kThreads, 0, (rt::stream_t)stream), "migan::pipe_flags_kernel");
  std::vector<int> f((size_t)height + width);
  rt_check(rt::memcpy_d2h(f.data(), scratch, f.size() * sizeof(int), (rt::stream_t)stream), "hipMemcpy (bbox flags)");
  int x_min = width, x_max = 0, y_min = height, y_max = 0;                         // :149-152 (min over [..., w], max over [..., 0])
  for (int x = 0; x < width; ++x) if (f[x]) { x_min = std::min(x_min, x); x_max = std::max(x_max, x); }
  for (int y = 0; y < height; ++y) if (f[width + y]) { y_min = std::min(y_min, y); y_max = std::max(y_max, y); }
  x_min = std::min(x_min, x_max); x_max = std::max(x_min, x_max);                  // :154-172
  y_min = std::min(y_min, y_max); y_max = std::max(y_min, y_max);
  const int cnt_x = (x_min + x_max) / 2, cnt_y = (y_min + y_max) / 2;              // :174-175
  int crop = std::max(x_max - x_min, y_max - y_min) + 2 * padding;                 // :177-180
  crop = std::max(crop, resolution);                                               // :181-184
  const int off = crop / 2;                                                        // :186
  x_min = std::max(cnt_x - off, 0); x_max = std::min(cnt_x + off, width);          // :187-202
  y_min = std::max(cnt_y - off, 0); y_max = std::min(cnt_y + off, height);
  const int xe = std::max(crop - (x_max - x_min), 0), ye = std::max(crop - (y_max - y_min), 0);   // :204-211
  x_min = std::max(x_min - xe, 0); x_max = std::min(x_max + xe, width);            // :213-229
  y_min = std::max(y_min - ye, 0); y_max = std::min(y_max + ye, height);
  bbox[0] = x_min; bbox[1] = x_max; bbox[2] = y_min; bbox[3] = y_max;
  MIGAN_API_END
}
// preprocess (:233-239) of image[:, y_min:y_max, x_min:x_max] -> the network input x [1][4][R][R]
int migan_pipeline_pre(const void* image_chw_u8, const void* mask_u8, int height, int width, const int bbox[4], int resolution, void* x_nchw,
                       void* stream) {
  MIGAN_API_BEGIN
  using namespace migan;
  migan_pipeline_check(mask_u8, height, width, resolution);
  migan_pipeline_check_bbox(bbox, height, width);
  MIGAN_CHECK(image_chw_u8 && x_nchw, MIGAN_EINVAL, "null tensor");
  PipeArgs a{};
  a.image = (unsigned char*)image_chw_u8; a.mask = (const unsigned char*)mask_u8; a.x = (float*)x_nchw;
  a.H = height; a.W = width; a.R = resolution; a.x_min = bbox[0]; a.x_max = bbox[1]; a.y_min = bbox[2]; a.y_max = bbox[3];
  rt_check(rt::launch(pipe_pre_kernel, a, (unsigned)cdiv(resolution * resolution, kThreads), kThreads, 0, (rt::stream_t)stream), "migan::pipe_pre_kernel");
  MIGAN_API_END
}
// postprocess (:241-250) and the paste back into the image (:263), in place
int migan_pipeline_post(void* image_chw_u8, const void* mask_u8, int height, int width, const int bbox[4], int resolution, const void* y_nchw,
                        const float* gauss25, void* scratch, void* stream) {
  MIGAN_API_BEGIN
  using namespace migan;
  migan_pipeline_check(mask_u8, height, width, resolution);
  migan_pipeline_check_bbox(bbox, height, width);
  MIGAN_CHECK(image_chw_u8 && y_nchw && scratch, MIGAN_EINVAL, "null tensor");
  PipeArgs a{};
  a.image = (unsigned char*)image_chw_u8; a.mask = (const unsigned char*)mask_u8; a.y = (const float*)y_nchw;
  a.pooled = (unsigned char*)scratch + align256((size_t)(height + width) * sizeof(int));
  a.H = height; a.W = width; a.R = resolution; a.x_min = bbox[0]; a.x_max = bbox[1]; a.y_min = bbox[2]; a.y_max = bbox[3];
  if (gauss25) {
    for (int i = 0; i < 25; ++i) a.gauss[i] = gauss25[i];
  } else {                                                                         // GaussianSmoothing(3, 5, 1) (:63-85)
    float g[5], sum = 0.0f;
    for (int i = 0; i < 5; ++i) { const float t = ((float)i - 2.0f) / 2.0f; g[i] = (float)(1.0 / std::sqrt(2.0 * M_PI)) * std::exp(-(t * t)); }
    for (int i = 0; i < 25; ++i) { a.gauss[i] = g[i / 5] * g[i % 5]; sum += a.gauss[i]; }
    for (int i = 0; i < 25; ++i) a.gauss[i] /= sum;
  }
  const unsigned grid = (unsigned)cdiv((bbox[1] - bbox[0]) * (bbox[3] - bbox[2]), kThreads);
  rt_check(rt::launch(pipe_maxpool_kernel, a, grid, kThreads, 0, (rt::stream_t)stream), "migan::pipe_maxpool_kernel");
  rt_check(rt::launch(pipe_post_kernel, a, grid, kThreads, 0, (rt::stream_t)stream), "migan::pipe_post_kernel");
  MIGAN_API_END
}


#ifdef MIGAN_PHASE_PROF
// debug builds only: cumulative cycles of thread 0 per phase [prologue, S1, S2, MFMA, acc->LDS, epilogue, -, -, #workgroups]
int migan_prof_read(unsigned long long out[16], int reset) {
  MIGAN_API_BEGIN
  migan::rt_check(rt::prof_read(migan::prof_buffer(), out, 16, reset != 0), "prof read");
  MIGAN_API_END
}
// phase counters of launch `index` of the last migan_forward_timed
int migan_prof_layer(int index, unsigned long long out[16]) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(index >= 0 && index < (int)migan::prof_layers().size(), MIGAN_EINVAL, "no such launch");
  for (int i = 0; i < 16; ++i) out[i] = migan::prof_layers()[index].v[i];
  MIGAN_API_END
}
#endif

// Process-wide tuning knobs (the MIGAN_* environment variables, settable at run time for experiments and tests).
// Affects handles created / re-planned afterwards.
int migan_set_tuning(const char* key, int value) {
  MIGAN_API_BEGIN
  using namespace migan;
  MIGAN_CHECK(key != nullptr, MIGAN_EINVAL, "null key");
  Tuning& t = tuning();
  const std::string k = key;
  if (k == "kc16") t.kc16 = value;
  else if (k == "kc16_minw") t.kc16_minw = std::min(4, std::max(2, value));
  else if (k == "w3") t.w3 = value;
  else if (k == "wide") t.wide = value;
  else if (k == "wide_up") t.wide_up = value;
  else if (k == "small") t.small = value;
  else if (k == "small_max_wgs") t.small_max_wgs = value;
  else if (k == "small_kc") t.small_kc = value;
  else if (k == "small_up32") t.small_up32 = value;
  else if (k == "small_dwfir") t.small_dwfir = value;
  else if (k == "small_ksplit") t.small_ksplit = value;
  else if (k == "nt256") t.nt256 = value != 0;
  else if (k == "persist_min") t.persist_min = std::max(1, value);
  else if (k == "persist_grid") t.persist_grid = std::max(8, value / 8 * 8);
  else if (k == "streams") t.streams = std::min(4, std::max(1, value));
  else if (k == "stagger") t.stagger = value;
  else if (k == "single_b") t.force_single_b = value != 0;
  else if (k == "debug_split") t.debug_split = value != 0;
  else if (k == "stagger_pct") t.stagger_pct = std::min(100, std::max(0, value));
  else if (k == "pipe") t.pipe = value;
  else if (k == "pipe_grid") t.pipe_grid = std::max(8, value / 8 * 8);
  else if (k == "pipe_na") t.pipe_na = value == 8 ? 8 : 4;
  else if (k == "pipe_na8") t.pipe_na8 = value;
  else if (k == "pipe_dna") t.pipe_dna = value;
  else if (k == "pipe_min_tiles") t.pipe_min_tiles = std::max(1, value);
  else if (k == "pipe_min_batch") t.pipe_min_batch = std::max(1, value);
  else if (k == "w2") t.w2 = std::min(2, std::max(0, value));
  else if (k == "w2_min_tiles") t.w2_min_tiles = std::max(1, value);
  else if (k == "w2_pw") t.w2_pw = value != 0;
  else throw Error(MIGAN_EINVAL, "unknown tuning key: " + k);
  MIGAN_API_END
}

const char* migan_last_error(void) { return migan::last_error_ref().c_str(); }
const char* migan_last_kernel(void) { return migan::last_kernel_ref(); }
const char* migan_nan_policy(void) {
#ifdef MIGAN_NAN_CLAMP
  return "clamp";
#else
  return "propagate";
#endif
}
const char* migan_backend(void) { return rt::backend_name(); }
const char* migan_gemm_variant(void) {
  const int g = migan::tuning().gemm;
  return g == 2 ? "f16x2" : (g == 1 ? "bf16x3" : "f32");
}
int migan_version(void) { return 2; }

}  // extern "C"
