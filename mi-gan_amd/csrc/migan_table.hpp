// Table of sepconv_kernel instantiations.  The product library compiles one translation unit per
// (GEMM variant, activation storage format) slice (migan_k_g*s*.hip, in parallel); the CPU test harness
// includes all slices into its single translation unit.  Needs migan_kernels.hpp.
#pragma once

namespace migan {

typedef void (*SepKernelFn)(const SepArgs);

struct KernelEntry {
  int mode, MT, NT, KC;
  bool fromrgb;
  int NI, MINW;
  bool maing, persist;
  int gemmv;
  bool torgb;
  int stv;
  SepKernelFn fn;
  const char* name;     // the symbol as rocprofv3 prints it
};

#define MIGAN_K(MODE, MT, NT, KC, RGB, NI, MINW, MAING, PERSIST, GEMMV, TORGB, STV)                                              \
  {MODE, MT, NT, KC, RGB, NI, MINW, MAING, PERSIST, GEMMV, TORGB, STV,                                                           \
   sepconv_kernel<MODE, MT, NT, KC, RGB, NI, MINW, MAING, PERSIST, GEMMV, TORGB, STV>,                                            \
   "migan::sepconv_kernel<" #MODE ", " #MT ", " #NT ", " #KC ", " #RGB ", " #NI ", " #MINW ", " #MAING ", " #PERSIST ", " #GEMMV ", " #TORGB ", " #STV ">"}

// Every tile geometry the host plan can pick (choose_geo), for one GEMM variant G and one storage format S.
//   MODE 0 plain / 2 FIR-up / 3 pointwise GEMM (second half of FIR-down layers); MT x NT GEMM tile; KC channels per K chunk;
//   RGB fused FromRGB; NI prefetch items; MINW workgroups per CU the kernel is built for; MAING compile-time 8x16 tiles;
//   PERSIST workgroups walk several tiles; TORGB fused ToRGB tail.
#define MIGAN_GEOMETRIES(G, S)                                                                                                   \
  /* plain layers: main 8x16 tiles (NI 6) and small-resolution multi-image tiles (NI 9) */                                      \
  MIGAN_K(0, 128, 128, 32, false, 6, 2, true, false, G, false, S), MIGAN_K(0, 128, 128, 32, false, 9, 2, false, false, G, false, S), \
  MIGAN_K(0, 128, 64, 32, false, 6, 2, true, false, G, false, S), MIGAN_K(0, 128, 64, 32, false, 9, 2, false, false, G, false, S),   \
  MIGAN_K(0, 128, 64, 32, false, 6, 2, true, true, G, false, S),                                                                 \
  /* first encoder layer: fused FromRGB */                                                                                       \
  MIGAN_K(0, 128, 128, 32, true, 6, 2, true, false, G, false, S), MIGAN_K(0, 128, 128, 32, true, 9, 2, false, false, G, false, S),   \
  MIGAN_K(0, 128, 64, 32, true, 6, 2, true, false, G, false, S), MIGAN_K(0, 128, 64, 32, true, 9, 2, false, false, G, false, S),     \
  MIGAN_K(0, 128, 64, 32, true, 6, 2, true, true, G, false, S),                                                                  \
  /* plain layers whose epilogue also produces the running RGB image (CO == NT) */                                              \
  MIGAN_K(0, 128, 128, 32, false, 6, 2, true, false, G, true, S), MIGAN_K(0, 128, 128, 32, false, 9, 2, false, false, G, true, S),   \
  MIGAN_K(0, 128, 64, 32, false, 6, 2, true, false, G, true, S), MIGAN_K(0, 128, 64, 32, false, 9, 2, false, false, G, true, S),     \
  MIGAN_K(0, 64, 256, 32, false, 4, 2, true, false, G, true, S),                                                                 \
  /* FIR-up layers */                                                                                                            \
  MIGAN_K(2, 128, 128, 32, false, 6, 2, true, false, G, false, S), MIGAN_K(2, 128, 128, 32, false, 9, 2, false, false, G, false, S), \
  MIGAN_K(2, 128, 64, 32, false, 6, 2, true, false, G, false, S), MIGAN_K(2, 128, 64, 32, false, 9, 2, false, false, G, false, S),   \
  MIGAN_K(2, 128, 64, 32, false, 6, 2, true, true, G, false, S),                                                                 \
  /* pointwise GEMM: second half of FIR-down layers */                                                                          \
  MIGAN_K(3, 128, 128, 32, false, 4, 2, true, false, G, false, S), MIGAN_K(3, 128, 128, 32, false, 4, 2, false, false, G, false, S), \
  MIGAN_K(3, 128, 128, 32, false, 4, 2, true, true, G, false, S),                                                                \
  MIGAN_K(3, 128, 64, 32, false, 4, 2, true, false, G, false, S), MIGAN_K(3, 128, 64, 32, false, 4, 2, false, false, G, false, S)

// 16-channel K chunks (fp16 GEMM variants only): half the K-loop LDS and prefetch registers of the 32-channel tiles, built for
// 3 or 4 workgroups per CU -- the 64-output-channel layers at 512x512 (encoder first layer, last FIR-up layer, last plain
// layer + ToRGB), which are latency- / issue-bound rather than matrix-bound.
#define MIGAN_GEOMETRIES_KC16(G, S, W)                                                                                             \
  MIGAN_K(0, 128, 64, 16, false, 3, W, true, false, G, true, S), MIGAN_K(0, 128, 64, 16, false, 3, W, true, false, G, false, S),   \
  MIGAN_K(0, 128, 64, 16, false, 3, W, true, true, G, false, S),                                                                 \
  MIGAN_K(0, 128, 64, 16, true, 3, W, true, false, G, false, S), MIGAN_K(0, 128, 64, 16, true, 3, W, true, true, G, false, S),     \
  MIGAN_K(2, 128, 64, 16, false, 3, W, true, false, G, false, S), MIGAN_K(2, 128, 64, 16, false, 3, W, true, true, G, false, S)

// The 64-output-channel main tiles with 32-channel chunks built for 3 workgroups per CU (<= 168 VGPRs, single-buffered 1x1 weight
// tile so that three 49 KB workgroups fit the 160 KB LDS): one more workgroup's loads in flight per CU.  Non-persistent only.
#define MIGAN_GEOMETRIES_W3(G, S)                                                                                                 \
  MIGAN_K(0, 128, 64, 32, false, 6, 3, true, false, G, false, S), MIGAN_K(0, 128, 64, 32, false, 6, 3, true, false, G, true, S),   \
  MIGAN_K(0, 128, 64, 32, true, 6, 3, true, false, G, false, S), MIGAN_K(2, 128, 64, 32, false, 6, 3, true, false, G, false, S)

// 32-row tiles (4x8 pixels, or two 4x4 images) x 128 output channels for launches that would leave most CUs idle: single-image latency,
// the <= 16x16 layers at any batch.  The stage ablation at batch 1 (profiles/r03_batch1_latency_experiments.txt) shows such a launch is
// bound by the instruction stream of its (mostly padded, or lone-on-its-CU) 128-row tile, not by memory: a quarter of the work per
// workgroup, four times the workgroups.  Plain and pointwise layers, f16x2 GEMM, fp32 storage, run-time geometry.
// FIR-up layers: 64-row tiles (8x8 grid of GEMM pixels, 6x6 of them interior) or 32-row tiles (4x8 grid, 2x6 interior) for the smallest
// launches.  KC 64: the same tiles with 64-channel K chunks.
#define MIGAN_GEOMETRIES_SMALL(G, S)                                                                                               \
  MIGAN_K(0, 32, 128, 32, false, 3, 2, false, false, G, false, S), MIGAN_K(3, 32, 128, 32, false, 1, 2, false, false, G, false, S), \
  MIGAN_K(2, 64, 128, 32, false, 4, 2, false, false, G, false, S),                                                                 \
  MIGAN_K(0, 32, 128, 64, false, 5, 2, false, false, G, false, S), MIGAN_K(3, 32, 128, 64, false, 2, 2, false, false, G, false, S), \
  MIGAN_K(2, 64, 128, 64, false, 7, 2, false, false, G, false, S),                                                                 \
  MIGAN_K(2, 32, 128, 32, false, 2, 2, false, false, G, false, S), MIGAN_K(2, 32, 128, 64, false, 4, 2, false, false, G, false, S), \
  /* 32 x 32 K-split tiles: the four waves share one output tile and split the K steps (the smallest launches) */                  \
  MIGAN_K(0, 32, 32, 64, false, 5, 2, false, false, G, false, S), MIGAN_K(3, 32, 32, 64, false, 2, 2, false, false, G, false, S),   \
  MIGAN_K(2, 32, 32, 64, false, 4, 2, false, false, G, false, S)

struct KernelSlice {
  const KernelEntry* entries;
  int n;
};

// sepconv_pipe_kernel instantiations (migan_pipe.hpp; translation unit migan_pipe.hip)
struct PipeEntry {
  int mode, NT, cin;
  bool fromrgb, torgb;
  int ring, na;         // LDS ring depth; waves of the depthwise group (the workgroup has na + 8 waves)
  SepKernelFn fn;
  const char* name;     // the symbol as rocprofv3 prints it
  size_t lds_bytes;
};
struct PipeSlice {
  const PipeEntry* entries;
  int n;
};
// sepconv_pipedown_kernel instantiations: the fused down=2 layer, one workgroup owns all NT = Cout output channels
struct DownEntry {
  int NT, cin, ring, na, nb;
  SepKernelFn fn;
  const char* name;
  size_t lds_bytes;
};
struct DownSlice {
  const DownEntry* entries;
  int n;
};

}  // namespace migan
