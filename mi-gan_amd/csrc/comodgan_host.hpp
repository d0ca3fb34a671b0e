// Host side of the Co-Mod-GAN path of libmigan_hip.so: state_dict schema, the launch sequence of one forward and the
// C ABI of include/comodgan_hip.h.  Included after migan_kernels.hpp, comodgan_kernels.hpp and migan_host.hpp.
//
// Reference being replaced: lib/model_zoo/comodgan.py (Generator.forward :435-455, Encoder :114-204,
// Synthesis :346-420) and the layers of lib/model_zoo/stylegan.py it is built from.
#pragma once

#include "../../include/comodgan_hip.h"

namespace migan {

enum CmRole { CR_CONV_W, CR_CONV_B, CR_DENSE_W, CR_DENSE_B, CR_AFFINE_W, CR_AFFINE_B, CR_RGB_W, CR_RGB_B, CR_FIR, CR_NOISE_CONST,
              CR_NOISE_STRENGTH, CR_W_AVG };

struct CmSlot {
  std::string name;
  int64_t shape[4] = {0, 0, 0, 0};
  int ndim = 0;
  bool is_buffer = false;
  CmRole role = CR_CONV_W;
  const float* ptr = nullptr;
  size_t numel() const {
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    return n;
  }
};

struct CmInfo {
  std::string layer, kernel;
  double flops = 0, mfma_flops = 0, bytes = 0;
};

struct CmDebugTensor {
  std::string name;
  size_t offset = 0;
  int64_t shape[4] = {0, 0, 0, 0};
  int ndim = 0;
};

typedef void (*CmConvFn)(const CmConvArgs);
struct CmConvEntry { int NT, KC, nine, MTI; CmConvFn fn; const char* name; int up4 = 0; };
#define CM_CONV_ENTRY(NT, KC, NIA, NINE, MTI) {NT, KC, NINE, MTI, cm_conv_kernel<NT, KC, NIA, NINE, MTI>, "migan::cm_conv_kernel<" #NT ", " #KC ", " #NIA ", " #NINE ", " #MTI ", false>"}
inline const std::vector<CmConvEntry>& cm_conv_table() {
  static const std::vector<CmConvEntry> t = {
      // 8 x 16 pixel tiles (MTI 2), 64 / 128 output channels: nine-tap unrolled K loop (plain: 10x18-pixel tile, 6 items per thread;
      // strided: 17x33 pixels at 16 channels, 9 items) and the generic tap list (single transposed-convolution phases)
      CM_CONV_ENTRY(64, 32, 6, true, 2), CM_CONV_ENTRY(128, 32, 6, true, 2),
      CM_CONV_ENTRY(64, 16, 9, true, 2), CM_CONV_ENTRY(128, 16, 9, true, 2),
      CM_CONV_ENTRY(64, 32, 6, false, 2), CM_CONV_ENTRY(128, 32, 6, false, 2),
      // 16 x 16 pixel tiles (MTI 4) x 256 output channels: plain 18x18 tile = 11 items; strided 33x33 at 16 channels = 18 items
      CM_CONV_ENTRY(256, 32, 11, true, 4), CM_CONV_ENTRY(256, 16, 18, true, 4), CM_CONV_ENTRY(256, 32, 11, false, 4),
      // all four transposed-convolution phases in one launch (nine taps, four accumulator sets)
      {64, 32, 1, 2, cm_conv_kernel<64, 32, 6, true, 2, true>, "migan::cm_conv_kernel<64, 32, 6, true, 2, true>", 1},
      {128, 32, 1, 2, cm_conv_kernel<128, 32, 6, true, 2, true>, "migan::cm_conv_kernel<128, 32, 6, true, 2, true>", 1},
  };
  return t;
}
inline const CmConvEntry& cm_pick_conv(int NT, int KC, bool nine, int MTI, bool up4 = false) {
  for (const auto& e : cm_conv_table())
    if (e.NT == NT && e.KC == KC && (e.nine != 0) == nine && e.MTI == MTI && (e.up4 != 0) == up4) return e;
  throw Error(MIGAN_EINVAL, "internal: no cm_conv_kernel instantiation for this tile");
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: once per device ordinal (the caller has made
// that device current)
inline void cm_prepare_kernels() {
  static std::vector<char> done;
  static std::mutex mu;
  int dev = 0;
  rt_check(rt::get_device(&dev), "hipGetDevice");
  std::lock_guard<std::mutex> lock(mu);
  if (dev >= 0 && dev < (int)done.size() && done[dev]) return;
  for (const auto& e : cm_conv_table()) rt_check(rt::allow_dynamic_lds((const void*)e.fn, 160 * 1024), "hipFuncSetAttribute");
  if (dev >= (int)done.size()) done.resize(dev + 1, 0);
  if (dev >= 0) done[dev] = 1;
}

enum : int { CM_CONV_NORMAL = 0, CM_CONV_DOWN = 1, CM_CONV_UP = 2, CM_CONV_UP4 = 3 };

}  // namespace migan

struct comodgan_handle {
  comodgan_config cfg{};
  int device = 0;
  bool committed = false, debug = false;
  std::vector<migan::CmSlot> slots;
  std::vector<migan::CmInfo> infos;
  std::vector<migan::CmDebugTensor> debug_tensors;
  std::vector<rt::event_t> events;
  int planned_batch = 0;
  size_t planned_need = 0;       // workspace bytes of planned_batch (0 = not planned)
  int trunc_cutoff = -1;       // comodgan_set_truncation_cutoff: -1 = None (every row of ws truncated), else rows [0, cutoff)
  // comodgan_assume_static_weights: skip the per-forward weight preparation while nothing it depends on has changed
  bool static_weights = false;
  const void* prepared_ws = nullptr;
  rt::stream_t prepared_stream{};   // the planes are only valid for work ordered after their preparation: same stream
  unsigned long long weights_epoch = 1, prepared_epoch = 0;
  // the mapping network (8 small dense layers on z only) runs on a library-owned stream beside the encoder
  rt::stream_t map_stream{};
  rt::event_t ev_fork{}, ev_map{};
  bool side_ready = false;

  int channels(int res) const { return std::min(cfg.ch_base / res, cfg.ch_max); }
  int slot_index(const std::string& n) const {
    for (size_t i = 0; i < slots.size(); ++i)
      if (slots[i].name == n) return (int)i;
    return -1;
  }
  const float* W(const std::string& n) const {
    const int i = slot_index(n);
    MIGAN_CHECK(i >= 0, MIGAN_EINVAL, "internal: no such weight " + n);
    return slots[i].ptr;
  }
  void add_slot(const std::string& n, std::initializer_list<int64_t> shp, bool is_buf, migan::CmRole role) {
    migan::CmSlot s;
    s.name = n;
    s.ndim = (int)shp.size();
    int i = 0;
    for (auto v : shp) s.shape[i++] = v;
    s.is_buffer = is_buf;
    s.role = role;
    slots.push_back(s);
  }
  void build_schema();
  size_t noise_floats() const {
    size_t n = 16;
    for (int res = 8; res <= cfg.resolution; res *= 2) n += 2 * (size_t)res * res;
    return n;
  }
  // One walk over the network: dry = only record launch infos / debug tensors and size the workspace; otherwise launch.
  size_t walk(int batch, const float* x, const float* z, float* y, float psi, int noise_mode, const float* noise, void* ws,
              rt::stream_t stream, bool dry, float* ms, int n_ms);
};

// mirror of mi-gan_amd/comodgan_schema.py::entries
inline void comodgan_handle::build_schema() {
  using namespace migan;
  slots.clear();
  const int wl = cfg.w_dim + cfg.w0_dim;
  for (int i = 0; i < cfg.map_layers; ++i) {
    const std::string p = "mapping.fc" + std::to_string(i);
    add_slot(p + ".weight", {cfg.w_dim, i == 0 ? cfg.z_dim : cfg.w_dim}, false, CR_DENSE_W);
    add_slot(p + ".bias", {cfg.w_dim}, false, CR_DENSE_B);
  }
  add_slot("mapping.w_avg", {cfg.w_dim}, true, CR_W_AVG);
  const int c4 = channels(4);
  auto add_syn_layer = [&](const std::string& p, int cin, int cout, int res, bool up) {
    add_slot(p + ".weight", {cout, cin, 3, 3}, false, CR_CONV_W);
    add_slot(p + ".bias", {cout}, false, CR_CONV_B);
    add_slot(p + ".noise_strength", {}, false, CR_NOISE_STRENGTH);
    if (up) add_slot(p + ".resample_filter", {4, 4}, true, CR_FIR);
    add_slot(p + ".noise_const", {res, res}, true, CR_NOISE_CONST);
    add_slot(p + ".affine.weight", {cin, wl}, false, CR_AFFINE_W);
    add_slot(p + ".affine.bias", {cin}, false, CR_AFFINE_B);
  };
  auto add_torgb = [&](const std::string& p, int c) {
    add_slot(p + ".weight", {3, c, 1, 1}, false, CR_RGB_W);
    add_slot(p + ".bias", {3}, false, CR_RGB_B);
    add_slot(p + ".affine.weight", {c, wl}, false, CR_AFFINE_W);
    add_slot(p + ".affine.bias", {c}, false, CR_AFFINE_B);
  };
  add_slot("synthesis.b4.fc.weight", {c4 * 16, cfg.w0_dim}, false, CR_DENSE_W);
  add_slot("synthesis.b4.fc.bias", {c4 * 16}, false, CR_DENSE_B);
  add_syn_layer("synthesis.b4.conv", c4, c4, 4, true);     // conv2d_layer default resample_filter: the buffer exists (stylegan.py:207,214)
  add_torgb("synthesis.b4.torgb", c4);
  for (int res = 8; res <= cfg.resolution; res *= 2) {
    const std::string b = bname("synthesis", res);
    const int ci = channels(res / 2), co = channels(res);
    add_slot(b + ".resample_filter", {4, 4}, true, CR_FIR);
    add_syn_layer(b + ".conv0", ci, co, res, true);
    add_syn_layer(b + ".conv1", co, co, res, false);
    add_torgb(b + ".torgb", co);
  }
  for (int res = cfg.resolution; res > 4; res /= 2) {
    const std::string b = bname("encoder", res);
    const int c = channels(res), cn = channels(res / 2);
    add_slot(b + ".resample_filter", {4, 4}, true, CR_FIR);
    if (res == cfg.resolution) {
      add_slot(b + ".fromrgb.weight", {c, 4, 1, 1}, false, CR_CONV_W);
      add_slot(b + ".fromrgb.bias", {c}, false, CR_CONV_B);
    }
    add_slot(b + ".conv0.weight", {c, c, 3, 3}, false, CR_CONV_W);
    add_slot(b + ".conv0.bias", {c}, false, CR_CONV_B);
    add_slot(b + ".conv1.weight", {cn, c, 3, 3}, false, CR_CONV_W);
    add_slot(b + ".conv1.bias", {cn}, false, CR_CONV_B);
    add_slot(b + ".conv1.resample_filter", {4, 4}, true, CR_FIR);
  }
  add_slot("encoder.b4.conv.weight", {c4, c4, 3, 3}, false, CR_CONV_W);
  add_slot("encoder.b4.conv.bias", {c4}, false, CR_CONV_B);
  add_slot("encoder.b4.fc.weight", {cfg.w0_dim, c4 * 16}, false, CR_DENSE_W);
  add_slot("encoder.b4.fc.bias", {cfg.w0_dim}, false, CR_DENSE_B);
}

inline size_t comodgan_handle::walk(int batch, const float* x, const float* z, float* y, float psi, int noise_mode,
                                    const float* noise, void* ws, rt::stream_t stream, bool dry, float* ms, int n_ms) {
  using namespace migan;
  const int R = cfg.resolution, B = batch;
  const bool timed = ms != nullptr;
  size_t cursor = 0;
  int nlaunch = 0;
  if (dry) {
    infos.clear();
    debug_tensors.clear();
  }
  char* base = static_cast<char*>(ws);
  auto alloc = [&](size_t bytes) -> float* {
    const size_t off = cursor;
    cursor += (bytes + 255) & ~(size_t)255;
    return reinterpret_cast<float*>(base + off);          // dry: never dereferenced
  };
  auto reg_debug = [&](const std::string& name, const float* p, std::initializer_list<int64_t> shp) {
    if (!dry || !debug) return;
    CmDebugTensor t;
    t.name = name;
    t.offset = (size_t)(reinterpret_cast<const char*>(p) - base);
    t.ndim = (int)shp.size();
    int i = 0;
    for (auto v : shp) t.shape[i++] = v;
    debug_tensors.push_back(t);
  };
  bool skip_launch = false;      // set around the weight-preparation launches when their results in the workspace are still valid
  rt::stream_t cur_stream = stream;   // the stream emit() launches on (the mapping network moves to map_stream)
  auto emit = [&](const std::string& layer, const char* kname, double flops, double mfma, double bytes, auto kernel, const auto& args,
                  unsigned grid, size_t lds) {
    if (dry) {
      CmInfo inf;
      inf.layer = layer; inf.kernel = kname; inf.flops = flops; inf.mfma_flops = mfma; inf.bytes = bytes;
      infos.push_back(inf);
    } else {
      if (timed) rt_check(rt::event_record(events[2 * nlaunch], stream), "hipEventRecord");
      if (!skip_launch) rt_check(rt::launch(kernel, args, grid, kThreads, lds, cur_stream), kname);
      if (timed) rt_check(rt::event_record(events[2 * nlaunch + 1], stream), "hipEventRecord");
#ifdef MIGAN_PHASE_PROF
      if (timed) {
        if ((int)prof_layers().size() <= nlaunch) prof_layers().resize(nlaunch + 1);
        rt_check(rt::prof_read(prof_buffer(), prof_layers()[nlaunch].v, 16, true), "prof read");
      }
#endif
    }
    ++nlaunch;
  };
  auto grid1d = [](size_t items) -> unsigned { return (unsigned)std::min<size_t>((items + kThreads - 1) / kThreads, 1u << 20); };

  // ---------------------------------------------------------------- weight preparation (every forward: weights are read in place)
  struct ConvW { std::string name; int co, ci; unsigned short* planes; float* amax; float* wsq; float* wn2; bool mod; };
  std::vector<ConvW> convs;
  auto add_conv = [&](const std::string& name, int co, int ci, bool mod) {
    ConvW c;
    c.name = name; c.co = co; c.ci = ci; c.mod = mod;
    const size_t plane_bytes = (size_t)2 * 9 * ci * co * sizeof(unsigned short);
    unsigned short* hdr = reinterpret_cast<unsigned short*>(alloc(16 + plane_bytes));
    c.planes = hdr + kSplitHeader;
    c.amax = alloc((size_t)co * 4);
    c.wsq = mod ? alloc((size_t)co * ci * 4) : nullptr;
    c.wn2 = mod ? alloc((size_t)co * 4) : nullptr;
    convs.push_back(c);
  };
  for (int res = R; res > 4; res /= 2) {
    add_conv(bname("encoder", res) + ".conv0", channels(res), channels(res), false);
    add_conv(bname("encoder", res) + ".conv1", channels(res / 2), channels(res), false);
  }
  add_conv("encoder.b4.conv", channels(4), channels(4), false);
  add_conv("synthesis.b4.conv", channels(4), channels(4), true);
  for (int res = 8; res <= R; res *= 2) {
    add_conv(bname("synthesis", res) + ".conv0", channels(res), channels(res / 2), true);
    add_conv(bname("synthesis", res) + ".conv1", channels(res), channels(res), true);
  }
  auto conv_of = [&](const std::string& name) -> const ConvW& {
    for (const auto& c : convs)
      if (c.name == name) return c;
    throw Error(MIGAN_EINVAL, "internal: no conv " + name);
  };
  skip_launch = !dry && static_weights && prepared_ws == ws && prepared_epoch == weights_epoch && prepared_stream == stream;
  if (!dry && !skip_launch) prepared_ws = nullptr;      // marked prepared again only after every preparation launch succeeded
  for (const auto& c : convs) {
    CmWprepArgs q{};
    q.w = dry ? nullptr : W(c.name + ".weight"); q.amax = c.amax; q.wsq = c.wsq; q.wn2 = c.wn2; q.CO = c.co; q.CI = c.ci;
    emit(c.name + ".wprep", "migan::cm_wprep_kernel", 0, 0, 4.0 * c.co * c.ci * (c.mod ? 10 : 9) / B, cm_wprep_kernel, q, (unsigned)c.co,
         8 * sizeof(float));
    CmSplitArgs a{};
    a.src = q.w; a.amax = c.amax; a.dst = c.planes; a.CO = c.co; a.CI = c.ci;
    emit(c.name + ".split", "migan::cm_split_conv_kernel", 0, 0, 8.0 * c.co * c.ci * 9 / B, cm_split_conv_kernel, a,
         grid1d((size_t)c.co * c.ci), 4 * sizeof(float));
  }
  skip_launch = false;
  if (!dry) {
    prepared_ws = ws;
    prepared_epoch = weights_epoch;
    prepared_stream = stream;
  }

  // ---------------------------------------------------------------- helpers for the layers
  auto dense = [&](const std::string& layer, const float* xin, const float* xin2, int K, int K1, const std::string& wname, int O,
                   float* out, float lr_multi, bool act, bool norm, int in_c, int out_c, const float* add, const float* lerp0,
                   float* out_raw = nullptr) {
    // (the kernel's NHWC-bottleneck path walks channels x 16 positions of ONE input tensor)
    MIGAN_CHECK(in_c == 0 || (K == 16 * in_c && K1 == K && xin2 == nullptr), MIGAN_EINVAL, "internal: bottleneck dense layer must read one [N][16][C] tensor");
    CmDenseArgs a{};
    a.x = xin; a.x2 = xin2; a.w = dry ? nullptr : W(wname + ".weight"); a.b = dry ? nullptr : W(wname + ".bias");
    a.add = add; a.lerp0 = lerp0; a.y = out; a.y_raw = lerp0 ? out_raw : nullptr;
    a.wgain = lr_multi / std::sqrt((float)K); a.bgain = lr_multi; a.psi = psi;
    a.N = B; a.K = K; a.K1 = K1; a.O = O; a.act = act; a.norm = norm; a.in_c = in_c; a.out_c = out_c;
    emit(layer, "migan::cm_dense_kernel", 2.0 * K * O, 0, 4.0 * ((double)K * O / B + K + O), cm_dense_kernel, a, (unsigned)cdiv(O, 8), 0);
  };
  auto conv = [&](const std::string& layer, int mode, int ey, int ex, const float* xin, float* out, const float* skip, const ConvW& cw,
                  const float* sa, const float* coef, float cgain, const float* bias, const float* nz, const float* nstr,
                  long long nz_bstride, int H, int Wd, int HO, int WO, bool raw) {
    CmConvArgs a{};
    a.x = xin; a.y = out; a.skip = skip; a.wsplit = cw.planes; a.sa = sa; a.coef = coef; a.bias = bias;
    a.noise = nz; a.noise_strength = nstr; a.noise_bstride = nz_bstride;
    a.a_scale = kCmF16Top / kCmInBound; a.cgain = cgain / a.a_scale;
    a.B = B; a.H = H; a.W = Wd; a.CI = cw.ci; a.CO = cw.co; a.HO = HO; a.WO = WO;
    a.oy_mul = 1; a.ox_mul = 1; a.oy_add = 0; a.ox_add = 0; a.raw = raw;
    a.prof = prof_buffer();
    // tile: 16 x 16 grid pixels (MTI 4) where the layer is large enough, else 8 x 16
    const char* mti_env = std::getenv("COMODGAN_MTI");                     // experiments / tests: force 2 or 4
    const int ghn = mode == CM_CONV_UP4 ? H + 1 : (mode == CM_CONV_UP ? H + (ey == 0) : HO);
    const int gwn = mode == CM_CONV_UP4 ? Wd + 1 : (mode == CM_CONV_UP ? Wd + (ex == 0) : WO);
    // 16 x 16 pixels x 256 channels per workgroup (one wave per SIMD, 128 x 128 wave tiles) pays where Cout allows it and the
    // launch still has two workgroups per CU (measured: +5..15 % at >= 64^2 with 256/512 channels, a loss on smaller launches
    // and with 128- or 64-column tiles)
    const size_t wgs16 = (size_t)cdiv(ghn, 16) * cdiv(gwn, 16) * B * (cw.co / 256);
    int MTI = (cw.co % 256 == 0 && std::min(ghn, gwn) >= 16 && wgs16 >= 512) ? 4 : 2;
    if (mti_env && std::atoi(mti_env) == 2) MTI = 2;
    if (mti_env && std::atoi(mti_env) == 4 && cw.co % 256 == 0) MTI = 4;     // the 16 x 16 tiles exist with 256 columns only
    if (mode == CM_CONV_UP4) MTI = 2;
    const int GH = 4 * MTI;
    if (mode == CM_CONV_NORMAL) {
      a.stride = 1;
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) { a.dy[a.ntaps] = ky - 1; a.dx[a.ntaps] = kx - 1; a.wtap[a.ntaps] = ky * 3 + kx; ++a.ntaps; }
      a.dymin = -1; a.dxmin = -1; a.IH = GH + 2; a.IW = 18; a.GHn = HO; a.GWn = WO;
    } else if (mode == CM_CONV_DOWN) {
      a.stride = 2;
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) { a.dy[a.ntaps] = ky; a.dx[a.ntaps] = kx; a.wtap[a.ntaps] = ky * 3 + kx; ++a.ntaps; }
      a.dymin = 0; a.dxmin = 0; a.IH = 2 * GH + 1; a.IW = 33; a.GHn = HO; a.GWn = WO;
    } else if (mode == CM_CONV_UP4) {
      // every phase of conv_transpose2d(stride 2) at once: tap (ky, kx) feeds the phase (ky == 1, kx == 1) from x[g - (k == 2)]
      a.stride = 1;
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) { a.dy[a.ntaps] = ky == 2 ? -1 : 0; a.dx[a.ntaps] = kx == 2 ? -1 : 0; a.wtap[a.ntaps] = ky * 3 + kx; ++a.ntaps; }
      a.dymin = -1; a.dxmin = -1; a.IH = GH + 1; a.IW = 17; a.GHn = H + 1; a.GWn = Wd + 1;
      a.oy_mul = 2; a.ox_mul = 2;
    } else {
      // output phase (ey, ex) of conv_transpose2d(stride 2): raw[2g + e] = sum over taps k with k = e (mod 2) of x[g - (k - e) / 2] w[k]
      a.stride = 1;
      const int nky = ey == 0 ? 2 : 1, nkx = ex == 0 ? 2 : 1;
      for (int iy = 0; iy < nky; ++iy)
        for (int ix = 0; ix < nkx; ++ix) {
          const int ky = ey == 0 ? 2 * iy : 1, kx = ex == 0 ? 2 * ix : 1;
          a.dy[a.ntaps] = -(ky - ey) / 2; a.dx[a.ntaps] = -(kx - ex) / 2; a.wtap[a.ntaps] = ky * 3 + kx; ++a.ntaps;
        }
      a.dymin = ey == 0 ? -1 : 0; a.dxmin = ex == 0 ? -1 : 0;
      a.IH = GH + (ey == 0); a.IW = 16 + (ex == 0);
      a.GHn = H + (ey == 0); a.GWn = Wd + (ex == 0);
      a.oy_mul = 2; a.ox_mul = 2; a.oy_add = ey; a.ox_add = ex;
    }
    int NT = (MTI == 4) ? 256 : ((cw.co % 128 == 0) ? 128 : 64);
    if (mode == CM_CONV_UP4) {
      const char* e = std::getenv("COMODGAN_UP4_NT");
      NT = (e && std::atoi(e) == 128 && cw.co % 128 == 0) ? 128 : 64;
    }
    const int KC = mode == CM_CONV_DOWN ? 16 : 32;          // the (2GH+1)x33-pixel tile of the strided mode is staged 16 channels at a time
    a.tiles_y = cdiv(a.GHn, GH); a.tiles_x = cdiv(a.GWn, 16); a.nchunks = cw.co / NT;
    const size_t pitch = (size_t)4 * KC + 16;                // LDS row: both fp16 planes of KC channels + 16 bytes of padding
    const size_t a_bytes = (size_t)a.IH * a.IW * pitch;
    a.off_b = (int)((a_bytes + 127) & ~(size_t)127);
    const size_t lds = std::max<size_t>((size_t)a.off_b + (size_t)2 * NT * pitch, (size_t)64 * (NT + 4) * 4);
    MIGAN_CHECK(lds <= 160 * 1024, MIGAN_EINVAL, "internal: LDS tile exceeds 160 KiB");
    const int nia = KC == 32 ? (MTI == 4 ? 11 : 6) : (MTI == 4 ? 18 : 9);
    MIGAN_CHECK(a.IH * a.IW * (KC / 4) <= 256 * nia, MIGAN_EINVAL, "internal: input tile exceeds the prefetch registers");
    const unsigned grid = (unsigned)((size_t)a.tiles_x * a.tiles_y * B * a.nchunks);
    // four-phase launch: the same multiply-adds and output pixels as the four single-phase launches together
    const double mf = mode == CM_CONV_UP4 ? 2.0 * cw.ci * cw.co * ((double)(H + 1) * (Wd + 1) * 4 + 2.0 * (H + 1) * Wd + 2.0 * H * (Wd + 1) + (double)H * Wd)
                                          : 2.0 * cw.ci * cw.co * a.ntaps * (double)a.GHn * a.GWn;
    const double by = mode == CM_CONV_UP4 ? 4.0 * ((double)cw.ci * H * Wd + (double)cw.co * (2.0 * H + 1) * (2.0 * Wd + 1))
                                          : 4.0 * ((double)cw.ci * H * Wd + (double)cw.co * a.GHn * a.GWn * (skip ? 2 : 1));
    const bool nine = a.ntaps == 9;
    MIGAN_CHECK(!nine || (cw.ci / KC) % 2 == 0, MIGAN_EINVAL, "internal: the nine-tap kernel walks channel chunks in pairs");
    MIGAN_CHECK(nine || KC == 32, MIGAN_EINVAL, "internal: no generic-tap-list kernel with 16-channel chunks");
    const CmConvEntry& ke = cm_pick_conv(NT, KC, nine, MTI, mode == CM_CONV_UP4);
    emit(layer, ke.name, mf, mf, by, ke.fn, a, grid, lds);
  };

  // ---------------------------------------------------------------- buffers
  size_t max_act = 0, max_tmp = 0;
  for (int res = 4; res <= R; res *= 2) {
    max_act = std::max(max_act, (size_t)res * res * channels(res));
    max_tmp = std::max(max_tmp, (size_t)(res + 1) * (res + 1) * channels(res));
  }
  float* bufA = debug ? nullptr : alloc(max_act * B * 4);
  float* bufB = debug ? nullptr : alloc(max_act * B * 4);
  float* tmp = alloc(max_tmp * B * 4);
  float* img[2] = {debug ? nullptr : alloc((size_t)3 * R * R * B * 4), debug ? nullptr : alloc((size_t)3 * R * R * B * 4)};
  auto act_out = [&](const std::string& name, float* pingpong, int res, int c) -> float* {
    float* p = debug ? alloc((size_t)res * res * c * B * 4) : pingpong;
    reg_debug(name, p, {B, res, res, c});
    return p;
  };
  std::vector<float*> feat(16, nullptr);
  for (int res = R; res >= 4; res /= 2) feat[ilog2(res)] = alloc((size_t)res * res * channels(res) * B * 4);

  // ---------------------------------------------------------------- mapping (stylegan.py:396-439)
  float* m0 = alloc((size_t)B * cfg.w_dim * 4);
  float* m1 = alloc((size_t)B * cfg.w_dim * 4);
  float* wlat = alloc((size_t)B * cfg.w_dim * 4);
  // truncation_cutoff (stylegan.py:436-437): only ws[:, :cutoff] are pulled towards w_avg; the layers reading later rows get the raw w
  // (the buffer is part of the workspace whenever a cutoff is set, whatever psi a forward passes: the planned size must not depend on it)
  float* wraw = trunc_cutoff >= 0 ? alloc((size_t)B * cfg.w_dim * 4) : nullptr;
  const bool cut = psi != 1.0f && trunc_cutoff >= 0;
  // The mapping network depends on z only and its eight launches are latency-bound (27 us each, 64 workgroups): they run on the
  // handle's own stream while the caller's stream goes on with the encoder; the affine layers (first reader of w) wait for it.
  // Ordering is by events only.  Timed / debug walks keep everything on the caller's stream.
  const bool side = !dry && !timed && !debug;
  if (side) {
    if (!side_ready) {
      rt_check(rt::stream_create(&map_stream), "hipStreamCreate");
      rt_check(rt::event_create_sync(&ev_fork), "hipEventCreate");
      rt_check(rt::event_create_sync(&ev_map), "hipEventCreate");
      side_ready = true;
    }
    rt_check(rt::event_record(ev_fork, stream), "hipEventRecord");          // after everything already queued by the caller (z, earlier forwards)
    rt_check(rt::stream_wait_event(map_stream, ev_fork), "hipStreamWaitEvent");
    cur_stream = map_stream;
  }
  {
    const float* cur = z;
    for (int i = 0; i < cfg.map_layers; ++i) {
      const bool last = i + 1 == cfg.map_layers;
      float* out = last ? wlat : ((i & 1) ? m1 : m0);
      dense("mapping.fc" + std::to_string(i), cur, nullptr, i == 0 ? cfg.z_dim : cfg.w_dim, i == 0 ? cfg.z_dim : cfg.w_dim,
            "mapping.fc" + std::to_string(i), cfg.w_dim, out, 0.01f, true, i == 0, 0, 0, nullptr,
            (last && psi != 1.0f) ? (dry ? nullptr : W("mapping.w_avg")) : nullptr, (last && cut) ? wraw : nullptr);
      cur = out;
    }
    reg_debug("mapping", wlat, {B, cfg.w_dim});
  }
  if (side) {
    rt_check(rt::event_record(ev_map, map_stream), "hipEventRecord");
    cur_stream = stream;
  }

  // ---------------------------------------------------------------- encoder (comodgan.py:192-204)
  float* w0 = alloc((size_t)B * cfg.w0_dim * 4);
  {
    const int c0 = channels(R);
    float* cur = debug ? alloc((size_t)R * R * c0 * B * 4) : bufA;
    {
      CmFromRgbArgs a{};
      a.x = x; a.w = dry ? nullptr : W(bname("encoder", R) + ".fromrgb.weight"); a.b = dry ? nullptr : W(bname("encoder", R) + ".fromrgb.bias");
      a.y = cur; a.wgain = 0.5f; a.B = B; a.R = R; a.C = c0;
      emit(bname("encoder", R) + ".fromrgb", "migan::cm_fromrgb_kernel", 2.0 * 4 * c0 * R * R, 0, 4.0 * (4 + c0) * R * R, cm_fromrgb_kernel, a,
           grid1d((size_t)B * R * R * (c0 / 4) / 8), 0);      // 8 pixels per thread: the weights are read once per thread
    }
    for (int res = R; res > 4; res /= 2) {
      const std::string b = bname("encoder", res);
      const int c = channels(res), cn = channels(res / 2);
      const ConvW& w0c = conv_of(b + ".conv0");
      const ConvW& w1c = conv_of(b + ".conv1");
      float* f = feat[ilog2(res)];
      reg_debug(b + ".conv0", f, {B, res, res, c});
      conv(b + ".conv0", CM_CONV_NORMAL, 0, 0, cur, f, nullptr, w0c, nullptr, nullptr, 1.0f / std::sqrt(9.0f * c),
           dry ? nullptr : W(b + ".conv0.bias"), nullptr, nullptr, 0, res, res, res, res, false);
      {
        CmFirArgs a{};
        a.x = f; a.y = tmp; a.B = B; a.H = res; a.W = res; a.C = c; a.HO = res + 1; a.WO = res + 1; a.pad = 2; a.fs = 0.125f;
        emit(b + ".conv1.fir", "migan::cm_fir_kernel<0>", 2.0 * 16 * c * (res + 1) * (res + 1), 0, 4.0 * c * (2.0 * res * res + 2 * res + 1),
             cm_fir_kernel<0>, a, grid1d((size_t)B * cdiv(res + 1, 2) * cdiv(res + 1, 4) * (c / 4)), 0);
      }
      float* out = act_out(b + ".conv1", bufA, res / 2, cn);
      conv(b + ".conv1", CM_CONV_DOWN, 0, 0, tmp, out, nullptr, w1c, nullptr, nullptr, 1.0f / std::sqrt(9.0f * c),
           dry ? nullptr : W(b + ".conv1.bias"), nullptr, nullptr, 0, res + 1, res + 1, res / 2, res / 2, false);
      cur = out;
    }
    const int c4 = channels(4);
    reg_debug("encoder.b4.conv", feat[2], {B, 4, 4, c4});
    conv("encoder.b4.conv", CM_CONV_NORMAL, 0, 0, cur, feat[2], nullptr, conv_of("encoder.b4.conv"), nullptr, nullptr,
         1.0f / std::sqrt(9.0f * c4), dry ? nullptr : W("encoder.b4.conv.bias"), nullptr, nullptr, 0, 4, 4, 4, 4, false);
    // fc over feat.flatten(1) of the NCHW tensor (comodgan.py:106): the kernel permutes the K index to our NHWC storage
    dense("encoder.b4.fc", feat[2], nullptr, c4 * 16, c4 * 16, "encoder.b4.fc", cfg.w0_dim, w0, 1.0f, true, false, c4, 0, nullptr, nullptr);
    reg_debug("encoder.b4.fc", w0, {B, cfg.w0_dim});
  }

  // ---------------------------------------------------------------- synthesis (comodgan.py:395-420)
  const int wl = cfg.w_dim + cfg.w0_dim;
  size_t noise_off = 0;                     // floats per image into the caller's random-noise blob
  // every affine layer (styles = affine(cat([w, w0])), stylegan.py:282,337) in one launch, ahead of the synthesis blocks
  struct Affine { std::string name; int c; float* styles; int widx; };   // widx: the row of ws the layer reads (comodgan.py:399-405)
  std::vector<Affine> affines;
  {
    auto add_aff = [&](const std::string& p, int c, int widx) { affines.push_back({p, c, alloc((size_t)B * c * 4), widx}); };
    add_aff("synthesis.b4.conv", channels(4), 0);
    add_aff("synthesis.b4.torgb", channels(4), 1);
    int widx = 1;
    for (int res = 8; res <= R; res *= 2) {
      add_aff(bname("synthesis", res) + ".conv0", channels(res / 2), widx);
      add_aff(bname("synthesis", res) + ".conv1", channels(res), widx + 1);
      add_aff(bname("synthesis", res) + ".torgb", channels(res), widx + 2);
      widx += 2;
    }
    MIGAN_CHECK((int)affines.size() <= kCmMaxAffine, MIGAN_EINVAL, "internal: too many affine layers");
    CmDenseMultiArgs a{};
    double fl = 0;
    int blk = 0;
    for (const auto& af : affines) {
      a.w[a.njobs] = dry ? nullptr : W(af.name + ".affine.weight");
      a.b[a.njobs] = dry ? nullptr : W(af.name + ".affine.bias");
      a.y[a.njobs] = af.styles; a.O[a.njobs] = af.c; a.blk0[a.njobs] = blk;
      if (cut && af.widx >= trunc_cutoff) a.alt_mask |= 1ull << a.njobs;
      blk += cdiv(af.c, 8);
      fl += 2.0 * wl * af.c;
      ++a.njobs;
    }
    a.blk0[a.njobs] = blk;
    a.x = wlat; a.x_alt = wraw; a.x2 = w0; a.wgain = 1.0f / std::sqrt((float)wl); a.N = B; a.K = wl; a.K1 = cfg.w_dim;
    if (side) rt_check(rt::stream_wait_event(stream, ev_map), "hipStreamWaitEvent");     // w from the mapping stream
    emit("synthesis.affine", "migan::cm_dense_multi_kernel", fl, 0, 2.0 * fl / B, cm_dense_multi_kernel, a, (unsigned)blk, 0);
  }
  auto styles_of = [&](const std::string& p) -> float* {
    for (const auto& af : affines)
      if (af.name == p) return af.styles;
    throw Error(MIGAN_EINVAL, "internal: no affine " + p);
  };
  // every style computation (input scales + demodulation coefficients of the modulated convs, modulated ToRGB weights) in one
  // launch ahead of the synthesis blocks: inputs are the affine outputs above and the per-tensor weight statistics
  struct Mod { float* sa; float* coef; };
  struct ModEntry { std::string name; Mod m; float* wm; };
  std::vector<ModEntry> mods;
  {
    CmStyleMultiArgs sm{};
    int blk = 0;
    size_t lds = 0;
    double fl = 0, by = 0;
    auto add_demod = [&](const std::string& p) {
      const ConvW& cw = conv_of(p);
      MIGAN_CHECK(sm.njobs < kCmMaxStyle, MIGAN_EINVAL, "internal: too many modulated layers");
      Mod m{alloc((size_t)B * cw.ci * 4), alloc((size_t)B * cw.co * 4)};
      CmStyleArgs& a = sm.job[sm.njobs];
      a.styles = styles_of(p); a.wsq = cw.wsq; a.wn2 = cw.wn2; a.sa = m.sa; a.coef = m.coef; a.B = B; a.CI = cw.ci; a.CO = cw.co; a.demod = 1;
      sm.blk0[sm.njobs++] = blk;
      blk += B * cdiv(cw.co, kCmStyleSlice);
      lds = std::max(lds, (size_t)(cw.ci + 8) * 4);
      fl += 2.0 * cw.ci * cw.co; by += 4.0 * ((double)cw.ci * cw.co + cw.ci + cw.co);
      mods.push_back({p, m, nullptr});
    };
    auto add_rgb = [&](const std::string& p, int c) {
      MIGAN_CHECK(sm.njobs < kCmMaxStyle, MIGAN_EINVAL, "internal: too many modulated layers");
      float* wm = alloc((size_t)B * 3 * c * 4);
      CmStyleArgs& a = sm.job[sm.njobs];
      a.styles = styles_of(p); a.w = dry ? nullptr : W(p + ".weight"); a.wm = wm; a.wgain = 1.0f / std::sqrt((float)c); a.B = B; a.CI = c; a.CO = 3; a.demod = 0;
      sm.blk0[sm.njobs++] = blk;
      blk += B;
      lds = std::max(lds, (size_t)(c + 8) * 4);
      fl += 6.0 * c; by += 4.0 * 7 * c;
      mods.push_back({p, Mod{nullptr, nullptr}, wm});
    };
    add_demod("synthesis.b4.conv");
    add_rgb("synthesis.b4.torgb", channels(4));
    for (int res = 8; res <= R; res *= 2) {
      add_demod(bname("synthesis", res) + ".conv0");
      add_demod(bname("synthesis", res) + ".conv1");
      add_rgb(bname("synthesis", res) + ".torgb", channels(res));
    }
    sm.blk0[sm.njobs] = blk;
    emit("synthesis.styles", "migan::cm_style_multi_kernel", fl, 0, by, cm_style_multi_kernel, sm, (unsigned)blk, lds);
  }
  auto mod_of = [&](const std::string& p) -> const ModEntry& {
    for (const auto& e : mods)
      if (e.name == p) return e;
    throw Error(MIGAN_EINVAL, "internal: no style job " + p);
  };
  auto style_demod = [&](const std::string& p, const ConvW&) -> Mod { return mod_of(p).m; };
  auto noise_of = [&](const std::string& p, int res, const float*& nz, long long& bstride) {
    nz = nullptr; bstride = 0;
    if (noise_mode == COMODGAN_NOISE_CONST) nz = dry ? reinterpret_cast<const float*>(base) : W(p + ".noise_const");
    else if (noise_mode == COMODGAN_NOISE_RANDOM) { nz = noise + noise_off * (size_t)B; bstride = (long long)res * res; }
    noise_off += (size_t)res * res;
  };
  auto torgb = [&](const std::string& p, const float* xin, int res, int c, const float* prev, float* out) {
    float* wm = mod_of(p).wm;
    CmRgbArgs a{};
    a.x = xin; a.wm = wm; a.bias = dry ? nullptr : W(p + ".bias"); a.img_prev = prev; a.img_out = out; a.B = B; a.H = res; a.W = res; a.C = c;
    const double fl = 2.0 * 3 * c * res * res, by = 4.0 * ((double)c * res * res + 3.75 * res * res);
    const auto grid_of = [&](int lpp) { return (unsigned)(((size_t)B * res * res * lpp + kThreads - 1) / kThreads); };
    if (c <= 64) emit(p, "migan::cm_torgb_kernel<4>", fl, 0, by, cm_torgb_kernel<4>, a, grid_of(4), 0);
    else if (c <= 128) emit(p, "migan::cm_torgb_kernel<8>", fl, 0, by, cm_torgb_kernel<8>, a, grid_of(8), 0);
    else emit(p, "migan::cm_torgb_kernel<16>", fl, 0, by, cm_torgb_kernel<16>, a, grid_of(16), 0);
  };
  {
    const int c4 = channels(4);
    // b4 (comodgan.py:232-257): x = fc(w0).view(N, C, 4, 4) + feat[4]; conv; torgb
    float* x4 = act_out("synthesis.b4.fc", bufA, 4, c4);
    dense("synthesis.b4.fc", w0, nullptr, cfg.w0_dim, cfg.w0_dim, "synthesis.b4.fc", c4 * 16, x4, 1.0f, true, false, 0, c4, feat[2], nullptr);
    const ConvW& cw = conv_of("synthesis.b4.conv");
    const Mod m = style_demod("synthesis.b4.conv", cw);
    const float* nz; long long nbs;
    noise_of("synthesis.b4.conv", 4, nz, nbs);
    float* xc = act_out("synthesis.b4.conv", bufB, 4, c4);
    conv("synthesis.b4.conv", CM_CONV_NORMAL, 0, 0, x4, xc, nullptr, cw, m.sa, m.coef, 1.0f, dry ? nullptr : W("synthesis.b4.conv.bias"), nz,
         dry ? nullptr : W("synthesis.b4.conv.noise_strength"), nbs, 4, 4, 4, 4, false);
    float* im = debug ? alloc((size_t)3 * 16 * B * 4) : img[0];
    reg_debug("synthesis.b4.img", im, {B, 3, 4, 4});
    torgb("synthesis.b4.torgb", xc, 4, c4, nullptr, (R == 4) ? y : im);
    float* xcur = xc;
    const float* imprev = im;
    int flip = 1;
    for (int res = 8; res <= R; res *= 2) {
      const std::string b = bname("synthesis", res);
      const int co = channels(res), h = res / 2;
      // conv0: modulated transposed convolution (4 output phases) -> FIR + noise + bias + activation, + skip (comodgan.py:329-331)
      const ConvW& c0w = conv_of(b + ".conv0");
      const Mod m0s = style_demod(b + ".conv0", c0w);
      // all four phases in one launch, on 64-column tiles: 4 x 32 accumulator registers per lane leave room for two waves per SIMD
      // (251 VGPRs), which the 128-column form (256 accumulators in AGPRs + 169 VGPRs, one wave per SIMD) does not.  Measured at
      // comodgan-512, batch 16 (profiles/r02_comodgan_up4_forms.txt): seven conv0 layers 3.28 ms (128-column four-phase launches +
      // four single-phase launches at 512^2) -> 2.86 ms.  COMODGAN_UP4=0|1 and COMODGAN_UP4_NT=64|128 force one form (experiments / tests).
      const char* up4_env = std::getenv("COMODGAN_UP4");
      if (up4_env ? std::atoi(up4_env) != 0 : true)
        conv(b + ".conv0.phases", CM_CONV_UP4, 0, 0, xcur, tmp, nullptr, c0w, m0s.sa, m0s.coef, 1.0f, nullptr, nullptr, nullptr, 0, h, h,
             res + 1, res + 1, true);
      else
        for (int ph = 0; ph < 4; ++ph)
          conv(b + ".conv0.phase" + std::to_string(ph), CM_CONV_UP, ph >> 1, ph & 1, xcur, tmp, nullptr, c0w, m0s.sa, m0s.coef, 1.0f, nullptr,
               nullptr, nullptr, 0, h, h, res + 1, res + 1, true);
      noise_of(b + ".conv0", res, nz, nbs);
      float* x0 = act_out(b + ".conv0", bufA, res, co);
      {
        CmFirArgs a{};
        a.x = tmp; a.y = x0; a.skip = feat[ilog2(res)]; a.bias = dry ? nullptr : W(b + ".conv0.bias"); a.noise = nz;
        a.noise_strength = dry ? nullptr : W(b + ".conv0.noise_strength"); a.noise_bstride = nbs;
        a.B = B; a.H = res + 1; a.W = res + 1; a.C = co; a.HO = res; a.WO = res; a.pad = 1; a.fs = 0.25f;
        emit(b + ".conv0.fir", "migan::cm_fir_kernel<1>", 2.0 * 16 * co * res * res, 0, 4.0 * co * ((res + 1.0) * (res + 1.0) + 2.0 * res * res),
             cm_fir_kernel<1>, a, grid1d((size_t)B * (res / 2) * cdiv(res, 4) * (co / 4)), 0);
      }
      // conv1
      const ConvW& c1w = conv_of(b + ".conv1");
      const Mod m1s = style_demod(b + ".conv1", c1w);
      noise_of(b + ".conv1", res, nz, nbs);
      float* x1 = act_out(b + ".conv1", bufB, res, co);
      conv(b + ".conv1", CM_CONV_NORMAL, 0, 0, x0, x1, nullptr, c1w, m1s.sa, m1s.coef, 1.0f, dry ? nullptr : W(b + ".conv1.bias"), nz,
           dry ? nullptr : W(b + ".conv1.noise_strength"), nbs, res, res, res, res, false);
      // img = upsample2d(img) + torgb(x) (comodgan.py:334-343)
      float* imo = (res == R) ? y : (debug ? alloc((size_t)3 * res * res * B * 4) : img[flip]);
      if (res != R) reg_debug(b + ".img", imo, {B, 3, res, res});
      torgb(b + ".torgb", x1, res, co, imprev, imo);
      imprev = imo;
      flip ^= 1;
      xcur = x1;
    }
  }
  MIGAN_CHECK(!timed || nlaunch <= n_ms, MIGAN_EINVAL, "launch_ms array too small");
  if (timed && !dry) {
    rt_check(rt::stream_sync(stream), "hipStreamSynchronize");
    for (int i = 0; i < nlaunch; ++i) rt_check(rt::event_elapsed(&ms[i], events[2 * i], events[2 * i + 1]), "hipEventElapsedTime");
  }
  return cursor;
}

extern "C" {

int comodgan_create(const comodgan_config* cfg, int device, comodgan_handle** out) {
  MIGAN_API_BEGIN
  using namespace migan;
  MIGAN_CHECK(cfg && out, MIGAN_EINVAL, "null argument");
  const int r = cfg->resolution;
  MIGAN_CHECK(r >= 8 && r <= 512 && (r & (r - 1)) == 0, MIGAN_EINVAL, "resolution must be a power of two in [8, 512]");
  MIGAN_CHECK(cfg->ch_base > 0 && cfg->ch_max > 0 && cfg->z_dim > 0 && cfg->w_dim > 0 && cfg->w0_dim > 0 && cfg->map_layers > 0 && cfg->num_ws > 0,
              MIGAN_EINVAL, "non-positive dimension");
  for (int res = 4; res <= r; res *= 2) {
    const int c = std::min(cfg->ch_base / res, cfg->ch_max);
    MIGAN_CHECK(c >= 64 && c % 64 == 0, MIGAN_EINVAL, "channel counts must be multiples of 64");
  }
  DeviceGuard guard(device);
  prepare_kernels();
  cm_prepare_kernels();
  comodgan_handle* h = new comodgan_handle();
  h->cfg = *cfg;
  h->device = device;
  h->build_schema();
  h->planned_batch = 1;
  h->planned_need = h->walk(1, nullptr, nullptr, nullptr, 1.0f, COMODGAN_NOISE_CONST, nullptr, nullptr, nullptr, true, nullptr, 0);
  *out = h;
  MIGAN_API_END
}

int comodgan_destroy(comodgan_handle* h) {
  MIGAN_API_BEGIN
  if (h) {
    for (auto& e : h->events) rt::event_destroy(e);
    if (h->side_ready) {
      rt::event_destroy(h->ev_fork);
      rt::event_destroy(h->ev_map);
      rt::stream_destroy(h->map_stream);
    }
    delete h;
  }
  MIGAN_API_END
}

int comodgan_num_weights(const comodgan_handle* h, int* n) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && n, MIGAN_EINVAL, "null argument");
  *n = (int)h->slots.size();
  MIGAN_API_END
}

int comodgan_weight_info(const comodgan_handle* h, int index, const char** name, int64_t shape[4], int* ndim, int* is_buffer) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  MIGAN_CHECK(index >= 0 && index < (int)h->slots.size(), MIGAN_EINVAL, "weight index out of range");
  const migan::CmSlot& s = h->slots[index];
  if (name) *name = s.name.c_str();
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = s.shape[i];
  if (ndim) *ndim = s.ndim;
  if (is_buffer) *is_buffer = s.is_buffer ? 1 : 0;
  MIGAN_API_END
}

int comodgan_set_weight(comodgan_handle* h, const char* name, const void* dev_ptr, const int64_t* shape, int ndim) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && name && dev_ptr, MIGAN_EINVAL, "null argument");
  const int i = h->slot_index(name);
  MIGAN_CHECK(i >= 0, MIGAN_EINVAL, std::string("unexpected key in state_dict: ") + name);
  migan::CmSlot& s = h->slots[i];
  bool same = ndim == s.ndim;
  for (int d = 0; same && d < ndim; ++d) same = shape[d] == s.shape[d];
  MIGAN_CHECK(same, MIGAN_EINVAL, std::string("size mismatch for ") + name);
  MIGAN_CHECK(((uintptr_t)dev_ptr % 4) == 0, MIGAN_EINVAL, std::string("misaligned tensor ") + name);
  s.ptr = static_cast<const float*>(dev_ptr);
  h->committed = false;
  ++h->weights_epoch;
  MIGAN_API_END
}

int comodgan_commit(comodgan_handle* h, void* stream) {
  MIGAN_API_BEGIN
  using namespace migan;
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  DeviceGuard guard(h->device);
  for (const auto& s : h->slots) MIGAN_CHECK(s.ptr != nullptr, MIGAN_ESTATE, std::string("missing key in state_dict: ") + s.name);
  static const double taps[4] = {1.0, 3.0, 3.0, 1.0};
  float host[16];
  for (const auto& s : h->slots) {
    if (s.role == CR_CONV_W && s.shape[2] == 3)
      MIGAN_CHECK(((uintptr_t)s.ptr % 16) == 0, MIGAN_EINVAL, s.name + " must be 16-byte aligned");
    if (s.role != CR_FIR) continue;
    rt_check(rt::memcpy_d2h(host, s.ptr, sizeof(host), (rt::stream_t)stream), "hipMemcpy (FIR check)");
    for (int ky = 0; ky < 4; ++ky)
      for (int kx = 0; kx < 4; ++kx)
        MIGAN_CHECK(std::fabs((double)host[ky * 4 + kx] - taps[ky] * taps[kx] / 64.0) <= 1e-6, MIGAN_EUNSUPPORTED,
                    s.name + " differs from setup_filter([1,3,3,1]); only the reference FIR is implemented");
  }
  h->committed = true;
  ++h->weights_epoch;
  MIGAN_API_END
}

int comodgan_assume_static_weights(comodgan_handle* h, int on) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  h->static_weights = on != 0;
  h->prepared_ws = nullptr;       // every call drops the planes prepared so far: (re-)asserting after an in-place write is the way to invalidate
  MIGAN_API_END
}

int comodgan_workspace_bytes(const comodgan_handle* h, int batch, size_t* bytes) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && bytes && batch > 0, MIGAN_EINVAL, "bad argument");
  comodgan_handle* m = const_cast<comodgan_handle*>(h);
  if (m->planned_batch != batch || m->planned_need == 0) {
    m->planned_need = m->walk(batch, nullptr, nullptr, nullptr, 1.0f, COMODGAN_NOISE_CONST, nullptr, nullptr, nullptr, true, nullptr, 0);
    m->planned_batch = batch;
  }
  *bytes = m->planned_need;
  MIGAN_API_END
}

int comodgan_noise_floats(const comodgan_handle* h, size_t* floats_per_image) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && floats_per_image, MIGAN_EINVAL, "null argument");
  *floats_per_image = h->noise_floats();
  MIGAN_API_END
}

static int comodgan_forward_impl(comodgan_handle* h, const void* x, const void* z, void* y, int batch, float psi, int noise_mode,
                                 const void* noise, void* ws, size_t ws_bytes, void* stream, float* ms, int n_ms) {
  MIGAN_API_BEGIN
  using namespace migan;
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  MIGAN_CHECK(h->committed, MIGAN_ESTATE, "comodgan_forward before comodgan_commit");
  MIGAN_CHECK(x && z && y && batch > 0, MIGAN_EINVAL, "null tensor or empty batch");
  MIGAN_CHECK(noise_mode == COMODGAN_NOISE_NONE || noise_mode == COMODGAN_NOISE_CONST || noise_mode == COMODGAN_NOISE_RANDOM, MIGAN_EINVAL,
              "noise_mode must be none, const or random");
  MIGAN_CHECK(noise_mode != COMODGAN_NOISE_RANDOM || noise != nullptr, MIGAN_EINVAL, "noise_mode random needs the noise tensor");
  MIGAN_CHECK(ws != nullptr && ((uintptr_t)ws % 256) == 0, MIGAN_EINVAL, "null or misaligned workspace (256 bytes)");
  MIGAN_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)z % 4) == 0, MIGAN_EINVAL, "misaligned tensor");
  // the launch list and the workspace size depend on the batch (and the debug flag) only: planned once per batch size
  if (h->planned_batch != batch || h->planned_need == 0) {
    h->planned_need = h->walk(batch, nullptr, nullptr, nullptr, psi, noise_mode, nullptr, nullptr, nullptr, true, nullptr, 0);
    h->planned_batch = batch;
  }
  const size_t need = h->planned_need;
  MIGAN_CHECK(ws_bytes >= need, MIGAN_EINVAL, "workspace too small for this batch");
  if (ms) {
    MIGAN_CHECK(n_ms >= (int)h->infos.size(), MIGAN_EINVAL, "launch_ms array too small");
    while (h->events.size() < 2 * h->infos.size()) {
      rt::event_t e;
      rt_check(rt::event_create(&e), "hipEventCreate");
      h->events.push_back(e);
    }
  }
  DeviceGuard guard(h->device);
  h->walk(batch, (const float*)x, (const float*)z, (float*)y, psi, noise_mode, (const float*)noise, ws, (rt::stream_t)stream, false, ms, n_ms);
  MIGAN_API_END
}

int comodgan_forward(comodgan_handle* h, const void* x, const void* z, void* y, int batch, float psi, int noise_mode, const void* noise,
                     void* ws, size_t ws_bytes, void* stream) {
  return comodgan_forward_impl(h, x, z, y, batch, psi, noise_mode, noise, ws, ws_bytes, stream, nullptr, 0);
}

int comodgan_forward_timed(comodgan_handle* h, const void* x, const void* z, void* y, int batch, float psi, int noise_mode, const void* noise,
                           void* ws, size_t ws_bytes, void* stream, float* launch_ms, int n_launch_ms) {
  if (!launch_ms) {
    migan::last_error_ref() = "null launch_ms";
    return MIGAN_EINVAL;
  }
  return comodgan_forward_impl(h, x, z, y, batch, psi, noise_mode, noise, ws, ws_bytes, stream, launch_ms, n_launch_ms);
}

int comodgan_num_launches(const comodgan_handle* h, int* n) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && n, MIGAN_EINVAL, "null argument");
  *n = (int)h->infos.size();
  MIGAN_API_END
}

int comodgan_launch_info(const comodgan_handle* h, int index, const char** layer, const char** kernel, double* flops, double* mfma_flops,
                         double* bytes) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  MIGAN_CHECK(index >= 0 && index < (int)h->infos.size(), MIGAN_EINVAL, "launch index out of range");
  const migan::CmInfo& L = h->infos[index];
  if (layer) *layer = L.layer.c_str();
  if (kernel) *kernel = L.kernel.c_str();
  if (flops) *flops = L.flops;
  if (mfma_flops) *mfma_flops = L.mfma_flops;
  if (bytes) *bytes = L.bytes;
  MIGAN_API_END
}

int comodgan_set_truncation_cutoff(comodgan_handle* h, int cutoff) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  MIGAN_CHECK(cutoff >= -1, MIGAN_EINVAL, "truncation_cutoff must be >= 0, or -1 for None");
  if (h->trunc_cutoff != cutoff) {
    h->trunc_cutoff = cutoff;
    h->planned_need = 0;                   // (one more [batch][w_dim] buffer in the workspace walk)
  }
  MIGAN_API_END
}

int comodgan_set_debug(comodgan_handle* h, int keep) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h, MIGAN_EINVAL, "null handle");
  h->debug = keep != 0;
  if (h->planned_batch <= 0) h->planned_batch = 1;
  h->planned_need = h->walk(h->planned_batch, nullptr, nullptr, nullptr, 1.0f, COMODGAN_NOISE_CONST, nullptr, nullptr, nullptr, true, nullptr, 0);
  MIGAN_API_END
}

int comodgan_debug_tensor(const comodgan_handle* h, int batch, const char* layer, size_t* byte_offset, int64_t shape[4], int* ndim) {
  MIGAN_API_BEGIN
  MIGAN_CHECK(h && layer && byte_offset && shape && ndim, MIGAN_EINVAL, "null argument");
  MIGAN_CHECK(h->debug, MIGAN_ESTATE, "comodgan_set_debug(h, 1) first");
  comodgan_handle* m = const_cast<comodgan_handle*>(h);
  if (m->planned_batch != batch || m->planned_need == 0) {
    m->planned_batch = batch;
    m->planned_need = m->walk(batch, nullptr, nullptr, nullptr, 1.0f, COMODGAN_NOISE_CONST, nullptr, nullptr, nullptr, true, nullptr, 0);
  }
  for (const auto& t : h->debug_tensors) {
    if (t.name != layer) continue;
    *byte_offset = t.offset;
    for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
    *ndim = t.ndim;
    return MIGAN_OK;
  }
  throw migan::Error(MIGAN_EINVAL, std::string("no such debug tensor: ") + layer);
  MIGAN_API_END
}

}  // extern "C"
