// Stand-alone driver of sepconv_stream_kernel (mi-gan_amd/csrc/migan_stream.hpp): correctness against a straightforward CPU
// evaluation of the same SeparableConv2d on a small ragged image, then timing at 32 x 512 x 512 x 64 beside a plain copy.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/stream_proto.hip -o gpurun_out/stream_proto && gpurun_out/stream_proto
#include "../../../mi-gan_amd/csrc/migan_rt_hip.h"
#include "stream_rt.h"
#include "../../../mi-gan_amd/csrc/migan_kernels.hpp"
#include "migan_stream.hpp"

#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using namespace migan;

static unsigned long long rng_state = 88172645463325252ull;
static float frand() {   // uniform (-1, 1)
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (float)((rng_state >> 11) * (1.0 / 9007199254740992.0)) * 2.0f - 1.0f;
}

__global__ void copy_kernel(const f4* __restrict__ a, f4* __restrict__ o, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(a[i], &o[i]);
}

static double act_ref(double v) {
  double t = v > 0 ? v : 0.2 * v;
  t *= 1.4142135623730951;
  return t > 256 ? 256 : (t < -256 ? -256 : t);
}

struct Host {
  std::vector<float> wdw, bdw, wpw, noise, frw, frb, trw, trb;
  float ns;
};

// A-operand planes + scales, as weight_absmax_kernel + stream_prep_kernel produce them
static void prep_host(const Host& hst, std::vector<unsigned short>& planes, float hdr[4]) {
  float m = 0;
  for (float v : hst.wpw) m = std::fmax(m, std::fabs(v));
  int e; std::frexp(m, &e); e -= 1;                       // floor(log2(m))
  hdr[2] = std::ldexp(1.0f, 13 - e);
  hdr[0] = std::ldexp(1.0f, e - 13 - 7);
  hdr[1] = m; hdr[3] = 0;
  planes.assign(2 * 4 * 2 * 64 * 8, 0);
  for (int mt = 0; mt < 2; ++mt) for (int ks = 0; ks < 4; ++ks) for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
    const int mm = l & 31, hh = l >> 5;
    const float v = hst.wpw[(32 * mt + mm) * 64 + 16 * ks + 8 * (j >> 2) + 4 * hh + (j & 3)] * hdr[2];
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    unsigned short uh, ul;
    std::memcpy(&uh, &hi, 2); std::memcpy(&ul, &lo, 2);
    planes[((((mt * 4 + ks) * 2 + 0) * 64) + l) * 8 + j] = uh;
    planes[((((mt * 4 + ks) * 2 + 1) * 64) + l) * 8 + j] = ul;
  }
}

template <int FROMRGB, int TORGB>
static int run_case(int B, int H, int W, bool check, int grid, int reps) {
  const int C = 64;
  Host hst;
  hst.wdw.resize(C * 9); hst.bdw.resize(C); hst.wpw.resize(C * C); hst.noise.resize((size_t)H * W);
  hst.frw.resize(C * 4); hst.frb.resize(C); hst.trw.resize(3 * C); hst.trb.resize(3);
  for (auto& v : hst.wdw) v = frand() * 0.5f;
  for (auto& v : hst.bdw) v = frand() * 0.5f;
  for (auto& v : hst.wpw) v = frand() * 0.2f;
  for (auto& v : hst.noise) v = frand();
  for (auto& v : hst.frw) v = frand();
  for (auto& v : hst.frb) v = frand() * 0.3f;
  for (auto& v : hst.trw) v = frand() * 0.2f;
  for (auto& v : hst.trb) v = frand();
  hst.ns = 0.3f;
  const size_t nin = FROMRGB ? (size_t)B * 4 * H * W : (size_t)B * H * W * C, nout = (size_t)B * H * W * C;
  std::vector<float> x(nin);
  for (auto& v : x) v = frand() * 2.0f;
  std::vector<float> prev((size_t)B * 3 * (H / 2) * (W / 2));
  for (auto& v : prev) v = frand();
  std::vector<unsigned short> planes; float hdr[4];
  prep_host(hst, planes, hdr);

  float *dx, *dy, *dwdw, *dbdw, *dnoise, *dns, *dfrw, *dfrb, *dtrw, *dtrb, *dprev, *dimg, *dhdr;
  unsigned short* dplanes;
  CK(hipMalloc(&dx, nin * 4)); CK(hipMalloc(&dy, nout * 4));
  CK(hipMalloc(&dwdw, C * 9 * 4)); CK(hipMalloc(&dbdw, C * 4)); CK(hipMalloc(&dnoise, (size_t)H * W * 4)); CK(hipMalloc(&dns, 4));
  CK(hipMalloc(&dfrw, C * 16)); CK(hipMalloc(&dfrb, C * 4)); CK(hipMalloc(&dtrw, 3 * C * 4)); CK(hipMalloc(&dtrb, 12));
  CK(hipMalloc(&dprev, prev.size() * 4 + 16)); CK(hipMalloc(&dimg, (size_t)B * 3 * H * W * 4)); CK(hipMalloc(&dhdr, 16));
  CK(hipMalloc(&dplanes, planes.size() * 2));
  CK(hipMemcpy(dx, x.data(), nin * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dwdw, hst.wdw.data(), C * 9 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbdw, hst.bdw.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dnoise, hst.noise.data(), (size_t)H * W * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dns, &hst.ns, 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dfrw, hst.frw.data(), C * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dfrb, hst.frb.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dtrw, hst.trw.data(), 3 * C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dtrb, hst.trb.data(), 12, hipMemcpyHostToDevice));
  CK(hipMemcpy(dprev, prev.data(), prev.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dhdr, hdr, 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(dplanes, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dy, 0xff, nout * 4)); CK(hipMemset(dimg, 0xff, (size_t)B * 3 * H * W * 4));

  StreamArgs a{};
  a.x = dx; a.y = dy; a.wdw = dwdw; a.bdw = dbdw; a.wstream = dplanes; a.acc_scale = dhdr;
  a.noise = FROMRGB ? nullptr : dnoise; a.noise_strength = dns;
  a.frgb_w = dfrw; a.frgb_b = dfrb; a.trgb_w = dtrw; a.trgb_b = dtrb; a.img_prev = dprev; a.img_out = dimg;
  a.B = B; a.H = H; a.W = W;
  a.nstrips = (W + kStreamValid - 1) / kStreamValid;
  const long long total = (long long)B * a.nstrips * H;
  if (grid <= 0) grid = 512;
  a.rows_per_wave = (int)((total + 4ll * grid - 1) / (4ll * grid));
  auto kern = sepconv_stream_kernel<FROMRGB, TORGB, 0>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kStreamLds * 4));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kStreamLds * 4, 0, a);
  CK(hipGetLastError()); CK(hipDeviceSynchronize());
  int bad = 0;
  if (check) {
    std::vector<float> y(nout), img((size_t)B * 3 * H * W);
    CK(hipMemcpy(y.data(), dy, nout * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(img.data(), dimg, img.size() * 4, hipMemcpyDeviceToHost));
    // reference in double
    std::vector<double> in((size_t)B * H * W * C);
    if (FROMRGB) {
      for (int b = 0; b < B; ++b) for (int yy = 0; yy < H; ++yy) for (int xx = 0; xx < W; ++xx) for (int c = 0; c < C; ++c) {
        double t = hst.frb[c];
        for (int i = 0; i < 4; ++i) t += (double)hst.frw[c * 4 + i] * x[(((size_t)b * 4 + i) * H + yy) * W + xx];
        in[(((size_t)b * H + yy) * W + xx) * C + c] = act_ref(t);
      }
    } else {
      for (size_t i = 0; i < in.size(); ++i) in[i] = x[i];
    }
    double maxerr = 0, maxref = 0, maxerr_rgb = 0;
    std::vector<double> dwv(C), outv(C);
    for (int b = 0; b < B; ++b) for (int yy = 0; yy < H; ++yy) for (int xx = 0; xx < W; ++xx) {
      for (int c = 0; c < C; ++c) {
        double t = hst.bdw[c];
        for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
          const int sy = yy + ky - 1, sx = xx + kx - 1;
          if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
          t += (double)hst.wdw[c * 9 + ky * 3 + kx] * in[(((size_t)b * H + sy) * W + sx) * C + c];
        }
        dwv[c] = act_ref(t);
      }
      double rgb[3] = {hst.trb[0], hst.trb[1], hst.trb[2]};
      for (int co = 0; co < C; ++co) {
        double t = 0;
        for (int c = 0; c < C; ++c) t += (double)hst.wpw[co * C + c] * dwv[c];
        if (!FROMRGB) t += (double)hst.noise[(size_t)yy * W + xx] * hst.ns;
        t = act_ref(t);
        outv[co] = t;
        const double got = y[(((size_t)b * H + yy) * W + xx) * C + co];
        maxerr = std::fmax(maxerr, std::fabs(got - t)); maxref = std::fmax(maxref, std::fabs(t));
        if (!(std::fabs(got - t) <= 2e-3) && bad < 10) { printf("  mismatch b%d y%d x%d co%d: got %g want %g\n", b, yy, xx, co, got, t); ++bad; }
        for (int ch = 0; ch < 3; ++ch) rgb[ch] += (double)hst.trw[ch * C + co] * t;
      }
      if (TORGB) {
        const int hp = H / 2, wp = W / 2;
        for (int ch = 0; ch < 3; ++ch) {
          // Upsample2d of the previous image: out[2i] = g[i-1]/4 + 3g[i]/4, out[2i+1] = 3g[i]/4 + g[i+1]/4 per axis
          double up = 0;
          const int iy = yy >> 1, ix = xx >> 1;
          const int y0 = (yy & 1) ? iy : iy - 1, x0 = (xx & 1) ? ix : ix - 1;
          const double wy[2] = {(yy & 1) ? 0.75 : 0.25, (yy & 1) ? 0.25 : 0.75}, wx[2] = {(xx & 1) ? 0.75 : 0.25, (xx & 1) ? 0.25 : 0.75};
          for (int dy = 0; dy < 2; ++dy) for (int dx2 = 0; dx2 < 2; ++dx2) {
            const int py = y0 + dy, pxx = x0 + dx2;
            if (py < 0 || py >= hp || pxx < 0 || pxx >= wp) continue;
            up += wy[dy] * wx[dx2] * prev[(((size_t)b * 3 + ch) * hp + py) * wp + pxx];
          }
          const double want = rgb[ch] + up, got = img[(((size_t)b * 3 + ch) * H + yy) * W + xx];
          maxerr_rgb = std::fmax(maxerr_rgb, std::fabs(got - want));
          if (!(std::fabs(got - want) <= 5e-3) && bad < 10) { printf("  rgb mismatch b%d y%d x%d ch%d: got %g want %g\n", b, yy, xx, ch, got, want); ++bad; }
        }
      }
    }
    printf("check FROMRGB=%d TORGB=%d B=%d H=%d W=%d grid=%d: max|y|=%.3f max err=%.3e rgb err=%.3e %s\n", (int)FROMRGB, (int)TORGB, B, H, W, grid, maxref,
           maxerr, maxerr_rgb, bad ? "FAIL" : "ok");
  }
  if (reps > 0) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kStreamLds * 4, 0, a);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kStreamLds * 4, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)nin * 4 + (double)nout * 4 + (TORGB ? (double)B * 3 * H * W * 4 * 1.25 : 0);
    printf("time FROMRGB=%d TORGB=%d B=%d %dx%d grid=%d: %.4f ms  %.2f TB/s algorithmic\n", (int)FROMRGB, (int)TORGB, B, H, W, grid, ms, bytes / ms * 1e-9);
    if (!FROMRGB) {
      const size_t n4 = nout / 4;
      for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (const f4*)dx, (f4*)dy, n4);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (const f4*)dx, (f4*)dy, n4);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      printf("  float4 copy of the same tensor: %.4f ms  %.2f TB/s\n", ms, 2.0 * nout * 4 / ms * 1e-9);
    }
  }
  for (void* q : {(void*)dx, (void*)dy, (void*)dwdw, (void*)dbdw, (void*)dnoise, (void*)dns, (void*)dfrw, (void*)dfrb, (void*)dtrw, (void*)dtrb, (void*)dprev, (void*)dimg, (void*)dhdr, (void*)dplanes}) CK(hipFree(q));
  return bad;
}

int main(int argc, char** argv) {
  int bad = 0;
  if (argc >= 5) {       // one timed configuration: <fromrgb> <torgb> <grid> <reps>   (for rocprofv3 passes)
    const int fr = atoi(argv[1]), tr = atoi(argv[2]), grid = atoi(argv[3]), reps = atoi(argv[4]);
    if (fr) run_case<1, 0>(32, 512, 512, false, grid, reps);
    else if (tr) run_case<0, 1>(32, 512, 512, false, grid, reps);
    else run_case<0, 0>(32, 512, 512, false, grid, reps);
    return 0;
  }
  bad += run_case<0, 0>(2, 38, 70, true, 16, 0);
  bad += run_case<0, 1>(2, 38, 70, true, 16, 0);
  bad += run_case<1, 0>(2, 38, 70, true, 16, 0);
  bad += run_case<0, 1>(1, 64, 64, true, 3, 0);
  if (argc > 1 && atoi(argv[1]) == 0) return bad != 0;
  for (int grid : {512, 1024, 2048, 4096}) {
    run_case<0, 0>(32, 512, 512, false, grid, 10);
    run_case<0, 1>(32, 512, 512, false, grid, 10);
    run_case<1, 0>(32, 512, 512, false, grid, 10);
  }
  return bad != 0;
}
