// The reference's DEPLOYED pre/post-processing around the generator forward (second half):
// scripts/create_onnx_pipeline.py::MIGAN_Pipeline (:118-264) -- masked bounding box, crop, bilinear resize to the network
// resolution, generator, bilinear resize back, 3x3 max-pool + 5x5 gaussian feathering of the mask, blend -- as gfx950 kernels.
// All of it is memory-bound elementwise / small-stencil work on one image: one thread per output pixel, coalesced rows.
//
// torch's arithmetic is followed operation by operation where it decides a rounding the test can see:
//   F.interpolate(mode="bilinear", align_corners=False): scale = float(in) / out, src = scale * (dst + 0.5) - 0.5 clamped at 0,
//     i0 = floor(src), i1 = min(i0 + 1, in - 1), l1 = src - i0, l0 = 1 - l1, value = l0y (l0x p00 + l1x p01) + l1y (l0x p10 + l1x p11)
//   F.interpolate(mode="nearest"): src = min(floor(dst * scale), in - 1)
//   torchvision's tensor resize rounds a uint8 image back with torch.round (half to even) -> rintf
// Compiled for the product and (tests/emu) for the CPU emulator.
#pragma once

namespace migan {

struct PipeArgs {
  unsigned char* image;        // [3][H][W] uint8 (CHW, the reference pipeline's layout); post: read and written in place
  const unsigned char* mask;   // [H][W] uint8, 255 = known pixel
  float* x;                    // pre: network input [4][R][R]
  const float* y;              // post: network output [3][R][R]
  unsigned char* pooled;       // post: 3x3 max-pool of the cropped mask [ch][cw] (scratch)
  int* flags;                  // bbox: [W] column flags then [H] row flags (scratch)
  int H, W, R;
  int x_min, x_max, y_min, y_max;
  float gauss[25];             // GaussianSmoothing(kernel_size=5, sigma=1) weights, row major (:63-85)
};

MIGAN_DEVICE MIGAN_INLINE void bilinear_coord(int dst, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.0f) src = 0.0f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l1 = fminf(fmaxf(l1, 0.0f), 1.0f);
  l0 = 1.0f - l1;
}
MIGAN_DEVICE MIGAN_INLINE float bilinear_mix(float p00, float p01, float p10, float p11, float l0x, float l1x, float l0y, float l1y) {
  const float h0 = MIGAN_FADD_RN(MIGAN_FMUL_RN(l0x, p00), MIGAN_FMUL_RN(l1x, p01));
  const float h1 = MIGAN_FADD_RN(MIGAN_FMUL_RN(l0x, p10), MIGAN_FMUL_RN(l1x, p11));
  return MIGAN_FADD_RN(MIGAN_FMUL_RN(l0y, h0), MIGAN_FMUL_RN(l1y, h1));
}

#ifndef MIGAN_TEMPLATE_KERNELS_ONLY
// get_masked_bbox (:132-147): which columns / rows contain a pixel that is not 255 (mean < 255 <=> any pixel < 255 for uint8 data)
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) pipe_flags_clear_kernel(const PipeArgs p) {
  const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
  if (i < p.H + p.W) p.flags[i] = 0;
}
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) pipe_flags_kernel(const PipeArgs p) {
  const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
  if (i >= p.H * p.W) return;
  if (p.mask[i] != 255) {
    p.flags[i % p.W] = 1;                 // (every writer stores the same value)
    p.flags[p.W + i / p.W] = 1;
  }
}

// MIGAN_Pipeline.forward's first line (:256): tvF.resize(mask, image size, NEAREST) = F.interpolate(mode="nearest").
// args: mask = source [y_max][x_max] (its height / width ride in y_max / x_max), pooled = destination [H][W]
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) pipe_mask_resize_kernel(const PipeArgs p) {
  const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
  if (i >= p.H * p.W) return;
  const int oy = i / p.W, ox = i % p.W, ih = p.y_max, iw = p.x_max;
  int sy = (int)floorf((float)oy * ((float)ih / (float)p.H)), sx = (int)floorf((float)ox * ((float)iw / (float)p.W));
  sy = sy < ih - 1 ? sy : ih - 1;
  sx = sx < iw - 1 ? sx : iw - 1;
  p.pooled[i] = p.mask[(size_t)sy * iw + sx];
}

// preprocess (:233-239) of the crop [y_min, y_max) x [x_min, x_max): bilinear resize of the uint8 image (rounded back to uint8 as
// torchvision does), nearest resize of the mask, x = cat([mask / 255 - 0.5, (image * 2 / 255 - 1) * mask / 255])
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) pipe_pre_kernel(const PipeArgs p) {
  const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
  if (i >= p.R * p.R) return;
  const int oy = i / p.R, ox = i % p.R;
  const int ch = p.y_max - p.y_min, cw = p.x_max - p.x_min;
  const float sy = (float)ch / (float)p.R, sx = (float)cw / (float)p.R;
  int y0, y1, x0, x1;
  float l0y, l1y, l0x, l1x;
  bilinear_coord(oy, sy, ch, y0, y1, l0y, l1y);
  bilinear_coord(ox, sx, cw, x0, x1, l0x, l1x);
  int ny = (int)floorf((float)oy * sy), nx = (int)floorf((float)ox * sx);
  ny = ny < ch - 1 ? ny : ch - 1;
  nx = nx < cw - 1 ? nx : cw - 1;
  const float m = (float)p.mask[(size_t)(p.y_min + ny) * p.W + p.x_min + nx] / 255.0f;
  const size_t plane = (size_t)p.H * p.W, oplane = (size_t)p.R * p.R;
  p.x[i] = m - 0.5f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const unsigned char* q = p.image + c * plane;
    const float p00 = (float)q[(size_t)(p.y_min + y0) * p.W + p.x_min + x0], p01 = (float)q[(size_t)(p.y_min + y0) * p.W + p.x_min + x1];
    const float p10 = (float)q[(size_t)(p.y_min + y1) * p.W + p.x_min + x0], p11 = (float)q[(size_t)(p.y_min + y1) * p.W + p.x_min + x1];
    float v = rintf(bilinear_mix(p00, p01, p10, p11, l0x, l1x, l0y, l1y));          // torch.round, then .to(uint8)
    v = (float)(unsigned char)(int)v;
    v = MIGAN_FSUB_RN(MIGAN_FMUL_RN(v, 2.0f) / 255.0f, 1.0f);                        // image.float() * 2 / 255 - 1
    p.x[(c + 1) * oplane + i] = MIGAN_FMUL_RN(v, m);
  }
}

// F.max_pool2d(mask, 3, stride=1, padding=1) of the cropped mask (:246): neighbours outside the crop do not count
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) pipe_maxpool_kernel(const PipeArgs p) {
  const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
  const int ch = p.y_max - p.y_min, cw = p.x_max - p.x_min;
  if (i >= ch * cw) return;
  const int py = i / cw, px = i % cw;
  int m = 0;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = py + dy, xx = px + dx;
      if (yy < 0 || yy >= ch || xx < 0 || xx >= cw) continue;
      const int v = p.mask[(size_t)(p.y_min + yy) * p.W + p.x_min + xx];
      m = v > m ? v : m;
    }
  p.pooled[i] = (unsigned char)m;
}

// postprocess (:241-250) + the paste back (:263): generator output -> [0, 255], bilinear resize to the crop, feathered mask
// (gaussian 5x5 on the max-pooled mask, reflect padding), composed = image * mask + output * (1 - mask), clamp, truncate to uint8
MIGAN_GLOBAL void MIGAN_LAUNCH_BOUNDS(256, 2) pipe_post_kernel(const PipeArgs p) {
  const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
  const int ch = p.y_max - p.y_min, cw = p.x_max - p.x_min;
  if (i >= ch * cw) return;
  const int py = i / cw, px = i % cw;
  // feathered mask.  The 25 products are summed in fp64 and rounded once: the fp32 weights sum to 1 - 3.7e-9, so a flat 255
  // neighbourhood blurs to exactly 255.0f and a known pixel far from the hole is returned unchanged (as ATen's conv2d does on the
  // reference's host); a plain fp32 running sum gives 254.99998 there and the truncation below would darken every known pixel by 1.
  double acc = 0.0;
  for (int ky = 0; ky < 5; ++ky) {
    int yy = py + ky - 2;
    yy = yy < 0 ? -yy : (yy >= ch ? 2 * ch - 2 - yy : yy);                          // F.pad(mode='reflect') (:114)
    for (int kx = 0; kx < 5; ++kx) {
      int xx = px + kx - 2;
      xx = xx < 0 ? -xx : (xx >= cw ? 2 * cw - 2 - xx : xx);
      acc += (double)p.gauss[ky * 5 + kx] * (double)p.pooled[yy * cw + xx];
    }
  }
  const float mk = (float)acc / 255.0f;
  // generator output resized to the crop
  const float sy = (float)p.R / (float)ch, sx = (float)p.R / (float)cw;
  int y0, y1, x0, x1;
  float l0y, l1y, l0x, l1x;
  bilinear_coord(py, sy, p.R, y0, y1, l0y, l1y);
  bilinear_coord(px, sx, p.R, x0, x1, l0x, l1x);
  const size_t plane = (size_t)p.H * p.W, oplane = (size_t)p.R * p.R;
  const size_t at = (size_t)(p.y_min + py) * p.W + p.x_min + px;
  auto to255 = [](float v) {
    float t = MIGAN_FMUL_RN(MIGAN_FADD_RN(MIGAN_FMUL_RN(v, 0.5f), 0.5f), 255.0f);    // ((y * 0.5 + 0.5) * 255)
    return fminf(fmaxf(t, 0.0f), 255.0f);
  };
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* q = p.y + c * oplane;
    const float o = bilinear_mix(to255(q[y0 * p.R + x0]), to255(q[y0 * p.R + x1]), to255(q[y1 * p.R + x0]), to255(q[y1 * p.R + x1]), l0x, l1x, l0y, l1y);
    const float img = (float)p.image[c * plane + at];
    float v = MIGAN_FADD_RN(MIGAN_FMUL_RN(img, mk), MIGAN_FMUL_RN(o, MIGAN_FSUB_RN(1.0f, mk)));
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    p.image[c * plane + at] = (unsigned char)(int)v;
  }
}
#endif  // MIGAN_TEMPLATE_KERNELS_ONLY

}  // namespace migan
