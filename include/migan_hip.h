/* migan_hip.h -- C ABI of libmigan_hip.so: MI355X (gfx950) MI-GAN generator forward.
 *
 * Drop-in boundary for the hot path of Picsart-AI-Research/MI-GAN,
 *   lib/model_zoo/migan_inference.py::Generator        (reference :355-369)
 * i.e. what a binding for that module would call instead of torch.nn.Conv2d / F.pad /
 * nn.Upsample.  Plain pointers and sizes only: no torch, no HIP types in the signatures
 * (`stream` is a hipStream_t passed as void*; 0 = the default stream).
 *
 * Conventions
 *   - every function returns 0 on success, a MIGAN_E* code otherwise; the message for the last
 *     failure on the calling thread is migan_last_error().  Nothing throws across the boundary.
 *   - all tensors are device pointers owned by the caller; parameters, the network input and the network output are
 *     fp32 (the reference's dtype).  Weights are passed in the
 *     reference's own state_dict layouts (conv weights [Co][Ci][kh][kw], etc.) and are read in
 *     place: they must stay valid and unmodified-in-address until the next migan_set_weight /
 *     migan_destroy.  The library allocates no device memory.
 *   - activations inside the library are NHWC; the network input/output keep the reference's
 *     NCHW contract ([N,4,R,R] -> [N,3,R,R], reference :362-369).
 *   - all launches are asynchronous on `stream`; no device synchronisation happens inside
 *     migan_forward.  A handle is not safe for concurrent forwards (same as an nn.Module whose
 *     forward uses in-place ops, reference :21,:167,:313); use one handle per stream.
 */
#ifndef MIGAN_HIP_H_
#define MIGAN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIGAN_OK 0
#define MIGAN_EINVAL 1      /* bad argument (shape, name, null pointer, unsupported resolution) */
#define MIGAN_ESTATE 2      /* call order: weights missing / not committed */
#define MIGAN_ERUNTIME 3    /* HIP runtime error (launch failure, copy failure) */
#define MIGAN_EUNSUPPORTED 4 /* checkpoint uses FIR taps other than setup_filter([1,3,3,1]) */

/* Storage format of the activation tensors BETWEEN layers (feature maps and skip tensors, NHWC inside the library).
 * Parameters, network input/output, the running RGB image and all arithmetic (depthwise, activation, FIR, 1x1 accumulate,
 * epilogue) are fp32 in every mode; the 16-bit modes round each SeparableConv2d output (after the skip add, where there is
 * one) once, to nearest even, when it is written -- BASELINE configs[1] ("bf16").  MIGAN_DTYPE_F32 is the reference's
 * own precision (configs[2], the <= 1e-3 parity configuration). */
#define MIGAN_DTYPE_F32 0
#define MIGAN_DTYPE_BF16 1
#define MIGAN_DTYPE_F16 2

/* How the 1x1 convolutions are multiplied (see migan_gemm_variant below). */
#define MIGAN_GEMM_DEFAULT (-1)
#define MIGAN_GEMM_F32 0
#define MIGAN_GEMM_BF16X3 1
#define MIGAN_GEMM_F16X2 2
#define MIGAN_GEMM_F16 3          /* 16-bit activation storage only (its default) */

typedef struct migan_handle migan_handle;

/* Generator(resolution) -- reference :356-360.  resolution must be a power of two in [8,512]
 * (the reference raises ValueError for non powers of two, :215-216,:330-331 -> MIGAN_EINVAL).
 * dtype: MIGAN_DTYPE_* (activation storage). */
int migan_create(int resolution, int dtype, int device, migan_handle** out);
int migan_destroy(migan_handle* h);

/* Per-handle GEMM variant.  fp32 storage: f32 / bf16x3 / f16x2 (initially the process default, environment MIGAN_GEMM), all
 * fp32-grade.  16-bit storage: "f16" (default: both operands of the 1x1 convolutions rounded to fp16 -- 11-bit significands,
 * after the same exact power-of-two scaling as f16x2 -- one v_mfma_f32_32x32x16_f16 per product, fp32 accumulate: the
 * "bf16 config" of BASELINE configs[1] / SURVEY 8d, whose rounding sits below the bf16 storage rounding) or f16x2 (operands exact).
 * Re-plans the launch sequence; the workspace size may change. */
int migan_set_gemm(migan_handle* h, int variant);
int migan_get_gemm(const migan_handle* h, int* variant);
/* on != 0: the caller asserts that the conv2 weights do not change between forwards.  The 16-bit operand planes of the
 * split GEMM variants (a per-forward preparation pass otherwise, so that in-place parameter updates are always seen) are
 * then written once -- by the first forward after this call, after any migan_set_weight / migan_commit, or when the
 * workspace pointer or stream differs from the one they were prepared on -- and reused.  In-place writes to a bound weight
 * tensor while the assertion is on are NOT seen: call migan_assume_static_weights(h, 1) again (or migan_commit) after them. */
int migan_assume_static_weights(migan_handle* h, int on);
/* streams = n > 1 (default 2, at most 4): a forward of >= 8 n images runs as n sub-batches (whole groups of 8 images) on n HIP
 * streams -- the caller's and n - 1 the handle owns, forked and joined with events, no host synchronisation -- each starting
 * when the previous one is a few layers in: the low-resolution layers of one sub-batch (a few dozen workgroups each) and the
 * tail of every launch overlap full-size layers of the others.  Results are bit-identical to streams = 1 (every launch on the
 * caller's stream).  Changes migan_workspace_bytes. */
int migan_set_streams(migan_handle* h, int streams);

/* state_dict schema (same keys, shapes and parameter/buffer split as
 * reference Generator(resolution).state_dict(); 177 entries at 512, 154 at 256). */
int migan_num_weights(const migan_handle* h, int* n);
int migan_weight_info(const migan_handle* h, int index, const char** name,
                      int64_t shape[4], int* ndim, int* is_buffer);

/* load_state_dict: bind one tensor by its reference key (reference scripts/demo.py:110).
 * Unknown key or shape mismatch -> MIGAN_EINVAL, as load_state_dict(strict=True) would raise. */
int migan_set_weight(migan_handle* h, const char* name, const void* dev_ptr,
                     const int64_t* shape, int ndim);
/* Check that every entry is bound and that the FIR parameters/buffers hold the constants the
 * kernels implement in closed form (filter.weight == setup_filter([1,3,3,1]) reference :71-72,
 * :95-96; filter_const == even/even pattern :83-85).  Copies those small tensors to the host
 * (synchronises `stream`).  Must be called after the last migan_set_weight. */
int migan_commit(migan_handle* h, void* stream);

/* Scratch (skip tensors, ping-pong activations, running RGB image) for a batch of n images. */
int migan_workspace_bytes(const migan_handle* h, int batch, size_t* bytes);

/* Generator.forward (reference :362-369).  x: [batch,4,R,R] = cat([mask-0.5, img*mask])
 * (reference scripts/demo.py:56-66); y: [batch,3,R,R].  x is not modified. */
int migan_forward(migan_handle* h, const void* x_nchw, void* y_nchw, int batch,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Fully convolutional forward (reference README.md:87): x [batch,4,height,width] ->
 * y [batch,3,height,width], height and width positive multiples of resolution / 4.  Block b<res> runs at
 * (height*res/resolution) x (width*res/resolution); the reference's two fixed-size constants are made "dynamic" the way
 * its README asks: filter_const (the zero-insertion mask of Upsample2d, :85) is implicit in the polyphase FIR, and each
 * noise_const [res,res] (:149) is tiled periodically and cropped to its layer's size (== the reference module run with
 * those buffers replaced by noise_const.repeat(...)[:h,:w]).  height = width = resolution is migan_forward. */
int migan_workspace_bytes_hw(migan_handle* h, int batch, int height, int width, size_t* bytes);
int migan_forward_hw(migan_handle* h, const void* x_nchw, void* y_nchw, int batch, int height, int width,
                     void* workspace, size_t workspace_bytes, void* stream);

/* uint8 in, uint8 out (reference scripts/demo.py:56-66,135-140 at network resolution):
 * img [batch][R][R][3] uint8 HWC + mask [batch][R][R] uint8 (255 = known pixel) -> composited uint8 image
 * [batch][R][R][3] = img where the mask is 255, (y*0.5+0.5).clamp(0,1)*255 -> uint8 elsewhere.  preprocess() is computed
 * inside the first layer's tile builder (no fp32 network input is materialised) and the post-processing + composition
 * inside the last ToRGB epilogue (no fp32 network output either); bit-identical to migan_pack_input -> migan_forward ->
 * migan_compose_output. */
typedef struct migan_io_u8 {
  const void* img;
  const void* mask;
  void* out;
} migan_io_u8;
int migan_forward_u8(migan_handle* h, const void* img_hwc_u8, const void* mask_u8, void* out_hwc_u8, int batch,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- the deployed pipeline around the generator (second half) ----
 * reference scripts/create_onnx_pipeline.py::MIGAN_Pipeline (:118-264), the module the reference exports as
 * migan_pipeline_v2.onnx: any-size uint8 image [3][H][W] (CHW) + mask [H][W] (255 = known pixel), batch 1.
 *   migan_pipeline_mask_resize  the nearest resize of the mask to the image's size (:256), when their sizes differ
 *   migan_pipeline_bbox   get_masked_bbox (:132-231): bbox = {x_min, x_max, y_min, y_max} of the masked region, padded
 *                         and grown to at least resolution x resolution, clipped to the image.  Row / column flags are
 *                         reduced on the device; the call synchronises `stream` to read them (the crop size decides the
 *                         launch sizes of everything after it -- the reference has the same dependency).
 *   migan_pipeline_pre    preprocess (:233-239) of the crop: bilinear resize (torchvision tensor resize: no antialias,
 *                         rounded back to uint8) of the image, nearest resize of the mask, x = cat([mask/255 - 0.5,
 *                         (image*2/255 - 1) * mask/255]) -> x [1][4][R][R] fp32, the generator's input
 *   migan_pipeline_post   postprocess (:241-250) + paste (:263), in place on the image: y -> ((y*0.5+0.5)*255).clamp,
 *                         bilinear resize to the crop, mask feathering (3x3 max-pool, 5x5 gaussian with reflect padding,
 *                         GaussianSmoothing :51-115), composed = image*mask + output*(1-mask), clamp, truncate to uint8.
 *                         gauss25: the module's 5x5 weight buffer (host pointer), or NULL for kernel_size=5, sigma=1.
 * `scratch` is migan_pipeline_scratch_bytes(H, W) bytes of device memory, shared by the three calls.  Results: bbox and
 * the uint8 resize are exact; fp32 sums may associate differently from ATen's (<= 1 uint8 step in the result). */
/* tvF.resize(mask, image size, NEAREST), the first line of MIGAN_Pipeline.forward (:256): only needed when the mask does not
 * already have the image's size.  out [height][width] uint8. */
int migan_pipeline_mask_resize(const void* mask_u8, int mask_height, int mask_width, void* out_u8, int height, int width,
                               void* stream);
int migan_pipeline_scratch_bytes(int height, int width, size_t* bytes);
int migan_pipeline_bbox(const void* mask_u8, int height, int width, int resolution, int padding, void* scratch,
                        int bbox[4], void* stream);
int migan_pipeline_pre(const void* image_chw_u8, const void* mask_u8, int height, int width, const int bbox[4],
                       int resolution, void* x_nchw, void* stream);
int migan_pipeline_post(void* image_chw_u8, const void* mask_u8, int height, int width, const int bbox[4], int resolution,
                        const void* y_nchw, const float* gauss25, void* scratch, void* stream);

/* ---- measurement and debugging ---------------------------------------------------------- */

/* The forward is a fixed sequence of kernel launches (one per SeparableConv2d, plus un-fused
 * ToRGB launches at low resolution). */
int migan_num_launches(const migan_handle* h, int* n);
/* layer: reference module path ("encoder.b512.conv1", "synthesis.b8.torgb", ...);
 * kernel: kernel symbol as rocprofv3 prints it; flops / bytes: algorithmic work PER IMAGE
 * (2*MAC of every fused stage; one read of each input, one write of each output). */
int migan_launch_info(const migan_handle* h, int index, const char** layer, const char** kernel,
                      double* flops_per_image, double* mfma_flops_per_image,
                      double* bytes_per_image, int* workgroups_batch1);
/* Sub-batch hand-over for multi-GPU callers ("split the shard into 2-4 micro-batches and gather chunk k while computing
 * k+1").  A forward of `batch` images runs as the sub-batches migan_forward_split reports (batches of >= 16 images: two, see
 * migan_set_streams).  migan_forward_parts is migan_forward with the hand-over made explicit: sub-batch k writes its images to
 * y_parts[k] ([part_batch[k]][3][R][R], any device address: e.g. this rank's slice of a collective's receive buffer, so that the
 * all-gather needs no local copy) and, for k >= 1, runs on the CALLER's stream part_streams[k - 1] instead of a stream of the handle.
 * The streams are NOT joined: whatever the caller enqueues on `stream` after the call is ordered behind sub-batch 0 only, work on
 * part_streams[k - 1] behind sub-batch k -- so the collective of sub-batch 0's shard can start while sub-batch 1 still computes.  The
 * caller must make `stream` wait for every part stream before the next forward on this handle / workspace.  No reference counterpart
 * (the reference has no multi-GPU inference path); mi-gan_amd/distributed.py::OutputGather.forward_and_submit is the caller. */
int migan_forward_split(const migan_handle* h, int batch, int part_batch[4], int* n_parts);
int migan_forward_parts(migan_handle* h, const void* x_nchw, void* const* y_parts, int batch,
                        void* workspace, size_t workspace_bytes, void* stream,
                        void* const* part_streams, int n_part_streams, int part_batch[4], int* n_parts);
/* Same as migan_forward, with a hipEvent pair around every launch on `stream`;
 * layer_ms[i] = duration of launch i.  Synchronises the stream before returning. */
int migan_forward_timed(migan_handle* h, const void* x_nchw, void* y_nchw, int batch,
                        void* workspace, size_t workspace_bytes, void* stream,
                        float* layer_ms, int n_layer_ms);
/* keep_intermediates != 0: every layer writes to its own workspace region (no ping-pong) so
 * tests can compare each reference module output.  Changes migan_workspace_bytes. */
int migan_set_debug(migan_handle* h, int keep_intermediates);
/* Location of a layer output inside the workspace after a forward with keep_intermediates:
 * NHWC [batch][r][r][c] for "<block>.conv1|conv2", planar [batch][3][r][r] for "<block>.img". */
int migan_debug_tensor(const migan_handle* h, int batch, const char* layer,
                       size_t* byte_offset, int64_t shape[4]);

/* ---- single operator --------------------------------------------------------------------- */

/* One SeparableConv2d.forward (reference :106-170) on NHWC tensors; used by the operator-level
 * parity tests.  Pointers follow the reference parameter layouts; optional ones may be null. */
typedef struct migan_sepconv_desc {
  const void* x;              /* NHWC [batch][res_in][res_in][cin]; with fromrgb: NCHW [batch][4][res_in][res_in] */
  void* y;                    /* NHWC [batch][res_out][res_out][cout] */
  const void* skip;           /* NHWC like y, added after the activation (reference :272,:305) */
  const void* conv1_weight;   /* [cin][1][3][3] */
  const void* conv1_bias;     /* [cin] */
  const void* conv2_weight;   /* [cout][cin][1][1] */
  const void* noise_const;    /* [res_out][res_out] or null */
  const void* noise_strength; /* scalar (device) */
  const void* fromrgb_weight; /* [cin][4][1][1] or null: x = act(fromrgb(x)) first (reference :193-196) */
  const void* fromrgb_bias;   /* [cin] */
  const void* torgb_weight;   /* [3][cout][1][1] or null: also produce img (reference :308-313); fused into the epilogue when one
                                 workgroup owns all of cout, else a second launch on y */
  const void* torgb_bias;     /* [3] */
  const void* img_prev;       /* planar [batch][3][res_out/2][res_out/2] or null */
  void* img_out;              /* planar [batch][3][res_out][res_out] */
  int batch, cin, cout, res_in;
  int down;                   /* 1 or 2 (Downsample2d, reference :58-76) */
  int up;                     /* 1 or 2 (Upsample2d, reference :79-103) */
  void* scratch;              /* down == 2 only: batch*(res_in/2)^2*cin floats (output of the depthwise+FIR kernel) */
  size_t scratch_bytes;
  void* wsplit;               /* optional: 16 + 3*cout*cin*2 bytes (16-byte aligned) for the 16-bit weight planes of the split GEMM
                                 variants; null or too small -> the exact fp32-MFMA kernels run */
  size_t wsplit_bytes;
  int gemm;                   /* MIGAN_GEMM_*; MIGAN_GEMM_DEFAULT (-1) = the process default */
  int dtype;                  /* MIGAN_DTYPE_*: storage format of x (unless fromrgb), y and skip; scratch stays fp32 */
  int width_in;               /* 0: square input (res_in x res_in); otherwise the input is res_in rows x width_in columns */
} migan_sepconv_desc;
int migan_sepconv_forward(const migan_sepconv_desc* d, void* stream);

/* ---- either side of the forward ---------------------------------- */

/* preprocess() of reference scripts/demo.py:56-66 at network resolution: uint8 RGB image [batch][R][R][3]
 * (HWC, np.array(PIL image)) and uint8 mask [batch][R][R] (255 = known pixel, everything else = hole,
 * demo.py:44,60) -> x = cat([mask - 0.5, (img * 2 / 255 - 1) * mask]) as [batch][4][R][R] fp32, bit-exact.
 * Pointers: uint8 tensors 4-byte aligned, fp32 tensors 16-byte aligned. */
int migan_pack_input(const void* img_hwc_u8, const void* mask_u8, void* x_nchw, int batch, int resolution, void* stream);
/* reference scripts/demo.py:135-140 at network resolution: (y * 0.5 + 0.5).clamp(0, 1) * 255 -> uint8,
 * composed = img * mask + result * (1 - mask); y [batch][3][R][R] fp32 -> out [batch][R][R][3] uint8, bit-exact. */
int migan_compose_output(const void* y_nchw, const void* img_hwc_u8, const void* mask_u8, void* out_hwc_u8,
                         int batch, int resolution, void* stream);

/* Process-wide tuning knobs, the run-time form of the MIGAN_* environment variables (experiments and tests):
 * "kc16" (bit mask: 16-channel K chunks for the 64-channel 512x512 layers), "kc16_minw", "w3" (the same mask: those layers on 32-channel chunks at 3 workgroups per CU), "wide", "nt256", "persist_min",
 * "persist_grid", "streams", "stagger", "stagger_pct", "single_b", "debug_split", "pipe" (bit mask: software-pipelined persistent kernels for
 * 1 plain, 2 fused-FromRGB, 4 FIR-up layers), "pipe_grid", "pipe_min_tiles".  Applies to handles created or re-planned afterwards. */
int migan_set_tuning(const char* key, int value);

const char* migan_last_error(void);
/* Symbol (as rocprofv3 prints it) of the fused-SeparableConv2d kernel the calling thread launched last, "" before the first launch:
 * which tile form / schedule the plan picked for a layer and batch (diagnostics and tests). */
const char* migan_last_kernel(void);
/* What the clamp of lrelu_agc (reference :21-23) does with a NaN in THIS build of the library: "propagate" (default build: a NaN stays a NaN,
 * like Tensor.clamp in the reference module -- SURVEY 8c "follow torch") or "clamp" (libmigan_hip_nanclamp.so, built with -DMIGAN_NAN_CLAMP:
 * v_med3_f32 maps it to -256, like the reference's CUDA plugin bias_act.cu:139; ~2 % faster).  Finite inputs give the same bits in both. */
const char* migan_nan_policy(void);
/* "hip:gfx950" for the product library. */
const char* migan_backend(void);
/* The process default of how the 1x1 convolutions are multiplied (environment MIGAN_GEMM=f32|bf16x3|f16x2, read once
 * per process; per handle: migan_set_gemm):
 *   "f32"    v_mfma_f32_32x32x2_f32, exact fp32 products;
 *   "bf16x3" each fp32 operand split into three bf16 pieces, the six products of order <= 2^-16
 *            on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-grade accuracy at 6/16 of the MFMA cost;
 *   "f16x2"  (default) operands scaled by exact powers of two into fp16's normal range (activations by the
 *            +-256 clamp of lrelu_agc, each weight tensor by its largest magnitude) and split into two fp16
 *            pieces; three products on v_mfma_f32_32x32x16_f16, fp32 accumulation: fp32-grade accuracy at
 *            3/16 of the MFMA cost. */
const char* migan_gemm_variant(void);
int migan_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MIGAN_HIP_H_ */
