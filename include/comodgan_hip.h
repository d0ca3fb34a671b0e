/* comodgan_hip.h -- C ABI of libmigan_hip.so, second model: MI355X (gfx950) Co-Mod-GAN generator forward
 * (BASELINE.json configs[4]).
 *
 * Drop-in boundary for
 *   lib/model_zoo/comodgan.py::Generator.forward            (reference :435-455)
 * = Mapping.forward (lib/model_zoo/stylegan.py:396-439) + Encoder.forward (comodgan.py:192-204) +
 *   Synthesis.forward (comodgan.py:395-420), i.e. what `--model-name comodgan-256|512` of the reference's
 *   scripts/demo.py:95-106 runs.  Same conventions as migan_hip.h: plain pointers and sizes, every function returns
 *   0 or a MIGAN_E* code (message: migan_last_error()), fp32 device tensors owned by the caller, weights bound by
 *   their reference state_dict key in the reference layout and read in place, the library allocates no device
 *   memory, all launches asynchronous on `stream`.
 */
#ifndef COMODGAN_HIP_H_
#define COMODGAN_HIP_H_

#include "migan_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct comodgan_handle comodgan_handle;

/* Constructor arguments that fix the tensor shapes; reference defaults in brackets. */
typedef struct comodgan_config {
  int resolution;   /* Encoder / Synthesis resolution, power of two in [8,512] (comodgan.py:115,349; ValueError otherwise :134-135) */
  int ch_base;      /* [32768] comodgan.py:119,351 */
  int ch_max;       /* [512]   comodgan.py:120,352 */
  int z_dim;        /* [512]   stylegan.py:358 */
  int w_dim;        /* [512]   stylegan.py:360 */
  int w0_dim;       /* [1024]  Encoder oc_n (comodgan.py:118) == Synthesis w0_dim (:348) */
  int map_layers;   /* [8]     stylegan.py:362 */
  int num_ws;       /* Mapping(num_ws=...): 14 at 256, 16 at 512 (scripts/demo.py:96,102; comodgan.py:367-370) */
} comodgan_config;

/* CoModGANGenerator(CoModGANMapping(num_ws), CoModGANEncoder(resolution), CoModGANSynthesis(resolution)) */
int comodgan_create(const comodgan_config* cfg, int device, comodgan_handle** out);
int comodgan_destroy(comodgan_handle* h);

/* state_dict schema: same keys, shapes and parameter/buffer split as the reference Generator(...).state_dict()
 * (204 entries at 512, 180 at 256). */
int comodgan_num_weights(const comodgan_handle* h, int* n);
int comodgan_weight_info(const comodgan_handle* h, int index, const char** name, int64_t shape[4], int* ndim, int* is_buffer);
/* load_state_dict (scripts/demo.py:110): unknown key or shape mismatch -> MIGAN_EINVAL */
int comodgan_set_weight(comodgan_handle* h, const char* name, const void* dev_ptr, const int64_t* shape, int ndim);
/* Every entry bound; every resample_filter buffer equals upfirdn2d.setup_filter([1,3,3,1]) (the kernels implement
 * that FIR in closed form; anything else -> MIGAN_EUNSUPPORTED).  Synchronises `stream`. */
int comodgan_commit(comodgan_handle* h, void* stream);

int comodgan_workspace_bytes(const comodgan_handle* h, int batch, size_t* bytes);

/* Every forward starts by writing the fp16 operand planes and the demodulation statistics of the 3x3 weights into the head of the
 * workspace (weights are read in place, so in-place parameter updates are always seen).  on != 0: the caller asserts that the
 * weight VALUES have not changed since the previous comodgan_forward and that the workspace passed then is passed again with its
 * contents intact; the preparation launches are then skipped (they are batch-size independent and sit at fixed workspace offsets).
 * comodgan_set_weight / comodgan_commit, another workspace pointer or on == 0 bring them back for the next forward.  Default 0. */
int comodgan_assume_static_weights(comodgan_handle* h, int on);

#define COMODGAN_NOISE_NONE 0    /* noise_mode='none'  (stylegan.py:283-289) */
#define COMODGAN_NOISE_CONST 1   /* noise_mode='const': noise_const * noise_strength */
#define COMODGAN_NOISE_RANDOM 2  /* noise_mode='random': `noise` holds, for every synthesis layer in forward order
                                    (b4.conv, b8.conv0, b8.conv1, ..., bR.conv1), a standard-normal [batch][res][res] block
                                    drawn by the caller (what torch.randn([N,1,res,res]) draws at stylegan.py:284-285);
                                    comodgan_noise_floats() floats per image in total */
int comodgan_noise_floats(const comodgan_handle* h, size_t* floats_per_image);

/* Generator.forward(x, z, c=None, truncation_psi, truncation_cutoff (comodgan_set_truncation_cutoff), noise_mode) (comodgan.py:435-455).
 * x: [batch,4,R,R] = cat([mask-0.5, img*mask]) (scripts/demo.py:56-66); z: [batch,z_dim]; y: [batch,3,R,R]. */
int comodgan_forward(comodgan_handle* h, const void* x_nchw, const void* z, void* y_nchw, int batch,
                     float truncation_psi, int noise_mode, const void* noise,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- measurement and debugging (same meaning as the migan_* counterparts) ---- */
int comodgan_num_launches(const comodgan_handle* h, int* n);
int comodgan_launch_info(const comodgan_handle* h, int index, const char** layer, const char** kernel,
                         double* flops_per_image, double* mfma_flops_per_image, double* bytes_per_image);
int comodgan_forward_timed(comodgan_handle* h, const void* x_nchw, const void* z, void* y_nchw, int batch,
                           float truncation_psi, int noise_mode, const void* noise,
                           void* workspace, size_t workspace_bytes, void* stream, float* launch_ms, int n_launch_ms);
/* keep_intermediates != 0: every layer output keeps its own workspace region. */
/* truncation_cutoff of Generator.forward / MappingNetwork.forward (comodgan.py:435,446; stylegan.py:403,432-437): with
 * truncation_psi != 1 only the first `cutoff` rows of ws are pulled towards w_avg; the layers that read later rows (comodgan.py:399-405:
 * b4.conv 0, b4.torgb 1, then conv0 / conv1 / torgb of block j at 1+2j, 2+2j, 3+2j) get the un-truncated w.  -1 = None (all rows).
 * Changes comodgan_workspace_bytes. */
int comodgan_set_truncation_cutoff(comodgan_handle* h, int cutoff);
int comodgan_set_debug(comodgan_handle* h, int keep_intermediates);
/* After a forward with keep_intermediates: NHWC [batch][r][r][c] for "encoder.bR.conv0|conv1", "encoder.b4.conv",
 * "synthesis.b4.conv", "synthesis.bR.conv0|conv1"; planar [batch][3][r][r] for "synthesis.bR.img";
 * [batch][d] for "mapping" (w) and "encoder.b4.fc" (the global code). */
int comodgan_debug_tensor(const comodgan_handle* h, int batch, const char* layer, size_t* byte_offset, int64_t shape[4], int* ndim);

#ifdef __cplusplus
}
#endif
#endif /* COMODGAN_HIP_H_ */
