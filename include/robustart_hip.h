/*
 * robustart_hip.h -- C-ABI of the MI355X (gfx950) hot path of RobustART's AddNoise:
 * ImageNet-C corruption kernels, adversarial step kernels and the eval-mode
 * ResNet-50 forward / backward-to-input engine.
 *
 * The reference (DIG-Beihang/RobustART) is pure Python and has no FFI of its own; this
 * header is the boundary a maintainer would bind with ctypes.  Each
 * entry point cites the reference function it replaces (paths relative to
 * RobustART/noise/utils/ unless stated).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless its name ends in _host;
 *  - the library never allocates or frees caller memory; scratch comes from the caller's
 *    `workspace` (size from the matching *_workspace_bytes query);
 *  - all work is enqueued on `stream` (a hipStream_t); no hidden device synchronisation;
 *  - return value: 0 = RART_OK, otherwise an rart_status; rart_last_error_string() gives
 *    the thread-local message.  Nothing throws across the boundary or calls exit();
 *  - randomness is counter based: a draw is a pure function of
 *    (seed, global sample index = sample_offset + i, element index), independent of launch
 *    geometry and of how samples are sharded over GPUs.  When `injected` is non-NULL the
 *    kernels consume the caller's draws instead (parity mode, fp64 where the reference is
 *    fp64) -- layouts are listed per corruption below.
 */
#ifndef ROBUSTART_HIP_H
#define ROBUSTART_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rart_stream_t; /* hipStream_t */

typedef enum rart_status {
  RART_OK = 0,
  RART_ERR_INVALID = 1,     /* bad argument (null pointer, severity outside 1..5, unknown id ...) */
  RART_ERR_UNSUPPORTED = 2, /* valid request the library does not implement (e.g. a conv geometry no kernel instance covers) */
  RART_ERR_WORKSPACE = 3,   /* workspace missing or too small */
  RART_ERR_HIP = 4          /* a HIP runtime call failed */
} rart_status;

/* corruption ids == index into corruption_tuple (imagenet_c/__init__.py:5-8) */
enum {
  RART_GAUSSIAN_NOISE = 0, RART_SHOT_NOISE = 1, RART_IMPULSE_NOISE = 2, RART_DEFOCUS_BLUR = 3,
  RART_GLASS_BLUR = 4, RART_MOTION_BLUR = 5, RART_ZOOM_BLUR = 6, RART_SNOW = 7, RART_FROST = 8,
  RART_FOG = 9, RART_BRIGHTNESS = 10, RART_CONTRAST = 11, RART_ELASTIC_TRANSFORM = 12,
  RART_PIXELATE = 13, RART_JPEG_COMPRESSION = 14, RART_SPECKLE_NOISE = 15, RART_GAUSSIAN_BLUR = 16,
  RART_SPATTER = 17, RART_SATURATE = 18, RART_NUM_CORRUPTIONS = 19
};

/* ABI version of THIS header.  Bumped whenever a struct layout or a signature changes (round 3 -> 4: rart_conv_desc grew
 * its tap arrays from 16 to 32 entries and gained dst_pair_off / res_pair_off; several attack entries gained a per-row
 * sample-index pointer; round 4 -> 5: rart_stencil_fixed_point_info added).  A caller compiled against another header must refuse to run:
 * compare with rart_version(). */
#define RART_ABI_VERSION 109
int rart_version(void);
const char* rart_last_error_string(void);
/* name of corruption id (static string), NULL if out of range */
const char* rart_corruption_name(int corruption_id);

/* ---------------------------------------------------------------------------------------
 * ImageNet-C corruptions.  Replaces corrupt() + the per-image Python loop of
 * add_noise_for_imagenet_c (add_noise_utils.py:22-31, imagenet_c/__init__.py:13-35,
 * imagenet_c/corruptions.py:122-424).
 *
 * in / out : uint8 NHWC (n, h, w, 3); may alias (in-place, as the reference mutates its input).
 *            Corruptions the reference hard-codes to 224x224 (glass_blur, fog, frost, snow,
 *            pixelate, elastic_transform) require h == w == 224 here too.
 * severity : 1..5.
 * injected : NULL (native counter-based RNG) or an array of device pointers holding what
 *            np.random returned inside the reference function, per image, C order:
 *   gaussian_noise, speckle_noise : [0] double noise[n][h][w][3]   (np.random.normal(scale=c))
 *   shot_noise                    : [0] int32  counts[n][h][w][3]  (np.random.poisson)
 *   impulse_noise                 : [0] uint8  code[n][h][w][3]    (0 keep, 1 salt, 2 pepper)
 *   glass_blur                    : [0] int8   dxdy[n][iters][224-2d][224-2d][2]   (every value in [-d, d): numpy's randint(-d, d);
 *                                       the overlapped copy chain packs dx + d, dy + d into nibbles -- the Python entry refuses others)
 *   motion_blur                   : [0] double angle[n]
 *   snow                          : [0] double layer[n][224][224]  [1] double angle[n]
 *   frost                         : [0] uint8  texture_crop[n][224][224][3]  (REQUIRED always: the
 *                                       reference's frost photos are not in its repository)
 *   fog                           : [0] double uniform[n][65535]   (plasma_fractal draws, call order)
 *   elastic_transform             : [0] float  jitter[n][3][2]  [1] double field_x[n][224][224]
 *                                   [2] double field_y[n][224][224]
 *                                   (severities 1-2: the field filter runs as fp64 matrix products against the folded reflect-filter
 *                                   matrix, resident on the device after the first call -- not scipy's summation order: the fields
 *                                   differ in the last bits, the images stay within the corruption's stated tolerance (<= 1 LSB on <= 1e-4
 *                                   of the pixels) and are NOT guaranteed bit-identical to the ordered kernels; RART_ELASTIC_ORDERED=1 in the
 *                                   environment selects the ordered, bit-exact kernels.  The first severity-1 / 2 call per device
 *                                   allocates: keep it outside stream capture)
 *   spatter                       : [0] double layer[n][224][224]
 *   others                        : ignored
 * workspace: rart_corrupt_workspace_bytes(...) bytes, 256-byte aligned.
 * ------------------------------------------------------------------------------------- */
size_t rart_corrupt_workspace_bytes(int corruption_id, int severity, int n, int h, int w);

int rart_corrupt_u8(const uint8_t* in, uint8_t* out, int n, int h, int w,
                    int corruption_id, int severity,
                    uint64_t seed, uint64_t sample_offset,
                    const void* const* injected_host_array, int n_injected,
                    void* workspace, size_t workspace_bytes, rart_stream_t stream);

/* gaussian_noise / speckle_noise at ns <= 8 (severity, seed) pairs of ONE source batch in one launch: every 1 KiB chunk is read once and
 * written ns times, each output with its own field.  outs[i] is bit-identical to rart_corrupt_u8(in, outs[i], n, h, w, corruption_id,
 * severities[i], seeds[i], sample_offset, NULL, 0, ...) with the native matrix-core generator.  ImageNet-C generation corrupts every
 * image at five severities (imagenet_c/__init__.py:13-35 once per (image, severity)): five launches move 10 x the batch through HBM,
 * this one 6 x.  outs: HOST array of ns device pointers.  RART_ERR_UNSUPPORTED unless h*w*3 % 1024 == 0 with 16-byte aligned
 * buffers and the matrix-core generator selected (callers then issue ns rart_corrupt_u8 calls). */
int rart_noise_multi_u8(const uint8_t* in, uint8_t* const* outs, int ns, int n, int h, int w, int corruption_id, const int* severities,
                        const uint64_t* seeds, uint64_t sample_offset, rart_stream_t stream);

/* frost from device-resident photographs (round 5).  The reference (imagenet_c/corruptions.py:249-266) opens one of its frost photographs
 * per image -- randint(5) over a six-entry list -- crops 224 x 224 at a random origin and blends; the photographs are not in its repository, so
 * the caller registers them (robustart_amd.noise.imagenet_c.set_frost_textures) and they live on the device as one stack
 * uint8 [k_tex][sh][sw][3] (each padded to the common size; dims_host: host int[2 k_tex] = height, width of each).  The texture index and the
 * crop origin of image i are the counter generator's uniforms of streams 8, 9, 10 at sample sample_offset + i, drawn in the kernel, which reads
 * the crop in place: bit-identical to rart_corrupt_u8(RART_FROST) with those crops injected, without the gather (n x 150 528 B written and
 * read back).  h = w = 224 only, as the reference; in / out 16-byte aligned.  in == out allowed. */
int rart_frost_textures_u8(const uint8_t* in, uint8_t* out, int n, int h, int w, int severity, const uint8_t* stack, int k_tex, int sh, int sw,
                           const int* dims_host, uint64_t seed, uint64_t sample_offset, rart_stream_t stream);

/* Host-only introspection of the fixed-point tables behind the matrix-core stencil paths (round 5; no GPU needed: the "-m 'not gpu'"
 * tests check the tables against the oracle's weights).  defocus_blur (corruptions.py:187-198: 17 x 17 disks, 21 x 21 at severity 5) runs as an
 * exact 2-D integer filter, gaussian_blur (corruptions.py:162-166) and glass_blur's two blurs (corruptions.py:169-184) as an exact
 * separable one: W = round(w 2^frac_bits) split into four signed base-256 digits laid out as v_mfma_i32_16x16x64_i8 A-operand
 * fragments -- fragment (step j, digit d), lane (m = lane & 15, g = lane >> 4), byte i:
 *   kind 1 (2-D):       kernel row 2 j + (g >> 1), window column 16 (g & 1) + i, tap b = column - m; rows m >= 33 - ksize are zero
 *                       (n_steps = (ksize + 1) / 2: 9 for 17 x 17, 11 for 21 x 21 whose row blocks hold 12 outputs)
 *   kind 2 (separable): tap t = 16 g + i - m of the 1-D kernel                                            (n_steps = 1)
 * An output whose fixed-point value lies within `band` units (of 2^-out_frac_bits of an output step) of an integer is recomputed in the
 * reference's fp64 order.  blur_index: 0 for defocus_blur / gaussian_blur; glass_blur has one sigma, index 0.
 * frags: NULL or a HOST buffer of n_steps * 4 * 64 * 16 bytes.  Returns RART_ERR_UNSUPPORTED (kind 0) where the fp64 kernels run. */
typedef struct rart_fixed_point_info {
  int kind, ksize, n_steps, frac_bits, out_frac_bits;
  long long corr, band;
  double max_abs_weight_error;      /* max |W / 2^frac_bits - w| over the taps */
  double sum_abs_weight_error;
} rart_fixed_point_info;
int rart_stencil_fixed_point_info(int corruption_id, int severity, rart_fixed_point_info* info, unsigned char* frags, size_t frags_bytes);

/* ImageNet-S resize operators (imagenet_s_gen.py:19-34,127-166): Pillow's Image.resize((resize_w, resize_h), filter) on
 * uint8 NHWC images followed by a crop, bit-exact with Pillow (Resample.c 22-bit fixed point; Geometry.c for
 * NEAREST).  filter = PIL.Image constant: 0 nearest, 1 bilinear, 2 bicubic, 3 box, 4 hamming, 5 lanczos.
 * out: uint8 [n][crop_h][crop_w][3] = resized[crop_y : crop_y+crop_h, crop_x : crop_x+crop_w]. */
size_t rart_pil_resize_workspace_bytes(int n, int h, int w, int resize_h, int resize_w, int filter, int crop_y, int crop_x,
                                       int crop_h, int crop_w);
int rart_pil_resize_u8(const uint8_t* in, uint8_t* out, int n, int h, int w, int resize_h, int resize_w, int filter,
                       int crop_y, int crop_x, int crop_h, int crop_w, void* workspace, size_t workspace_bytes,
                       rart_stream_t stream);
/* The 'opencv-*' ImageNet-S resize operators (imagenet_s_gen.py:27-33,138-146): cv2.resize(img, (resize_w, resize_h),
 * interpolation) on uint8 NHWC images followed by a crop.  interpolation = cv2 constant: 0 INTER_NEAREST, 1 INTER_LINEAR,
 * 2 INTER_CUBIC, 3 INTER_AREA, 4 INTER_LANCZOS4.  Restates opencv/modules/imgproc/src/resize.cpp (4.x) -- OpenCV is an
 * unpinned dependency absent from the build container: parity unpinned.  out: uint8 [n][crop_h][crop_w][3]. */
size_t rart_cv_resize_workspace_bytes(int n, int h, int w, int resize_h, int resize_w, int interpolation, int crop_y,
                                      int crop_x, int crop_h, int crop_w);
int rart_cv_resize_u8(const uint8_t* in, uint8_t* out, int n, int h, int w, int resize_h, int resize_w, int interpolation,
                      int crop_y, int crop_x, int crop_h, int crop_w, void* workspace, size_t workspace_bytes,
                      rart_stream_t stream);

/* uint8 NHWC -> ImageNet-normalised tensor for the model ((x/255 - mean)/std), i.e. the ToTensor +
 * Normalize step that follows AddNoise in the reference's eval pipeline
 * (exprs/exp/imagenet_c_loop_mini/config_vit_base.yaml:85-99).
 * out_dtype : 0 = fp32, 1 = bf16.   out_layout : 0 = NCHW, 1 = NHWC. */
int rart_u8_to_normalized(const uint8_t* in, void* out, int n, int h, int w,
                          int out_dtype, int out_layout, rart_stream_t stream);
/* x01 fp32 NCHW = u8 NHWC * (1 / 255) (bit-identical to torch's u8.permute(0, 3, 1, 2).float().div(255), which multiplies by the fp32
 * reciprocal of a scalar divisor): the hand-over from
 * the uint8 images of the corruption kernels / datasets to the attack tensors (fp32 NCHW in [0,1], adv/attack.py:20-23) in one kernel. */
int rart_u8_to_unit_f32_nchw(const uint8_t* in, float* out, int n, int h, int w, rart_stream_t stream);

/* Fill helpers exposing the library's counter-based generator (used by the parity tests to
 * replay the native noise field on the host oracle, and by the attack random starts).
 * stream_id selects an independent sub-stream (0..15). */
int rart_rng_uniform_u32(uint32_t* out, int n_samples, size_t elems_per_sample,
                         uint64_t seed, uint64_t sample_offset, int stream_id, rart_stream_t stream);
int rart_rng_normal_f32(float* out, int n_samples, size_t elems_per_sample,
                        uint64_t seed, uint64_t sample_offset, int stream_id, rart_stream_t stream);
/* Normal generator of gaussian_noise / speckle_noise: 1 (default) = matrix-core CLT generator (int8 MFMA
 * against a Hadamard matrix, one random byte per normal, cubic kurtosis correction) whenever a sample is a
 * whole number of 1 KiB chunks; 0 = Threefry + Box-Muller everywhere.  rart_rng_noise_field_f32 replays the
 * N(0,1) field those two corruptions use under the current setting. */
int rart_set_normal_generator(int kind);
/* Measurement aid of bench.py (hbm_roofline_gaussian_noise): a plain device copy of `bytes` (multiple of 16, 16-byte aligned buffers) --
 * variant 0: one 16-byte vector per lane, one wave per 1 KiB, the noise kernels' launch geometry; variant 1: a grid-stride loop over 1 024
 * workgroups.  What a kernel that only moves a launch's bytes reaches at that size is the ceiling the noise kernels are read against. */
int rart_copy_calibration(const void* in, void* out, size_t bytes, int variant, rart_stream_t stream);
int rart_get_normal_generator(void);
int rart_rng_noise_field_f32(float* out, int n_samples, size_t elems_per_sample, uint64_t seed,
                             uint64_t sample_offset, rart_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Adversarial step kernels (x, x0, g: fp32, batch-major, n_per_sample elements per sample,
 * values of x in [0,1] BEFORE normalisation, as in the reference).
 * ------------------------------------------------------------------------------------- */

/* Workspace size (bytes) that satisfies every attack entry point below for `batch` samples. */
size_t rart_attack_workspace_bytes(int batch);

/* x = clip(x0 + U(-eps, eps), lo, hi).  foolbox LinfPGD random start (adv/attack.py:20-23);
 * with clip_lo > clip_hi the clip is skipped = MIM start (Attacks/imfgsm_attack.py:73-74).
 * injected_u: NULL or fp32 U(-eps,eps) draws shaped like x. */
/* row_samples (here and in the entries below that take it): NULL, or int64 [batch] on the device -- the GLOBAL sample index of every
 * row.  Row b then draws at row_samples[b] instead of sample_offset + b: inside AutoAttack the sub-attacks receive the still-robust
 * SUBSET of a batch (autoattack.py:117-136), and with its rows' own indices the draws of a sample do not depend on which other
 * samples survived, i.e. on how the dataset is batched or sharded over GPUs. */
int rart_attack_init_linf(float* x, const float* x0, int batch, size_t n_per_sample, float eps,
                          float clip_lo, float clip_hi, uint64_t seed, uint64_t sample_offset, const int64_t* row_samples,
                          const float* injected_u, rart_stream_t stream);

/* x <- clip(x0 + clip(x + alpha*sign(g) - x0, -eps, eps), 0, 1): one fused PGD-Linf / FGSM step
 * (foolbox BaseGradientDescent.run, adv/attack.py:20-23,30-33). */
int rart_pgd_step_linf(float* x, const float* g, const float* x0, size_t n_elems,
                       float eps, float alpha, rart_stream_t stream);

/* PGD-L2 step (adv/attack.py:25-28): x += alpha*g/max(|g|_2,1e-12); delta *= min(1, eps/max(|delta|_2,1e-12));
 * clip [0,1].  workspace: rart_attack_workspace_bytes(batch). */
int rart_pgd_step_l2(float* x, const float* g, const float* x0, int batch, size_t n_per_sample,
                     float eps, float alpha, void* workspace, size_t workspace_bytes, rart_stream_t stream);
/* PGD-L1 step of ART's ProjectedGradientDescentPyTorch(norm=1) (adv/attack.py:44-49; ART is unpinned in the
 * reference's requirements.txt:25, this is the 1.x algorithm): x = clip(x + eps_step*g/(sum|g| + 1e-7), 0, 1);
 * x = x0 + (x - x0)*min(1, eps/(sum|x - x0| + 1e-7)).  workspace: rart_attack_workspace_bytes(batch). */
int rart_pgd_step_l1(float* x, const float* g, const float* x0, int batch, size_t n_per_sample, float eps, float eps_step,
                     void* workspace, size_t workspace_bytes, rart_stream_t stream);
/* ART random_sphere(norm=1) start: x = clip(x0 + r * s_i e_i / sum e, 0, 1), e ~ Exp(1), s = +-1, r = sqrt(U(0, eps^2)).
 * injected_signed_exp [batch][n] (s_i*e_i) and injected_radius [batch] replace the native draws (both or neither). */
int rart_random_start_l1(float* x, const float* x0, int batch, size_t n_per_sample, float eps, uint64_t seed,
                         uint64_t sample_offset, const int64_t* row_samples, const float* injected_signed_exp, const float* injected_radius,
                         void* workspace, size_t workspace_bytes, rart_stream_t stream);

/* MIM step (Attacks/imfgsm_attack.py:85-90): g/=mean|g| per sample; m = decay*m + g;
 * x = clip(x0 + clip(x + step*sign(m) - x0, +-eps), 0, 1).  workspace: rart_attack_workspace_bytes(batch). */
int rart_mim_step(float* x, float* momentum, const float* g, const float* x0, int batch,
                  size_t n_per_sample, float eps, float step_size, float decay,
                  void* workspace, size_t workspace_bytes, rart_stream_t stream);

/* APGD random start (Attacks/autoattack/autopgd_base.py:213-220,237):
 * x = clip(x0 + eps * t / (max|t| + 1e-12), 0, 1) for Linf, t ~ U(-1,1);
 * x = clip(x0 + eps * t / (|t|_2 + 1e-12), 0, 1) for L2, t ~ N(0,1).
 * norm: 0 = Linf, 1 = L2, 2 = L1 (normal draws divided by their L1 norm: the random start of FAB's restarts, fab_base.py:133-166, which
 * has the same x0 + r t / ||t|| form with r = eps / 2).  injected_t: NULL or the fp32 draws.  workspace: rart_attack_workspace_bytes(batch). */
int rart_apgd_init(float* x, const float* x0, int batch, size_t n_per_sample, int norm, float eps,
                   uint64_t seed, uint64_t sample_offset, const int64_t* row_samples, const float* injected_t,
                   void* workspace, size_t workspace_bytes, rart_stream_t stream);

/* APGD step with momentum and double projection (autopgd_base.py:327-348).
 * x_adv (in/out), x_adv_old (in/out: receives the pre-step x_adv), grad, x0; step_size: per-sample
 * fp32[batch]; a = 1.0 on the first iteration, 0.75 afterwards.  norm: 0 = Linf, 1 = L2.
 * workspace: rart_attack_workspace_bytes(batch) (L2 only). */
int rart_apgd_step(float* x_adv, float* x_adv_old, const float* grad, const float* x0,
                   const float* step_size, int batch, size_t n_per_sample, int norm, float eps, float a,
                   void* workspace, size_t workspace_bytes, rart_stream_t stream);

/* Square attack, Linf (Attacks/autoattack/square.py:228-258).  Start: x_best = clamp(x0 + eps*sign, 0, 1) with one
 * random sign per (image, channel, column); injected_sign: NULL or fp32 [batch][c][w] of +-1.
 * Proposal: x_new = clamp(min(max(x_best + delta, x0-eps), x0+eps), 0, 1), delta = 2*eps*sign_host[ch] inside the
 * s x s window at (vh, vw), shared by the whole batch like the reference. */
int rart_square_init_linf(float* x_best, const float* x0, int batch, int c, int h, int w, float eps, uint64_t seed,
                          uint64_t sample_offset, const int64_t* row_samples, const float* injected_sign, rart_stream_t stream);
/* out[b][j] = +1 / -1: the sign rows Square's L2 / L1 proposals multiply their windows with (square.py:343-345, 455-457), one per
 * (image, channel), from the counter generator on the device: bit 31 of the first Threefry word of counter (index_base + j, stream_id) of
 * row b's sample (the value noise/rng.py host_uniform(...) >= 0.5 gives).  Replaces a numpy draw + host-to-device copy per query.
 * rart_rng_normal_rows_f32: rart_rng_normal_f32 for explicit rows (out[b][:] = the field of sample row_samples[b]). */
int rart_rng_signs_f32(float* out, int batch, int n_per_row, uint64_t seed, uint64_t sample_offset, const int64_t* row_samples,
                       int stream_id, uint32_t index_base, rart_stream_t stream);
int rart_rng_normal_rows_f32(float* out, int batch, size_t elems, uint64_t seed, const int64_t* row_samples, int stream_id,
                             rart_stream_t stream);
int rart_square_propose_linf(float* x_new, const float* x_best, const float* x0, int batch, int c, int h, int w,
                             float eps, int vh, int vw, int s, const float* sign_host, rart_stream_t stream);

/* Expectation over transformation of APGD (autopgd_base.py:271-289, :367-384; AutoAttack version 'rand'): mode 0: acc += g over n fp32
 * elements; mode 1: acc /= divisor (the reference divides by float(eot_iter)). */
int rart_eot_accumulate(float* acc, const float* g, size_t n, int mode, float divisor, rart_stream_t stream);

/* Square attack, norm 2 = L2 / 1 = L1 (Attacks/autoattack/square.py:123-190, 296-530); all pointers are device pointers.
 * rart_square_init_lp: the start perturbation -- tiles of side s on a tiles_h x tiles_w grid from (sp, sp), tile t carries
 *   eta2[transposed[t]] (eta2 = [2][s*s]: eta(s) and its transpose, :172-190) times signs[t][image][channel] (+-1).  L2: out = the start
 *   point clamp(x0 + delta / (||delta||_2 + 1e-12) * eps, 0, 1); L1: out = delta (add rart_l1_project(x0, delta, eps (1 - 1e-6)), :425-426).
 * rart_square_propose_lp: one query -- window (vh, vw) of side s gets eta * signs[image][channel] + delta / (1e-12 + |delta|_window),
 *   rescaled to the budget left (:347-362 / :461-476), window (vh2, vw2) is cleared.  L2: out = the candidate point; L1: out = the new
 *   delta (then rart_l1_project, :483-484).  One workgroup per image; c <= 4. */
int rart_square_init_lp(float* out, const float* x0, int batch, int c, int h, int w, float eps, int norm, int s, int sp, int tiles_h,
                        int tiles_w, const float* eta2, const uint8_t* transposed, const float* signs, rart_stream_t stream);
int rart_square_propose_lp(float* out, const float* x_best, const float* x0, int batch, int c, int h, int w, float eps, int norm, int vh,
                           int vw, int vh2, int vw2, int s, const float* eta, const float* signs, rart_stream_t stream);

/* FAB, Linf (Attacks/autoattack/fab_projections.py:7-59, fab_base.py:168-245).
 * rart_fab_project_linf: per row r, the box-constrained Linf projection step d of points[r] onto {x: w[r].x = b[r]};
 *   rowmax_out[r] = max|d[r]| (nullable).  One workgroup per row, bisection on the monotone piecewise-linear
 *   constraint instead of the reference's argsort (same solution up to fp32 summation order).
 * rart_row_dot / rart_row_absmax_diff: per-row <a,b> and max|a-b|.
 * rart_fab_update: x1 = clamp((x1 + eta*d1)*(1-alpha[r]) + (x0 + eta*d2)*alpha[r], 0, 1).
 * rart_fab_backoff: rows with mask != 0: x1 = x0 + (x1 - x0)*beta. */
int rart_fab_project_linf(const float* points, const float* w, const float* b, float* d_out, float* rowmax_out, int rows,
                          size_t n, rart_stream_t stream);
/* rart_fab_project: the same step for norm 0 = Linf (projection_linf), 1 = L1 (projection_l1, fab_projections.py:120-166),
 * 2 = L2 (projection_l2, :62-117); rownorm_out[r] = ||d[r]||_norm (fab_base.py:194-203).  Sort-free: per row a bisection on the
 * bit pattern of the multiplier (L2) / of the |w| threshold (L1), then the exact solve of the segment.
 * rart_row_norm_diff: out[r] = ||a[r] - b[r]||_norm (fab_base.py:226-236). */
int rart_fab_project(const float* points, const float* w, const float* b, float* d_out, float* rownorm_out, int rows, size_t n,
                     int norm, rart_stream_t stream);
int rart_row_norm_diff(const float* a, const float* b, float* out, int rows, size_t n, int norm, rart_stream_t stream);
int rart_row_dot(const float* a, const float* b, float* out, int rows, size_t n, rart_stream_t stream);
int rart_row_absmax_diff(const float* a, const float* b, float* out, int rows, size_t n, rart_stream_t stream);
int rart_fab_update(float* x1, const float* x0, const float* d1, const float* d2, const float* alpha, int batch,
                    size_t n_per_sample, float eta, rart_stream_t stream);
int rart_fab_backoff(float* x1, const float* x0, const uint8_t* mask, int batch, size_t n_per_sample, float beta,
                     rart_stream_t stream);

/* ---- APGD with the L1 threat model (Attacks/autoattack/autopgd_base.py:19-83, 222-226, 351-364, 431-441) ----
 * rart_l1_project: the reference's L1_projection(x2 = x, y2 = y, eps1 = eps) for `rows` rows of n fp32 elements: delta with
 * ||y + delta||_1 <= eps and 0 <= x + y + delta <= 1.  out = delta (point_out = 0) or the projected point x + y + delta
 * (point_out = 1; clamp01 additionally clamps it to [0,1] as attack_single_run :237 does).  out may alias y.
 * rart_row_kth_abs: thr[r] = k[r]-th smallest (0-based, clamped to [0, n-1]) of |g[r][.]|, exact (radix select).
 * rart_apgd_l1_move: delta_u = x_adv + step_size[r] * sign(g * [|g| >= thr[r]]) / (count_r + 1e-10) - x0 (:355-360).
 * rart_row_count_diff: out[r] = #{i : a[r][i] != b[r][i]} as fp32 (L0_norm, other_utils.py:42-43). */
int rart_l1_project(const float* x, const float* y, float* out, int rows, size_t n, float eps, int point_out, int clamp01,
                    rart_stream_t stream);
int rart_row_kth_abs(const float* g, const int64_t* k, float* thr, int rows, size_t n, rart_stream_t stream);
int rart_apgd_l1_move(const float* x_adv, const float* grad, const float* x0, const float* thr, const float* step_size,
                      float* delta_u, int rows, size_t n, rart_stream_t stream);
int rart_row_count_diff(const float* a, const float* b, float* out, int rows, size_t n, rart_stream_t stream);

/* Per-sample select: dst[i] = src[i] where mask[i] != 0 (rows of n_per_sample floats).
 * The x_best / x_best_adv / grad_best bookkeeping of autopgd_base.py:389-406,426-427. */
int rart_select_rows(float* dst, const float* src, const uint8_t* mask, int batch, size_t n_per_sample,
                     rart_stream_t stream);

/* Row-wise losses on logits [batch][classes] fp32 (autopgd_base.py:198-204,599-604; CE of
 * foolbox / imfgsm_attack.py:83).  kind: 0 = CE, 1 = DLR, 2 = targeted DLR (y_target required),
 * 3 = margin z_y - max_{j != y} z_j (Attacks/autoattack/square.py:68-86),
 * 4 = FAB targeted difference -(z_y - z_t) (Attacks/autoattack/fab_pt.py:102-117).
 * loss_out[batch] (nullable), dlogits_out[batch][classes] = d(sum_i loss_i * scale)/dlogits (nullable),
 * pred_out[batch] int32 argmax (nullable). */
int rart_logit_loss(const float* logits, const int64_t* y, const int64_t* y_target, int batch, int classes,
                    int kind, float scale, float* loss_out, float* dlogits_out, int32_t* pred_out,
                    rart_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Eval-mode model engine (ResNet-50 forward / backward-to-input, the `f_model(x)` + autograd step
 * every attack iteration performs: adv/attack.py:21-22, autopgd_base.py:271-289,367-384,
 * imfgsm_attack.py:82-84; model definition: RobustART/model/__init__.py:1 -> absent submodule,
 * public ResNet-50 v1.5).  BatchNorm is folded into the conv weights/bias on the host (eval mode),
 * activations are bf16 NHWC, accumulation fp32.
 *
 * rart_conv_igemm_bf16: C[m][n] = sum_k A[m][k] * W[n][k] with A gathered from `src`:
 *   row m      <-> (image, oy, ox) over batch x grid_h x grid_w
 *   k          =  tap * k_per_tap + c,  tap t reads source pixel (oy*sy + tap_dy[t], ox*sx + tap_dx[t])
 *                 of the (src_h x src_w) image at element offset tap_src_off[t] + pixel*src_pix_stride + c
 *                 (zeros outside the image)
 *   W          :  bf16 [round_up(n_cols, 128 if n_cols > 64 else 64)][n_taps * k_per_tap], K contiguous
 *   destination:  element ((image*dst_h + oy*dst_sy + dst_oy)*dst_w + ox*dst_sx + dst_ox)*dst_pix_stride + n
 *   epilogue   :  v = acc + bias[n]; v += res[dst index]; if (mask) v = mask[dst index] > 0 ? v : 0;
 *                 if (flags & 1) v = max(v, 0); if (flags & 4) v = gelu(v); store bf16 (or fp32 if flags & 2).
 *                 flags & 8: `mask` holds a GELU pre-activation u instead and v = v * gelu'(u) (no sign masking).
 *                 `res` may alias `dst`.
 * Covers forward convs, backward-to-input of stride-1 convs, each input-parity class of a stride-2
 * conv's backward, the 7x7 stem on the padded 4-channel hi/lo image, and fully connected layers.
 * ------------------------------------------------------------------------------------- */
typedef struct rart_conv_desc {
  const void* src;
  const void* wgt;
  const float* bias;      /* fp32 [n_cols] or NULL */
  const void* res;        /* bf16, indexed like dst, or NULL */
  const void* mask;       /* bf16, indexed like dst, or NULL */
  void* dst;
  int32_t batch, grid_h, grid_w;
  int32_t src_h, src_w, src_pix_stride;
  int32_t k_per_tap, n_taps;
  int32_t sy, sx;
  int32_t tap_dy[32], tap_dx[32];
  int64_t tap_src_off[32];
  int32_t n_cols;
  int32_t dst_h, dst_w, dst_sy, dst_sx, dst_oy, dst_ox, dst_pix_stride;
  int32_t flags;          /* 1 = ReLU, 2 = fp32 output, 4 = exact GELU, 8 = GELU' of `mask`, 16 = `mask` is a 1-bit tensor,
                           * 32 = split-bf16 tensors (dst_pair_off / res_pair_off below), 64 = GELU with the pre-activation kept,
                           * 128 = the 1-bit `mask` (with 16) applies to the RESIDUAL instead of the result: dst = acc + res . [mask]
                           * (train engine: the skip gradient d_out . [out > 0] without materialising it) */
  /* batched problems (attention: one GEMM per (image, head)): problem z in [0, n_batched) splits into
   * zo = z / z_inner, zi = z % z_inner; zo*_z_outer + zi*_z_inner elements are added to src / wgt / dst (res and
   * mask follow dst).  n_batched <= 1 = a single problem.  wgt_row_stride: elements between consecutive weight
   * rows (0 = n_taps * k_per_tap), so K / V can be read in place from the fused qkv activation. */
  int32_t n_batched, z_inner, wgt_row_stride, reserved_;
  int64_t src_z_outer, src_z_inner, wgt_z_outer, wgt_z_inner, dst_z_outer, dst_z_inner;
  /* 1-bit-per-element sign tensors, indexed like dst (byte = element / 8, bit = element % 8): the eval engine's backward
   * needs activations only for their ReLU sign, so the forward GEMM writes (output > 0) here (nullable) and the backward
   * GEMM reads it through `mask` with flag 16 -- 1/16 of the bytes of a bf16 mask.  bf16 output, unbatched problems. */
  void* sign_out;
  /* flag 32, the reference-precision ("bf16x3") mode: every activation is a PAIR of bf16 planes, value = hi + lo with
   * hi = bf16(v), lo = bf16(v - hi) (16 significand bits), every weight likewise; the caller lists the three products
   * x_hi.w_hi + x_hi.w_lo + x_lo.w_hi as 3x the taps (tap_src_off selects the operand plane, the weight row is the
   * concatenation [w_hi | w_lo | w_hi]) so they accumulate in the same fp32 registers.  dst / res then name the hi plane and
   * the lo plane sits dst_pair_off / res_pair_off ELEMENTS behind it (multiples of 8); `mask` must be a 1-bit tensor;
   * with flag 2 the output is plain fp32 and no residual / mask is applied.  Reference arithmetic: fp32 everywhere
   * (adv/attack.py:20-23, autopgd_base.py:271-289); this mode reproduces it to ~1e-5 of the logit scale. */
  int64_t dst_pair_off, res_pair_off;
  /* train-mode forward (nullable): per 128-row tile of the output, the column sums and sums of squares of the bf16 values as stored,
   * fp32 [ceil(rows / 128)][2][n_cols] -- the batch statistics of the BatchNorm that follows (rart_bn_train_forward_bf16's
   * stats_partial), from the accumulators instead of a pass over the tensor.  Plain bf16 convolution only: no bias / residual / mask /
   * flags, unbatched. */
  float* bn_stats_out;
} rart_conv_desc;

int rart_conv_igemm_bf16(const rart_conv_desc* desc_host, rart_stream_t stream);
/* Tuning knob: problems with n_taps*k_per_tap >= k (and k_per_tap % 64 == 0) use the 128x{128,64}x64 pipeline
 * (default 1024); others the x32 pipeline with loads two K steps ahead. */
int rart_igemm_set_bk64_min_k(long long k);
/* Tuning knob: plain row-major products (one tap, unit strides, no batching, flags within GELU / GELU') with at least 512 tiles of
 * 256 x 256 run on the 8-wave 256 x 256 x 64 kernel with direct-to-LDS tiles (the transformer layers): 2 (default) = on the ping-pong
 * schedule of round 6 (k_gemm256_pp: the two halves of the workgroup alternate memory and matrix phases, counted vmcnt, vector-free
 * buffer_load ... lds issue), 1 = on round 2's two-stage loop (bit-identical outputs), 0 disables the kernel. */
/* Small-M product C[m][n] = A[m][k] . W[n][k]^T (+ bias[n]) for the classifier head and its backward (m = the batch): one workgroup per
 * 32 x 32 output tile, four waves split k (k % 64 == 0), fragments straight from global memory.  out: fp32 (out_is_f32) or bf16, leading
 * dimension ldo.  Replaces the 16-workgroup implicit-GEMM launch of the fc layer (RobustART/model -> public ResNet-50 `fc`). */
int rart_gemm_small_m_bf16(const void* a, int lda, const void* w, int ldw, const float* bias, void* out, int ldo, int out_is_f32, int m,
                           int n, int k, rart_stream_t stream);
int rart_igemm_set_gemm256(int enable);
/* 1 when a plain product of `rows` x k (leading dimension src_ld) by n_cols columns takes the 256 x 256 GEMM kernel -- the only kernel that
 * serves flag 64 (GELU with the pre-activation kept: dst = gelu(u), `mask` RECEIVES the bf16 pre-activation u; ViT fc1 in keep mode). */
int rart_gemm256_supported(long long rows, int k, int n_cols, int src_ld, int dst_ld);

/* Split-bf16 ("fp32x" / "bf16x3") GEMM -- the dense contraction of the REFERENCE-PRECISION engines (csrc/gemm_pair.hip).  The
 * reference computes fp32 (adv/attack.py:20-23, autopgd_base.py:271-289; exprs/exp/imagenet_c_loop_mini/config_vit_base.yaml:1-9 has
 * no precision key).  Every operand is a PAIR of bf16 planes, value = hi + lo (16 significand bits):
 *     C[m][n] = sum_k A[m][k] * W[n][k] ~= A_lo.W_hi + A_hi.W_lo + A_hi.W_hi        (fp32 accumulation, + bias[n])
 * 256 x 256 tiles, the four operand planes of a 32-deep K step staged once in LDS (direct-to-LDS loads), three MFMAs per fragment pair.
 * a_* : [rows][lda] bf16, K contiguous; w_* : [w_rows][ldw] bf16 (rows = output columns; rows >= w_rows read as zeros, so N may exceed
 *       the rows that exist: attention products whose "weights" are activations); K % 32 == 0, N % 8 == 0, ld* % 8 == 0.
 * dst : pair planes [rows][ldc] (hi = bf16(v), lo = bf16(v - hi)), or fp32 [rows][ldc] in dst_hi with flag 2 (dst_lo unused).
 * res : optional residual PAIR indexed like dst, added in fp32.
 * 256- or 128-row tiles x 64 / 128 / 256-column tiles (chosen from the shape; tile_m / tile_n override).
 * flags: 1 ReLU; 2 fp32 output; 4 exact GELU (of the value the pair of the pre-activation represents, as flag 64 computes it); 64 GELU with the pre-activation kept (aux RECEIVES the pair u, dst = gelu(u_hi + u_lo));
 *        8 GELU' (v *= gelu'(aux_hi + aux_lo), aux indexed like dst);
 *        16 (round 5) the weight table is INTERLEAVED: a row holds, per 32-deep K step, the 32 hi elements followed by the 32 lo elements
 *        (ldw >= 2 K, w_lo = w_hi + 32 elements): one 128-byte line per row and step instead of two half lines K apart.
 * Row re-basing (0 = none): output row m of image m / rows_per_image reads source row img * src_rows_per_image + m % rows_per_image +
 * src_row_off and writes destination row img * dst_rows_per_image + m % rows_per_image + dst_row_off (ViT's class-token slot).
 * Batched (n_batched > 1, blockIdx.y = z -> zo = z / z_inner, zi = z % z_inner): element offsets zo * *_z_outer + zi * *_z_inner are
 * added to a_* / w_* / (dst, res, aux). */
typedef struct rart_gemm_pair_desc {
  const void *a_hi, *a_lo, *w_hi, *w_lo;
  const float* bias;
  const void *res_hi, *res_lo;
  void *dst_hi, *dst_lo;
  void *aux_hi, *aux_lo;
  int M, N, K, lda, ldw, ldc, w_rows;
  int rows_per_image, src_rows_per_image, src_row_off, dst_rows_per_image, dst_row_off;
  int flags, n_batched, z_inner;
  int64_t a_z_outer, a_z_inner, w_z_outer, w_z_inner, c_z_outer, c_z_inner;
  /* conv = 1: implicit-GEMM convolution, the row / tap / destination conventions of rart_conv_desc: row m = (image, oy, ox) of a
   * batch x grid_h x grid_w grid, k = tap * k_per_tap + c (k_per_tap a power of two >= 32, <= 16 taps), source pixel
   * (oy sy + tap_dy, ox sx + tap_dx) of a src_h x src_w image with lda elements per pixel (zeros outside), destination pixel
   * (oy dst_sy + dst_oy, ox dst_sx + dst_ox) of a dst_h x dst_w image with ldc elements per pixel; M and K are derived.
   * mask_bits: 1-bit tensor indexed like dst (byte (off + col) / 8, bit col % 8), the value is zeroed where the bit is clear (the
   * ReLU mask of a backward-to-input); sign_out receives (output hi plane > 0) in the same indexing.  Unbatched, no row re-basing. */
  int conv, batch, grid_h, grid_w, src_h, src_w, sy, sx, k_per_tap, n_taps;
  int tap_dy[16], tap_dx[16];
  int dst_h, dst_w, dst_sy, dst_sx, dst_oy, dst_ox;
  const void* mask_bits;
  void* sign_out;
  int tile_n;               /* 0 = automatic (64 / 128 / 256 by N); tests / sweeps force a column tile */
  int tile_m;               /* 0 = automatic (256; 128 for short-K or small convolutions; 224 where it saves a pass); 128 / 224 / 256 force it */
} rart_gemm_pair_desc;
int rart_gemm_pair_bf16(const rart_gemm_pair_desc* desc_host, rart_stream_t stream);
/* Schedule of the 256-row tiles with 128 / 256 columns (round 6, csrc/gemm_pair_pp.hip): 1 (default; RART_PAIR_SCHEDULE in the environment
 * sets the initial value) = ping-pong -- the two halves of the workgroup alternate memory and matrix phases, counted vmcnt, no drain in the
 * steady state; 2 (opt-in: measured slower than 1, see the kernel's header) = ping-pong, and one-tap problems (1x1 convolutions, plain
 * products without row re-basing / batching / GELU) with at least two 256 x 128 tiles per CU on the PERSISTENT kernel, which writes a tile
 * out under the next tile's K loop; 3 (opt-in: measured equal to 1) = ping-pong, and launches of more than one tile per CU WALKED by one
 * workgroup per CU (the next tile's first stage requested before the epilogue); 0 = the two-stage loop of round 4 everywhere.  Under 1 - 3 a
 * plain product whose 256 x 256 tiles fill every XCD's CUs a whole number of times and then less than half of them once more is issued as
 * two launches, the remaining rows on 256 x 128 tiles (RART_PAIR_SPLIT=0 in the environment disables it), and a launch whose row tiles
 * fill an XCD's 32 CUs badly runs tiles that step 224 rows (RART_PAIR_ROWS224=0 disables it).  Outputs are bit-identical under
 * all of these (same products, same order, same point-wise code). */
int rart_gemm_pair_set_schedule(int mode);
int rart_gemm_pair_get_schedule(void);

/* The second half of a ResNet Bottleneck of the reference-precision engine as ONE launch (csrc/conv_tail_pair.hip): 3x3 stride-1 pad-1
 * convolution (c_mid -> c_mid channels, c_mid = 64 or 128) + point-wise step + the 1x1 expansion (c_mid -> 4 c_mid) + skip pair +
 * point-wise step, on hi + lo planes of bf16 with three MFMA products per contraction; the c_mid-channel intermediate stays in registers.
 *   forward  (Bottleneck.forward, conv2-bn2-relu-conv3-bn3-add-relu):  relu_mid = relu_out = 1, biases set, sign_* receive the 1-bit
 *            (value > 0) tensors the backward masks with;
 *   backward (autograd of attack.py:21-22 / autopgd_base.py:271-289 through conv1^T after conv2^T): tables transposed / taps flipped,
 *            mask_mid / mask_out = those 1-bit tensors, res = the gradient arriving over the identity skip.
 * w_*: the 3x3's table, row = output channel, k = tap * c_mid + c, row stride ldw elements.  t_*: the 1x1's [4 c_mid][c_mid] matrix
 * in MFMA fragment order: element ((blk * (c_mid / 16) + s) * 64 + h * 32 + r) * 8 + e = T[blk * 32 + r][16 s + 8 h + e].
 * 1-bit tensors: byte (position * channels + channel) / 8, bit channel % 8. */
typedef struct rart_conv_tail_desc {
  const void* a_hi;  const void* a_lo;       /* input pair [batch][h][w][c_mid] */
  const void* w_hi;  const void* w_lo;
  const void* t_hi;  const void* t_lo;
  const float* bias_mid;                     /* [c_mid] or null */
  const float* bias_out;                     /* [4 c_mid] or null */
  const void* mask_mid;  const void* mask_out;
  void* sign_mid;  void* sign_out;
  const void* res_hi;  const void* res_lo;   /* skip pair [batch][h][w][4 c_mid] or null */
  void* dst_hi;  void* dst_lo;               /* [batch][h][w][4 c_mid] */
  int batch, h, w, c_mid, ldw, relu_mid, relu_out;
  int tap_dy[9], tap_dx[9];                  /* source pixel of tap i = (y + tap_dy[i], x + tap_dx[i]); zero outside the image */
  /* optional (n_hi != null): the NEIGHBOURING block's 1x1 reduction (4 c_mid -> c_mid) of the tile just produced -- forward: the next
   * block's conv1 + bias + ReLU (+ sign tensor); backward: the previous block's conv3^T + mask.  n_*: its [c_mid][4 c_mid] matrix in
   * fragment order by 64-wide K chunk: element (((chunk * (c_mid / 32) + blk) * 4 + s) * 64 + h * 32 + r) * 8 + e =
   * N[blk * 32 + r][64 chunk + 16 s + 8 h + e].  dstn_*: [batch][h][w][c_mid]. */
  const void* n_hi;  const void* n_lo;
  const float* bias_next;
  const void* mask_next;
  void* sign_next;
  void* dstn_hi;  void* dstn_lo;
  int relu_next;
} rart_conv_tail_desc;
int rart_conv3x3_tail_pair_supported(int c_mid);
int rart_conv3x3_tail_pair(const rart_conv_tail_desc* desc_host, rart_stream_t stream);

/* Row / elementwise kernels of the reference-precision ViT-B/16 engine (csrc/vit_pair.hip): every tensor a pair of bf16 planes, the
 * arithmetic in fp32 as the fp32 module (timm ViT-B/16, RobustART/model/__init__.py:1 -> absent submodule) does it.
 * add_pos_cls: x[b][0][:] = cls_pos0, x[b][t][:] += pos[t].  layernorm: eps inside the root, biased variance; rows of dim <= 1024.
 * layernorm_bwd: dx = rstd (g - mean(g) - xhat mean(g xhat)) [+ res pair], g = dy gamma.  softmax_rows: P = softmax(scale S) of fp32
 * scores (the pair GEMM's fp32 output), zero beyond n_valid up to ld_out (<= 256 columns).  softmax_bwd_rows: dS = scale P (dP - <P, dP>),
 * dP fp32.  unpatchify_from_f32: fp32 patch gradients -> fp32 NCHW image gradient / std. */
int rart_vit_add_pos_cls_pair(void* x_hi, void* x_lo, const float* cls_pos0, const float* pos, int n, int tokens, int dim,
                              rart_stream_t stream);
int rart_layernorm_pair(const void* x_hi, const void* x_lo, const float* gamma, const float* beta, void* out_hi, void* out_lo, int rows,
                        int dim, int64_t in_row_stride, int64_t out_row_stride, float eps, rart_stream_t stream);
int rart_layernorm_bwd_pair(const void* dy_hi, const void* dy_lo, const void* x_hi, const void* x_lo, const float* gamma,
                            const void* res_hi, const void* res_lo, void* dx_hi, void* dx_lo, int rows, int dim, int64_t dy_row_stride,
                            int64_t x_row_stride, int64_t res_row_stride, int64_t dx_row_stride, float eps, rart_stream_t stream);
int rart_softmax_rows_pair(const float* scores, void* probs_hi, void* probs_lo, int64_t rows, int n_valid, int ld_in, int ld_out,
                           float scale, rart_stream_t stream);
int rart_softmax_bwd_rows_pair(const void* probs_hi, const void* probs_lo, const float* dprobs, void* ds_hi, void* ds_lo, int64_t rows,
                               int n_valid, int ld_p, int ld_dp, int ld_out, float scale, rart_stream_t stream);
int rart_vit_unpatchify_from_f32(const float* dpatches, float* grad, int n, int h, int w, int patch, int64_t ld, const float* std_host,
                                 rart_stream_t stream);
/* Fused multi-head self-attention on pairs (head_dim 64, <= 224 tokens), one workgroup per (image, head): qkv pair [n*tokens][3*D]
 * (q | k | v column blocks, heads of 64 inside each) -> out pair [n*tokens][D] = softmax(q k^T / 8) v, every contraction three MFMA
 * products, the soft-max in fp32 registers (timm Attention.forward; the reference runs it in fp32). */
int rart_vit_attention_pair(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, int n, int tokens, int heads, int head_dim,
                            rart_stream_t stream);
/* Its backward (two launches, csrc/vit_pair.hip): dqkv pair [n*tokens][3*D] (dq | dk | dv) from the qkv pair, the forward's output pair, and
 * the gradient pair of that output.  stats: scratch of n * heads * 32 * ceil(tokens / 32) * 4 floats (per query: max, 1 / sum, delta). */
int rart_vit_attention_bwd_pair(const void* qkv_hi, const void* qkv_lo, const void* out_hi, const void* out_lo, const void* dout_hi,
                                const void* dout_lo, void* dqkv_hi, void* dqkv_lo, float* stats, int n, int tokens, int heads, int head_dim,
                                rart_stream_t stream);

/* 3x3 stride-1 "same" convolution, channels in = channels out = 64, 128 or 256 (256: images of at most 224 positions, one
 * image per workgroup), bf16 NHWC, with the input halo tile resident in LDS (csrc/conv3x3_halo.hip): ResNet-50's layer1 /
 * layer2 / layer3 conv2, forward (taps (r-1, s-1), bias + ReLU, sign_out =
 * 1-bit mask of the output) and backward-to-input (taps (1-r, 1-s), weights [cin][tap*cout+co], mask_bits = 1-bit ReLU
 * mask of the destination).  wgt: the bf16 table [channels][9*channels] (k = tap*channels + c) RE-ORDERED by
 * rart_conv3x3_pack_frag_bf16 into MFMA-fragment order (fragment (K step st of 64, column tile wn, ks) = 64 lanes x 8
 * elements, lane l = w[wn*32 + (l & 31)][st*64 + ks*16 + (l >> 5)*8 ..]: a fragment load reads 1 KiB contiguous).  Any pointer
 * of bias / mask_bits / sign_out may be NULL.  rart_conv3x3_halo_supported: 1 when the geometry fits the LDS tile
 * (otherwise use rart_conv_igemm_bf16, which computes the same function). */
int rart_conv3x3_halo_supported(int channels, int h, int w);
int rart_conv3x3_pack_frag_bf16(const void* w_rows, void* w_frag, int channels, rart_stream_t stream);
/* The same re-ordering for any bf16 [rows][k] row-major matrix (rows % 32 == 0, k % 64 == 0): fragment (st, wn, ks) as above. */
int rart_pack_frag_bf16(const void* w_rows, void* w_frag, int rows, int k, rart_stream_t stream);
int rart_conv3x3_halo_bf16(const void* src, const void* wgt, const float* bias, const void* mask_bits, void* sign_out,
                           void* dst, int n, int h, int w, int channels, const int* tap_dy, const int* tap_dx, int relu,
                           rart_stream_t stream);

/* One identity Bottleneck (1x1 c_io -> c_mid, 3x3 c_mid -> c_mid, 1x1 c_mid -> c_io, + the block input) as ONE kernel with
 * both intermediates kept on chip (csrc/bottleneck_fused.hip); supported geometry: c_io 256, c_mid 64, 56 x 56 (ResNet-50
 * layer1 blocks 1-2).  Replaces three rart_conv_igemm_bf16 / rart_conv3x3_halo_bf16 launches and computes the same function.
 * x, out: bf16 NHWC [n][h][w][c_io], out != x.  w1: bf16 [c_mid][c_io]; w2: the 3x3 table [c_mid][9*c_mid] (k = tap*c_mid + c)
 * re-ordered by rart_conv3x3_pack_frag_bf16 into MFMA-fragment order, with its taps; w3: bf16 [c_io][c_mid]; b1..b3 fp32 or NULL.  m1 / m2 / m3: 1 bit per element of the stage-1 / stage-2 / final
 * result (byte (pos*C + ch) >> 3, bit ch & 7).
 *   backward = 0: out = relu(w3.relu(w2*relu(w1.x + b1) + b2) + b3 + x); m1..m3 are OUTPUTS (value > 0), each may be NULL.
 *   backward = 1: x is the gradient at the block output (already masked by that output's ReLU), w1 / w2 / w3 the transposed
 *                 tables of conv3 / conv2 (flipped taps) / conv1, no biases; out = m3 . (w3.(m2.(w2*(m1.(w1.x)))) + x) with
 *                 m1 = sign of the forward's conv2 output, m2 = of its conv1 output, m3 = of the block input (or NULL). */
int rart_bottleneck_fused_supported(int c_io, int c_mid, int h, int w);
int rart_bottleneck_fused_bf16(const void* x, const void* w1, const void* w2, const void* w3, const float* b1, const float* b2,
                               const float* b3, void* m1, void* m2, void* m3, void* out, int n, int h, int w, int c_io,
                               int c_mid, const int* tap_dy, const int* tap_dx, int backward, rart_stream_t stream);

/* The FIRST Bottleneck of the layer (projection shortcut, stride 1) as one kernel; supported geometry: c_in 64, c_mid 64,
 * c_out 256, 56 x 56 (ResNet-50 layer1 block 0).  Same kernel family and conventions as rart_bottleneck_fused_bf16.
 *   backward = 0: x bf16 [n][h][w][c_in]; w1 [c_mid][c_in]; w2 fragment-ordered 3x3 table; w3 [c_out][c_mid]; w4 = the
 *                 shortcut's [c_out][c_in] table in fragment order (rart_pack_frag_bf16(rows c_out, k c_in)); b3 = conv3 bias +
 *                 shortcut bias; out [n][h][w][c_out] = relu(w3.a2 + w4.x + b3); m1..m3 sign OUTPUTS or NULL.
 *   backward = 1: x = gradient at the block output [n][h][w][c_out]; w1 = conv3's transposed table [c_mid][c_out]; w2 = conv2's
 *                 backward table (fragment order, flipped taps); w3 = the shortcut's transposed table [c_in][c_out]; w4 = conv1's
 *                 transposed table [c_in][c_mid] (row-major); out [n][h][w][c_in] = m3 . (w4.(m2.(w2*(m1.(w1.x)))) + w3.x). */
int rart_bottleneck_first_supported(int c_in, int c_mid, int c_out, int h, int w);
int rart_bottleneck_first_bf16(const void* x, const void* w1, const void* w2, const void* w3, const void* w4, const float* b1,
                               const float* b2, const float* b3, void* m1, void* m2, void* m3, void* out, int n, int h, int w,
                               int c_in, int c_mid, int c_out, const int* tap_dy, const int* tap_dx, int backward,
                               rart_stream_t stream);

/* The FIRST Bottleneck of layer2 / layer3 -- stride 2 on the 3x3, projection shortcut -- FORWARD as one kernel
 * (csrc/bottleneck_s2_fused.hip): out = relu(w3 . relu(w2 *s2 relu(w1 . x + b1) + b2) + wd . x[::2, ::2] + b3), b3 = conv3 bias +
 * shortcut bias.  Supported: c_in 256 -> c_mid 128 -> c_out 512 at 56 x 56 and 512 -> 256 -> 1024 at 28 x 28 (x is h x w, out
 * h/2 x w/2).  All four tables in fragment order: w1 = rart_pack_frag_bf16(rows c_mid, k c_in) of [c_mid][c_in]; w2 = of
 * [c_mid][9*c_mid] with k = (r*3+s)*c_mid + c and taps (r-1, s-1); w3 = of [c_out][c_mid]; wd = of [c_out][c_in].
 * m1 / m2 / m3 (nullable): 1-bit sign OUTPUTS of a1 [n][h][w][c_mid/8], a2 [n][h/2][w/2][c_mid/8], out [n][h/2][w/2][c_out/8].
 * Replaces four rart_conv_igemm_bf16 launches and computes the same function (the shortcut is not rounded to bf16 on its own).
 * Reference: Bottleneck.forward with a downsample branch inside `model(x)` of adv/attack.py:21-22, autopgd_base.py:271-289. */
int rart_bottleneck_s2_fwd_supported(int c_in, int c_mid, int c_out, int h, int w);
int rart_bottleneck_s2_fwd_bf16(const void* x, const void* w1, const void* w2, const void* w3, const void* wd, const float* b1,
                                const float* b2, const float* b3, void* m1, void* m2, void* m3, void* out, int n, int h, int w,
                                int c_in, int c_mid, int c_out, rart_stream_t stream);

/* The backward-to-input of the same stride-2 blocks as one kernel:
 *   dx = m0 . ( w1t . ( m1 . ( w2t *s2^T ( m2 . ( w3t . g ) ) ) ) + wdt . g [at even / even input positions] ),
 * g = bf16 gradient at the block output [n][h/2][w/2][c_out] (already masked by that output's ReLU), dx [n][h][w][c_in].
 * Tables in fragment order (rart_pack_frag_bf16): w3t = conv3's transposed table [c_mid][c_out]; w1t = conv1's [c_in][c_mid];
 * wdt = the shortcut's [c_in][c_out]; w2t = the four input-parity-class tables of the 3x3 / 2's transpose, class order (0,0) (0,1)
 * (1,0) (1,1), class (ph, pw) = [c_mid][ntaps * c_mid] with k = tap * c_mid + c and its taps = the filter taps (r, s) with
 * (ph + 1 - r), (pw + 1 - s) even, r then s ascending (1 / 2 / 2 / 4 taps), each packed on its own and concatenated (element offsets
 * 0, 1, 3, 5 times c_mid^2).  m2 / m1 / m0: the 1-bit sign tensors of a2 / a1 / the block input written by the forward (m0 nullable).
 * Replaces seven rart_conv_igemm_bf16 launches.  Reference: autograd of the same Bottleneck (adv/attack.py:21-22). */
int rart_bottleneck_s2_bwd_bf16(const void* g, const void* w3t, const void* w2t, const void* w1t, const void* wdt, const void* m2,
                                const void* m1, const void* m0, void* dx, int n, int h, int w, int c_in, int c_mid, int c_out,
                                rart_stream_t stream);

/* One identity Bottleneck at 14 x 14 (ResNet-50 layer3 blocks 1-5: c_io 1024, c_mid 256) as one kernel, ONE IMAGE PER WORKGROUP
 * (csrc/bottleneck14_fused.hip).  Same conventions as rart_bottleneck_fused_bf16 (forward / backward, biases, the three 1-bit
 * mask tensors m1 [P][c_mid/8], m2 [P][c_mid/8], m3 [P][c_io/8]), except that ALL THREE weight tables are passed in fragment order:
 * w1 = rart_pack_frag_bf16(rows c_mid, k c_io) of [c_mid][c_io]; w2 = rart_conv3x3_pack_frag_bf16 of [c_mid][9*c_mid];
 * w3 = rart_pack_frag_bf16(rows c_io, k c_mid) of [c_io][c_mid]. */
int rart_bottleneck14_fused_supported(int c_io, int c_mid, int h, int w);
int rart_bottleneck14_fused_bf16(const void* x, const void* w1, const void* w2, const void* w3, const float* b1, const float* b2,
                                 const float* b3, void* m1, void* m2, void* m3, void* out, int n, int h, int w, int c_io,
                                 int c_mid, const int* tap_dy, const int* tap_dx, int backward, rart_stream_t stream);

/* One identity Bottleneck at 28 x 28 (ResNet-50 layer2 blocks 1-3: c_io 512, c_mid 128) as one kernel, a 14 x 14 quarter of an
 * image per workgroup, two workgroups per CU (csrc/bottleneck28_fused.hip).  Conventions and table formats as for
 * rart_bottleneck14_fused_bf16: w1 = rart_pack_frag_bf16(rows c_mid, k c_io), w2 = rart_conv3x3_pack_frag_bf16,
 * w3 = rart_pack_frag_bf16(rows c_io, k c_mid); masks m1 / m2 [P][c_mid/8], m3 [P][c_io/8]. */
int rart_bottleneck28_fused_supported(int c_io, int c_mid, int h, int w);
int rart_bottleneck28_fused_bf16(const void* x, const void* w1, const void* w2, const void* w3, const float* b1, const float* b2,
                                 const float* b3, void* m1, void* m2, void* m3, void* out, int n, int h, int w, int c_io,
                                 int c_mid, const int* tap_dy, const int* tap_dx, int backward, rart_stream_t stream);

/* One identity Bottleneck at 7 x 7 (ResNet-50 layer4 blocks 1-2: c_io 2048, c_mid 512) as one kernel, one image per workgroup
 * (csrc/bottleneck7_fused.hip).  Conventions as for rart_bottleneck14_fused_bf16; all three tables in fragment order:
 * w1 = rart_pack_frag_bf16(rows c_mid, k c_io), w2 = rart_pack_frag_bf16(rows c_mid, k 9*c_mid), w3 = rart_pack_frag_bf16(rows c_io,
 * k c_mid); masks m1 / m2 [P][c_mid/8], m3 [P][c_io/8]. */
int rart_bottleneck7_fused_supported(int c_io, int c_mid, int h, int w);
int rart_bottleneck7_fused_bf16(const void* x, const void* w1, const void* w2, const void* w3, const float* b1, const float* b2,
                                const float* b3, void* m1, void* m2, void* m3, void* out, int n, int h, int w, int c_io,
                                int c_mid, const int* tap_dy, const int* tap_dx, int backward, rart_stream_t stream);

/* src: fp32 NCHW in [0,1] (src_is_u8 = 0) or uint8 NHWC (src_is_u8 = 1) -> (x - mean)/std as two bf16
 * planes hi, lo (hi + lo ~ fp32 value), each [n][h+8][w+8][4] with the image at (3,3) and zeros around:
 * the stem convolution's operand (normalisation of imfgsm_attack.py:14-23 / autoattack.py:17-20 fused). */
int rart_engine_prep_input(const void* src, int src_is_u8, void* hi, void* lo, int n, int h, int w,
                           const float* mean_host, const float* std_host, rart_stream_t stream);
/* 3x3 stride-2 pad-1 max pool on bf16 NHWC; argmax_out (nullable): uint8 [n][h/2][w/2][c], the window
 * position ky*3+kx of the first maximum (PyTorch's rule).  Backward, fused with the ReLU mask of the pool's
 * input y: dz = (y > 0) * sum of dpool over the windows whose argmax is this pixel. */
int rart_engine_maxpool(const void* in, void* out, void* argmax_out, int n, int h, int w, int c,
                        rart_stream_t stream);
/* The same pool, also writing sign_out (nullable): uint8 [n][h/2][w/2][c/8], bit j of byte k = (pooled channel 8k+j > 0)
 * -- the 1-bit ReLU mask the backward GEMMs read with rart_conv_desc flag 16.  argmax code 15 = window maximum <= 0. */
int rart_engine_maxpool_keep(const void* in, void* out, void* argmax_out, void* sign_out, int n, int h, int w, int c,
                             rart_stream_t stream);
int rart_engine_maxpool_bwd(const void* y, const void* argmax, const void* dpool, void* dz, int n, int h, int w,
                            int c, rart_stream_t stream);
/* global average pool [n][hw][c] -> [n][c], and dz = (y > 0) ? dpool / hw : 0. */
int rart_engine_avgpool(const void* in, void* out, int n, int hw, int c, rart_stream_t stream);
int rart_engine_avgpool_bwd(const void* y, const void* dpool, void* dz, int n, int hw, int c, rart_stream_t stream);
/* stem backward: bf16 patches [n][h/2][w/2][patch_cols] (column (r*7+s)*3+c) -> fp32 NCHW gradient w.r.t.
 * the [0,1] image (the 1/std of the normalisation applied). */
int rart_engine_stem_col2im(const void* patches, float* grad, int n, int h, int w, int patch_cols,
                            const float* std_host, rart_stream_t stream);
/* ---- reference-precision ("bf16x3") forms of the pools / converters (csrc/engine_aux_pair.hip) -----------------------------
 * Every tensor is a PAIR of bf16 planes, value = hi + lo (see rart_conv_desc flag 32): `*_hi` names the hi plane and the lo
 * plane sits `*_lo_off` ELEMENTS behind it (a positive multiple of 8).  Arithmetic: unpack to fp32 (hi + lo is exact),
 * compute as the bf16 kernel of the same name, split again.  Reference: the fp32 max_pool2d / adaptive_avg_pool2d and their
 * autograd inside `model(x)` of adv/attack.py:20-23, autopgd_base.py:271-289. */
int rart_engine_maxpool_pair(const void* in_hi, long long in_lo_off, void* out_hi, long long out_lo_off, void* argmax_out,
                             void* sign_out, int n, int h, int w, int c, rart_stream_t stream);
/* dz (at the pool input, masked by the ReLU in front of the pool) from the argmax codes alone: code 15 marks windows whose
 * maximum is <= 0, any other code names a position whose value is that positive maximum. */
int rart_engine_maxpool_bwd_pair(const void* argmax, const void* dpool_hi, long long dpool_lo_off, void* dz_hi,
                                 long long dz_lo_off, int n, int h, int w, int c, rart_stream_t stream);
int rart_engine_avgpool_pair(const void* in_hi, long long in_lo_off, void* out_hi, long long out_lo_off, int n, int hw, int c,
                             rart_stream_t stream);
/* y_sign_bits: uint8 [n][hw][c/8], the 1-bit (y > 0) tensor the last block's forward GEMM wrote through sign_out. */
int rart_engine_avgpool_bwd_pair(const void* y_sign_bits, const void* dpool_hi, long long dpool_lo_off, void* dz_hi,
                                 long long dz_lo_off, int n, int hw, int c, rart_stream_t stream);
/* fp32 [rows][cols] -> pair [rows][dst_cols] (zero padded): the loss gradient entering the fc backward. */
int rart_f32_to_pair_rows(const float* src, void* dst_hi, long long dst_lo_off, int rows, int cols, int dst_cols,
                          rart_stream_t stream);
/* rart_engine_stem_col2im for fp32 patches (the stem's backward GEMM run with flags 32 | 2); h % 16 == 0, w % 32 == 0. */
int rart_engine_stem_col2im_f32(const float* patches, float* grad, int n, int h, int w, int patch_cols, const float* std_host,
                                rart_stream_t stream);
/* The same stem backward as ONE kernel: pooled gradient dpool [n][h/4][w/4][64] bf16 + the max pool's argmax codes
 * (rart_engine_maxpool; code 15 = window maximum <= 0) -> fp32 NCHW gradient w.r.t. the [0,1] image.  Max-pool backward,
 * the ReLU mask and the transposed 7x7/2 convolution are fused: an implicit GEMM over the 4x4 neighbourhood of
 * stem-output positions (K = 16 taps x 64 channels, N = 4 input parities x 3 channels) on MFMA with the gathered
 * operand built in LDS.  wtab: bf16 [16][1024], row (py*2+px)*3+c, column ((dp+1)*4+(dq+1))*64+k = W[k][c][py+3-2dp][px+3-2dq]
 * (0 where the tap index leaves 0..6; rows 12..15 zero).  h, w multiples of 4.
 * (autograd of attack.py:21-22 / autopgd_base.py:271-289 through conv1-bn1-relu-maxpool) */
int rart_engine_stem_bwd_fused(const void* dpool, const void* argmax, const void* wtab, float* grad, int n, int h, int w,
                               const float* std_host, rart_stream_t stream);
/* The reference-precision stem FORWARD as one persistent kernel (csrc/stem_pair.hip): normalisation + hi / lo split + 7x7/2 convolution
 * (three MFMA products per K step) + bias + ReLU + hi / lo split + 3x3/2 max pool of the pair values; bit-identical to
 * rart_engine_prep_input -> rart_gemm_pair_bf16 (7 row taps) -> rart_engine_maxpool_pair.  wgt_*: [64][224] bf16, column r * 32 + px * 4 + c
 * (px < 7, c < 3; zero elsewhere).  p1_*: [n][h/4][w/4][64]; argmax_out / sign_out as rart_engine_maxpool_pair (may be null). */
int rart_engine_stem_fwd_fused_pair(const void* src, int src_is_u8, const void* wgt_hi, const void* wgt_lo, const float* bias, void* p1_hi,
                                    void* p1_lo, void* argmax_out, void* sign_out, int n, int h, int w, const float* mean_host,
                                    const float* std_host, rart_stream_t stream);
/* The reference-precision form (csrc/stem_pair.hip): pooled gradient and weight table as hi + lo planes of bf16, three MFMA products per
 * fragment pair; replaces rart_engine_maxpool_bwd_pair -> rart_gemm_pair_bf16 (patches, fp32) -> rart_engine_stem_col2im_f32. */
int rart_engine_stem_bwd_fused_pair(const void* dpool_hi, const void* dpool_lo, const void* argmax, const void* wtab_hi, const void* wtab_lo,
                                    float* grad, int n, int h, int w, const float* std_host, rart_stream_t stream);
/* The stem FORWARD as one persistent kernel: normalisation + hi/lo bf16 split (rart_engine_prep_input), 7x7/2 convolution
 * with the folded BatchNorm bias + ReLU (the K = 448 row-tap GEMM) and the 3x3/2 max pool (rart_engine_maxpool_keep) fused;
 * the 112 x 112 x 64 stem output never reaches HBM.  wgt: bf16 [64][wgt_row_stride], row n, column r*32 + s*4 + c =
 * W[n][c][r][s] (s = 7 and c = 3 zero) -- the hi half of the stem table; outputs as rart_engine_maxpool_keep
 * (p1 bf16 [n][h/4][w/4][64], argmax codes and sign bits nullable).  h, w multiples of 4. */
int rart_engine_stem_fwd_fused(const void* src, int src_is_u8, const void* wgt, int wgt_row_stride, const float* bias,
                               void* p1, void* argmax_out, void* sign_out, int n, int h, int w, const float* mean_host,
                               const float* std_host, rart_stream_t stream);
/* fp32 [rows][cols] -> bf16 [rows][dst_cols] (zero padded): dlogits -> GEMM operand. */
int rart_f32_to_bf16_rows(const float* src, void* dst, int rows, int cols, int dst_cols, rart_stream_t stream);

/* ---- ViT-B/16 forward (model `vit_base`; every matmul incl. attention runs on rart_conv_igemm_bf16) ---- */
/* image -> normalised patches [n][(h/p)*(w/p)][3*p*p] (k = c*p*p + r*p + s, the Conv2d(3, D, p, p) weight order),
 * bf16 hi/lo planes. */
int rart_vit_patchify(const void* src, int src_is_u8, void* hi, void* lo, int n, int h, int w, int patch,
                      const float* mean_host, const float* std_host, rart_stream_t stream);
/* x[b][0] = cls_pos0 (class token + its position embedding); x[b][t] += pos[t] for t >= 1.  x bf16 [n][tokens][dim]. */
int rart_vit_add_pos_cls(void* x, const float* cls_pos0, const float* pos, int n, int tokens, int dim,
                         rart_stream_t stream);
int rart_layernorm_bf16(const void* x, const float* gamma, const float* beta, void* out, int rows, int dim,
                        int64_t in_row_stride, int64_t out_row_stride, float eps, rart_stream_t stream);
/* probs[r][0..n_valid) = softmax(scale * scores[r][0..n_valid)), zeros up to ld_out (the K padding of P.V). */
int rart_softmax_rows_bf16(const void* scores, void* probs, int64_t rows, int n_valid, int ld_in, int ld_out,
                           float scale, rart_stream_t stream);
/* Fused multi-head attention on the fused qkv activation [n][tokens][3*heads*head_dim] (q | k | v, heads contiguous,
 * timm layout): out[n][tokens][heads*head_dim] = softmax(q k^T / sqrt(head_dim)) v, bf16 in/out, fp32 statistics.
 * head_dim == 64, tokens <= 224.  (timm Attention.forward; model `vit_base`.) */
int rart_vit_attention(const void* qkv, void* out, int n, int tokens, int heads, int head_dim, rart_stream_t stream);
/* Backward of the same attention: dqkv[n][tokens][3*heads*head_dim] (dQ | dK | dV, qkv's layout) from qkv, the forward's
 * output out[n][tokens][heads*head_dim] and its gradient dout; one fused kernel per (image, head), nothing score-sized
 * leaves the chip. */
int rart_vit_attention_bwd(const void* qkv, const void* out, const void* dout, void* dqkv, int n, int tokens, int heads,
                           int head_dim, rart_stream_t stream);
/* vt[n][heads][head_dim][t_pad] <- V slice of the fused qkv activation [n][tokens][qkv_ld] (zero padded). */
int rart_vit_transpose_v(const void* qkv, void* vt, int n, int tokens, int heads, int head_dim, int qkv_ld, int v_off,
                         int t_pad, rart_stream_t stream);

/* ViT-B/16 backward-to-input pieces (autograd through timm's VisionTransformer for the attack gradient); the
 * contractions run on rart_conv_igemm_bf16 (batched for the per-head products).  bf16 storage, fp32 math. */
int rart_gelu_bf16(const void* u, void* out, size_t n, rart_stream_t stream);            /* exact (erf) GELU */
int rart_gelu_bwd_bf16(const void* dh, const void* u, void* du, size_t n, rart_stream_t stream);   /* du = dh*gelu'(u) */
/* dx = LayerNorm backward to the input (statistics recomputed from x) [+ res]; strides in elements per row. */
int rart_layernorm_bwd_bf16(const void* dy, const void* x, const float* gamma, const void* res, void* dx, int rows, int dim,
                            int64_t dy_row_stride, int64_t x_row_stride, int64_t res_row_stride, int64_t dx_row_stride,
                            float eps, rart_stream_t stream);
/* dS = scale * P * (dP - rowsum(P*dP)) on [0, n_valid), zeros up to ld_out (soft-max backward, scale = 1/sqrt(head_dim)). */
int rart_softmax_bwd_rows_bf16(const void* probs, const void* dprobs, void* dscores, int64_t rows, int n_valid, int ld_p,
                               int ld_dp, int ld_out, float scale, rart_stream_t stream);
/* grad[b][c][y][x] (fp32, w.r.t. the [0,1] image) from the patch-embedding input gradient [b*patches][ld >= 3*p*p]. */
int rart_vit_unpatchify_f32(const void* dpatches, float* grad, int n, int h, int w, int patch, int64_t ld,
                            const float* std_host, rart_stream_t stream);
/* ViT training: LayerNorm backward to the input (dx nullable) plus dgamma = sum_rows dy*xhat, dbeta = sum_rows dy
 * (written, or added when accumulate != 0); deterministic two-level reduction.  workspace: rart_layernorm_bwd_workspace_bytes. */
size_t rart_layernorm_bwd_workspace_bytes(int dim);
int rart_layernorm_bwd_full_bf16(const void* dy, const void* x, const float* gamma, const void* res, void* dx, int rows,
                                 int dim, int64_t dy_row_stride, int64_t x_row_stride, int64_t res_row_stride,
                                 int64_t dx_row_stride, float eps, float* dgamma, float* dbeta, int accumulate, void* workspace,
                                 size_t workspace_bytes, rart_stream_t stream);
/* out[c] (+)= sum_r x[r][c] for a bf16 [rows][cols] matrix with row stride ld (Linear bias gradients, position embedding). */
size_t rart_colsum_workspace_bytes(int rows, int cols);
int rart_colsum_bf16(const void* x, int64_t ld, int rows, int cols, float* out, int accumulate, void* workspace,
                     size_t workspace_bytes, rart_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Training-side step kernels.  The solver the reference launches
 * (RobustART/train/__init__.py:1 -> absent submodule) is configured by
 * exprs/nips_benchmark/pgd_adv_train/resnet50/config.yaml:11-33 (SGD nesterov 0.9, wd 1e-4, label_smooth 0.1,
 * EMA 0.9999) and exprs/nips_benchmark/new_adv_train/vit_base/config.yaml:11-38 (AdamW); the arithmetic is
 * torch.optim.SGD / torch.optim.AdamW / F.cross_entropy(label_smoothing).  All buffers are flat fp32 arenas,
 * 16-byte aligned, updated in place.
 *   g' = grad*grad_scale (+ weight_decay*param for SGD); grad_scale folds the 1/world_size of the gradient mean.
 *   ema (nullable): ema = ema_decay*ema + (1-ema_decay)*param_new.   zero_grad != 0: grad is reset to 0.
 * Hyper-parameters are doubles (Python floats in the reference's configs); the kernels compute in fp32 and form
 * 1-beta, 1-lr*wd, 1-decay in double rounded once, as torch does with its scalar arguments.
 * ------------------------------------------------------------------------------------- */
int rart_sgd_step_f32(float* param, float* grad, float* momentum_buf, float* ema, size_t n, double lr, double momentum,
                      double weight_decay, int nesterov, double grad_scale, double ema_decay, int zero_grad,
                      rart_stream_t stream);
/* step counts from 1 (bias corrections 1-beta^step); decoupled weight decay (param *= 1 - lr*wd). */
int rart_adamw_step_f32(float* param, float* grad, float* exp_avg, float* exp_avg_sq, float* ema, size_t n, double lr,
                        double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale,
                        double ema_decay, int zero_grad, rart_stream_t stream);
int rart_ema_update_f32(float* ema, const float* param, size_t n, double decay, rart_stream_t stream);
/* loss_out[batch] (nullable) = (1-s)*(-log p_y) + s*mean_c(-log p_c);
 * dlogits_out[batch][classes] (nullable) = scale * (softmax - (1-s)*onehot - s/classes). */
int rart_label_smooth_ce_f32(const float* logits, const int64_t* labels, int batch, int classes, double smoothing,
                             double scale, float* loss_out, float* dlogits_out, rart_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Train-mode conv + BatchNorm support (cls_solver training step): the contractions run on
 * rart_conv_igemm_bf16; these are the HBM-bound kernels around it.  Activations are bf16 [rows][channels]
 * (NHWC, rows = batch*h*w), statistics fp32.  channels: a power of two in [8, 2048].
 * Reference arithmetic: torch.nn.BatchNorm2d (training mode) and conv2d's autograd.
 * ------------------------------------------------------------------------------------- */
size_t rart_bn_workspace_bytes(size_t rows, int channels);
/* batch statistics of z, y = [relu](z*scale + shift [+ res]); running stats updated in place when non-NULL
 * (running_var with the unbiased variance); mean_out / invstd_out [channels] are kept for the backward;
 * scale_shift [2][channels] receives gamma*invstd and beta - mean*gamma*invstd.  sign_out (nullable): [rows][channels / 8] bytes,
 * bit j of byte (row, c / 8) = (y[row][c] > 0): the backward's ReLU mask at 1/16 of y's bytes.  stats_partial (nullable):
 * [stats_chunks][2][channels] fp32 column sums / sums of squares of z from the producing convolution (rart_conv_desc.bn_stats_out)
 * -- then no statistics pass reads z. */
int rart_bn_train_forward_bf16(const void* z, const void* res, void* y, void* sign_out, size_t rows, int channels, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, double momentum, double eps,
                               int relu, float* mean_out, float* invstd_out, float* scale_shift, const float* stats_partial,
                               int stats_chunks, void* workspace, size_t workspace_bytes, rart_stream_t stream);
/* g = dy * [ymask > 0] (ymask NULL: g = dy; ymask_is_bits: ymask is the forward's sign_out instead of the bf16 activation);
 * dgamma = sum g*xhat, dbeta = sum g (written, or added when accumulate != 0); dz = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat));
 * g_out (nullable) receives g.  coef: [3][channels] fp32 scratch. */
int rart_bn_train_backward_bf16(const void* dy, const void* ymask, int ymask_is_bits, const void* z, void* dz, void* g_out, size_t rows,
                                int channels, const float* gamma, const float* mean, const float* invstd, float* dgamma,
                                float* dbeta, int accumulate, float* coef, void* workspace, size_t workspace_bytes,
                                rart_stream_t stream);
/* T[(t*channels + c)][m] = src[img(m), oy*sy + tap_dy[t], ox*sx + tap_dx[t], c] for m < batch*grid_h*grid_w, zero
 * outside the image and for the padding up to rows_padded (multiple of 64): the K(=pixel)-contiguous operand of the
 * weight-gradient GEMM.  Stored as one compact slab per K split, dst[m / chunk][row][m % chunk] with rows_total rows
 * per slab (chunk 0 = a single slab; rows_total 0 = taps*channels): a split then touches a few MB instead of one
 * 128-byte piece from each of hundreds of multi-MB rows (TLB / DRAM-page locality).
 * channels: 4 (stem's padded plane) or a multiple of 8. */
int rart_transpose_gather_bf16(const void* src, void* dst, int batch, int src_h, int src_w, int channels, int grid_h,
                               int grid_w, int sy, int sx, int n_taps, const int* tap_dy, const int* tap_dx,
                               long long rows_padded, int chunk, int rows_total, rart_stream_t stream);
/* Weight gradient straight from the NHWC activations (csrc/wgrad_direct.hip): partial[z][t * channels + c][n] = sum over the positions m of
 * K split z (chunk positions each) of x[pixel(m) + tap t][c] * dz[m][n], fp32, the layout rart_wgrad_reduce_f32 folds -- without the two
 * rart_transpose_gather_bf16 passes: the position-major tiles are read out of LDS transposed (ds_read_b64_tr_b16).  x [batch][in_h][in_w]
 * [channels], dz [batch][grid_h][grid_w][dz_cols] bf16; channels 64 or a multiple of 128, dz_cols a multiple of 64, 1..9 taps -- or
 * channels 4 (the stem's padded plane: 32 taps x 4 channels per tile, 4-byte direct loads), dz_cols 64, 1..49 taps;
 * chunk a multiple of 32 with splits * chunk >= batch * grid_h * grid_w; ld_n = dz_cols.  (loss.backward() of cls_solver.py:183-215) */
int rart_wgrad_direct_supported(int channels, int dz_cols, int n_taps);
int rart_wgrad_direct_bf16(const void* x, const void* dz, float* partial, int batch, int in_h, int in_w, int channels, int grid_h, int grid_w,
                           int dz_cols, int stride_y, int stride_x, int n_taps, const int* tap_dy, const int* tap_dx, int splits, int chunk,
                           int ld_n, rart_stream_t stream);
/* grad[n][c][t] (torch conv weight layout, t = r*S + s) (+)= sum_z partial[z][t*channels_padded + c][n];
 * partial: fp32 [splits][taps*channels_padded][ld_n], the split-K output of rart_conv_igemm_bf16; it is scratch:
 * more than 16 splits are first folded 16:1 in place. */
int rart_wgrad_reduce_f32(float* partial, int splits, int taps, int channels, int channels_padded, int n_out,
                          int ld_n, float* grad, int accumulate, rart_stream_t stream);
/* fp32 master weight [n_out][channels][r][s] -> bf16 igemm table over the listed taps:
 * transpose 0: out[n][ti*channels + c] (forward), 1: out[c][ti*n_out + n] (backward to input); rows zero-padded.
 * out_channel_scale (nullable, [n_out]): weight[n] is multiplied by it first -- the eval-mode BatchNorm fold
 * gamma*rsqrt(running_var + eps), so the attack engine can be re-folded from the live parameters every iteration. */
int rart_pack_conv_weight_bf16(const float* weight, const float* out_channel_scale, void* out, int n_out, int channels,
                               int r, int s, int n_taps, const int* tap_r, const int* tap_s, int transpose,
                               int rows_padded, rart_stream_t stream);
/* Many table jobs in ONE launch: the adversarial-training step (cifar10/code/train.py:96-127: attack the CURRENT weights, then update
 * them) re-packs every conv table of the train engine and re-folds every table of the attack engine after each optimizer step -- about
 * 500 launches of 2-10 us kernels.  kind 0 = the job of rart_pack_conv_weight_bf16 (weight / scale / out and the geometry; <= 16 taps),
 * kind 1 = the job of rart_pack_frag_bf16 (src16 -> out, rows x k).  jobs_device: an array of rart_pack_job in DEVICE memory, built once
 * by the caller (every pointer in it must stay valid); blocks_per_job workgroups walk each job grid-strided. */
typedef struct rart_pack_job {
  int kind, n_out, channels, r, s, n_taps, transpose, rows_padded, rows, k;
  int tap_r[16], tap_s[16];
  const float* weight;
  const float* out_channel_scale;
  const void* src16;
  void* out;
} rart_pack_job;
int rart_pack_jobs_bf16(const rart_pack_job* jobs_device, int n_jobs, int blocks_per_job, rart_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ROBUSTART_HIP_H */
