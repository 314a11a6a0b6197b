// C-ABI entry points: argument validation + dispatch to the kernel families.
// Boundary mirrored: RobustART/noise/utils/imagenet_c/__init__.py:13-35 (corrupt) and
// RobustART/noise/utils/add_noise_utils.py:22-31 (the per-image batch loop).
#include <stdarg.h>
#include <string.h>
#include "rart_common.h"
#include <map>
#include <mutex>
#include <utility>

static thread_local char g_err[512] = "";

void rart_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

bool rart_raise_dynamic_lds(const void* kernel, size_t bytes, const char* what) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> raised;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = raised[std::make_pair(kernel, dev)];
  if (bytes <= have) return true;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    rart_set_error("%s: cannot raise the dynamic LDS limit to %zu bytes", what, bytes);
    return false;
  }
  have = bytes;
  return true;
}

static const char* kNames[RART_NUM_CORRUPTIONS] = {
    "gaussian_noise", "shot_noise", "impulse_noise", "defocus_blur", "glass_blur", "motion_blur",
    "zoom_blur", "snow", "frost", "fog", "brightness", "contrast", "elastic_transform", "pixelate",
    "jpeg_compression", "speckle_noise", "gaussian_blur", "spatter", "saturate"};

enum Family { F_POINT, F_RESAMPLE, F_JPEG, F_STENCIL, F_COMPOSITE };

static Family family_of(int id) {
  switch (id) {
    case RART_GAUSSIAN_NOISE: case RART_SHOT_NOISE: case RART_IMPULSE_NOISE: case RART_SPECKLE_NOISE:
    case RART_CONTRAST: case RART_BRIGHTNESS: case RART_SATURATE: case RART_FROST:
      return F_POINT;
    case RART_PIXELATE: case RART_ZOOM_BLUR:
      return F_RESAMPLE;
    case RART_JPEG_COMPRESSION:
      return F_JPEG;
    case RART_GAUSSIAN_BLUR: case RART_DEFOCUS_BLUR: case RART_MOTION_BLUR: case RART_GLASS_BLUR:
      return F_STENCIL;
    default:
      return F_COMPOSITE;  // fog, snow, elastic_transform, spatter
  }
}

// Corruptions whose kernels gather from neighbouring pixels cannot run in place; when the caller
// aliases in == out (the reference mutates its input array) the input is first staged into the
// head of the workspace.
static bool needs_staging(int id) {
  switch (id) {
    case RART_ZOOM_BLUR: case RART_DEFOCUS_BLUR: case RART_MOTION_BLUR:
    case RART_GAUSSIAN_BLUR:   // the fused separable kernel reads a tile's halo from `in` while neighbouring tiles write `out`
      return true;   // every other gathering corruption already goes through its own intermediate buffer
    default:
      return false;
  }
}

extern "C" {

int rart_version(void) { return RART_VERSION; }
const char* rart_last_error_string(void) { return g_err; }
const char* rart_corruption_name(int id) {
  return (id >= 0 && id < RART_NUM_CORRUPTIONS) ? kNames[id] : nullptr;
}

size_t rart_corrupt_workspace_bytes(int id, int severity, int n, int h, int w) {
  if (id < 0 || id >= RART_NUM_CORRUPTIONS || n <= 0 || h <= 0 || w <= 0) return 0;
  const size_t stage = needs_staging(id) ? rart_align_up((size_t)n * h * w * 3, 256) : 0;
  switch (family_of(id)) {
    case F_POINT: return stage + rart_ws_pointwise(id, severity, n, h, w);
    case F_RESAMPLE: return stage + rart_ws_resample(id, severity, n, h, w);
    case F_JPEG: return stage + rart_ws_jpeg(severity, n, h, w);
    case F_STENCIL: return stage + rart_ws_stencil(id, severity, n, h, w);
    case F_COMPOSITE: return stage + rart_ws_composite(id, severity, n, h, w);
  }
  return 0;
}

int rart_corrupt_u8(const uint8_t* in, uint8_t* out, int n, int h, int w, int corruption_id, int severity,
                    uint64_t seed, uint64_t sample_offset, const void* const* injected, int n_injected,
                    void* workspace, size_t workspace_bytes, rart_stream_t stream) {
  g_err[0] = 0;
  RART_CHECK_ARG(corruption_id >= 0 && corruption_id < RART_NUM_CORRUPTIONS, "unknown corruption id %d",
                 corruption_id);
  RART_CHECK_ARG(severity >= 1 && severity <= 5, "severity %d outside 1..5", severity);
  RART_CHECK_ARG(n >= 0 && h > 0 && w > 0, "bad shape n=%d h=%d w=%d", n, h, w);
  if (n == 0) return RART_OK;  // empty batch: the reference's loop body never runs
  RART_CHECK_ARG(in != nullptr && out != nullptr, "null image pointer");
  RART_CHECK_ARG(n_injected >= 0 && (n_injected == 0 || injected != nullptr), "bad injected array");
  const size_t need = rart_corrupt_workspace_bytes(corruption_id, severity, n, h, w);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need)) {
    rart_set_error("%s: workspace of %zu bytes required, got %zu", kNames[corruption_id], need, workspace_bytes);
    return RART_ERR_WORKSPACE;
  }
  RartCorruptArgs a{in, out, n, h, w, severity, seed, sample_offset, n_injected ? injected : nullptr,
                    n_injected, workspace, workspace_bytes, (hipStream_t)stream};
  if (needs_staging(corruption_id)) {
    const size_t stage = rart_align_up((size_t)n * h * w * 3, 256);
    if (in == out) {
      if (hipMemcpyAsync(workspace, in, (size_t)n * h * w * 3, hipMemcpyDeviceToDevice, a.stream) != hipSuccess) {
        rart_set_error("%s: staging copy failed", kNames[corruption_id]);
        return RART_ERR_HIP;
      }
      a.in = (const uint8_t*)workspace;
    }
    a.workspace = (uint8_t*)workspace + stage;
    a.workspace_bytes = workspace_bytes - stage;
  }
  switch (family_of(corruption_id)) {
    case F_POINT: return rart_launch_pointwise(corruption_id, a);
    case F_RESAMPLE: return rart_launch_resample(corruption_id, a);
    case F_JPEG: return rart_launch_jpeg(a);
    case F_STENCIL: return rart_launch_stencil(corruption_id, a);
    case F_COMPOSITE: return rart_launch_composite(corruption_id, a);
  }
  return RART_ERR_INVALID;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// RNG fills and the u8 -> normalised tensor kernel
// ---------------------------------------------------------------------------------------
namespace {
constexpr int kBlock = 256;

// word pair p of a sample = threefry(ctr0(p, stream), sample); element 2p, 2p+1
__global__ __launch_bounds__(kBlock) void k_fill_u32(uint32_t* __restrict__ out, uint32_t elems, uint32_t k0,
                                                     uint32_t k1, uint32_t sample_base, int stream_id) {
  const uint32_t sample = blockIdx.y;
  uint32_t* dst = out + (size_t)sample * elems;
  const uint32_t npair = (elems + 1) / 2;
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < npair; p += gridDim.x * kBlock) {
    const uint2 w = threefry2x32(k0, k1, rart_ctr0(p, stream_id), sample_base + sample);
    dst[2 * p] = w.x;
    if (2 * p + 1 < elems) dst[2 * p + 1] = w.y;
  }
}

// quad q of a sample = rart_normal4(q, stream): elements 4q..4q+3 (the field the noise kernels use)
__global__ __launch_bounds__(kBlock) void k_fill_normal(float* __restrict__ out, uint32_t elems, uint32_t k0,
                                                        uint32_t k1, uint32_t sample_base, int stream_id) {
  const uint32_t sample = blockIdx.y;
  float* dst = out + (size_t)sample * elems;
  const uint32_t nquad = (elems + 3) / 4;
  for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < nquad; q += gridDim.x * kBlock) {
    const float4 z = rart_normal4(k0, k1, q, stream_id, sample_base + sample);
    const float zz[4] = {z.x, z.y, z.z, z.w};
    for (int j = 0; j < 4; ++j)
      if (q * 4 + j < elems) dst[q * 4 + j] = zz[j];
  }
}

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// (x/255 - mean)/std, layouts NCHW / NHWC, dtypes fp32 / bf16.  One thread per pixel.
template <typename T, bool NHWC>
__global__ __launch_bounds__(kBlock) void k_u8_to_norm(const uint8_t* __restrict__ in, T* __restrict__ out,
                                                       uint32_t hw, uint32_t n) {
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float istd[3] = {1.0f / 0.229f, 1.0f / 0.224f, 1.0f / 0.225f};
  const size_t total = (size_t)hw * n;
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < total; p += (size_t)gridDim.x * kBlock) {
    const size_t img = p / hw, pix = p % hw;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = ((float)in[p * 3 + c] * (1.0f / 255.0f) - mean[c]) * istd[c];
      const size_t o = NHWC ? p * 3 + c : (img * 3 + c) * hw + pix;
      if constexpr (sizeof(T) == 4) out[o] = v; else out[o] = f32_to_bf16_rne(v);
    }
  }
}
// x01[n][c][h][w] = u8[n][h][w][c] * (1 / 255) -- torch's u8.permute(0, 3, 1, 2).float().div(255) multiplies by the fp32 reciprocal of a
// scalar divisor, so this is bit-identical to that expression (tests/test_corruptions_gpu.py): the attack
// tensors are fp32 NCHW in [0,1] (adv/attack.py:20-23), the corruption kernels and datasets hand over uint8 NHWC.  Four pixels of one
// channel per thread: 12 contiguous input bytes per 4 pixels are shared by the three channel threads through L1, stores are 16 bytes.
__global__ __launch_bounds__(kBlock) void k_u8_to_unit_nchw(const uint8_t* __restrict__ in, float* __restrict__ out, uint32_t hw,
                                                            uint32_t n) {
  const uint32_t q = hw / 4;                             // host: hw % 4 == 0
  const size_t total = (size_t)n * 3 * q;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const uint32_t p4 = (uint32_t)(i % q), c = (uint32_t)((i / q) % 3), img = (uint32_t)(i / ((size_t)3 * q));
    const uint8_t* s = in + ((size_t)img * hw + (size_t)p4 * 4) * 3 + c;
    const float r = 1.0f / 255.0f;
    const float4 v = make_float4((float)s[0] * r, (float)s[3] * r, (float)s[6] * r, (float)s[9] * r);
    reinterpret_cast<float4*>(out)[((size_t)img * 3 + c) * q + p4] = v;
  }
}
}  // namespace

static dim3 fill_grid(uint32_t items, int n) {
  uint32_t gx = (items + kBlock - 1) / kBlock;
  uint32_t cap = (uint32_t)(2048 / (n < 1 ? 1 : n));
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return dim3(gx, (uint32_t)n, 1);
}

extern "C" {

int rart_rng_uniform_u32(uint32_t* out, int n_samples, size_t elems, uint64_t seed, uint64_t sample_offset,
                         int stream_id, rart_stream_t stream) {
  g_err[0] = 0;
  RART_CHECK_ARG(out && n_samples > 0 && elems > 0 && elems < (1ull << 29), "bad rng fill arguments");
  RART_CHECK_ARG(stream_id >= 0 && stream_id < 16, "stream_id outside 0..15");
  hipLaunchKernelGGL(k_fill_u32, fill_grid((uint32_t)((elems + 1) / 2), n_samples), dim3(kBlock), 0,
                     (hipStream_t)stream, out, (uint32_t)elems, (uint32_t)seed, (uint32_t)(seed >> 32),
                     (uint32_t)sample_offset, stream_id);
  RART_CHECK_LAUNCH("rart_rng_uniform_u32");
  return RART_OK;
}

int rart_rng_normal_f32(float* out, int n_samples, size_t elems, uint64_t seed, uint64_t sample_offset,
                        int stream_id, rart_stream_t stream) {
  g_err[0] = 0;
  RART_CHECK_ARG(out && n_samples > 0 && elems > 0 && elems < (1ull << 30), "bad rng fill arguments");
  RART_CHECK_ARG(stream_id >= 0 && stream_id < 16, "stream_id outside 0..15");
  hipLaunchKernelGGL(k_fill_normal, fill_grid((uint32_t)((elems + 3) / 4), n_samples), dim3(kBlock), 0,
                     (hipStream_t)stream, out, (uint32_t)elems, (uint32_t)seed, (uint32_t)(seed >> 32),
                     (uint32_t)sample_offset, stream_id);
  RART_CHECK_LAUNCH("rart_rng_normal_f32");
  return RART_OK;
}

int rart_u8_to_normalized(const uint8_t* in, void* out, int n, int h, int w, int out_dtype, int out_layout,
                          rart_stream_t stream) {
  g_err[0] = 0;
  RART_CHECK_ARG(in && out && n > 0 && h > 0 && w > 0, "bad arguments");
  RART_CHECK_ARG((out_dtype == 0 || out_dtype == 1) && (out_layout == 0 || out_layout == 1),
                 "out_dtype must be 0 (fp32) / 1 (bf16), out_layout 0 (NCHW) / 1 (NHWC)");
  const uint32_t hw = (uint32_t)h * w;
  const dim3 g(rart_grid_for((size_t)hw * n));
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == 0 && out_layout == 0)
    hipLaunchKernelGGL((k_u8_to_norm<float, false>), g, dim3(kBlock), 0, s, in, (float*)out, hw, (uint32_t)n);
  else if (out_dtype == 0)
    hipLaunchKernelGGL((k_u8_to_norm<float, true>), g, dim3(kBlock), 0, s, in, (float*)out, hw, (uint32_t)n);
  else if (out_layout == 0)
    hipLaunchKernelGGL((k_u8_to_norm<uint16_t, false>), g, dim3(kBlock), 0, s, in, (uint16_t*)out, hw, (uint32_t)n);
  else
    hipLaunchKernelGGL((k_u8_to_norm<uint16_t, true>), g, dim3(kBlock), 0, s, in, (uint16_t*)out, hw, (uint32_t)n);
  RART_CHECK_LAUNCH("rart_u8_to_normalized");
  return RART_OK;
}

int rart_u8_to_unit_f32_nchw(const uint8_t* in, float* out, int n, int h, int w, rart_stream_t stream) {
  g_err[0] = 0;
  RART_CHECK_ARG(in && out && n > 0 && h > 0 && w > 0 && ((size_t)h * w) % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                 "rart_u8_to_unit_f32_nchw: h*w must be a multiple of 4 and the output 16-byte aligned");
  hipLaunchKernelGGL(k_u8_to_unit_nchw, dim3(rart_grid_for((size_t)n * 3 * h * w / 4, kBlock, 256 * 16)), dim3(kBlock), 0,
                     (hipStream_t)stream, in, out, (uint32_t)(h * w), (uint32_t)n);
  RART_CHECK_LAUNCH("rart_u8_to_unit_f32_nchw");
  return RART_OK;
}

}  // extern "C"
