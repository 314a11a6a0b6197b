// Non-GEMM kernels of the ViT-B/16 forward engine (gfx950): patch extraction (+normalise, bf16 hi/lo split),
// class-token / position-embedding add, LayerNorm, attention soft-max rows and the V transpose that lets
// P.V run on the implicit-GEMM kernel.  bf16 storage, fp32 statistics.  Reference: RobustART/model/__init__.py:1
// -> absent submodule; `vit_base` = timm ViT-B/16 (jx_vit_base_p16_224): 12 blocks, width 768,
// 12 heads, qkv bias, LayerNorm eps 1e-6, exact GELU.
#include "rart_common.h"

namespace {
constexpr int kBlock = 256;
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
struct Norm3 {
  float mean[3], istd[3];
};

// out[b][p][c*ps*ps + r*ps + s] = (x[b][c][py*ps + r][px*ps + s] - mean)/std, as hi and lo bf16 planes
template <bool SRC_U8>
__global__ __launch_bounds__(kBlock) void k_patchify(const void* __restrict__ src, uint16_t* __restrict__ hi,
                                                     uint16_t* __restrict__ lo, int n, int h, int w, int ps, Norm3 nm) {
  const int gw = w / ps, gh = h / ps, kk = 3 * ps * ps;
  const size_t total = (size_t)n * gh * gw * kk;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int k = (int)(i % kk);
    const size_t pidx = i / kk;
    const int px = (int)(pidx % gw), py = (int)((pidx / gw) % gh), img = (int)(pidx / ((size_t)gw * gh));
    const int c = k / (ps * ps), r = (k / ps) % ps, s = k % ps;
    const int y = py * ps + r, x = px * ps + s;
    float v01;
    if (SRC_U8)
      v01 = (float)((const uint8_t*)src)[(((size_t)img * h + y) * w + x) * 3 + c] * (1.0f / 255.0f);
    else
      v01 = ((const float*)src)[(((size_t)img * 3 + c) * h + y) * w + x];
    const float v = (v01 - nm.mean[c]) * nm.istd[c];
    const uint16_t hv = f2bf(v);
    hi[i] = hv;
    lo[i] = f2bf(v - bf2f(hv));
  }
}

// x[b][0][:] = cls_pos0; x[b][t][:] += pos[t] (t >= 1)
__global__ __launch_bounds__(kBlock) void k_add_pos_cls(uint16_t* __restrict__ x, const float* __restrict__ cls_pos0,
                                                        const float* __restrict__ pos, int n, int t, int d) {
  const size_t total = (size_t)n * t * d;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % d), tok = (int)((i / d) % t);
    x[i] = tok == 0 ? f2bf(cls_pos0[c]) : f2bf(bf2f(x[i]) + pos[(size_t)tok * d + c]);
  }
}

// LayerNorm over the last dim, one wave per row, fp32 two-pass statistics
__global__ __launch_bounds__(kBlock) void k_layernorm(const uint16_t* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ b, uint16_t* __restrict__ out,
                                                      int rows, int d, long long in_stride, long long out_stride,
                                                      float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint16_t* xr = x + (size_t)row * in_stride;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += bf2f(xr[c]);
  const float mean = rart_wave_sum(s) / (float)d;
  float v = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float t = bf2f(xr[c]) - mean;
    v += t * t;
  }
  const float rstd = rsqrtf(rart_wave_sum(v) / (float)d + eps);
  uint16_t* o = out + (size_t)row * out_stride;
  for (int c = lane; c < d; c += 64) o[c] = f2bf((bf2f(xr[c]) - mean) * rstd * g[c] + b[c]);
}

// P[row][0..n_valid) = softmax(scale * S[row][0..n_valid)), P[row][n_valid..ld_out) = 0; one wave per row
__global__ __launch_bounds__(kBlock) void k_softmax_rows(const uint16_t* __restrict__ sm, uint16_t* __restrict__ pm,
                                                         long long rows, int n_valid, int ld_in, int ld_out,
                                                         float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint16_t* sr = sm + row * ld_in;
  uint16_t* pr = pm + row * ld_out;
  float mx = -INFINITY;
  for (int c = lane; c < n_valid; c += 64) mx = fmaxf(mx, bf2f(sr[c]) * scale);
  mx = rart_wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < n_valid; c += 64) sum += __expf(bf2f(sr[c]) * scale - mx);
  const float inv = 1.0f / rart_wave_sum(sum);
  for (int c = lane; c < ld_out; c += 64) pr[c] = c < n_valid ? f2bf(__expf(bf2f(sr[c]) * scale - mx) * inv) : (uint16_t)0;
}

// vt[b][h][dd][t] = qkv[b][t][v_off + h*hd + dd] (t < T), 0 for T <= t < t_pad
__global__ __launch_bounds__(kBlock) void k_transpose_v(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ vt,
                                                        int n, int t, int heads, int hd, int ld, int v_off, int t_pad) {
  const size_t total = (size_t)n * heads * hd * t_pad;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int tok = (int)(i % t_pad), dd = (int)((i / t_pad) % hd), hh = (int)((i / ((size_t)t_pad * hd)) % heads);
    const int img = (int)(i / ((size_t)t_pad * hd * heads));
    vt[i] = tok < t ? qkv[((size_t)img * t + tok) * ld + v_off + hh * hd + dd] : (uint16_t)0;
  }
}
int grid_for(size_t items) { return rart_grid_for(items, kBlock, 256 * 16); }
}  // namespace

extern "C" {

int rart_vit_patchify(const void* src, int src_is_u8, void* hi, void* lo, int n, int h, int w, int patch,
                      const float* mean_host, const float* std_host, rart_stream_t stream) {
  RART_CHECK_ARG(src && hi && lo && n > 0 && patch > 0 && h % patch == 0 && w % patch == 0, "rart_vit_patchify: bad arguments");
  Norm3 nm;
  for (int c = 0; c < 3; ++c) {
    nm.mean[c] = mean_host ? mean_host[c] : 0.f;
    nm.istd[c] = std_host ? 1.0f / std_host[c] : 1.f;
  }
  const size_t total = (size_t)n * (h / patch) * (w / patch) * 3 * patch * patch;
  if (src_is_u8)
    hipLaunchKernelGGL(k_patchify<true>, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, src, (uint16_t*)hi,
                       (uint16_t*)lo, n, h, w, patch, nm);
  else
    hipLaunchKernelGGL(k_patchify<false>, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, src,
                       (uint16_t*)hi, (uint16_t*)lo, n, h, w, patch, nm);
  RART_CHECK_LAUNCH("rart_vit_patchify");
  return RART_OK;
}

int rart_vit_add_pos_cls(void* x, const float* cls_pos0, const float* pos, int n, int tokens, int dim,
                         rart_stream_t stream) {
  RART_CHECK_ARG(x && cls_pos0 && pos && n > 0 && tokens > 0 && dim > 0, "rart_vit_add_pos_cls: bad arguments");
  hipLaunchKernelGGL(k_add_pos_cls, dim3(grid_for((size_t)n * tokens * dim)), dim3(kBlock), 0, (hipStream_t)stream,
                     (uint16_t*)x, cls_pos0, pos, n, tokens, dim);
  RART_CHECK_LAUNCH("rart_vit_add_pos_cls");
  return RART_OK;
}

int rart_layernorm_bf16(const void* x, const float* gamma, const float* beta, void* out, int rows, int dim,
                        int64_t in_row_stride, int64_t out_row_stride, float eps, rart_stream_t stream) {
  RART_CHECK_ARG(x && gamma && beta && out && rows > 0 && dim > 0, "rart_layernorm_bf16: bad arguments");
  hipLaunchKernelGGL(k_layernorm, dim3((rows + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint16_t*)x, gamma, beta, (uint16_t*)out, rows, dim, (long long)in_row_stride,
                     (long long)out_row_stride, eps);
  RART_CHECK_LAUNCH("rart_layernorm_bf16");
  return RART_OK;
}

int rart_softmax_rows_bf16(const void* scores, void* probs, int64_t rows, int n_valid, int ld_in, int ld_out,
                           float scale, rart_stream_t stream) {
  RART_CHECK_ARG(scores && probs && rows > 0 && n_valid > 0 && ld_in >= n_valid && ld_out >= n_valid,
                 "rart_softmax_rows_bf16: bad arguments");
  hipLaunchKernelGGL(k_softmax_rows, dim3((uint32_t)((rows + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                     (hipStream_t)stream, (const uint16_t*)scores, (uint16_t*)probs, (long long)rows, n_valid, ld_in,
                     ld_out, scale);
  RART_CHECK_LAUNCH("rart_softmax_rows_bf16");
  return RART_OK;
}

int rart_vit_transpose_v(const void* qkv, void* vt, int n, int tokens, int heads, int head_dim, int qkv_ld, int v_off,
                         int t_pad, rart_stream_t stream) {
  RART_CHECK_ARG(qkv && vt && n > 0 && tokens > 0 && t_pad >= tokens, "rart_vit_transpose_v: bad arguments");
  hipLaunchKernelGGL(k_transpose_v, dim3(grid_for((size_t)n * heads * head_dim * t_pad)), dim3(kBlock), 0,
                     (hipStream_t)stream, (const uint16_t*)qkv, (uint16_t*)vt, n, tokens, heads, head_dim, qkv_ld, v_off,
                     t_pad);
  RART_CHECK_LAUNCH("rart_vit_transpose_v");
  return RART_OK;
}

}  // extern "C"
