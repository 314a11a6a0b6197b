// ViT-B/16 backward-to-input support kernels for gfx950: the pieces of `autograd(loss, x)` through timm's
// VisionTransformer (model `vit_base`; reference use: the gradient step of every attack, adv/attack.py:21-22,
// autopgd_base.py:271-289) that are not contractions.  The contractions -- all dgrad GEMMs and the five per-head
// products of the attention backward (dP = dO V^T, dQ = dS K, dK = dS^T Q, dV = P^T dO, and the recomputed
// S = Q K^T) -- run on rart_conv_igemm_bf16 as batched problems; transposed operands come from
// rart_transpose_gather_bf16 / rart_vit_transpose_v.  HBM-bound row / elementwise kernels, bf16 storage, fp32 math.
#include "rart_common.h"

namespace {
constexpr int kBlock = 256;
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float v) {
  // d/dv [v * Phi(v)] = Phi(v) + v * phi(v)
  return 0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
}

template <int BWD>
__global__ __launch_bounds__(kBlock) void k_gelu(const uint4* __restrict__ a, const uint4* __restrict__ u,
                                                 uint4* __restrict__ out, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n8; i += (size_t)gridDim.x * kBlock) {
    const uint4 uv = u[i];
    const uint32_t uw[4] = {uv.x, uv.y, uv.z, uv.w};
    uint32_t aw[4] = {0, 0, 0, 0};
    if (BWD) {
      const uint4 av = a[i];
      aw[0] = av.x; aw[1] = av.y; aw[2] = av.z; aw[3] = av.w;
    }
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u0 = bf2f((uint16_t)(uw[j] & 0xFFFF)), u1 = bf2f((uint16_t)(uw[j] >> 16));
      float r0, r1;
      if (BWD) {
        r0 = bf2f((uint16_t)(aw[j] & 0xFFFF)) * gelu_grad_f(u0);
        r1 = bf2f((uint16_t)(aw[j] >> 16)) * gelu_grad_f(u1);
      } else {
        r0 = gelu_f(u0);
        r1 = gelu_f(u1);
      }
      o[j] = f2bf(r0) | ((uint32_t)f2bf(r1) << 16);
    }
    out[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = __uint_as_float(w[j] << 16);
    f[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
  }
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16), f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16),
                    f2bf(f[4]) | ((uint32_t)f2bf(f[5]) << 16), f2bf(f[6]) | ((uint32_t)f2bf(f[7]) << 16));
}
__device__ __forceinline__ void load_row(const uint16_t* row, int nv, int lane, float r[2][8]) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int v = lane + 64 * k;
    uint4 q = make_uint4(0, 0, 0, 0);
    if (v < nv) q = *reinterpret_cast<const uint4*>(row + (size_t)v * 8);
    unpack8(q, r[k]);
  }
}

// LayerNorm backward to the input, one wave per row, rows held in registers as 16-byte vectors (every tensor is read
// once); statistics recomputed from x in fp32:
//   xhat = (x - mean) * rstd;  g = dy * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ res]
__global__ __launch_bounds__(kBlock) void k_layernorm_bwd(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x,
                                                          const float* __restrict__ gamma, const uint16_t* __restrict__ res,
                                                          uint16_t* __restrict__ dx, int rows, int d, long long dy_stride,
                                                          long long x_stride, long long res_stride, long long dx_stride,
                                                          float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = d / 8;
  float xr[2][8], gr[2][8];
  load_row(x + (size_t)row * x_stride, nv, lane, xr);
  load_row(dy + (size_t)row * dy_stride, nv, lane, gr);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += xr[k][j];
  const float mean = rart_wave_sum(s) / (float)d;
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (lane + 64 * k < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = xr[k][j] - mean;
        v += t * t;
      }
    }
  const float rstd = rsqrtf(rart_wave_sum(v) / (float)d + eps);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int vi = lane + 64 * k;
    if (vi < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        gr[k][j] *= gamma[vi * 8 + j];
        xr[k][j] = (xr[k][j] - mean) * rstd;                     // xhat
        sg += gr[k][j];
        sgx += gr[k][j] * xr[k][j];
      }
    }
  }
  const float mg = rart_wave_sum(sg) / (float)d, mgx = rart_wave_sum(sgx) / (float)d;
  uint16_t* o = dx + (size_t)row * dx_stride;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int vi = lane + 64 * k;
    if (vi < nv) {
      float r[8], rs[8];
      if (res) unpack8(*reinterpret_cast<const uint4*>(res + (size_t)row * res_stride + (size_t)vi * 8), rs);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        r[j] = rstd * (gr[k][j] - mg - xr[k][j] * mgx);
        if (res) r[j] += rs[j];
      }
      *reinterpret_cast<uint4*>(o + (size_t)vi * 8) = pack8(r);
    }
  }
}

// dS[row][c] = scale * P[row][c] * (dP[row][c] - sum_k P[row][k] dP[row][k]), c < n_valid; zeros up to ld_out
__global__ __launch_bounds__(kBlock) void k_softmax_bwd_rows(const uint16_t* __restrict__ p, const uint16_t* __restrict__ dp,
                                                             uint16_t* __restrict__ ds, long long rows, int n_valid, int ld_p,
                                                             int ld_dp, int ld_out, float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float pr[2][8], dr[2][8];
  load_row(p + row * ld_p, ld_p / 8, lane, pr);
  load_row(dp + row * ld_dp, ld_dp / 8, lane, dr);
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if ((lane + 64 * k) * 8 + j < n_valid) dot += pr[k][j] * dr[k][j];
  dot = rart_wave_sum(dot);
  uint16_t* o = ds + row * ld_out;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int vi = lane + 64 * k;
    if (vi < ld_out / 8) {
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (vi * 8 + j < n_valid) ? scale * pr[k][j] * (dr[k][j] - dot) : 0.f;
      *reinterpret_cast<uint4*>(o + (size_t)vi * 8) = pack8(r);
    }
  }
}

// grad[b][c][y][x] = dpatch[b][patch(y, x)][c*ps*ps + (y % ps)*ps + (x % ps)] * istd[c]   (fp32 NCHW, pixels in [0,1])
struct Istd3 { float v[3]; };
__global__ __launch_bounds__(kBlock) void k_unpatchify(const uint16_t* __restrict__ dp, float* __restrict__ grad, int n, int h,
                                                       int w, int ps, long long ld, Istd3 is) {
  // eight consecutive x of one image row (ps % 8 == 0, ld % 8 == 0: they sit in one patch row): one 16-byte load, two 16-byte stores
  const uint32_t gw = (uint32_t)(w / ps), gh = (uint32_t)(h / ps), w8 = (uint32_t)w / 8;
  const uint32_t total = (uint32_t)n * 3u * (uint32_t)h * w8;              // host: < 2^32
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
    const uint32_t x8 = i % w8, t = i / w8, y = t % (uint32_t)h, t2 = t / (uint32_t)h, c = t2 % 3u, img = t2 / 3u;
    const uint32_t x = x8 * 8, px = x / (uint32_t)ps, py = y / (uint32_t)ps;
    const size_t prow = (size_t)img * gh * gw + (size_t)py * gw + px;
    const uint4 v = *reinterpret_cast<const uint4*>(dp + prow * ld + (size_t)c * ps * ps + (y % (uint32_t)ps) * ps + (x % (uint32_t)ps));
    const float sc = is.v[c];
    float4* o = reinterpret_cast<float4*>(grad + (size_t)i * 8);
    o[0] = make_float4(__uint_as_float(v.x << 16) * sc, __uint_as_float(v.x & 0xFFFF0000u) * sc, __uint_as_float(v.y << 16) * sc,
                       __uint_as_float(v.y & 0xFFFF0000u) * sc);
    o[1] = make_float4(__uint_as_float(v.z << 16) * sc, __uint_as_float(v.z & 0xFFFF0000u) * sc, __uint_as_float(v.w << 16) * sc,
                       __uint_as_float(v.w & 0xFFFF0000u) * sc);
  }
}
int grid_for(size_t items) { return rart_grid_for(items, kBlock, 256 * 16); }
}  // namespace

extern "C" {

int rart_gelu_bf16(const void* u, void* out, size_t n, rart_stream_t stream) {
  RART_CHECK_ARG(u && out && n > 0 && n % 8 == 0, "rart_gelu_bf16: n must be a positive multiple of 8");
  hipLaunchKernelGGL(k_gelu<0>, dim3(grid_for(n / 8)), dim3(kBlock), 0, (hipStream_t)stream, nullptr, (const uint4*)u,
                     (uint4*)out, n / 8);
  RART_CHECK_LAUNCH("rart_gelu_bf16");
  return RART_OK;
}

int rart_gelu_bwd_bf16(const void* dh, const void* u, void* du, size_t n, rart_stream_t stream) {
  RART_CHECK_ARG(dh && u && du && n > 0 && n % 8 == 0, "rart_gelu_bwd_bf16: n must be a positive multiple of 8");
  hipLaunchKernelGGL(k_gelu<1>, dim3(grid_for(n / 8)), dim3(kBlock), 0, (hipStream_t)stream, (const uint4*)dh,
                     (const uint4*)u, (uint4*)du, n / 8);
  RART_CHECK_LAUNCH("rart_gelu_bwd_bf16");
  return RART_OK;
}

int rart_layernorm_bwd_bf16(const void* dy, const void* x, const float* gamma, const void* res, void* dx, int rows, int dim,
                            int64_t dy_row_stride, int64_t x_row_stride, int64_t res_row_stride, int64_t dx_row_stride,
                            float eps, rart_stream_t stream) {
  RART_CHECK_ARG(dy && x && gamma && dx && rows > 0 && dim > 0, "rart_layernorm_bwd_bf16: bad arguments");
  RART_CHECK_ARG(dim % 8 == 0 && dim <= 1024 && dy_row_stride % 8 == 0 && x_row_stride % 8 == 0 && res_row_stride % 8 == 0 &&
                     dx_row_stride % 8 == 0, "rart_layernorm_bwd_bf16: dim a multiple of 8, at most 1024; strides multiples of 8");
  hipLaunchKernelGGL(k_layernorm_bwd, dim3((rows + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint16_t*)dy, (const uint16_t*)x, gamma, (const uint16_t*)res, (uint16_t*)dx, rows, dim,
                     (long long)dy_row_stride, (long long)x_row_stride, (long long)res_row_stride, (long long)dx_row_stride, eps);
  RART_CHECK_LAUNCH("rart_layernorm_bwd_bf16");
  return RART_OK;
}

int rart_softmax_bwd_rows_bf16(const void* probs, const void* dprobs, void* dscores, int64_t rows, int n_valid, int ld_p,
                               int ld_dp, int ld_out, float scale, rart_stream_t stream) {
  RART_CHECK_ARG(probs && dprobs && dscores && rows > 0 && n_valid > 0 && ld_p >= n_valid && ld_dp >= n_valid &&
                     ld_out >= n_valid, "rart_softmax_bwd_rows_bf16: bad arguments");
  RART_CHECK_ARG(ld_p % 8 == 0 && ld_dp % 8 == 0 && ld_out % 8 == 0 && ld_p <= 1024 && ld_dp <= 1024 && ld_out <= 1024,
                 "rart_softmax_bwd_rows_bf16: leading dimensions must be multiples of 8, at most 1024");
  hipLaunchKernelGGL(k_softmax_bwd_rows, dim3((uint32_t)((rows + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                     (hipStream_t)stream, (const uint16_t*)probs, (const uint16_t*)dprobs, (uint16_t*)dscores,
                     (long long)rows, n_valid, ld_p, ld_dp, ld_out, scale);
  RART_CHECK_LAUNCH("rart_softmax_bwd_rows_bf16");
  return RART_OK;
}

int rart_vit_unpatchify_f32(const void* dpatches, float* grad, int n, int h, int w, int patch, int64_t ld,
                            const float* std_host, rart_stream_t stream) {
  RART_CHECK_ARG(dpatches && grad && std_host && n > 0 && patch > 0 && h % patch == 0 && w % patch == 0 &&
                     ld >= 3 * patch * patch, "rart_vit_unpatchify_f32: bad arguments");
  RART_CHECK_ARG(patch % 8 == 0 && ld % 8 == 0 && (size_t)n * 3 * h * w / 8 < (1ull << 32),
                 "rart_vit_unpatchify_f32: patch side and row stride must be multiples of 8 (eight pixels per thread)");
  Istd3 is;
  for (int c = 0; c < 3; ++c) is.v[c] = 1.0f / std_host[c];
  hipLaunchKernelGGL(k_unpatchify, dim3(grid_for((size_t)n * 3 * h * w / 8)), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint16_t*)dpatches, grad, n, h, w, patch, (long long)ld, is);
  RART_CHECK_LAUNCH("rart_vit_unpatchify_f32");
  return RART_OK;
}

}  // extern "C"

// ---- parameter-gradient reductions for ViT training (LayerNorm gamma / beta, Linear biases, position embedding) ----
namespace {
// LayerNorm backward to the input AND per-wave partial sums of dgamma = sum_rows dy*xhat, dbeta = sum_rows dy:
// a wave walks rows (grid stride), so its 2 x 8 column slots accumulate in registers; partial[wave][2][d] fp32.
__global__ __launch_bounds__(kBlock) void k_layernorm_bwd_full(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x,
                                                               const float* __restrict__ gamma, const uint16_t* __restrict__ res,
                                                               uint16_t* __restrict__ dx, int rows, int d, long long dy_stride,
                                                               long long x_stride, long long res_stride, long long dx_stride,
                                                               float eps, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int wave_g = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), n_waves = gridDim.x * (kBlock / 64);
  const int nv = d / 8;
  float ag[2][8], ab[2][8], gm[2][8];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ag[k][j] = ab[k][j] = 0.f;
      const int vi = lane + 64 * k;
      gm[k][j] = vi < nv ? gamma[vi * 8 + j] : 0.f;
    }
  for (int row = wave_g; row < rows; row += n_waves) {
    float xr[2][8], gr[2][8];
    load_row(x + (size_t)row * x_stride, nv, lane, xr);
    load_row(dy + (size_t)row * dy_stride, nv, lane, gr);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += xr[k][j];
    const float mean = rart_wave_sum(s) / (float)d;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (lane + 64 * k < nv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = xr[k][j] - mean;
          v += t * t;
        }
      }
    const float rstd = rsqrtf(rart_wave_sum(v) / (float)d + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (lane + 64 * k < nv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xr[k][j] = (xr[k][j] - mean) * rstd;               // xhat
          ag[k][j] += gr[k][j] * xr[k][j];
          ab[k][j] += gr[k][j];
          gr[k][j] *= gm[k][j];
          sg += gr[k][j];
          sgx += gr[k][j] * xr[k][j];
        }
      }
    const float mg = rart_wave_sum(sg) / (float)d, mgx = rart_wave_sum(sgx) / (float)d;
    if (dx) {
      uint16_t* o = dx + (size_t)row * dx_stride;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int vi = lane + 64 * k;
        if (vi < nv) {
          float r[8], rs[8];
          if (res) unpack8(*reinterpret_cast<const uint4*>(res + (size_t)row * res_stride + (size_t)vi * 8), rs);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            r[j] = rstd * (gr[k][j] - mg - xr[k][j] * mgx);
            if (res) r[j] += rs[j];
          }
          *reinterpret_cast<uint4*>(o + (size_t)vi * 8) = pack8(r);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int vi = lane + 64 * k;
    if (vi < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        partial[((size_t)wave_g * 2 + 0) * d + vi * 8 + j] = ag[k][j];
        partial[((size_t)wave_g * 2 + 1) * d + vi * 8 + j] = ab[k][j];
      }
    }
  }
}
// out[s][c] (+)= sum_k partial[k][s][c], fixed order; s < n_sets
__global__ __launch_bounds__(kBlock) void k_reduce_partials(const float* __restrict__ partial, int n_part, int n_sets, int d,
                                                            float* __restrict__ out0, float* __restrict__ out1, int accumulate) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= d) return;
  for (int s = 0; s < n_sets; ++s) {
    double a = 0.0;
#pragma unroll 8
    for (int k = 0; k < n_part; ++k) a += (double)partial[((size_t)k * n_sets + s) * d + c];   // independent loads, fixed order
    float* o = s == 0 ? out0 : out1;
    if (o) o[c] = accumulate ? o[c] + (float)a : (float)a;
  }
}
// column sums of a bf16 [rows][cols] matrix (row stride ld): thread = one 8-column vector, block = a chunk of rows
__global__ __launch_bounds__(kBlock) void k_colsum_bf16(const uint16_t* __restrict__ x, long long ld, int rows, int cols,
                                                        int rows_per_chunk, float* __restrict__ partial) {
  const int v = blockIdx.x * kBlock + threadIdx.x;
  if (v * 8 >= cols) return;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.f;
  for (int r = r0; r < r1; ++r) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(x + (size_t)r * ld + (size_t)v * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += f[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) partial[(size_t)blockIdx.y * cols + v * 8 + j] = a[j];
}
}  // namespace

extern "C" {

// 512 waves walk the rows: enough to fill the GPU, few enough that the fixed-order final reduction (one thread per
// column over all partials) stays short
size_t rart_layernorm_bwd_workspace_bytes(int dim) { return (size_t)512 * 2 * dim * sizeof(float); }

int rart_layernorm_bwd_full_bf16(const void* dy, const void* x, const float* gamma, const void* res, void* dx, int rows,
                                 int dim, int64_t dy_row_stride, int64_t x_row_stride, int64_t res_row_stride,
                                 int64_t dx_row_stride, float eps, float* dgamma, float* dbeta, int accumulate, void* workspace,
                                 size_t workspace_bytes, rart_stream_t stream) {
  RART_CHECK_ARG(dy && x && gamma && dgamma && dbeta && rows > 0 && dim > 0, "rart_layernorm_bwd_full_bf16: bad arguments");
  RART_CHECK_ARG(dim % 8 == 0 && dim <= 1024 && dy_row_stride % 8 == 0 && x_row_stride % 8 == 0 && res_row_stride % 8 == 0 &&
                     dx_row_stride % 8 == 0, "rart_layernorm_bwd_full_bf16: dim a multiple of 8, at most 1024; strides multiples of 8");
  const size_t need = rart_layernorm_bwd_workspace_bytes(dim);
  if (!workspace || workspace_bytes < need) {
    rart_set_error("rart_layernorm_bwd_full_bf16: workspace of %zu bytes required", need);
    return RART_ERR_WORKSPACE;
  }
  int blocks = (rows + kBlock / 64 - 1) / (kBlock / 64);
  if (blocks > 128) blocks = 128;                                  // 512 waves
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_layernorm_bwd_full, dim3(blocks), dim3(kBlock), 0, st, (const uint16_t*)dy, (const uint16_t*)x, gamma,
                     (const uint16_t*)res, (uint16_t*)dx, rows, dim, (long long)dy_row_stride, (long long)x_row_stride,
                     (long long)res_row_stride, (long long)dx_row_stride, eps, (float*)workspace);
  hipLaunchKernelGGL(k_reduce_partials, dim3((dim + kBlock - 1) / kBlock), dim3(kBlock), 0, st, (const float*)workspace,
                     blocks * (kBlock / 64), 2, dim, dgamma, dbeta, accumulate);
  RART_CHECK_LAUNCH("rart_layernorm_bwd_full_bf16");
  return RART_OK;
}

size_t rart_colsum_workspace_bytes(int rows, int cols) {
  const int chunks = rows < 128 ? 1 : (rows / 64 > 256 ? 256 : rows / 64);   // ~64 rows per thread, at most 256 partials
  return (size_t)chunks * cols * sizeof(float);
}

int rart_colsum_bf16(const void* x, int64_t ld, int rows, int cols, float* out, int accumulate, void* workspace,
                     size_t workspace_bytes, rart_stream_t stream) {
  RART_CHECK_ARG(x && out && rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && ld >= cols, "rart_colsum_bf16: bad arguments");
  const size_t need = rart_colsum_workspace_bytes(rows, cols);
  if (!workspace || workspace_bytes < need) {
    rart_set_error("rart_colsum_bf16: workspace of %zu bytes required", need);
    return RART_ERR_WORKSPACE;
  }
  const int chunks = (int)(need / ((size_t)cols * sizeof(float)));
  const int rpc = (rows + chunks - 1) / chunks;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_colsum_bf16, dim3((cols / 8 + kBlock - 1) / kBlock, chunks), dim3(kBlock), 0, st, (const uint16_t*)x,
                     (long long)ld, rows, cols, rpc, (float*)workspace);
  hipLaunchKernelGGL(k_reduce_partials, dim3((cols + kBlock - 1) / kBlock), dim3(kBlock), 0, st, (const float*)workspace, chunks,
                     1, cols, out, nullptr, accumulate);
  RART_CHECK_LAUNCH("rart_colsum_bf16");
  return RART_OK;
}

}  // extern "C"
