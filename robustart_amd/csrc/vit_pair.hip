// Non-GEMM kernels of the REFERENCE-PRECISION ("fp32x" / "bf16x3") ViT-B/16 engine for gfx950: every activation and gradient is a
// PAIR of bf16 planes (value = hi + lo, 16 significand bits; exact in fp32), every contraction runs on rart_gemm_pair_bf16
// (csrc/gemm_pair.hip), and what is left -- class-token / position add, LayerNorm forward and backward-to-input, the attention
// soft-max rows and their backward -- are the HBM-bound row kernels below: read the pair, compute in fp32 exactly as the fp32 module
// does (two-pass LayerNorm statistics, max-subtracted soft-max with libm's expf), write the pair.  The reference runs ViT in fp32
// (exprs/exp/imagenet_c_loop_mini/config_vit_base.yaml:1-9: no precision key; adv/attack.py:20-23; autopgd_base.py:271-289); model
// `vit_base` = timm ViT-B/16 (RobustART/model/__init__.py:1 -> absent submodule; robustart_amd/model/vit_torch.py).
#include "rart_common.h"

namespace {
constexpr int kBlock = 256;

__device__ __forceinline__ uint32_t pk2(float lo, float hi) {   // round to nearest even (v_cvt_pk_bf16_f32)
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
  f2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2_t));
}
__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = pk2(v[2 * j], v[2 * j + 1]);
    l[j] = pk2(v[2 * j] - __uint_as_float(h[j] << 16), v[2 * j + 1] - __uint_as_float(h[j] & 0xFFFF0000u));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void join8(const uint4& hi, const uint4& lo, float* v) {
  const uint32_t h[4] = {hi.x, hi.y, hi.z, hi.w}, l[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[2 * j] = __uint_as_float(h[j] << 16) + __uint_as_float(l[j] << 16);
    v[2 * j + 1] = __uint_as_float(h[j] & 0xFFFF0000u) + __uint_as_float(l[j] & 0xFFFF0000u);
  }
}
// one wave per row, the row in registers as 16-byte vectors (lane l holds vectors l and l + 64): rows of up to 1024 elements
__device__ __forceinline__ void load_row_pair(const uint16_t* hi, const uint16_t* lo, int nv, int lane, float r[2][8]) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int v = lane + 64 * k;
    uint4 qh = make_uint4(0, 0, 0, 0), ql = make_uint4(0, 0, 0, 0);
    if (v < nv) {
      qh = *reinterpret_cast<const uint4*>(hi + (size_t)v * 8);
      ql = *reinterpret_cast<const uint4*>(lo + (size_t)v * 8);
    }
    join8(qh, ql, r[k]);
  }
}

// x[b][0][:] = cls_pos0; x[b][t][:] += pos[t] (t >= 1); eight channels per thread (d % 8 == 0)
__global__ __launch_bounds__(kBlock) void k_add_pos_cls_pair(uint16_t* __restrict__ xh, uint16_t* __restrict__ xl,
                                                             const float* __restrict__ cls_pos0, const float* __restrict__ pos, int n,
                                                             int t, int d) {
  const uint32_t d8 = (uint32_t)d / 8, total = (uint32_t)n * t * d8;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
    const uint32_t c8 = i % d8, tok = (i / d8) % (uint32_t)t;
    const float* add = tok == 0 ? cls_pos0 + c8 * 8 : pos + (size_t)tok * d + c8 * 8;
    const float4 a0 = reinterpret_cast<const float4*>(add)[0], a1 = reinterpret_cast<const float4*>(add)[1];
    float v[8];
    join8(reinterpret_cast<const uint4*>(xh)[i], reinterpret_cast<const uint4*>(xl)[i], v);
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (tok == 0 ? 0.f : v[j]) + av[j];
    uint4 oh, ol;
    split8(v, oh, ol);
    reinterpret_cast<uint4*>(xh)[i] = oh;
    reinterpret_cast<uint4*>(xl)[i] = ol;
  }
}

// LayerNorm over the last dim, fp32 two-pass statistics (torch's layer_norm: biased variance, eps inside the root)
__global__ __launch_bounds__(kBlock) void k_layernorm_pair(const uint16_t* __restrict__ xh, const uint16_t* __restrict__ xl,
                                                           const float* __restrict__ g, const float* __restrict__ b,
                                                           uint16_t* __restrict__ oh, uint16_t* __restrict__ ol, int rows, int d,
                                                           long long in_stride, long long out_stride, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = d / 8;
  float xr[2][8];
  load_row_pair(xh + (size_t)row * in_stride, xl + (size_t)row * in_stride, nv, lane, xr);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += xr[k][j];                   // padding vectors are zero
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
  const float rstd = 1.0f / sqrtf(rart_wave_sum(v) / (float)d + eps);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int vi = lane + 64 * k;
    if (vi < nv) {
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (xr[k][j] - mean) * rstd * g[vi * 8 + j] + b[vi * 8 + j];
      uint4 ph, pl;
      split8(r, ph, pl);
      *reinterpret_cast<uint4*>(oh + (size_t)row * out_stride + (size_t)vi * 8) = ph;
      *reinterpret_cast<uint4*>(ol + (size_t)row * out_stride + (size_t)vi * 8) = pl;
    }
  }
}

// LayerNorm backward to the input: xhat = (x - mean) rstd; g = dy gamma; dx = rstd (g - mean(g) - xhat mean(g xhat)) [+ res]
__global__ __launch_bounds__(kBlock) void k_layernorm_bwd_pair(const uint16_t* __restrict__ dyh, const uint16_t* __restrict__ dyl,
                                                               const uint16_t* __restrict__ xh, const uint16_t* __restrict__ xl,
                                                               const float* __restrict__ gamma, const uint16_t* __restrict__ rh,
                                                               const uint16_t* __restrict__ rl, uint16_t* __restrict__ oh,
                                                               uint16_t* __restrict__ ol, int rows, int d, long long dy_stride,
                                                               long long x_stride, long long res_stride, long long dx_stride, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = d / 8;
  float xr[2][8], gr[2][8];
  load_row_pair(xh + (size_t)row * x_stride, xl + (size_t)row * x_stride, nv, lane, xr);
  load_row_pair(dyh + (size_t)row * dy_stride, dyl + (size_t)row * dy_stride, nv, lane, gr);
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
  const float rstd = 1.0f / sqrtf(rart_wave_sum(v) / (float)d + eps);
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
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int vi = lane + 64 * k;
    if (vi < nv) {
      float r[8], rs[8];
      if (rh) join8(*reinterpret_cast<const uint4*>(rh + (size_t)row * res_stride + (size_t)vi * 8),
                    *reinterpret_cast<const uint4*>(rl + (size_t)row * res_stride + (size_t)vi * 8), rs);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        r[j] = rstd * (gr[k][j] - mg - xr[k][j] * mgx);
        if (rh) r[j] += rs[j];
      }
      uint4 ph, pl;
      split8(r, ph, pl);
      *reinterpret_cast<uint4*>(oh + (size_t)row * dx_stride + (size_t)vi * 8) = ph;
      *reinterpret_cast<uint4*>(ol + (size_t)row * dx_stride + (size_t)vi * 8) = pl;
    }
  }
}

__device__ __forceinline__ void split4(const float* v, uint2& hi, uint2& lo) {
  const uint32_t h0 = pk2(v[0], v[1]), h1 = pk2(v[2], v[3]);
  hi = make_uint2(h0, h1);
  lo = make_uint2(pk2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xFFFF0000u)),
                  pk2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xFFFF0000u)));
}

// P[row][0..n_valid) = softmax(scale * S[row][0..n_valid)) as a pair, P[row][n_valid..ld_out) = 0; S fp32 (the pair GEMM's fp32
// output).  One wave per row, four consecutive columns per lane (rows of up to 256 columns)
__global__ __launch_bounds__(kBlock) void k_softmax_rows_pair(const float* __restrict__ sm, uint16_t* __restrict__ ph,
                                                              uint16_t* __restrict__ pl, long long rows, int n_valid, int ld_in,
                                                              int ld_out, float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c0 = lane * 4;
  float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c0 < ld_in) sv = *reinterpret_cast<const float4*>(sm + row * ld_in + c0);
  float s[4] = {sv.x, sv.y, sv.z, sv.w};
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (c0 + j < n_valid) mx = fmaxf(mx, s[j] * scale);
  mx = rart_wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s[j] = (c0 + j < n_valid) ? expf(s[j] * scale - mx) : 0.f;
    sum += s[j];
  }
  sum = rart_wave_sum(sum);
  if (c0 < ld_out) {
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = s[j] / sum;
    uint2 oh, ol;
    split4(s, oh, ol);
    *reinterpret_cast<uint2*>(ph + row * ld_out + c0) = oh;
    *reinterpret_cast<uint2*>(pl + row * ld_out + c0) = ol;
  }
}

// dS[row][c] = scale * P[row][c] * (dP[row][c] - sum_k P[row][k] dP[row][k]), c < n_valid; zeros up to ld_out.  P pair, dP fp32.
__global__ __launch_bounds__(kBlock) void k_softmax_bwd_rows_pair(const uint16_t* __restrict__ ph, const uint16_t* __restrict__ pl,
                                                                  const float* __restrict__ dp, uint16_t* __restrict__ dh,
                                                                  uint16_t* __restrict__ dl, long long rows, int n_valid, int ld_p,
                                                                  int ld_dp, int ld_out, float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c0 = lane * 4;
  float p[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f};
  if (c0 < ld_p) {
    const uint2 h = *reinterpret_cast<const uint2*>(ph + row * ld_p + c0), l = *reinterpret_cast<const uint2*>(pl + row * ld_p + c0);
    p[0] = __uint_as_float(h.x << 16) + __uint_as_float(l.x << 16);
    p[1] = __uint_as_float(h.x & 0xFFFF0000u) + __uint_as_float(l.x & 0xFFFF0000u);
    p[2] = __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16);
    p[3] = __uint_as_float(h.y & 0xFFFF0000u) + __uint_as_float(l.y & 0xFFFF0000u);
  }
  if (c0 < ld_dp) {
    const float4 gv = *reinterpret_cast<const float4*>(dp + row * ld_dp + c0);
    g[0] = gv.x; g[1] = gv.y; g[2] = gv.z; g[3] = gv.w;
  }
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (c0 + j < n_valid) dot += p[j] * g[j];
  dot = rart_wave_sum(dot);
  if (c0 < ld_out) {
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (c0 + j < n_valid) ? scale * p[j] * (g[j] - dot) : 0.f;
    uint2 oh, ol;
    split4(r, oh, ol);
    *reinterpret_cast<uint2*>(dh + row * ld_out + c0) = oh;
    *reinterpret_cast<uint2*>(dl + row * ld_out + c0) = ol;
  }
}

// grad[b][c][y][x] = dpatch[b][patch(y, x)][c*ps*ps + (y % ps)*ps + (x % ps)] * istd[c]   (fp32 in, fp32 NCHW out)
struct Istd3 { float v[3]; };
__global__ __launch_bounds__(kBlock) void k_unpatchify_from_f32(const float* __restrict__ dp, float* __restrict__ grad, int n, int h,
                                                                int w, int ps, long long ld, Istd3 is) {
  const uint32_t gw = (uint32_t)(w / ps), gh = (uint32_t)(h / ps), w4 = (uint32_t)w / 4;
  const uint32_t total = (uint32_t)n * 3u * (uint32_t)h * w4;              // host: < 2^32
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
    const uint32_t x4 = i % w4, t = i / w4, y = t % (uint32_t)h, t2 = t / (uint32_t)h, c = t2 % 3u, img = t2 / 3u;
    const uint32_t x = x4 * 4, px = x / (uint32_t)ps, py = y / (uint32_t)ps;
    const size_t prow = (size_t)img * gh * gw + (size_t)py * gw + px;
    const float4 v = *reinterpret_cast<const float4*>(dp + prow * ld + (size_t)c * ps * ps + (y % (uint32_t)ps) * ps + (x % (uint32_t)ps));
    const float sc = is.v[c];
    reinterpret_cast<float4*>(grad)[i] = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
  }
}
int grid_for(size_t items) { return rart_grid_for(items, kBlock, 256 * 16); }
}  // namespace

extern "C" {

int rart_vit_add_pos_cls_pair(void* x_hi, void* x_lo, const float* cls_pos0, const float* pos, int n, int tokens, int dim,
                              rart_stream_t stream) {
  RART_CHECK_ARG(x_hi && x_lo && cls_pos0 && pos && n > 0 && tokens > 0 && dim > 0 && dim % 8 == 0 &&
                     (size_t)n * tokens * dim / 8 < (1ull << 32), "rart_vit_add_pos_cls_pair: bad arguments");
  hipLaunchKernelGGL(k_add_pos_cls_pair, dim3(grid_for((size_t)n * tokens * dim / 8)), dim3(kBlock), 0, (hipStream_t)stream,
                     (uint16_t*)x_hi, (uint16_t*)x_lo, cls_pos0, pos, n, tokens, dim);
  RART_CHECK_LAUNCH("rart_vit_add_pos_cls_pair");
  return RART_OK;
}

int rart_layernorm_pair(const void* x_hi, const void* x_lo, const float* gamma, const float* beta, void* out_hi, void* out_lo, int rows,
                        int dim, int64_t in_row_stride, int64_t out_row_stride, float eps, rart_stream_t stream) {
  RART_CHECK_ARG(x_hi && x_lo && gamma && beta && out_hi && out_lo && rows > 0 && dim > 0, "rart_layernorm_pair: bad arguments");
  RART_CHECK_ARG(dim % 8 == 0 && dim <= 1024 && in_row_stride % 8 == 0 && out_row_stride % 8 == 0,
                 "rart_layernorm_pair: dim a multiple of 8, at most 1024; strides multiples of 8");
  hipLaunchKernelGGL(k_layernorm_pair, dim3((rows + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint16_t*)x_hi, (const uint16_t*)x_lo, gamma, beta, (uint16_t*)out_hi, (uint16_t*)out_lo, rows, dim,
                     (long long)in_row_stride, (long long)out_row_stride, eps);
  RART_CHECK_LAUNCH("rart_layernorm_pair");
  return RART_OK;
}

int rart_layernorm_bwd_pair(const void* dy_hi, const void* dy_lo, const void* x_hi, const void* x_lo, const float* gamma,
                            const void* res_hi, const void* res_lo, void* dx_hi, void* dx_lo, int rows, int dim, int64_t dy_row_stride,
                            int64_t x_row_stride, int64_t res_row_stride, int64_t dx_row_stride, float eps, rart_stream_t stream) {
  RART_CHECK_ARG(dy_hi && dy_lo && x_hi && x_lo && gamma && dx_hi && dx_lo && rows > 0 && dim > 0 && ((res_hi == nullptr) == (res_lo == nullptr)),
                 "rart_layernorm_bwd_pair: bad arguments");
  RART_CHECK_ARG(dim % 8 == 0 && dim <= 1024 && dy_row_stride % 8 == 0 && x_row_stride % 8 == 0 && res_row_stride % 8 == 0 &&
                     dx_row_stride % 8 == 0, "rart_layernorm_bwd_pair: dim a multiple of 8, at most 1024; strides multiples of 8");
  hipLaunchKernelGGL(k_layernorm_bwd_pair, dim3((rows + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint16_t*)dy_hi, (const uint16_t*)dy_lo, (const uint16_t*)x_hi, (const uint16_t*)x_lo, gamma,
                     (const uint16_t*)res_hi, (const uint16_t*)res_lo, (uint16_t*)dx_hi, (uint16_t*)dx_lo, rows, dim,
                     (long long)dy_row_stride, (long long)x_row_stride, (long long)res_row_stride, (long long)dx_row_stride, eps);
  RART_CHECK_LAUNCH("rart_layernorm_bwd_pair");
  return RART_OK;
}

int rart_softmax_rows_pair(const float* scores, void* probs_hi, void* probs_lo, int64_t rows, int n_valid, int ld_in, int ld_out,
                           float scale, rart_stream_t stream) {
  RART_CHECK_ARG(scores && probs_hi && probs_lo && rows > 0 && n_valid > 0 && ld_in >= n_valid && ld_out >= n_valid,
                 "rart_softmax_rows_pair: bad arguments");
  RART_CHECK_ARG(ld_in % 4 == 0 && ld_out % 4 == 0 && ld_in <= 256 && ld_out <= 256,
                 "rart_softmax_rows_pair: leading dimensions must be multiples of 4, at most 256");
  hipLaunchKernelGGL(k_softmax_rows_pair, dim3((uint32_t)((rows + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                     (hipStream_t)stream, scores, (uint16_t*)probs_hi, (uint16_t*)probs_lo, (long long)rows, n_valid, ld_in, ld_out, scale);
  RART_CHECK_LAUNCH("rart_softmax_rows_pair");
  return RART_OK;
}

int rart_softmax_bwd_rows_pair(const void* probs_hi, const void* probs_lo, const float* dprobs, void* ds_hi, void* ds_lo, int64_t rows,
                               int n_valid, int ld_p, int ld_dp, int ld_out, float scale, rart_stream_t stream) {
  RART_CHECK_ARG(probs_hi && probs_lo && dprobs && ds_hi && ds_lo && rows > 0 && n_valid > 0 && ld_p >= n_valid && ld_dp >= n_valid &&
                     ld_out >= n_valid, "rart_softmax_bwd_rows_pair: bad arguments");
  RART_CHECK_ARG(ld_p % 4 == 0 && ld_dp % 4 == 0 && ld_out % 4 == 0 && ld_p <= 256 && ld_dp <= 256 && ld_out <= 256,
                 "rart_softmax_bwd_rows_pair: leading dimensions must be multiples of 4, at most 256");
  hipLaunchKernelGGL(k_softmax_bwd_rows_pair, dim3((uint32_t)((rows + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                     (hipStream_t)stream, (const uint16_t*)probs_hi, (const uint16_t*)probs_lo, dprobs, (uint16_t*)ds_hi, (uint16_t*)ds_lo,
                     (long long)rows, n_valid, ld_p, ld_dp, ld_out, scale);
  RART_CHECK_LAUNCH("rart_softmax_bwd_rows_pair");
  return RART_OK;
}

int rart_vit_unpatchify_from_f32(const float* dpatches, float* grad, int n, int h, int w, int patch, int64_t ld, const float* std_host,
                                 rart_stream_t stream) {
  RART_CHECK_ARG(dpatches && grad && std_host && n > 0 && patch > 0 && h % patch == 0 && w % patch == 0 && ld >= 3 * patch * patch,
                 "rart_vit_unpatchify_from_f32: bad arguments");
  RART_CHECK_ARG(patch % 4 == 0 && ld % 4 == 0 && (size_t)n * 3 * h * w / 4 < (1ull << 32),
                 "rart_vit_unpatchify_from_f32: patch side and row stride must be multiples of 4 (four pixels per thread)");
  Istd3 is;
  for (int c = 0; c < 3; ++c) is.v[c] = 1.0f / std_host[c];
  hipLaunchKernelGGL(k_unpatchify_from_f32, dim3(grid_for((size_t)n * 3 * h * w / 4)), dim3(kBlock), 0, (hipStream_t)stream, dpatches,
                     grad, n, h, w, patch, (long long)ld, is);
  RART_CHECK_LAUNCH("rart_vit_unpatchify_from_f32");
  return RART_OK;
}

}  // extern "C"
