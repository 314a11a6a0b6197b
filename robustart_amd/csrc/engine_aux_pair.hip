// Non-GEMM kernels of the eval-mode ResNet-50 engine in its REFERENCE-PRECISION mode (gfx950).
//
// The reference runs fp32 everywhere (RobustART/noise/utils/adv/attack.py:20-23, Attacks/autoattack/autopgd_base.py:271-289:
// fp32 logits and gradients).  The engine's "bf16x3" mode keeps every activation / gradient as a PAIR of bf16 planes,
// value = hi + lo with hi = bf16(v), lo = bf16(v - hi) -- 16 significand bits in the bytes of one fp32 -- so that the MFMA
// kernels can form x.w as x_hi.w_hi + x_hi.w_lo + x_lo.w_hi with fp32 accumulation (conv_igemm.hip, flag 32).  These are
// the pool / converter kernels on such pairs: HBM-bound, 16-byte (8-channel) vectors per lane and plane.
// hi + lo is exact in fp32 (two 8-bit significands, |lo| <= half an ulp of hi), so every kernel unpacks to fp32,
// computes as the bf16 kernels of engine_aux.hip do, and splits again.
#include "rart_common.h"

namespace {
constexpr int kBlock = 256;

__device__ __forceinline__ float bf2f(uint32_t v16) { return __uint_as_float(v16 << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);   // round to nearest even (finite inputs)
  return u >> 16;
}
// value = hi + lo of the 8 channels of one 16-byte vector per plane; lo_off in uint4 units
__device__ __forceinline__ void load_pair8(const uint4* __restrict__ hi, size_t lo_off, size_t i, float* f) {
  const uint4 a = hi[i], b = hi[i + lo_off];
  const uint32_t wa[4] = {a.x, a.y, a.z, a.w}, wb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = bf2f(wa[j] & 0xFFFFu) + bf2f(wb[j] & 0xFFFFu);
    f[2 * j + 1] = bf2f(wa[j] >> 16) + bf2f(wb[j] >> 16);
  }
}
__device__ __forceinline__ void store_pair8(uint4* __restrict__ hi, size_t lo_off, size_t i, const float* f) {
  uint32_t h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    h[j] = f2bf(f[j]);
    l[j] = f2bf(f[j] - bf2f(h[j]));
  }
  hi[i] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
  hi[i + lo_off] = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

// 3x3 stride-2 pad-1 max pool on pairs; argmax codes / sign bits exactly as k_maxpool_fwd (engine_aux.hip): first
// maximum in scan order (PyTorch's rule), code 15 = window maximum <= 0
__global__ __launch_bounds__(kBlock) void k_maxpool_fwd_pair(const uint4* __restrict__ in, size_t in_lo, uint4* __restrict__ out,
                                                             size_t out_lo, uint2* __restrict__ arg, uint8_t* __restrict__ sign,
                                                             int n, int h, int w, int c8) {
  const int oh = h / 2, ow = w / 2;
  const size_t total = (size_t)n * oh * ow * c8;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % c8);
    size_t t = i / c8;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh);
    const int img = (int)(t / oh);
    float m[8];
    uint32_t code[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; code[j] = 0; }
    for (int ky = 0; ky < 3; ++ky) {
      const int y = oy * 2 - 1 + ky;
      if ((unsigned)y >= (unsigned)h) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int x = ox * 2 - 1 + kx;
        if ((unsigned)x >= (unsigned)w) continue;
        float f[8];
        load_pair8(in, in_lo, (((size_t)img * h + y) * w + x) * c8 + c, f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (f[j] > m[j]) { m[j] = f[j]; code[j] = (uint32_t)(ky * 3 + kx); }
      }
    }
    store_pair8(out, out_lo, i, m);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!(m[j] > 0.f)) code[j] = 15u;
    if (sign) {
      uint32_t sb = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) sb |= (m[j] > 0.f ? 1u : 0u) << j;
      sign[i] = (uint8_t)sb;
    }
    if (arg)
      arg[i] = make_uint2(code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24),
                          code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24));
  }
}

// backward of the pool fused with the ReLU mask of its input: code 15 already marks windows whose maximum is <= 0 and a
// window's argmax position holds that maximum, so the codes alone decide (no read of y):
// dz[h,w,c] = sum over the <= 4 windows containing (h,w) whose recorded argmax is (h,w)
__global__ __launch_bounds__(kBlock) void k_maxpool_bwd_pair(const uint2* __restrict__ arg, const uint4* __restrict__ dpool,
                                                             size_t dpool_lo, uint4* __restrict__ dz, size_t dz_lo, int n, int h,
                                                             int w, int c8) {
  const int oh = h / 2, ow = w / 2;
  const size_t total = (size_t)n * h * w * c8;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % c8);
    size_t t = i / c8;
    const int x = (int)(t % w);
    t /= w;
    const int yy = (int)(t % h);
    const int img = (int)(t / h);
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
    const int oy0 = yy >> 1, ox0 = x >> 1;
    const int nys = (yy & 1) ? 2 : 1, nxs = (x & 1) ? 2 : 1;
    for (int a = 0; a < nys; ++a) {
      const int oy = oy0 + a;
      if (oy >= oh) continue;
      const uint32_t ky = (uint32_t)(yy - (2 * oy - 1));
      for (int b = 0; b < nxs; ++b) {
        const int ox = ox0 + b;
        if (ox >= ow) continue;
        const uint32_t mine = ky * 3 + (uint32_t)(x - (2 * ox - 1));
        const size_t wi = (((size_t)img * oh + oy) * ow + ox) * c8 + c;
        const uint2 cd = arg[wi];
        float dp[8];
        load_pair8(dpool, dpool_lo, wi, dp);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t cj = ((j < 4 ? cd.x : cd.y) >> (8 * (j & 3))) & 0xFFu;
          if (cj == mine) g[j] += dp[j];
        }
      }
    }
    store_pair8(dz, dz_lo, i, g);
  }
}

// global average pool [n][hw][c] -> [n][c], fp32 accumulation in position order
__global__ __launch_bounds__(kBlock) void k_avgpool_fwd_pair(const uint4* __restrict__ in, size_t in_lo, uint4* __restrict__ out,
                                                             size_t out_lo, int n, int hw, int c8) {
  const size_t total = (size_t)n * c8;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % c8), img = (int)(i / c8);
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    for (int p = 0; p < hw; ++p) {
      float f[8];
      load_pair8(in, in_lo, ((size_t)img * hw + p) * c8 + c, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += f[j];
    }
    const float inv = 1.0f / (float)hw;
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] *= inv;
    store_pair8(out, out_lo, i, s);
  }
}

// dz[n][p][c] = sign bit of y[n][p][c] ? dpool[n][c] / hw : 0     (sign: 1 byte per 8 channels)
__global__ __launch_bounds__(kBlock) void k_avgpool_bwd_pair(const uint8_t* __restrict__ sign, const uint4* __restrict__ dpool,
                                                             size_t dpool_lo, uint4* __restrict__ dz, size_t dz_lo, int n, int hw,
                                                             int c8) {
  const size_t total = (size_t)n * hw * c8;
  const float inv = 1.0f / (float)hw;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % c8), img = (int)(i / ((size_t)hw * c8));
    float d[8];
    load_pair8(dpool, dpool_lo, (size_t)img * c8 + c, d);
    const uint32_t sb = sign[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = ((sb >> j) & 1u) ? d[j] * inv : 0.f;
    store_pair8(dz, dz_lo, i, d);
  }
}

// fp32 rows -> zero-padded pair rows (dlogits for the fc backward)
__global__ __launch_bounds__(kBlock) void k_f32_to_pair_rows(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t dst_lo,
                                                             int rows, int cols, int dst_cols) {
  const size_t total = (size_t)rows * dst_cols;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % dst_cols);
    const size_t r = i / dst_cols;
    const float v = c < cols ? src[r * cols + c] : 0.f;
    const uint32_t h = f2bf(v);
    dst[i] = (uint16_t)h;
    dst[i + dst_lo] = (uint16_t)f2bf(v - bf2f(h));
  }
}

// Stem backward from fp32 patches [n][oh][ow][pc] (column (r*7+s)*3+c, pc = 152) -> grad fp32 NCHW; k_stem_col2im (engine_aux.hip)
// with the fp32 patch rows that reach a TH x TW pixel tile ((TH/2+3) x (TW/2+3) x 608 B) in dynamic LDS, NT threads.
constexpr int C2F_PC = 152;
template <int TH, int TW, int NT>
__global__ __launch_bounds__(NT) void k_stem_col2im_f32(const float* __restrict__ patches, float* __restrict__ grad, int n,
                                                        int h, int w, float istd0, float istd1, float istd2) {
  constexpr int PH = TH / 2 + 3, PW = TW / 2 + 3;
  extern __shared__ __attribute__((aligned(16))) float c2f_sp[];
  float* const sp = c2f_sp;
  const int oh = h / 2, ow = w / 2;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, img = blockIdx.z;
  const int p0 = y0 / 2 - 1, q0 = x0 / 2 - 1;
  constexpr int VEC = C2F_PC / 4;                                     // 38 sixteen-byte vectors per patch row
#pragma unroll 4
  for (int i = threadIdx.x; i < PH * PW * VEC; i += NT) {
    const int v = i % VEC, pos = i / VEC;
    const int pp = p0 + pos / PW, qq = q0 + pos % PW;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)pp < (unsigned)oh && (unsigned)qq < (unsigned)ow)
      val = *reinterpret_cast<const float4*>(patches + (((size_t)img * oh + pp) * ow + qq) * C2F_PC + v * 4);
    *reinterpret_cast<float4*>(sp + (size_t)pos * C2F_PC + v * 4) = val;
  }
  __syncthreads();
  const size_t plane = (size_t)h * w;
  for (int i = threadIdx.x; i < TH * TW; i += NT) {
    const int x = x0 + i % TW, y = y0 + i / TW;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int r = (y + 3) & 1; r < 7; r += 2) {
      const int pp = (y + 3 - r) >> 1;
      if (y + 3 - r < 0 || pp >= oh) continue;
      for (int sft = (x + 3) & 1; sft < 7; sft += 2) {
        const int qq = (x + 3 - sft) >> 1;
        if (x + 3 - sft < 0 || qq >= ow) continue;
        const float* pt = sp + ((size_t)(pp - p0) * PW + (qq - q0)) * C2F_PC + (r * 7 + sft) * 3;
        g0 += pt[0];
        g1 += pt[1];
        g2 += pt[2];
      }
    }
    float* o = grad + (size_t)img * 3 * plane + (size_t)y * w + x;
    o[0] = g0 * istd0;
    o[plane] = g1 * istd1;
    o[2 * plane] = g2 * istd2;
  }
}
template <int TH, int TW, int NT>
int launch_c2f(const float* patches, float* grad, int n, int h, int w, float i0, float i1, float i2, hipStream_t stream) {
  constexpr int LDS = (TH / 2 + 3) * (TW / 2 + 3) * C2F_PC * 4;
  if (h % TH || w % TW) return RART_ERR_INVALID;
  if (!rart_raise_dynamic_lds((const void*)k_stem_col2im_f32<TH, TW, NT>, LDS, "rart_engine_stem_col2im_f32")) return RART_ERR_HIP;
  hipLaunchKernelGGL((k_stem_col2im_f32<TH, TW, NT>), dim3(w / TW, h / TH, n), dim3(NT), LDS, stream, patches, grad, n, h, w, i0, i1, i2);
  return RART_OK;
}

int grid_for(size_t items) { return rart_grid_for(items, kBlock, 256 * 16); }
bool lo_ok(long long off) { return off > 0 && off % 8 == 0; }
}  // namespace

extern "C" {

int rart_engine_maxpool_pair(const void* in_hi, long long in_lo_off, void* out_hi, long long out_lo_off, void* argmax_out,
                             void* sign_out, int n, int h, int w, int c, rart_stream_t stream) {
  RART_CHECK_ARG(in_hi && out_hi && n > 0 && h % 2 == 0 && w % 2 == 0 && c % 8 == 0 && lo_ok(in_lo_off) && lo_ok(out_lo_off),
                 "rart_engine_maxpool_pair: bad arguments");
  hipLaunchKernelGGL(k_maxpool_fwd_pair, dim3(grid_for((size_t)n * (h / 2) * (w / 2) * (c / 8))), dim3(kBlock), 0,
                     (hipStream_t)stream, (const uint4*)in_hi, (size_t)in_lo_off / 8, (uint4*)out_hi, (size_t)out_lo_off / 8,
                     (uint2*)argmax_out, (uint8_t*)sign_out, n, h, w, c / 8);
  RART_CHECK_LAUNCH("rart_engine_maxpool_pair");
  return RART_OK;
}

int rart_engine_maxpool_bwd_pair(const void* argmax, const void* dpool_hi, long long dpool_lo_off, void* dz_hi,
                                 long long dz_lo_off, int n, int h, int w, int c, rart_stream_t stream) {
  RART_CHECK_ARG(argmax && dpool_hi && dz_hi && n > 0 && h % 2 == 0 && w % 2 == 0 && c % 8 == 0 && lo_ok(dpool_lo_off) &&
                     lo_ok(dz_lo_off),
                 "rart_engine_maxpool_bwd_pair: bad arguments");
  hipLaunchKernelGGL(k_maxpool_bwd_pair, dim3(grid_for((size_t)n * h * w * (c / 8))), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint2*)argmax, (const uint4*)dpool_hi, (size_t)dpool_lo_off / 8, (uint4*)dz_hi, (size_t)dz_lo_off / 8,
                     n, h, w, c / 8);
  RART_CHECK_LAUNCH("rart_engine_maxpool_bwd_pair");
  return RART_OK;
}

int rart_engine_avgpool_pair(const void* in_hi, long long in_lo_off, void* out_hi, long long out_lo_off, int n, int hw, int c,
                             rart_stream_t stream) {
  RART_CHECK_ARG(in_hi && out_hi && n > 0 && hw > 0 && c % 8 == 0 && lo_ok(in_lo_off) && lo_ok(out_lo_off),
                 "rart_engine_avgpool_pair: bad arguments");
  hipLaunchKernelGGL(k_avgpool_fwd_pair, dim3(grid_for((size_t)n * (c / 8))), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint4*)in_hi, (size_t)in_lo_off / 8, (uint4*)out_hi, (size_t)out_lo_off / 8, n, hw, c / 8);
  RART_CHECK_LAUNCH("rart_engine_avgpool_pair");
  return RART_OK;
}

int rart_engine_avgpool_bwd_pair(const void* y_sign_bits, const void* dpool_hi, long long dpool_lo_off, void* dz_hi,
                                 long long dz_lo_off, int n, int hw, int c, rart_stream_t stream) {
  RART_CHECK_ARG(y_sign_bits && dpool_hi && dz_hi && n > 0 && hw > 0 && c % 8 == 0 && lo_ok(dpool_lo_off) && lo_ok(dz_lo_off),
                 "rart_engine_avgpool_bwd_pair: bad arguments");
  hipLaunchKernelGGL(k_avgpool_bwd_pair, dim3(grid_for((size_t)n * hw * (c / 8))), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint8_t*)y_sign_bits, (const uint4*)dpool_hi, (size_t)dpool_lo_off / 8, (uint4*)dz_hi,
                     (size_t)dz_lo_off / 8, n, hw, c / 8);
  RART_CHECK_LAUNCH("rart_engine_avgpool_bwd_pair");
  return RART_OK;
}

int rart_f32_to_pair_rows(const float* src, void* dst_hi, long long dst_lo_off, int rows, int cols, int dst_cols,
                          rart_stream_t stream) {
  RART_CHECK_ARG(src && dst_hi && rows > 0 && cols > 0 && dst_cols >= cols && dst_lo_off > 0, "rart_f32_to_pair_rows: bad arguments");
  hipLaunchKernelGGL(k_f32_to_pair_rows, dim3(grid_for((size_t)rows * dst_cols)), dim3(kBlock), 0, (hipStream_t)stream, src,
                     (uint16_t*)dst_hi, (size_t)dst_lo_off, rows, cols, dst_cols);
  RART_CHECK_LAUNCH("rart_f32_to_pair_rows");
  return RART_OK;
}

int rart_engine_stem_col2im_f32(const float* patches, float* grad, int n, int h, int w, int patch_cols, const float* std_host,
                                rart_stream_t stream) {
  RART_CHECK_ARG(patches && grad && n > 0 && n <= 65535 && h % 16 == 0 && w % 32 == 0 && patch_cols == C2F_PC,
                 "rart_engine_stem_col2im_f32: h %% 16 == 0, w %% 32 == 0, patch_cols == 152, n <= 65535");
  const float i0 = std_host ? 1.0f / std_host[0] : 1.f, i1 = std_host ? 1.0f / std_host[1] : 1.f,
              i2 = std_host ? 1.0f / std_host[2] : 1.f;
  // 16 x 32 pixel tiles, 1 024 threads (measured at B = 256: 0.78 ms; 8 x 16 / 256 threads 1.29, 16 x 32 / 256 threads 1.86,
  // 8 x 32 / 512 0.84, 16 x 16 / 512 0.80: the fill phase wants many loads in flight per CU, the tile few halo re-reads)
  const int rc = launch_c2f<16, 32, 1024>(patches, grad, n, h, w, i0, i1, i2, (hipStream_t)stream);
  if (rc != RART_OK) return rc;
  RART_CHECK_LAUNCH("rart_engine_stem_col2im_f32");
  return RART_OK;
}

}  // extern "C"
