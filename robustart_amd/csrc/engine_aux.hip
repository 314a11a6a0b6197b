// Non-GEMM kernels of the eval-mode ResNet-50 engine (gfx950): input preparation (normalise +
// bf16 hi/lo split + spatial/channel padding for the 7x7 stem), 3x3/2 max-pool forward and
// backward (+ReLU mask), global average pool forward/backward, the stem's col2im (backward to the
// fp32 image) and an fp32 -> padded bf16 row converter.  All HBM-bound elementwise/gather work on
// NHWC bf16 tensors, 16-byte (8-channel) vectors per lane.
#include "rart_common.h"

namespace {
constexpr int kBlock = 256;

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ void unpack8(const uint4 v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = bf2f((uint16_t)(w[j] & 0xFFFF));
    f[2 * j + 1] = bf2f((uint16_t)(w[j] >> 16));
  }
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 o;
  o.x = f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16);
  o.y = f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16);
  o.z = f2bf(f[4]) | ((uint32_t)f2bf(f[5]) << 16);
  o.w = f2bf(f[6]) | ((uint32_t)f2bf(f[7]) << 16);
  return o;
}

struct Norm3 {
  float mean[3], istd[3];
};

#pragma clang fp contract(off)   // op-by-op rounding: the fused stem forward (stem_fused.hip) restates this arithmetic
// src: fp32 NCHW in [0,1] (SRC_U8 = false) or uint8 NHWC (SRC_U8 = true).
// dst hi/lo: bf16 [n][h+8][w+8][4], image at offset (3,3), zero elsewhere; v = (x - mean)/std,
// hi = bf16(v), lo = bf16(v - hi): hi + lo carries ~16 mantissa bits so eps-sized PGD steps survive.
template <bool SRC_U8>
__global__ __launch_bounds__(kBlock) void k_prep_input(const void* __restrict__ src, uint2* __restrict__ hi,
                                                       uint2* __restrict__ lo, int n, int h, int w, Norm3 nm) {
  const int ph = h + 8, pw = w + 8;
  const size_t total = (size_t)n * ph * pw;
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < total; p += (size_t)gridDim.x * kBlock) {
    const int px = (int)(p % pw), py = (int)((p / pw) % ph);
    const int img = (int)(p / ((size_t)pw * ph));
    const int x = px - 3, y = py - 3;
    uint16_t hv[4] = {0, 0, 0, 0}, lv[4] = {0, 0, 0, 0};
    if ((unsigned)x < (unsigned)w && (unsigned)y < (unsigned)h) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v01;
        if (SRC_U8)
          v01 = (float)((const uint8_t*)src)[(((size_t)img * h + y) * w + x) * 3 + c] * (1.0f / 255.0f);
        else
          v01 = ((const float*)src)[(((size_t)img * 3 + c) * h + y) * w + x];
        const float v = (v01 - nm.mean[c]) * nm.istd[c];
        hv[c] = f2bf(v);
        lv[c] = f2bf(v - bf2f(hv[c]));
      }
    }
    hi[p] = make_uint2(hv[0] | ((uint32_t)hv[1] << 16), hv[2] | ((uint32_t)hv[3] << 16));
    lo[p] = make_uint2(lv[0] | ((uint32_t)lv[1] << 16), lv[2] | ((uint32_t)lv[3] << 16));
  }
}

#pragma clang fp contract(fast)
// 3x3 stride-2 pad-1 max pool, NHWC bf16, one thread per (output pixel, 8 channels).  Also records, per
// channel, WHICH of the 9 window positions (ky*3+kx) holds the first maximum in scan order (PyTorch's
// argmax rule) as one byte, so the backward is a <= 4-window gather instead of a 36-tap search.
__global__ __launch_bounds__(kBlock) void k_maxpool_fwd(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                        uint2* __restrict__ arg, uint8_t* __restrict__ sign, int n, int h,
                                                        int w, int c8) {
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
        unpack8(in[(((size_t)img * h + y) * w + x) * c8 + c], f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (f[j] > m[j]) { m[j] = f[j]; code[j] = (uint32_t)(ky * 3 + kx); }
      }
    }
    out[i] = pack8(m);
    // code 15 = the window maximum is <= 0: behind a ReLU its gradient is dead, so the fused stem backward
    // (stem_fused.hip) needs no second look at y; it never equals a window position, k_maxpool_bwd is unaffected
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!(m[j] > 0.f)) code[j] = 15u;
    if (sign) {       // 1 bit per channel: pooled value > 0 (the ReLU mask of the first block's input)
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

// backward of the pool fused with the ReLU mask of its input y (= relu(stem conv)):
// dz[h,w,c] = (y > 0) * sum over the <= 4 windows containing (h,w) whose recorded argmax is (h,w).
__global__ __launch_bounds__(kBlock) void k_maxpool_bwd(const uint4* __restrict__ y, const uint2* __restrict__ arg,
                                                        const uint4* __restrict__ dpool, uint4* __restrict__ dz, int n,
                                                        int h, int w, int c8) {
  const int oh = h / 2, ow = w / 2;
  const size_t total = (size_t)n * h * w * c8;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % c8);
    size_t t = i / c8;
    const int x = (int)(t % w);
    t /= w;
    const int yy = (int)(t % h);
    const int img = (int)(t / h);
    float self[8], g[8];
    unpack8(y[i], self);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
    // windows o with 2o-1 <= pos <= 2o+1: o = pos/2 (always) and o = (pos+1)/2 when pos is odd
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
        unpack8(dpool[wi], dp);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t cj = ((j < 4 ? cd.x : cd.y) >> (8 * (j & 3))) & 0xFFu;
          if (cj == mine) g[j] += dp[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!(self[j] > 0.f)) g[j] = 0.f;
    dz[i] = pack8(g);
  }
}

// global average pool [n][hw][c] -> [n][c] (bf16), fp32 accumulation
__global__ __launch_bounds__(kBlock) void k_avgpool_fwd(const uint4* __restrict__ in, uint4* __restrict__ out, int n,
                                                        int hw, int c8) {
  const size_t total = (size_t)n * c8;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % c8), img = (int)(i / c8);
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    for (int p = 0; p < hw; ++p) {
      float f[8];
      unpack8(in[((size_t)img * hw + p) * c8 + c], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += f[j];
    }
    const float inv = 1.0f / (float)hw;
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] *= inv;
    out[i] = pack8(s);
  }
}

// dz[n][p][c] = (y > 0) ? dpool[n][c] / hw : 0
__global__ __launch_bounds__(kBlock) void k_avgpool_bwd(const uint4* __restrict__ y, const uint4* __restrict__ dpool,
                                                        uint4* __restrict__ dz, int n, int hw, int c8) {
  const size_t total = (size_t)n * hw * c8;
  const float inv = 1.0f / (float)hw;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % c8), img = (int)(i / ((size_t)hw * c8));
    float f[8], d[8];
    unpack8(y[i], f);
    unpack8(dpool[(size_t)img * c8 + c], d);
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = f[j] > 0.f ? d[j] * inv : 0.f;
    dz[i] = pack8(d);
  }
}

// Stem backward: patches[n][oh][ow][pc] (bf16, column (r*7+s)*3+c, pc >= 147) -> grad fp32 NCHW
// dx[n][c][h][w] = istd[c] * sum_{r,s: (h+3-r), (w+3-s) even, p,q in range} patches[n][p][q][(r*7+s)*3+c]
// One workgroup = a 16 x 32 tile of image pixels: the 11 x 19 patch rows that reach it (304 B each) are staged in
// LDS with coalesced 16-byte loads, then every pixel gathers its <= 16 overlapping taps from LDS (the first version
// issued 48 scattered 2-byte global loads per pixel: 0.79 ms per launch at B = 256, 6.5 % of a gradient evaluation).
constexpr int C2I_TH = 16, C2I_TW = 32, C2I_PH = C2I_TH / 2 + 3, C2I_PW = C2I_TW / 2 + 3, C2I_PC = 152;
__global__ __launch_bounds__(kBlock) void k_stem_col2im(const uint16_t* __restrict__ patches, float* __restrict__ grad,
                                                        int n, int h, int w, int pc, Norm3 nm) {
  __shared__ __attribute__((aligned(16))) uint16_t sp[C2I_PH * C2I_PW * C2I_PC];
  const int oh = h / 2, ow = w / 2;
  const int x0 = blockIdx.x * C2I_TW, y0 = blockIdx.y * C2I_TH, img = blockIdx.z;
  const int p0 = y0 / 2 - 1, q0 = x0 / 2 - 1;                       // first patch row / column that reaches the tile
  constexpr int VEC = C2I_PC / 8;                                     // 19 sixteen-byte vectors per patch row
  for (int i = threadIdx.x; i < C2I_PH * C2I_PW * VEC; i += kBlock) {
    const int v = i % VEC, pos = i / VEC;
    const int pp = p0 + pos / C2I_PW, qq = q0 + pos % C2I_PW;
    uint4 val = make_uint4(0, 0, 0, 0);
    if ((unsigned)pp < (unsigned)oh && (unsigned)qq < (unsigned)ow)
      val = *reinterpret_cast<const uint4*>(patches + (((size_t)img * oh + pp) * ow + qq) * pc + v * 8);
    *reinterpret_cast<uint4*>(sp + (size_t)pos * C2I_PC + v * 8) = val;
  }
  __syncthreads();
  const size_t plane = (size_t)h * w;
  for (int i = threadIdx.x; i < C2I_TH * C2I_TW; i += kBlock) {
    const int x = x0 + i % C2I_TW, y = y0 + i / C2I_TW;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int r = (y + 3) & 1; r < 7; r += 2) {
      const int pp = (y + 3 - r) >> 1;                                // same parity by construction; negative near the top edge
      if (y + 3 - r < 0 || pp >= oh) continue;
      for (int sft = (x + 3) & 1; sft < 7; sft += 2) {
        const int qq = (x + 3 - sft) >> 1;
        if (x + 3 - sft < 0 || qq >= ow) continue;
        const uint16_t* pt = sp + ((size_t)(pp - p0) * C2I_PW + (qq - q0)) * C2I_PC + (r * 7 + sft) * 3;
        g0 += bf2f(pt[0]);
        g1 += bf2f(pt[1]);
        g2 += bf2f(pt[2]);
      }
    }
    float* o = grad + (size_t)img * 3 * plane + (size_t)y * w + x;
    o[0] = g0 * nm.istd[0];
    o[plane] = g1 * nm.istd[1];
    o[2 * plane] = g2 * nm.istd[2];
  }
}

__global__ __launch_bounds__(kBlock) void k_f32_to_bf16_rows(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                             int rows, int cols, int dst_cols) {
  const size_t total = (size_t)rows * dst_cols;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % dst_cols);
    const size_t r = i / dst_cols;
    dst[i] = c < cols ? f2bf(src[r * cols + c]) : (uint16_t)0;
  }
}

Norm3 make_norm(const float* mean, const float* std) {
  Norm3 nm;
  for (int c = 0; c < 3; ++c) {
    nm.mean[c] = mean ? mean[c] : 0.f;
    nm.istd[c] = std ? 1.0f / std[c] : 1.f;
  }
  return nm;
}
int grid_for(size_t items) { return rart_grid_for(items, kBlock, 256 * 16); }

// ---- small-M GEMM: C[M][N] = A[M][K] . W[N][K]^T (+ bias) for the classifier head and its backward (M = the batch, 256) ------------------
// The implicit-GEMM kernel gives such a product 2 x 8 tiles of 128 x 128: 16 workgroups for 256 CUs, each walking the whole K (43 us for
// 1 GFLOP).  Here a workgroup owns ONE 32 x 32 output tile (8 x 32 = 256 of them for the head) and its four waves split K; the operands
// are used once per workgroup, so the MFMA fragments come straight from global memory (A is 1 MB, W 4 MB: L2 resident), 16 bytes per
// lane, and the four partial tiles are summed through LDS in a fixed order.
typedef __attribute__((ext_vector_type(8))) __bf16 sm_bf16x8;
typedef __attribute__((ext_vector_type(16))) float sm_f32x16;
__global__ __launch_bounds__(kBlock) void k_gemm_small_m(const uint16_t* __restrict__ a, int lda, const uint16_t* __restrict__ wgt, int ldw,
                                                         const float* __restrict__ bias, void* __restrict__ out, int ldo, int out_f32,
                                                         int M, int N, int K) {
  __shared__ float red[4][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int row_a = min(m0 + l31, M - 1), row_w = min(n0 + l31, N - 1);       // rows past the edge recompute the last one; never stored
  const int kq = K / 4;                                                        // host: K % 64 == 0
  const uint16_t* pa = a + (size_t)row_a * lda + wave * kq + hh * 8;
  const uint16_t* pw = wgt + (size_t)row_w * ldw + wave * kq + hh * 8;
  sm_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < kq; k += 16) {
    const uint4 av = *reinterpret_cast<const uint4*>(pa + k), wv = *reinterpret_cast<const uint4*>(pw + k);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sm_bf16x8, av), __builtin_bit_cast(sm_bf16x8, wv), acc, 0, 0, 0);
  }
  // acc[r] = partial C[m0 + (r&3) + 8*(r>>2) + 4*hh][n0 + l31]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = wave * 4 + q;
    const float v = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hh, n = n0 + l31;
    if (m < M && n < N) {
      const float y = v + (bias ? bias[n] : 0.f);
      if (out_f32) reinterpret_cast<float*>(out)[(size_t)m * ldo + n] = y;
      else reinterpret_cast<uint16_t*>(out)[(size_t)m * ldo + n] = f2bf(y);
    }
  }
}
}  // namespace

extern "C" {

int rart_engine_prep_input(const void* src, int src_is_u8, void* hi, void* lo, int n, int h, int w,
                           const float* mean_host, const float* std_host, rart_stream_t stream) {
  RART_CHECK_ARG(src && hi && lo && n > 0 && h > 0 && w > 0, "rart_engine_prep_input: bad arguments");
  const Norm3 nm = make_norm(mean_host, std_host);
  const size_t total = (size_t)n * (h + 8) * (w + 8);
  if (src_is_u8)
    hipLaunchKernelGGL(k_prep_input<true>, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, src, (uint2*)hi,
                       (uint2*)lo, n, h, w, nm);
  else
    hipLaunchKernelGGL(k_prep_input<false>, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, src,
                       (uint2*)hi, (uint2*)lo, n, h, w, nm);
  RART_CHECK_LAUNCH("rart_engine_prep_input");
  return RART_OK;
}

int rart_engine_maxpool_keep(const void* in, void* out, void* argmax_out, void* sign_out, int n, int h, int w, int c,
                             rart_stream_t stream) {
  RART_CHECK_ARG(in && out && n > 0 && h % 2 == 0 && w % 2 == 0 && c % 8 == 0, "rart_engine_maxpool: bad arguments");
  hipLaunchKernelGGL(k_maxpool_fwd, dim3(grid_for((size_t)n * (h / 2) * (w / 2) * (c / 8))), dim3(kBlock), 0,
                     (hipStream_t)stream, (const uint4*)in, (uint4*)out, (uint2*)argmax_out, (uint8_t*)sign_out, n, h, w,
                     c / 8);
  RART_CHECK_LAUNCH("rart_engine_maxpool");
  return RART_OK;
}

int rart_engine_maxpool(const void* in, void* out, void* argmax_out, int n, int h, int w, int c,
                        rart_stream_t stream) {
  return rart_engine_maxpool_keep(in, out, argmax_out, nullptr, n, h, w, c, stream);
}

int rart_engine_maxpool_bwd(const void* y, const void* argmax, const void* dpool, void* dz, int n, int h, int w, int c,
                            rart_stream_t stream) {
  RART_CHECK_ARG(y && argmax && dpool && dz && n > 0 && h % 2 == 0 && w % 2 == 0 && c % 8 == 0,
                 "rart_engine_maxpool_bwd: bad arguments");
  hipLaunchKernelGGL(k_maxpool_bwd, dim3(grid_for((size_t)n * h * w * (c / 8))), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint4*)y, (const uint2*)argmax, (const uint4*)dpool, (uint4*)dz, n, h, w, c / 8);
  RART_CHECK_LAUNCH("rart_engine_maxpool_bwd");
  return RART_OK;
}

int rart_engine_avgpool(const void* in, void* out, int n, int hw, int c, rart_stream_t stream) {
  RART_CHECK_ARG(in && out && n > 0 && hw > 0 && c % 8 == 0, "rart_engine_avgpool: bad arguments");
  hipLaunchKernelGGL(k_avgpool_fwd, dim3(grid_for((size_t)n * (c / 8))), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint4*)in, (uint4*)out, n, hw, c / 8);
  RART_CHECK_LAUNCH("rart_engine_avgpool");
  return RART_OK;
}

int rart_engine_avgpool_bwd(const void* y, const void* dpool, void* dz, int n, int hw, int c, rart_stream_t stream) {
  RART_CHECK_ARG(y && dpool && dz && n > 0 && hw > 0 && c % 8 == 0, "rart_engine_avgpool_bwd: bad arguments");
  hipLaunchKernelGGL(k_avgpool_bwd, dim3(grid_for((size_t)n * hw * (c / 8))), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint4*)y, (const uint4*)dpool, (uint4*)dz, n, hw, c / 8);
  RART_CHECK_LAUNCH("rart_engine_avgpool_bwd");
  return RART_OK;
}

int rart_engine_stem_col2im(const void* patches, float* grad, int n, int h, int w, int patch_cols,
                            const float* std_host, rart_stream_t stream) {
  RART_CHECK_ARG(patches && grad && n > 0 && n <= 65535 && h % 16 == 0 && w % 32 == 0 && patch_cols == 152,
                 "rart_engine_stem_col2im: h %% 16 == 0, w %% 32 == 0, patch_cols == 152 (147 rounded up to 8), n <= 65535");
  const Norm3 nm = make_norm(nullptr, std_host);
  hipLaunchKernelGGL(k_stem_col2im, dim3(w / C2I_TW, h / C2I_TH, n), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint16_t*)patches, grad, n, h, w, patch_cols, nm);
  RART_CHECK_LAUNCH("rart_engine_stem_col2im");
  return RART_OK;
}

int rart_f32_to_bf16_rows(const float* src, void* dst, int rows, int cols, int dst_cols, rart_stream_t stream) {
  RART_CHECK_ARG(src && dst && rows > 0 && cols > 0 && dst_cols >= cols, "rart_f32_to_bf16_rows: bad arguments");
  hipLaunchKernelGGL(k_f32_to_bf16_rows, dim3(grid_for((size_t)rows * dst_cols)), dim3(kBlock), 0, (hipStream_t)stream,
                     src, (uint16_t*)dst, rows, cols, dst_cols);
  RART_CHECK_LAUNCH("rart_f32_to_bf16_rows");
  return RART_OK;
}

int rart_gemm_small_m_bf16(const void* a, int lda, const void* w, int ldw, const float* bias, void* out, int ldo, int out_is_f32, int m,
                           int n, int k, rart_stream_t stream) {
  RART_CHECK_ARG(a && w && out && m > 0 && n > 0 && k > 0 && k % 64 == 0 && lda >= k && ldw >= k && lda % 8 == 0 && ldw % 8 == 0 && ldo >= n,
                 "rart_gemm_small_m_bf16: k must be a multiple of 64, leading dimensions multiples of 8 and at least k (a, w) / n (out)");
  RART_CHECK_ARG((m + 31) / 32 <= 65535, "rart_gemm_small_m_bf16: m must stay below 2^21 rows (this is the small-M kernel)");
  hipLaunchKernelGGL(k_gemm_small_m, dim3((n + 31) / 32, (m + 31) / 32), dim3(kBlock), 0, (hipStream_t)stream, (const uint16_t*)a, lda,
                     (const uint16_t*)w, ldw, bias, out, ldo, out_is_f32, m, n, k);
  RART_CHECK_LAUNCH("rart_gemm_small_m_bf16");
  return RART_OK;
}

}  // extern "C"
