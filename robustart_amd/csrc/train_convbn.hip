// Train-mode conv + BatchNorm support kernels for gfx950 (ResNet-50 training step of cls_solver; SURVEY.md 8a M2,
// BASELINE config 5).  The dense contractions stay on rart_conv_igemm_bf16; this file holds the HBM-bound
// pieces around it:
//   * batch statistics of a conv output z[M][C] (bf16 NHWC): two-level deterministic column sums, finalised
//     into mean / invstd / fused scale+shift and the running-statistics update (nn.BatchNorm2d, momentum 0.1,
//     unbiased running variance);
//   * y = relu(z*scale + shift + residual);
//   * BatchNorm backward: g = dy * [y > 0]; reductions sum(g), sum(g*xhat); dz = gamma*invstd*(g - mean(g) -
//     xhat*mean(g*xhat)); dgamma = sum(g*xhat), dbeta = sum(g) written straight into the gradient arena;
//   * weight gradient as a split-K GEMM on the igemm kernel: both operands are needed K(=pixel)-contiguous, so
//     dz and the (implicit) im2col matrix are transposed by `rart_transpose_gather_bf16`
//     (out[(tap, c)][m] = x[pixel(m) + tap][c]), the igemm writes fp32 partials per K split, and
//     `rart_wgrad_reduce_f32` sums the splits in fixed order into the torch weight layout [N][C][R][S];
//   * packing of the fp32 master weights into the bf16 forward / backward-to-input tables of the igemm.
// Reference arithmetic: torch.nn.BatchNorm2d / conv2d autograd (the reference trains with PyTorch:
// cifar10/code/train.py:96-127; model RobustART/model/__init__.py:1 -> public ResNet-50).
#include "rart_common.h"
#include <cstdlib>

namespace {
constexpr int kBlock = 256;

__device__ __forceinline__ float bf2f(uint32_t v16) { return __uint_as_float(v16 << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
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
  uint4 o;
  o.x = f2bf(f[0]) | (f2bf(f[1]) << 16);
  o.y = f2bf(f[2]) | (f2bf(f[3]) << 16);
  o.z = f2bf(f[4]) | (f2bf(f[5]) << 16);
  o.w = f2bf(f[6]) | (f2bf(f[7]) << 16);
  return o;
}
// 8 per-channel "keep" flags from a post-ReLU activation vector: bits > 0 as int16
__device__ __forceinline__ void keep8(const uint4& v, bool* k) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    k[2 * j] = (short)(w[j] & 0xFFFFu) > 0;
    k[2 * j + 1] = (short)(w[j] >> 16) > 0;
  }
}

// ... or from the 1-bit form (bit j of the byte = channel j of the group > 0), written by k_bn_apply
__device__ __forceinline__ void keep8_bits(uint32_t byte, bool* k) {
#pragma unroll
  for (int j = 0; j < 8; ++j) k[j] = (byte >> j) & 1u;
}

// ---- two-level column reductions over rows of a [M][C] bf16 matrix ------------------------------------------
// MODE 0: a = sum z, b = sum z^2.   MODE 1: g = dy*[ymask>0]; a = sum g, b = sum g*xhat  (xhat from z, mean, invstd)
// Block = 256 threads = (C/8 channel groups) x (2048/C row lanes); partial[chunk][0/1][C] fp32.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_colsum2(const uint4* __restrict__ z, const uint4* __restrict__ dy,
                                                     const uint4* __restrict__ ymask, const uint8_t* __restrict__ ybits,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd, size_t M, int C,
                                                     size_t rows_per_chunk, float* __restrict__ partial) {
  __shared__ float sh[2][2048];
  const int c8n = C / 8;
  const int rows_par = kBlock / c8n;
  const int tid = threadIdx.x, cg = tid % c8n, rl = tid / c8n;
  const size_t r0 = (size_t)blockIdx.x * rows_per_chunk;
  const size_t r1 = r0 + rows_per_chunk < M ? r0 + rows_per_chunk : M;
  float a[8], b[8], mu[8], is[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = b[j] = 0.f;
    mu[j] = MODE == 1 ? mean[cg * 8 + j] : 0.f;
    is[j] = MODE == 1 ? invstd[cg * 8 + j] : 0.f;
  }
  for (size_t r = r0 + rl; r < r1; r += rows_par) {
    const size_t idx = r * c8n + cg;
    float zf[8];
    unpack8(z[idx], zf);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[j] += zf[j];
        b[j] = fmaf(zf[j], zf[j], b[j]);
      }
    } else {
      float gf[8];
      unpack8(dy[idx], gf);
      bool kp[8];
      if (ybits) keep8_bits(ybits[idx], kp);
      else if (ymask) keep8(ymask[idx], kp);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = ((!ymask && !ybits) || kp[j]) ? gf[j] : 0.f;
        a[j] += g;
        b[j] = fmaf(g, (zf[j] - mu[j]) * is[j], b[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sh[0][rl * C + cg * 8 + j] = a[j];
    sh[1][rl * C + cg * 8 + j] = b[j];
  }
  __syncthreads();
  for (int c = tid; c < C; c += kBlock) {
    float sa = 0.f, sb = 0.f;
    for (int r = 0; r < rows_par; ++r) {
      sa += sh[0][r * C + c];
      sb += sh[1][r * C + c];
    }
    partial[((size_t)blockIdx.x * 2 + 0) * C + c] = sa;
    partial[((size_t)blockIdx.x * 2 + 1) * C + c] = sb;
  }
}

// Finalisers: FIN_CH channels x FIN_LANES chunk lanes per 256-thread block; fixed summation order -> deterministic.  A single thread walking
// 2 048 partials was a 2 048-deep chain of dependent-latency loads (~0.5 ms per layer); 16 lanes per channel still left 128 loads per lane in
// 32 dependent batches -- 14.5 us per launch, 106 launches per training step; 64 lanes per channel (4 channels per block, C / 4 blocks) leave
// eight batches.
constexpr int FIN_LANES = 64, FIN_CH = 256 / FIN_LANES;
__device__ __forceinline__ void reduce_partials(const float* __restrict__ partial, int chunks, int C, int c, int lane,
                                                double (*sh)[FIN_LANES][FIN_CH], double& s, double& q) {
  double a = 0.0, b = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int k = lane; k < chunks; k += FIN_LANES) {
      a += (double)partial[((size_t)k * 2 + 0) * C + c];
      b += (double)partial[((size_t)k * 2 + 1) * C + c];
    }
  }
  const int cl = threadIdx.x % FIN_CH;
  sh[0][lane][cl] = a;
  sh[1][lane][cl] = b;
  __syncthreads();
  s = q = 0.0;
  if (lane == 0) {
    for (int k = 0; k < FIN_LANES; ++k) {
      s += sh[0][k][cl];
      q += sh[1][k][cl];
    }
  }
}

__global__ __launch_bounds__(256) void k_bn_finalize_fwd(const float* __restrict__ partial, int chunks, int C, double inv_m,
                                                         double unbias, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float momentum,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                                         float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                         float* __restrict__ scale_out, float* __restrict__ shift_out) {
  __shared__ double sh[2][FIN_LANES][FIN_CH];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH), lane16 = threadIdx.x / FIN_CH;
  double s, q;
  reduce_partials(partial, chunks, C, c, lane16, sh, s, q);
  if (lane16 != 0 || c >= C) return;
  const double mu = s * inv_m;
  double var = q * inv_m - mu * mu;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * is;
  mean_out[c] = (float)mu;
  invstd_out[c] = is;
  scale_out[c] = sc;
  shift_out[c] = beta[c] - (float)mu * sc;
  if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mu;
  if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(var * unbias);
}

// backward finalise: dgamma / dbeta into the gradient arena, k1 = mean(g), k2 = mean(g*xhat), sg = gamma*invstd
__global__ __launch_bounds__(256) void k_bn_finalize_bwd(const float* __restrict__ partial, int chunks, int C, double inv_m,
                                                         const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                         float* __restrict__ coef /* [3][C]: k1, k2, gamma*invstd */) {
  __shared__ double sh[2][FIN_LANES][FIN_CH];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH), lane16 = threadIdx.x / FIN_CH;
  double s, q;
  reduce_partials(partial, chunks, C, c, lane16, sh, s, q);
  if (lane16 != 0 || c >= C) return;
  if (accumulate) {
    dbeta[c] += (float)s;
    dgamma[c] += (float)q;
  } else {
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
  }
  coef[c] = (float)(s * inv_m);
  coef[C + c] = (float)(q * inv_m);
  coef[2 * C + c] = gamma[c] * invstd[c];
}

// The grid stride (gridDim.x * 256 vectors) is a multiple of the channel-group count (a power of two <= 256), so a
// thread meets the same 8 channels in every iteration: its per-channel constants are loaded once, outside the loop.
// round 6: the BatchNorm apply kernels' streams are non-temporal accesses (tensors of 100-800 MB read or written once per kernel):
// adv_train 4 992 -> 5 069 images/s in a same-box A/B of lab builds (the implicit GEMM's bf16 output store likewise: 5 002 -> 5 077)
typedef __attribute__((ext_vector_type(4))) uint32_t bn_u4;
__device__ __forceinline__ uint4 bn_ld(const uint4* p) { const bn_u4 v = __builtin_nontemporal_load(reinterpret_cast<const bn_u4*>(p)); return make_uint4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void bn_st(uint4* p, const uint4& v) { bn_u4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<bn_u4*>(p)); }
__global__ __launch_bounds__(kBlock) void k_bn_apply(const uint4* __restrict__ z, const uint4* __restrict__ res,
                                                     uint4* __restrict__ y, uint8_t* __restrict__ sign, size_t n8, int c8n,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     int relu) {
  const size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const int cg = (int)(i0 & (size_t)(c8n - 1));
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[cg * 8 + j];
    sh[j] = shift[cg * 8 + j];
  }
  for (size_t i = i0; i < n8; i += (size_t)gridDim.x * kBlock) {
    float zf[8], rf[8];
    unpack8(bn_ld(z + i), zf);
    if (res) unpack8(bn_ld(res + i), rf);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = fmaf(zf[j], sc[j], sh[j]);
      if (res) v += rf[j];
      if (relu) v = fmaxf(v, 0.f);
      zf[j] = v;
    }
    const uint4 o = pack8(zf);
    bn_st(y + i, o);
    if (sign) {             // (stored value > 0), one byte per 8 channels: the backward's ReLU mask at 1/16 of y's bytes
      bool kp[8];
      keep8(o, kp);
      uint32_t sb = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) sb |= (kp[j] ? 1u : 0u) << j;
      sign[i] = (uint8_t)sb;
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_bn_bwd_apply(const uint4* __restrict__ dy, const uint4* __restrict__ ymask,
                                                         const uint8_t* __restrict__ ybits,
                                                         const uint4* __restrict__ z, uint4* __restrict__ dz,
                                                         uint4* __restrict__ g_out, size_t n8, int c8n, int C,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ coef) {
  const size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const int cg = (int)(i0 & (size_t)(c8n - 1));
  float mu[8], is[8], k1[8], k2[8], sg[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    mu[j] = mean[c];
    is[j] = invstd[c];
    k1[j] = coef[c];
    k2[j] = coef[C + c];
    sg[j] = coef[2 * C + c];
  }
  for (size_t i = i0; i < n8; i += (size_t)gridDim.x * kBlock) {
    float gf[8], zf[8];
    unpack8(bn_ld(dy + i), gf);
    unpack8(bn_ld(z + i), zf);
    bool kp[8];
    if (ybits) keep8_bits(ybits[i], kp);
    else if (ymask) keep8(ymask[i], kp);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g = ((!ymask && !ybits) || kp[j]) ? gf[j] : 0.f;
      gf[j] = g;
      const float xh = (zf[j] - mu[j]) * is[j];
      zf[j] = sg[j] * (g - k1[j] - xh * k2[j]);
    }
    bn_st(dz + i, pack8(zf));
    if (g_out) bn_st(g_out + i, pack8(gf));
  }
}

// ---- transposed (im2col) copy: out[(t*C + c)][m] = x[img(m), oy*sy + dy_t, ox*sx + dx_t, c]; zeros outside the
//      image and for m >= M.  64 pixels x 64 channels per workgroup through LDS.
struct GatherArgs {
  int batch, src_h, src_w, C, grid_h, grid_w, sy, sx, n_taps;
  int tap_dy[49], tap_dx[49];
  long long M, M_pad;
  int chunk, rows_total;   // output layout [M_pad / chunk][rows_total][chunk]: one compact slab per K split
};
__global__ __launch_bounds__(kBlock) void k_transpose_gather(const uint16_t* __restrict__ x, uint16_t* __restrict__ out,
                                                             const GatherArgs a) {
  __shared__ uint16_t tile[64][72];
  const int tid = threadIdx.x;
  const long long m0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64, t = blockIdx.z;
  const int dy = a.tap_dy[t], dx = a.tap_dx[t];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = tid / 8 + 32 * i, ch = tid % 8;
    const long long m = m0 + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m < a.M && c0 + ch * 8 < a.C) {
      const int ox = (int)(m % a.grid_w);
      const long long q = m / a.grid_w;
      const int oy = (int)(q % a.grid_h), img = (int)(q / a.grid_h);
      const int iy = oy * a.sy + dy, ix = ox * a.sx + dx;
      if ((unsigned)iy < (unsigned)a.src_h && (unsigned)ix < (unsigned)a.src_w)
        v = *reinterpret_cast<const uint4*>(x + (((size_t)img * a.src_h + iy) * a.src_w + ix) * a.C + c0 + ch * 8);
    }
    *reinterpret_cast<uint4*>(&tile[r][ch * 8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid / 8 + 32 * i, mch = tid % 8;
    if (c0 + c >= a.C) continue;
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      w[j] = (uint32_t)tile[mch * 8 + 2 * j][c] | ((uint32_t)tile[mch * 8 + 2 * j + 1][c] << 16);
    const long long slab = m0 / a.chunk, mk = m0 - slab * a.chunk;        // 64-pixel tiles never straddle a chunk
    *reinterpret_cast<uint4*>(out + ((size_t)slab * a.rows_total + (size_t)t * a.C + c0 + c) * a.chunk + mk + mch * 8) =
        make_uint4(w[0], w[1], w[2], w[3]);
  }
}
// C == 4 (the stem's padded hi plane): one thread per (tap, pixel), four output rows
__global__ __launch_bounds__(kBlock) void k_transpose_gather_c4(const uint2* __restrict__ x, uint16_t* __restrict__ out,
                                                                const GatherArgs a) {
  const long long m = (long long)blockIdx.x * kBlock + threadIdx.x;
  const int t = blockIdx.y;
  if (m >= a.M_pad) return;
  uint2 v = make_uint2(0, 0);
  if (m < a.M) {
    const int ox = (int)(m % a.grid_w);
    const long long q = m / a.grid_w;
    const int oy = (int)(q % a.grid_h), img = (int)(q / a.grid_h);
    const int iy = oy * a.sy + a.tap_dy[t], ix = ox * a.sx + a.tap_dx[t];
    if ((unsigned)iy < (unsigned)a.src_h && (unsigned)ix < (unsigned)a.src_w)
      v = x[((size_t)img * a.src_h + iy) * a.src_w + ix];
  }
  const long long slab = m / a.chunk, mk = m - slab * a.chunk;
  uint16_t* o = out + ((size_t)slab * a.rows_total + (size_t)t * 4) * a.chunk + mk;
  o[0] = (uint16_t)(v.x & 0xFFFF);
  o[a.chunk] = (uint16_t)(v.x >> 16);
  o[2 * (size_t)a.chunk] = (uint16_t)(v.y & 0xFFFF);
  o[3 * (size_t)a.chunk] = (uint16_t)(v.y >> 16);
}

// ---- split-K reduce: grad[n][c][t] (torch [N][C][R][S], t = r*S + s) (+)= sum_z partial[z][(t*Cp + c)][n]
// One workgroup = 32 output channels n x (up to 32 input channels c) x all taps: partial rows are read 128 B at a
// time along n (coalesced), the sums are transposed through LDS, and each n writes its contiguous (c, t) run.
// Layers with a tiny weight tensor run up to 256 K splits (the GEMM needs the parallelism); those are first folded
// 16:1 by a fully parallel streaming kernel (in place, into the first slot of each group), fixed order.
constexpr int kWgNT = 32, kWgRow = 296, kFold = 16;
__global__ __launch_bounds__(kBlock) void k_wgrad_fold(float* __restrict__ partial, int splits, size_t zstride) {
  const int groups = (splits + kFold - 1) / kFold;
  const size_t total = (size_t)groups * zstride;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int gidx = (int)(i / zstride);
    const size_t e = i - (size_t)gidx * zstride;
    float* p = partial + (size_t)gidx * kFold * zstride + e;
    const int cnt = splits - gidx * kFold < kFold ? splits - gidx * kFold : kFold;
    float s = 0.f;
#pragma unroll 4
    for (int j = 0; j < cnt; ++j) s += p[(size_t)j * zstride];
    p[0] = s;
  }
}
__global__ __launch_bounds__(kBlock) void k_wgrad_reduce(const float* __restrict__ partial, int splits, int zstep, int taps,
                                                         int C, int Cp, int N, int ldn, int ct, float* __restrict__ grad,
                                                         int accumulate) {
  __shared__ float tile[kWgNT][kWgRow + 1];
  const int tid = threadIdx.x, nl = tid & 31, cl = tid >> 5;          // 32 n lanes x 8 c lanes
  const int n0 = blockIdx.x * kWgNT, c0 = blockIdx.y * ct;
  const int cn = C - c0 < ct ? C - c0 : ct;                            // channels in this tile
  const size_t zstride = (size_t)taps * Cp * ldn * zstep;
  const int n = n0 + nl;
  for (int j = cl; j < cn * taps; j += 8) {       // j = c * taps + t: the position inside n's contiguous run
    const int c = j / taps, t = j - c * taps;
    float s = 0.f;
    if (n < N) {
      const float* p = partial + ((size_t)t * Cp + c0 + c) * ldn + n;
      for (int z = 0; z < splits; ++z) s += p[(size_t)z * zstride];
    }
    tile[nl][j] = s;
  }
  __syncthreads();
  const int run = cn * taps;                                           // contiguous floats per n
  for (int r = 0; r < kWgNT; ++r) {
    if (n0 + r >= N) break;
    float* g = grad + ((size_t)(n0 + r) * C + c0) * taps;
    for (int i = tid; i < run; i += kBlock) g[i] = accumulate ? g[i] + tile[r][i] : tile[r][i];
  }
}

// ---- fp32 master weights [N][C][R][S] -> bf16 igemm tables
//   transpose == 0: out[n][k], k = ti*C + c          (forward; rows padded to rows_pad with zeros)
//   transpose == 1: out[c][k], k = ti*N + n          (backward to input)
// tap list: (r, s) pairs, ti enumerates them.
struct PackArgs {
  int N, C, R, S, n_taps, transpose, rows_pad;
  int tap_r[49], tap_s[49];
};
__global__ __launch_bounds__(kBlock) void k_pack_weight(const float* __restrict__ w, const float* __restrict__ row_scale,
                                                        uint16_t* __restrict__ out, const PackArgs a) {
  const int inner = a.transpose ? a.N : a.C;
  const size_t K = (size_t)a.n_taps * inner;
  const size_t total = (size_t)a.rows_pad * K;
  const int rows = a.transpose ? a.C : a.N;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int row = (int)(i / K);
    const size_t k = i % K;
    const int ti = (int)(k / inner), in = (int)(k % inner);
    float v = 0.f;
    if (row < rows) {
      const int n = a.transpose ? in : row, c = a.transpose ? row : in;
      v = w[(((size_t)n * a.C + c) * a.R + a.tap_r[ti]) * a.S + a.tap_s[ti]];
      if (row_scale) v *= row_scale[n];      // eval-mode BatchNorm folded into the conv: gamma * rsqrt(var + eps)
    }
    out[i] = (uint16_t)f2bf(v);
  }
}

// Many small table jobs in ONE launch (blockIdx.y = job): the adversarial-training step re-packs every conv table of the train engine
// and re-folds + re-packs every table of the attack engine after each optimizer step -- ~500 launches of 2-10 us kernels whose dispatch
// gaps cost more than their work.  The job list lives in device memory (built once: the tensors are persistent).
// kind 0 = k_pack_weight's job (fp32 [N][C][R][S] -> bf16 igemm table, optional per-row scale), kind 1 = the fragment re-order of
// conv3x3_halo.hip's k_pack_frag (w [rows][k] row-major -> fragment-major 16-byte chunks).
struct PackJobDev {
  int kind, N, C, R, S, n_taps, transpose, rows_pad, rows, k;
  int tap_r[16], tap_s[16];
  const float* w;
  const float* row_scale;
  const uint16_t* src16;
  uint16_t* out;
};
__global__ __launch_bounds__(kBlock) void k_pack_jobs(const PackJobDev* __restrict__ jobs) {
  const PackJobDev& a = jobs[blockIdx.y];
  if (a.kind == 0) {
    const int inner = a.transpose ? a.N : a.C;
    const size_t K = (size_t)a.n_taps * inner;
    const size_t total = (size_t)a.rows_pad * K;
    const int rows = a.transpose ? a.C : a.N;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
      const int row = (int)(i / K);
      const size_t k = i % K;
      const int ti = (int)(k / inner), in = (int)(k % inner);
      float v = 0.f;
      if (row < rows) {
        const int n = a.transpose ? in : row, c = a.transpose ? row : in;
        v = a.w[(((size_t)n * a.C + c) * a.R + a.tap_r[ti]) * a.S + a.tap_s[ti]];
        if (a.row_scale) v *= a.row_scale[n];
      }
      a.out[i] = (uint16_t)f2bf(v);
    }
  } else {
    const int chunks = a.rows * a.k / 8, nwn = a.rows / 32;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < chunks; i += gridDim.x * kBlock) {
      const int l = i & 63, ks = (i >> 6) & 3, f = i >> 8, wn = f % nwn, st = f / nwn;
      *reinterpret_cast<uint4*>(a.out + (size_t)i * 8) =
          *reinterpret_cast<const uint4*>(a.src16 + (size_t)(wn * 32 + (l & 31)) * a.k + st * 64 + ks * 16 + (l >> 5) * 8);
    }
  }
}

inline unsigned grid_for(size_t n) {
  size_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > 256u * 32u) b = 256u * 32u;
  return (unsigned)b;
}
inline int chunks_for(size_t M, int C, size_t* rows_per_chunk) {
  // ~1024 workgroups (4 per CU), each at least four passes of its row lanes
  const size_t rows_par = (size_t)(kBlock / (C / 8));
  size_t chunks = 1024;      // (2 048 measured 0.35 ms per adv_train step slower: the finalisers read every partial row)
  size_t rpc = (M + chunks - 1) / chunks;
  if (rpc < rows_par * 4) rpc = rows_par * 4;
  chunks = (M + rpc - 1) / rpc;
  *rows_per_chunk = rpc;
  return (int)chunks;
}
}  // namespace

extern "C" size_t rart_bn_workspace_bytes(size_t rows, int channels) {
  if (channels < 8 || channels % 8 != 0 || channels > 2048) return 0;
  size_t rpc;
  const int chunks = chunks_for(rows, channels, &rpc);
  return (size_t)chunks * 2 * channels * sizeof(float);
}

// [chunks][2][C] -> [ceil(chunks / fold)][2][C]: groups of `fold` partial rows summed in fixed order (the convolution's per-tile statistics
// of a large layer are thousands of rows; the finaliser walks at most 1 024)
__global__ __launch_bounds__(kBlock) void k_stats_fold(const float* __restrict__ in, float* __restrict__ out, int chunks, int C2, int fold) {
  const int groups = (chunks + fold - 1) / fold;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)groups * C2; i += (size_t)gridDim.x * kBlock) {
    const int gidx = (int)(i / C2);
    const int e = (int)(i - (size_t)gidx * C2);
    const int cnt = chunks - gidx * fold < fold ? chunks - gidx * fold : fold;
    float s = 0.f;
#pragma unroll 4
    for (int k = 0; k < cnt; ++k) s += in[((size_t)gidx * fold + k) * C2 + e];
    out[i] = s;
  }
}

extern "C" int rart_bn_train_forward_bf16(const void* z, const void* res, void* y, void* sign_out, size_t rows, int channels,
                                          const float* gamma, const float* beta, float* running_mean, float* running_var,
                                          double momentum, double eps, int relu, float* mean_out, float* invstd_out,
                                          float* scale_shift /* [2][channels] */, const float* stats_partial, int stats_chunks,
                                          void* workspace, size_t workspace_bytes, rart_stream_t stream) {
  RART_CHECK_ARG(z && y && gamma && beta && mean_out && invstd_out && scale_shift && rows > 1,
                 "rart_bn_train_forward_bf16: null pointer or fewer than 2 rows");
  RART_CHECK_ARG(channels >= 8 && channels % 8 == 0 && channels <= 2048 && 2048 % channels == 0,
                 "rart_bn_train_forward_bf16: channels must be a power of two in [8, 2048]");
  const size_t need = rart_bn_workspace_bytes(rows, channels);
  if (!workspace || workspace_bytes < need) {
    rart_set_error("rart_bn_train_forward_bf16: workspace of %zu bytes required", need);
    return RART_ERR_WORKSPACE;
  }
  size_t rpc;
  const int chunks = chunks_for(rows, channels, &rpc);
  hipStream_t st = (hipStream_t)stream;
  const float* part = (const float*)workspace;
  int n_part = chunks;
  RART_CHECK_ARG(!stats_partial || stats_chunks >= 1, "rart_bn_train_forward_bf16: stats_chunks must be >= 1");
  // more tiles than the finaliser should walk: fold them `fold`:1 into the workspace first (as many slots as it holds, at most 1 024)
  const size_t slots = workspace_bytes / ((size_t)2 * channels * sizeof(float));
  const int keep = (int)(slots < 1024 ? slots : 1024);
  const bool use_partial = stats_partial && (stats_chunks <= 1024 || keep >= 64);
  if (use_partial) {          // the producing convolution's per-tile sums (rart_conv_desc.bn_stats_out): no pass over z
    part = stats_partial;
    n_part = stats_chunks;
    if (n_part > 1024) {
      const int fold = (n_part + keep - 1) / keep;
      const int groups = (n_part + fold - 1) / fold;
      hipLaunchKernelGGL(k_stats_fold, dim3(grid_for((size_t)groups * 2 * channels)), dim3(kBlock), 0, st, stats_partial,
                         (float*)workspace, n_part, 2 * channels, fold);
      part = (const float*)workspace;
      n_part = groups;
    }
  } else {
    hipLaunchKernelGGL(k_colsum2<0>, dim3(chunks), dim3(kBlock), 0, st, (const uint4*)z, nullptr, nullptr, nullptr, nullptr, nullptr,
                       rows, channels, rpc, (float*)workspace);
  }
  hipLaunchKernelGGL(k_bn_finalize_fwd, dim3((channels + FIN_CH - 1) / FIN_CH), dim3(256), 0, st, part, n_part,
                     channels, 1.0 / (double)rows, (double)rows / (double)(rows - 1), gamma, beta, (float)eps,
                     (float)momentum, running_mean, running_var, mean_out, invstd_out, scale_shift,
                     scale_shift + channels);
  const size_t n8 = rows * (size_t)(channels / 8);
  hipLaunchKernelGGL(k_bn_apply, dim3(grid_for(n8)), dim3(kBlock), 0, st, (const uint4*)z, (const uint4*)res, (uint4*)y,
                     (uint8_t*)sign_out, n8, channels / 8, scale_shift, scale_shift + channels, relu);
  RART_CHECK_LAUNCH("rart_bn_train_forward_bf16");
  return RART_OK;
}

extern "C" int rart_bn_train_backward_bf16(const void* dy, const void* ymask, int ymask_is_bits, const void* z, void* dz, void* g_out,
                                           size_t rows, int channels, const float* gamma, const float* mean,
                                           const float* invstd, float* dgamma, float* dbeta, int accumulate,
                                           float* coef /* [3][channels] scratch */, void* workspace,
                                           size_t workspace_bytes, rart_stream_t stream) {
  RART_CHECK_ARG(dy && z && dz && gamma && mean && invstd && dgamma && dbeta && coef && rows > 0,
                 "rart_bn_train_backward_bf16: null pointer");
  RART_CHECK_ARG(channels >= 8 && channels % 8 == 0 && channels <= 2048 && 2048 % channels == 0,
                 "rart_bn_train_backward_bf16: channels must be a power of two in [8, 2048]");
  const size_t need = rart_bn_workspace_bytes(rows, channels);
  if (!workspace || workspace_bytes < need) {
    rart_set_error("rart_bn_train_backward_bf16: workspace of %zu bytes required", need);
    return RART_ERR_WORKSPACE;
  }
  size_t rpc;
  const int chunks = chunks_for(rows, channels, &rpc);
  hipStream_t st = (hipStream_t)stream;
  const uint4* ym = ymask_is_bits ? nullptr : (const uint4*)ymask;
  const uint8_t* yb = ymask_is_bits ? (const uint8_t*)ymask : nullptr;
  hipLaunchKernelGGL(k_colsum2<1>, dim3(chunks), dim3(kBlock), 0, st, (const uint4*)z, (const uint4*)dy, ym, yb, mean, invstd, rows,
                     channels, rpc, (float*)workspace);
  hipLaunchKernelGGL(k_bn_finalize_bwd, dim3((channels + FIN_CH - 1) / FIN_CH), dim3(256), 0, st, (const float*)workspace, chunks,
                     channels, 1.0 / (double)rows, gamma, invstd, dgamma, dbeta, accumulate, coef);
  const size_t n8 = rows * (size_t)(channels / 8);
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(grid_for(n8)), dim3(kBlock), 0, st, (const uint4*)dy, ym, yb, (const uint4*)z, (uint4*)dz,
                     (uint4*)g_out, n8, channels / 8, channels, mean, invstd, coef);
  RART_CHECK_LAUNCH("rart_bn_train_backward_bf16");
  return RART_OK;
}

extern "C" int rart_transpose_gather_bf16(const void* src, void* dst, int batch, int src_h, int src_w, int channels,
                                          int grid_h, int grid_w, int sy, int sx, int n_taps, const int* tap_dy,
                                          const int* tap_dx, long long rows_padded, int chunk, int rows_total,
                                          rart_stream_t stream) {
  RART_CHECK_ARG(src && dst && batch > 0 && grid_h > 0 && grid_w > 0 && tap_dy && tap_dx,
                 "rart_transpose_gather_bf16: bad arguments");
  RART_CHECK_ARG(n_taps >= 1 && n_taps <= 49, "rart_transpose_gather_bf16: 1..49 taps");
  RART_CHECK_ARG(channels == 4 || channels % 8 == 0, "rart_transpose_gather_bf16: channels must be 4 or a multiple of 8");
  GatherArgs a;
  a.batch = batch; a.src_h = src_h; a.src_w = src_w; a.C = channels; a.grid_h = grid_h; a.grid_w = grid_w;
  a.sy = sy; a.sx = sx; a.n_taps = n_taps;
  for (int i = 0; i < n_taps; ++i) { a.tap_dy[i] = tap_dy[i]; a.tap_dx[i] = tap_dx[i]; }
  a.M = (long long)batch * grid_h * grid_w;
  a.M_pad = rows_padded;
  RART_CHECK_ARG(a.M_pad >= a.M && a.M_pad % 64 == 0, "rart_transpose_gather_bf16: rows_padded must be >= rows and a multiple of 64");
  a.chunk = chunk > 0 ? chunk : (int)a.M_pad;
  a.rows_total = rows_total > 0 ? rows_total : n_taps * channels;
  RART_CHECK_ARG(a.chunk % 64 == 0 && a.M_pad % a.chunk == 0 && a.rows_total >= n_taps * channels,
                 "rart_transpose_gather_bf16: chunk must be a multiple of 64 dividing rows_padded; rows_total >= taps*channels");
  hipStream_t st = (hipStream_t)stream;
  if (channels == 4)
    hipLaunchKernelGGL(k_transpose_gather_c4, dim3((unsigned)((a.M_pad + kBlock - 1) / kBlock), n_taps), dim3(kBlock), 0, st,
                       (const uint2*)src, (uint16_t*)dst, a);
  else
    hipLaunchKernelGGL(k_transpose_gather, dim3((unsigned)(a.M_pad / 64), (channels + 63) / 64, n_taps), dim3(kBlock), 0, st,
                       (const uint16_t*)src, (uint16_t*)dst, a);
  RART_CHECK_LAUNCH("rart_transpose_gather_bf16");
  return RART_OK;
}

extern "C" int rart_wgrad_reduce_f32(float* partial, int splits, int taps, int channels, int channels_padded,
                                     int n_out, int ld_n, float* grad, int accumulate, rart_stream_t stream) {
  RART_CHECK_ARG(partial && grad && splits >= 1 && taps >= 1 && channels >= 1 && channels_padded >= channels &&
                     n_out >= 1 && ld_n >= n_out, "rart_wgrad_reduce_f32: bad arguments");
  RART_CHECK_ARG(taps <= kWgRow, "rart_wgrad_reduce_f32: too many taps");
  // channel-tile width: as wide as the LDS row allows for big tensors, narrower for small ones so that at least
  // ~512 workgroups share the (latency-bound) partial reads
  int ct = kWgRow / taps;
  if (ct > 32) ct = 32;
  const long long want = (long long)channels * ((n_out + kWgNT - 1) / kWgNT) / 512;
  if (ct > want) ct = (int)(want < 1 ? 1 : want);
  if (ct > channels) ct = channels;
  int zstep = 1;
  if (splits > kFold) {   // in place: the caller's partial buffer is scratch
    const size_t zs = (size_t)taps * channels_padded * ld_n;
    const int groups = (splits + kFold - 1) / kFold;
    hipLaunchKernelGGL(k_wgrad_fold, dim3(grid_for((size_t)groups * zs)), dim3(kBlock), 0, (hipStream_t)stream,
                       partial, splits, zs);
    splits = groups;
    zstep = kFold;
  }
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((n_out + kWgNT - 1) / kWgNT, (channels + ct - 1) / ct), dim3(kBlock), 0,
                     (hipStream_t)stream, partial, splits, zstep, taps, channels, channels_padded, n_out, ld_n, ct, grad,
                     accumulate);
  RART_CHECK_LAUNCH("rart_wgrad_reduce_f32");
  return RART_OK;
}

extern "C" int rart_pack_conv_weight_bf16(const float* weight, const float* out_channel_scale, void* out, int n_out,
                                          int channels, int r, int s, int n_taps, const int* tap_r, const int* tap_s,
                                          int transpose, int rows_padded, rart_stream_t stream) {
  RART_CHECK_ARG(weight && out && n_out > 0 && channels > 0 && r > 0 && s > 0 && tap_r && tap_s,
                 "rart_pack_conv_weight_bf16: bad arguments");
  RART_CHECK_ARG(n_taps >= 1 && n_taps <= 49, "rart_pack_conv_weight_bf16: 1..49 taps");
  RART_CHECK_ARG(rows_padded >= (transpose ? channels : n_out), "rart_pack_conv_weight_bf16: rows_padded too small");
  PackArgs a;
  a.N = n_out; a.C = channels; a.R = r; a.S = s; a.n_taps = n_taps; a.transpose = transpose; a.rows_pad = rows_padded;
  for (int i = 0; i < n_taps; ++i) {
    RART_CHECK_ARG(tap_r[i] >= 0 && tap_r[i] < r && tap_s[i] >= 0 && tap_s[i] < s, "rart_pack_conv_weight_bf16: tap outside the filter");
    a.tap_r[i] = tap_r[i]; a.tap_s[i] = tap_s[i];
  }
  const size_t total = (size_t)rows_padded * n_taps * (transpose ? n_out : channels);
  hipLaunchKernelGGL(k_pack_weight, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, weight, out_channel_scale,
                     (uint16_t*)out, a);
  RART_CHECK_LAUNCH("rart_pack_conv_weight_bf16");
  return RART_OK;
}


// The jobs of rart_pack_conv_weight_bf16 (kind 0) / rart_pack_frag_bf16 (kind 1) for MANY tables in one launch.  jobs: DEVICE array of
// rart_pack_job (the caller builds it once -- every pointer in it must stay valid); blocks_per_job workgroups walk each job grid-strided.
extern "C" int rart_pack_jobs_bf16(const rart_pack_job* jobs_device, int n_jobs, int blocks_per_job, rart_stream_t stream) {
  static_assert(sizeof(rart_pack_job) == sizeof(PackJobDev), "rart_pack_job and its device mirror must have one layout");
  RART_CHECK_ARG(jobs_device && n_jobs > 0 && n_jobs <= 65535 && blocks_per_job > 0 && blocks_per_job <= 4096, "rart_pack_jobs_bf16: bad arguments");
  hipLaunchKernelGGL(k_pack_jobs, dim3((unsigned)blocks_per_job, (unsigned)n_jobs), dim3(kBlock), 0, (hipStream_t)stream,
                     reinterpret_cast<const PackJobDev*>(jobs_device));
  RART_CHECK_LAUNCH("rart_pack_jobs_bf16");
  return RART_OK;
}
