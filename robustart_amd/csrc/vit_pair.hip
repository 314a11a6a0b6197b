// Non-GEMM kernels of the REFERENCE-PRECISION ("fp32x" / "bf16x3") ViT-B/16 engine for gfx950: every activation and gradient is a
// PAIR of bf16 planes (value = hi + lo, 16 significand bits; exact in fp32), every contraction runs on rart_gemm_pair_bf16
// (csrc/gemm_pair.hip), and what is left -- class-token / position add, LayerNorm forward and backward-to-input, the attention
// soft-max rows and their backward -- are the HBM-bound row kernels below: read the pair, compute in fp32 exactly as the fp32 module
// does (two-pass LayerNorm statistics, max-subtracted soft-max with libm's expf), write the pair.  The reference runs ViT in fp32
// (exprs/exp/imagenet_c_loop_mini/config_vit_base.yaml:1-9: no precision key; adv/attack.py:20-23; autopgd_base.py:271-289); model
// `vit_base` = timm ViT-B/16 (RobustART/model/__init__.py:1 -> absent submodule; robustart_amd/model/vit_torch.py).
#include "rart_common.h"
#include <stdlib.h>
#include "rart_lds_dma.h"

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
template <bool NT = false>
__device__ __forceinline__ void load_row_pair(const uint16_t* hi, const uint16_t* lo, int nv, int lane, float r[2][8]) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int v = lane + 64 * k;
    uint4 qh = make_uint4(0, 0, 0, 0), ql = make_uint4(0, 0, 0, 0);
    if (v < nv) {
      qh = NT ? rart_nt_load16(hi + (size_t)v * 8) : *reinterpret_cast<const uint4*>(hi + (size_t)v * 8);
      ql = NT ? rart_nt_load16(lo + (size_t)v * 8) : *reinterpret_cast<const uint4*>(lo + (size_t)v * 8);
    }
    join8(qh, ql, r[k]);
  }
}
#define RART_LNB_LOAD load_row_pair<true>      // the LayerNorm backward reads its inputs for the last time: non-temporal (65.63 -> 65.34 ms)

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
  RART_LNB_LOAD(xh + (size_t)row * x_stride, xl + (size_t)row * x_stride, nv, lane, xr);
  RART_LNB_LOAD(dyh + (size_t)row * dy_stride, dyl + (size_t)row * dy_stride, nv, lane, gr);
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

// ---- fused multi-head attention on pairs (head_dim 64, tokens <= NKT*32), one workgroup per (image, head) -------------------------
// The structure of k_vit_attention (csrc/vit_aux.hip) with every contraction as three MFMA products: K_hi / K_lo rows and V_hi / V_lo
// transposed are resident in LDS (122 KB at 224 tokens: one workgroup per CU), a wave owns 32 queries; S^T = K Q^T lands with lane
// (l & 31) = the QUERY, so the soft-max statistics of a query live in one lane pair and the whole score row stays in registers (fp32, as
// the fp32 module computes it); the un-normalised probabilities e = exp(s - max) are split into hi + lo and fed straight back as the
// B operand of O^T = V^T P^T in the key order the accumulator registers already have; O is scaled by 1 / sum and written as a pair.
// Replaces, per layer and forward pass, the batched S = Q K^T product (fp32 scores, 0.5 GB at B = 256), the soft-max row kernel, two
// V transposes and the batched P V product of the unfused path (ViTEngine.fused_attention = False keeps that path as the cross-check).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int PATT_HD = 64, PATT_LDK = PATT_HD + 8;
// Round 6: EIGHT waves per workgroup (512 threads) instead of four.  The LDS images keep these kernels at one workgroup per CU, so with four
// waves every SIMD held ONE wave -- nothing ran beside its soft-max VALU work, its LDS gathers or its K / V staging, the matrix pipes were
// busy 12-13 % of a launch (profiles/r05_x3_counters.json) -- and the seven 32-query tiles of ViT-B/16's 197 tokens took two rounds over
// four waves.  With eight waves the seven tiles run in one round, two waves per SIMD, and one wave's MFMAs cover the other's vector work.
constexpr int kAttBlock = 512;

// (Round 5: the persistent form of k_vit_attention -- next item's K / V / Q of both planes prefetched into registers -- does not fit here: 112 score
//  registers + 32 output + 64-96 of prefetch exceed the 256 a 7-wave workgroup has (98-132 VGPRs spilled); a split commit -- K after S = K Q^T, V at the
//  item boundary, the next Q into the registers the current Q released -- was written too and spilled 162: the allocator keeps the staged planes and the
//  score registers apart for the whole item.  What is left for this kernel (5 % of a reference-precision ViT-B/16 gradient evaluation) is a prefetch
//  through LDS, i.e. K / V tiles small enough for two buffers.)
#ifdef RART_ATT_STAMPS
__device__ unsigned long long g_att_stamps[4096][8][4];       // lab build, per workgroup and wave: [0] staging, [1] Q load + S, [2] soft-max + PV, [3] stores, [4] wave-tiles, [5] workgroups, [6] whole
#define ATT_T(V) const unsigned long long V = __builtin_amdgcn_s_memtime();
#else
#define ATT_T(V)
#endif
template <int NKT>
__global__ __launch_bounds__(kAttBlock, 1) void k_vit_attention_pair(const uint16_t* __restrict__ qkv_h, const uint16_t* __restrict__ qkv_l,
                                                                  uint16_t* __restrict__ att_h, uint16_t* __restrict__ att_l, int T, int H,
                                                                  int ld, int D, float scale_log2e) {
  constexpr int TP = NKT * 32, LDV = TP + 4;     // 228-element rows: conflict-free 8-byte reads across 32 lanes
  __shared__ __attribute__((aligned(16))) uint16_t sK[2][TP * PATT_LDK];
  __shared__ __attribute__((aligned(16))) uint16_t sVt[2][PATT_HD * LDV];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
  const size_t boff = (size_t)b * T * ld + h * PATT_HD;
  const uint16_t* const base[2] = {qkv_h + boff, qkv_l + boff};
  ATT_T(t_a)
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    for (int i = tid; i < TP * 8; i += kAttBlock) {                       // K rows: 16-byte chunks, coalesced
      const int t = i >> 3, c = i & 7;
      uint4 kv = make_uint4(0, 0, 0, 0);
      if (t < T) kv = *reinterpret_cast<const uint4*>(base[p] + (size_t)t * ld + D + c * 8);
      *reinterpret_cast<uint4*>(&sK[p][t * PATT_LDK + c * 8]) = kv;
    }
    // V transposed: a thread takes 8 channels of FOUR consecutive tokens and writes eight 8-byte runs (one per channel)
    for (int i = tid; i < (TP / 4) * 8; i += kAttBlock) {
      const int tq = i % (TP / 4), c = i / (TP / 4);
      uint32_t w[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        uint4 vv = make_uint4(0, 0, 0, 0);
        if (tq * 4 + u < T) vv = *reinterpret_cast<const uint4*>(base[p] + (size_t)(tq * 4 + u) * ld + 2 * D + c * 8);
        w[u][0] = vv.x; w[u][1] = vv.y; w[u][2] = vv.z; w[u][3] = vv.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint16_t* row = &sVt[p][(c * 8 + 2 * j) * LDV + tq * 4];
        *reinterpret_cast<uint2*>(row) = make_uint2(__builtin_amdgcn_perm(w[1][j], w[0][j], 0x05040100u),
                                                    __builtin_amdgcn_perm(w[3][j], w[2][j], 0x05040100u));
        *reinterpret_cast<uint2*>(row + LDV) = make_uint2(__builtin_amdgcn_perm(w[1][j], w[0][j], 0x07060302u),
                                                          __builtin_amdgcn_perm(w[3][j], w[2][j], 0x07060302u));
      }
    }
  }
  __syncthreads();
  ATT_T(t_b)
#ifdef RART_ATT_STAMPS
#endif
  for (int qt = wave; qt < NKT; qt += kAttBlock / 64) {
    ATT_T(t_c)
    const int q = qt * 32 + l31;
    bf16x8 bqh[4], bql[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      uint4 vh = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
      if (q < T) {
        vh = *reinterpret_cast<const uint4*>(base[0] + (size_t)q * ld + kb * 16 + hh * 8);
        vl = *reinterpret_cast<const uint4*>(base[1] + (size_t)q * ld + kb * 16 + hh * 8);
      }
      bqh[kb] = *reinterpret_cast<bf16x8*>(&vh);
      bql[kb] = *reinterpret_cast<bf16x8*>(&vl);
    }
    f32x16 sacc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sK[0][(kt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8]);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sK[1][(kt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8]);
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bqh[kb], sacc[kt], 0, 0, 0);
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bql[kb], sacc[kt], 0, 0, 0);
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bqh[kb], sacc[kt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);          // keep the K fragments of later key tiles out of the register file
    }
    // sacc[kt][r] = q . k for key kt*32 + (r&3) + 8*(r>>2) + 4*hh of query q.  Keys past the sequence must not win the maximum: only
    // the LAST key tile can hold any (the host picks NKT = ceil(T / 32))
    ATT_T(t_d)
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((NKT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= T) sacc[NKT - 1][r] = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sacc[kt][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mneg = -m * scale_log2e;            // scale > 0: max(s) * scale == max(s * scale)
    float sum = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[nt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      uint32_t ph[8], pl[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float e0 = __builtin_amdgcn_exp2f(fmaf(sacc[kt][2 * j], scale_log2e, mneg));
        const float e1 = __builtin_amdgcn_exp2f(fmaf(sacc[kt][2 * j + 1], scale_log2e, mneg));
        sum += e0 + e1;
        ph[j] = pk2(e0, e1);
        pl[j] = pk2(e0 - __uint_as_float(ph[j] << 16), e1 - __uint_as_float(ph[j] & 0xFFFF0000u));
      }
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        uint4 pvh = make_uint4(ph[4 * kb2], ph[4 * kb2 + 1], ph[4 * kb2 + 2], ph[4 * kb2 + 3]);
        uint4 pvl = make_uint4(pl[4 * kb2], pl[4 * kb2 + 1], pl[4 * kb2 + 2], pl[4 * kb2 + 3]);
        const bf16x8 pbh = *reinterpret_cast<bf16x8*>(&pvh), pbl = *reinterpret_cast<bf16x8*>(&pvl);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int vo = (nt * 32 + l31) * LDV + kt * 32 + 16 * kb2 + 4 * hh;
          const uint2 h0 = *reinterpret_cast<const uint2*>(&sVt[0][vo]), h1 = *reinterpret_cast<const uint2*>(&sVt[0][vo + 8]);
          const uint2 l0 = *reinterpret_cast<const uint2*>(&sVt[1][vo]), l1 = *reinterpret_cast<const uint2*>(&sVt[1][vo + 8]);
          uint4 avh = make_uint4(h0.x, h0.y, h1.x, h1.y), avl = make_uint4(l0.x, l0.y, l1.x, l1.y);
          const bf16x8 ah = *reinterpret_cast<bf16x8*>(&avh), al = *reinterpret_cast<bf16x8*>(&avl);
          o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, pbh, o[nt], 0, 0, 0);
          o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pbl, o[nt], 0, 0, 0);
          o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pbh, o[nt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    sum += __shfl_xor(sum, 32, 64);
    ATT_T(t_e)
    const float inv = 1.0f / sum;
    // o[nt][r] = O[query q][d = nt*32 + (r&3) + 8*(r>>2) + 4*hh]: a lane owns runs of four channels of ITS query -> 8-byte stores per plane
    if (q < T) {
      const size_t ro = ((size_t)b * T + q) * D + h * PATT_HD;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v[4] = {o[nt][4 * g] * inv, o[nt][4 * g + 1] * inv, o[nt][4 * g + 2] * inv, o[nt][4 * g + 3] * inv};
          uint2 vh, vl;
          split4(v, vh, vl);
          *reinterpret_cast<uint2*>(att_h + ro + nt * 32 + 8 * g + 4 * hh) = vh;
          *reinterpret_cast<uint2*>(att_l + ro + nt * 32 + 8 * g + 4 * hh) = vl;
        }
    }
#ifdef RART_ATT_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    ATT_T(t_f)
    if (lane == 0 && blockIdx.x < 4096) {
      unsigned long long* o = g_att_stamps[blockIdx.x][wave];
      o[0] = t_b - t_a; o[1] = t_d - t_c; o[2] = t_e - t_d; o[3] = t_f - t_e;
    }
#endif
  }
}

// ---- the forward again, WALKING (round 6): one workgroup per CU takes a run of (image, head) items; K and V of the NEXT item arrive while the
//      current one computes.  The phase stamps of k_vit_attention_pair (scratch/r6/att_stamps.py) put 45 % of a workgroup's time in the K / V
//      staging, with nothing running beside it: 16 k of 35 k cycles.  Here wave 7 -- the one without a query tile at 197 tokens -- is the loader:
//      both operands go to LDS ROW-MAJOR ([token][72], the K image of the kernel above) with buffer_load ... lds (rart_lds_dma.h: lane -> 16-byte
//      chunk c of the padded image, token c / 9, chunk c % 9, the ninth chunk and the tokens past T zero-filled by the range check; the
//      per-lane offsets are item-invariant, the item moves the scalar offset), K of item i + 1 during soft-max + P V of item i, V of item i + 1
//      during S = K Q^T of item i + 1, two barriers per item.  V is no longer transposed on the way in: the A fragments of O^T = V^T P^T are
//      gathered from the row-major image with 2-byte reads (patt_tr_frag, as the backward kernels do).  Same products in the same order as
//      k_vit_attention_pair: bit-identical output.  Taken for seven key tiles (193 .. 224 tokens) and at least four items per CU.
__device__ __forceinline__ bf16x8 patt_tr_frag(const uint16_t* s, int t0, int d);
template <int NKT>
__global__ __launch_bounds__(kAttBlock, 1) void k_vit_attention_pair_walk(const uint16_t* __restrict__ qkv_h, const uint16_t* __restrict__ qkv_l,
                                                                       uint16_t* __restrict__ att_h, uint16_t* __restrict__ att_l, int T, int H,
                                                                       int ld, int D, float scale_log2e, int n_items) {
  constexpr int TP = NKT * 32;
  constexpr int NI = (TP * 9 + 63) / 64;                   // DMA instructions per plane (1 KiB of the padded image each)
  constexpr int PLANE = NI * 1024 / 2;                     // elements per plane buffer (the last instruction's tail included)
  static_assert(NKT <= 7, "wave 7 is the loader");
  __shared__ __attribute__((aligned(16))) uint16_t sK[2][PLANE];
  __shared__ __attribute__((aligned(16))) uint16_t sV[2][PLANE];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave == 7;
  // items blockIdx.x, blockIdx.x + gridDim.x, ...: at any moment neighbouring workgroups hold the heads of ONE image (the 128-byte pieces of a
  // token row they read lie in one DRAM page), as the dispatch order of the one-item kernel had it
  const int first = blockIdx.x, last = n_items, step = gridDim.x;
  if (first >= last) return;
  const uint32_t k_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)&sK[0][0];
  const uint32_t v_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)&sV[0][0];
  const rart_srd_t srd_h = rart_dma_srd(qkv_h), srd_l = rart_dma_srd(qkv_l);
  // byte offset of item `it`'s (image, head) block inside a plane
#define RART_AW_ITEM_OFF(IT) ((uint32_t)((((long long)((IT) / H) * T) * ld + ((IT) % H) * PATT_HD) * 2))
#ifdef RART_ATT_NT       // lab build: dQ / dK / dV as non-temporal stores (measured: no change)
#define RART_ATT_ST8(P, V) __builtin_nontemporal_store(*reinterpret_cast<const unsigned long long*>(&(V)), reinterpret_cast<unsigned long long*>(P))
#else
#define RART_ATT_ST8(P, V) (*reinterpret_cast<uint2*>(P) = (V))
#endif
#define RART_AW_DMA rart_dma_load16_nt      // K / V of an item: no other workgroup reads them (203 -> 199 us per launch)
#define RART_AW_LOAD(LDS_, IT, COL)                                                                              \
  {                                                                                                              \
    const uint32_t so_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(RART_AW_ITEM_OFF(IT) + (uint32_t)((COL)*2)));   \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {      /* (the loader wave has nothing else to do: offsets on the fly) */ \
      const int c_ = 64 * i + lane, t_ = c_ / 9, cc_ = c_ - 9 * t_;                                              \
      const uint32_t vo_ = (cc_ < 8 && t_ < T) ? (uint32_t)((t_ * ld + cc_ * 8) * 2) : RART_DMA_OOR;             \
      RART_AW_DMA(vo_, srd_h, so_, (LDS_) + i * 1024);                                                           \
      RART_AW_DMA(vo_, srd_l, so_, (LDS_) + PLANE * 2 + i * 1024);                                               \
    }                                                                                                            \
  }
#define RART_AW_BARRIER()                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
  __builtin_amdgcn_s_barrier();                                                                                  \
  asm volatile("" ::: "memory");
  if (loader) {
    // ---- the loader wave: K and V of the first item, then per item K of the next one during phase 2 and its V during the next phase 1
    RART_AW_LOAD(k_lds, first, D)
    RART_AW_LOAD(v_lds, first, 2 * D)
    rart_dma_wait<0>();
    RART_AW_BARRIER()
    for (int it = first; it < last; it += step) {
      const bool more = it + step < last;
      rart_dma_wait<0>();                               // V of this item
      RART_AW_BARRIER()                                 // (B1) every wave is done with K; V is visible
      if (more) RART_AW_LOAD(k_lds, it + step, D)
      rart_dma_wait<0>();                               // K of the next item
      RART_AW_BARRIER()                                 // (B2) every wave is done with V; the next K is visible
      if (more) RART_AW_LOAD(v_lds, it + step, 2 * D)
    }
    return;
  }
  RART_AW_BARRIER()
  const bool tile = wave < NKT;
  const int q = wave * 32 + l31;
  bf16x8 bqh[4], bql[4];                                // the wave's Q fragments of the CURRENT item (requested during the previous item's stores)
#define RART_AW_LOAD_Q(IT)                                                                                       \
  if (tile) {                                                                                                    \
    const size_t bo_ = ((size_t)((IT) / H) * T) * ld + ((IT) % H) * PATT_HD;                                     \
    _Pragma("unroll") for (int kb = 0; kb < 4; ++kb) {                                                          \
      uint4 vh = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);                                            \
      if (q < T) {                                                                                               \
        vh = *reinterpret_cast<const uint4*>(qkv_h + bo_ + (size_t)q * ld + kb * 16 + hh * 8);                   \
        vl = *reinterpret_cast<const uint4*>(qkv_l + bo_ + (size_t)q * ld + kb * 16 + hh * 8);                   \
      }                                                                                                          \
      bqh[kb] = *reinterpret_cast<bf16x8*>(&vh);                                                                 \
      bql[kb] = *reinterpret_cast<bf16x8*>(&vl);                                                                 \
    }                                                                                                            \
  }
  for (int it = first; it < last; it += step) {
    RART_AW_LOAD_Q(it)
    // ---- phase 1: S^T = K Q^T
    f32x16 sacc[NKT];
    if (tile) {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sK[0][(kt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8]);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sK[1][(kt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8]);
          sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bqh[kb], sacc[kt], 0, 0, 0);
          sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bql[kb], sacc[kt], 0, 0, 0);
          sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bqh[kb], sacc[kt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    RART_AW_BARRIER()                                   // (B1)
    // ---- phase 2: soft-max + O^T = V^T P^T + stores
    if (tile) {
      float m = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if ((NKT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= T) sacc[NKT - 1][r] = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, sacc[kt][r]);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      const float mneg = -m * scale_log2e;
      float sum = 0.f;
      f32x16 o[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nt][r] = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        uint32_t ph[8], pl[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float e0 = __builtin_amdgcn_exp2f(fmaf(sacc[kt][2 * j], scale_log2e, mneg));
          const float e1 = __builtin_amdgcn_exp2f(fmaf(sacc[kt][2 * j + 1], scale_log2e, mneg));
          sum += e0 + e1;
          ph[j] = pk2(e0, e1);
          pl[j] = pk2(e0 - __uint_as_float(ph[j] << 16), e1 - __uint_as_float(ph[j] & 0xFFFF0000u));
        }
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
          uint4 pvh = make_uint4(ph[4 * kb2], ph[4 * kb2 + 1], ph[4 * kb2 + 2], ph[4 * kb2 + 3]);
          uint4 pvl = make_uint4(pl[4 * kb2], pl[4 * kb2 + 1], pl[4 * kb2 + 2], pl[4 * kb2 + 3]);
          const bf16x8 pbh = *reinterpret_cast<bf16x8*>(&pvh), pbl = *reinterpret_cast<bf16x8*>(&pvl);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const bf16x8 ah = patt_tr_frag(sV[0], kt * 32 + 16 * kb2 + 4 * hh, nt * 32 + l31);
            const bf16x8 al = patt_tr_frag(sV[1], kt * 32 + 16 * kb2 + 4 * hh, nt * 32 + l31);
            o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, pbh, o[nt], 0, 0, 0);
            o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pbl, o[nt], 0, 0, 0);
            o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pbh, o[nt], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.0f / sum;
      if (q < T) {
        const size_t ro = ((size_t)(it / H) * T + q) * D + (it % H) * PATT_HD;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float v[4] = {o[nt][4 * g] * inv, o[nt][4 * g + 1] * inv, o[nt][4 * g + 2] * inv, o[nt][4 * g + 3] * inv};
            uint2 vh, vl;
            split4(v, vh, vl);
            *reinterpret_cast<uint2*>(att_h + ro + nt * 32 + 8 * g + 4 * hh) = vh;
            *reinterpret_cast<uint2*>(att_l + ro + nt * 32 + 8 * g + 4 * hh) = vl;
          }
      }
    }
    RART_AW_BARRIER()                                   // (B2)
  }
#undef RART_AW_BARRIER
#undef RART_AW_LOAD_Q
#undef RART_AW_LOAD
#undef RART_AW_ITEM_OFF
}

// ---- fused attention BACKWARD on pairs: two kernels, one workgroup per (image, head) each -----------------------------------------------
// dQ, dK, dV of softmax(Q K^T / sqrt(d)) V from the Q, K, V, O (the forward's output) and dO pairs.  The bf16 kernel (k_vit_attention_bwd,
// csrc/vit_aux.hip) keeps Q, K, V, dO and two transposed copies in 158 KB of LDS; as pairs that is twice what a CU has, so the two phases
// of that kernel are two launches here, each with HALF of the operands resident (129 KB):
//   k_vit_attention_bwd_q_pair  (a wave owns 32 QUERIES; K, V pairs resident): S^T = K Q^T and dP^T = V dO^T land with lane = query, so max,
//       1 / sum and delta = rowsum(dO * O) need one shuffle; dS (fp32, split into hi + lo) feeds straight back as the B operand of
//       dQ^T = K^T dS^T, whose A operand K^T is gathered from the row-major K image with 2-byte reads in the accumulator's key order (a
//       transposed copy would not fit); writes dQ and the per-query statistics (max, 1 / sum, delta);
//   k_vit_attention_bwd_kv_pair (a wave owns 32 KEYS; Q, dO pairs resident): S = Q K^T and dP = dO V^T are recomputed with lane = key
//       (statistics from LDS), P and dS are the B operands of dV^T = dO^T P and dK^T = Q^T dS (A operands gathered from the row-major Q / dO
//       images), accumulated in registers over the query tiles; writes dK and dV.
// Every contraction is three MFMA products; the soft-max and its derivative are evaluated in fp32 registers.  Replaces, per layer, the
// decomposition into five batched pair products with fp32 score-sized temporaries, two soft-max row kernels and ten transposes.
__device__ __forceinline__ void patt_load_rows(uint16_t* s, const uint16_t* g, int T, int TP, int ld, int tid) {
  for (int i = tid; i < TP * 8; i += kAttBlock) {                         // [t][72]: 16-byte chunks, coalesced
    const int t = i >> 3, c = i & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t < T) v = *reinterpret_cast<const uint4*>(g + (size_t)t * ld + c * 8);
    *reinterpret_cast<uint4*>(s + t * PATT_LDK + c * 8) = v;
  }
}
// A-operand fragment of the TRANSPOSE of a row-major [t][72] image: rows = channel d, k = the eight tokens t0 + {0,1,2,3,8,9,10,11}
__device__ __forceinline__ bf16x8 patt_tr_frag(const uint16_t* s, int t0, int d) {
#ifndef RART_PATT_GATHER
  // round 6: two ds_read_b64_tr_b16 instead of eight 2-byte reads.  A 16-lane group reads a [4 tokens][16 channels] block of the row-major
  // image -- lane a supplies the address of channels 4 (a & 3) .. + 3 of token a >> 2 -- and lane a receives channel a of the four tokens
  // (the mapping csrc/wgrad_direct.hip uses; scratch/r4/probe_tr16.hip prints it).  Callers pass d = 32 nt + (lane & 31) and a t0 that is
  // uniform over a 16-lane group, so a = d & 15 and the group's channel base is d - a.  144-byte rows: 8-byte aligned addresses.
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int a = d & 15;
  const uint16_t* p = s + (t0 + (a >> 2)) * PATT_LDK + (d - a) + 4 * (a & 3);
  const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 8 * PATT_LDK));
  const s16x8 r = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  return __builtin_bit_cast(bf16x8, r);
#else
  const uint16_t* p = s + t0 * PATT_LDK + d;
  uint4 v;
  v.x = (uint32_t)p[0] | ((uint32_t)p[PATT_LDK] << 16);
  v.y = (uint32_t)p[2 * PATT_LDK] | ((uint32_t)p[3 * PATT_LDK] << 16);
  v.z = (uint32_t)p[8 * PATT_LDK] | ((uint32_t)p[9 * PATT_LDK] << 16);
  v.w = (uint32_t)p[10 * PATT_LDK] | ((uint32_t)p[11 * PATT_LDK] << 16);
  return *reinterpret_cast<bf16x8*>(&v);
#endif
}
__device__ __forceinline__ void patt_split8(const float* v, bf16x8& hi, bf16x8& lo) {
  uint4 h, l;
  split8(v, h, l);
  hi = *reinterpret_cast<bf16x8*>(&h);
  lo = *reinterpret_cast<bf16x8*>(&l);
}
#define RART_MFMA3(ACC, AH, AL, BH, BL)                                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL, BH, ACC, 0, 0, 0);    \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BL, ACC, 0, 0, 0);    \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BH, ACC, 0, 0, 0);

template <int NKT>
__global__ __launch_bounds__(kAttBlock, 1) void k_vit_attention_bwd_q_pair(const uint16_t* __restrict__ qkv_h, const uint16_t* __restrict__ qkv_l,
                                                                        const uint16_t* __restrict__ o_h, const uint16_t* __restrict__ o_l,
                                                                        const uint16_t* __restrict__ do_h, const uint16_t* __restrict__ do_l,
                                                                        uint16_t* __restrict__ dq_h, uint16_t* __restrict__ dq_l,
                                                                        float4* __restrict__ stats, int T, int H, int ld, int D, float scale,
                                                                        float scale_log2e) {
  constexpr int TP = NKT * 32;
  __shared__ __attribute__((aligned(16))) uint16_t sK[2][TP * PATT_LDK];
  __shared__ __attribute__((aligned(16))) uint16_t sV[2][TP * PATT_LDK];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
  const size_t boff = (size_t)b * T * ld + h * PATT_HD, doff = (size_t)b * T * D + h * PATT_HD;
  const uint16_t* const qb[2] = {qkv_h + boff, qkv_l + boff};
  const uint16_t* const ob[2] = {o_h + doff, o_l + doff};
  const uint16_t* const db[2] = {do_h + doff, do_l + doff};
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    patt_load_rows(sK[p], qb[p] + D, T, TP, ld, tid);
    patt_load_rows(sV[p], qb[p] + 2 * D, T, TP, ld, tid);
  }
  __syncthreads();
  for (int qt = wave; qt < NKT; qt += kAttBlock / 64) {
    const int q = qt * 32 + l31;
    bf16x8 bq[2][4], bdo[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        uint4 vq = make_uint4(0, 0, 0, 0), vd = make_uint4(0, 0, 0, 0);
        if (q < T) {
          vq = *reinterpret_cast<const uint4*>(qb[p] + (size_t)q * ld + kb * 16 + hh * 8);
          vd = *reinterpret_cast<const uint4*>(db[p] + (size_t)q * D + kb * 16 + hh * 8);
        }
        bq[p][kb] = *reinterpret_cast<bf16x8*>(&vq);
        bdo[p][kb] = *reinterpret_cast<bf16x8*>(&vd);
      }
    // delta_q = sum_d dO[q][d] O[q][d]: a lane sums its half of the channels, one shuffle adds the halves
    float delta = 0.f;
    if (q < T) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float ov[8], dv[8];
        join8(*reinterpret_cast<const uint4*>(ob[0] + (size_t)q * D + hh * 32 + c * 8), *reinterpret_cast<const uint4*>(ob[1] + (size_t)q * D + hh * 32 + c * 8), ov);
        join8(*reinterpret_cast<const uint4*>(db[0] + (size_t)q * D + hh * 32 + c * 8), *reinterpret_cast<const uint4*>(db[1] + (size_t)q * D + hh * 32 + c * 8), dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) delta = fmaf(ov[j], dv[j], delta);
      }
    }
    delta += __shfl_xor(delta, 32, 64);
    f32x16 sacc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sK[0][(kt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8]);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sK[1][(kt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8]);
        RART_MFMA3(sacc[kt], ah, al, bq[0][kb], bq[1][kb])
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((NKT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= T) sacc[NKT - 1][r] = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sacc[kt][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64)) * scale_log2e;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], scale_log2e, -m));
        sacc[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (hh == 0) stats[(size_t)blockIdx.x * TP + q] = make_float4(m, inv, delta, 0.f);
    const float sinv = scale * inv;
    f32x16 dq[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[nt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x16 dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sV[0][(kt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8]);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sV[1][(kt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8]);
        RART_MFMA3(dp, ah, al, bdo[0][kb], bdo[1][kb])
      }
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        float dsv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dsv[j] = sinv * sacc[kt][8 * kb2 + j] * (dp[8 * kb2 + j] - delta);
        bf16x8 dsh, dsl;
        patt_split8(dsv, dsh, dsl);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int t0 = kt * 32 + 16 * kb2 + 4 * hh, dch = nt * 32 + l31;
          const bf16x8 ah = patt_tr_frag(sK[0], t0, dch), al = patt_tr_frag(sK[1], t0, dch);
          RART_MFMA3(dq[nt], ah, al, dsh, dsl)
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // one key tile at a time
    }
    // dq[nt][r] = dQ[query q][d = nt*32 + (r&3) + 8*(r>>2) + 4*hh]
    if (q < T) {
      const size_t ro = boff + (size_t)q * ld;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v[4] = {dq[nt][4 * g], dq[nt][4 * g + 1], dq[nt][4 * g + 2], dq[nt][4 * g + 3]};
          uint2 vh, vl;
          split4(v, vh, vl);
          RART_ATT_ST8(dq_h + ro + nt * 32 + 8 * g + 4 * hh, vh);
          RART_ATT_ST8(dq_l + ro + nt * 32 + 8 * g + 4 * hh, vl);
        }
    }
  }
}

template <int NKT>
__global__ __launch_bounds__(kAttBlock, 1) void k_vit_attention_bwd_q_pair_walk(const uint16_t* __restrict__ qkv_h, const uint16_t* __restrict__ qkv_l,
                                                                        const uint16_t* __restrict__ o_h, const uint16_t* __restrict__ o_l,
                                                                        const uint16_t* __restrict__ do_h, const uint16_t* __restrict__ do_l,
                                                                        uint16_t* __restrict__ dq_h, uint16_t* __restrict__ dq_l,
                                                                        float4* __restrict__ stats, int T, int H, int ld, int D, float scale,
                                                                        float scale_log2e, int n_items) {
  constexpr int TP = NKT * 32;
  constexpr int NI = (TP * 9 + 63) / 64;                   // DMA instructions per plane of a padded [token][72] image
  constexpr int PLANE = NI * 512;                          // elements per plane buffer (the last instruction's tail included)
  static_assert(NKT == 7, "wave 7 is the loader: one query tile per compute wave");
  __shared__ __attribute__((aligned(16))) uint16_t sK[2][PLANE];
  __shared__ __attribute__((aligned(16))) uint16_t sV[2][PLANE];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int first = blockIdx.x, step = gridDim.x;          // items first, first + step, ...: neighbouring workgroups hold the heads of one image
  if (first >= n_items) return;
#define RART_BQ_BARRIER()                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
  __builtin_amdgcn_s_barrier();                                                                                  \
  asm volatile("" ::: "memory");
  if (wave == 7) {
    // ---- the loader wave.  Phase B (dP, dS, dQ) walks the key tiles in order, every wave the same, and is the LAST reader of a key tile's
    //      K and V rows: once all waves are past key tiles 2k, 2k + 1 (barriers S1 / S3 / S5) the NEXT item's 64-row slice of K and of V goes
    //      there.  The last slice (rows 192 ..) follows the item boundary: K's is needed by the seventh key tile of S = K Q^T (barrier A6
    //      in front of it), V's not before phase B.
    const uint32_t k_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)&sK[0][0];
    const uint32_t v_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)&sV[0][0];
    const rart_srd_t srd_h = rart_dma_srd(qkv_h), srd_l = rart_dma_srd(qkv_l);
#define RART_BQ_SLICE(IT, I0, I1)                                                                                \
    {                                                                                                            \
      const int it_ = (IT);                                                                                      \
      const uint32_t ib_ = (uint32_t)((((long long)(it_ / H) * T) * ld + (it_ % H) * PATT_HD) * 2);              \
      const uint32_t ko_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ib_ + (uint32_t)(D * 2)));             \
      const uint32_t vo2_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ib_ + (uint32_t)(D * 4)));            \
      _Pragma("unroll") for (int i = (I0); i < (I1); ++i) {                                                     \
        const int c_ = 64 * i + lane, t_ = c_ / 9, cc_ = c_ - 9 * t_;                                            \
        const uint32_t vo_ = (cc_ < 8 && t_ < T) ? (uint32_t)((t_ * ld + cc_ * 8) * 2) : RART_DMA_OOR;           \
        rart_dma_load16_nt(vo_, srd_h, ko_, k_lds + i * 1024);                                                   \
        rart_dma_load16_nt(vo_, srd_l, ko_, k_lds + PLANE * 2 + i * 1024);                                       \
        rart_dma_load16_nt(vo_, srd_h, vo2_, v_lds + i * 1024);                                                  \
        rart_dma_load16_nt(vo_, srd_l, vo2_, v_lds + PLANE * 2 + i * 1024);                                      \
      }                                                                                                          \
    }
    RART_BQ_SLICE(first, 0, 27)
    for (int it = first; it < n_items; it += step) {
      const bool more = it + step < n_items;
      rart_dma_wait<0>();                               // rows 0 .. 191 of this item's K and V
      RART_BQ_BARRIER()                                 // (E) the item starts
      RART_BQ_SLICE(it, 27, NI)                         // rows 192 ..: free since the previous item's last key tile
      rart_dma_wait<0>();
      RART_BQ_BARRIER()                                 // (A6) in front of the seventh key tile of S = K Q^T
      RART_BQ_BARRIER()                                 // (S1) phase B is past key tiles 0, 1
      if (more) RART_BQ_SLICE(it + step, 0, 9)
      RART_BQ_BARRIER()                                 // (S3)
      if (more) RART_BQ_SLICE(it + step, 9, 18)
      RART_BQ_BARRIER()                                 // (S5)
      if (more) RART_BQ_SLICE(it + step, 18, 27)
    }
    rart_dma_wait<0>();
    return;
#undef RART_BQ_SLICE
  }
  for (int it = first; it < n_items; it += step) {
  const int b = it / H, h = it - b * H;
  const size_t boff = (size_t)b * T * ld + h * PATT_HD, doff = (size_t)b * T * D + h * PATT_HD;
  const uint16_t* const qb[2] = {qkv_h + boff, qkv_l + boff};
  const uint16_t* const ob[2] = {o_h + doff, o_l + doff};
  const uint16_t* const db[2] = {do_h + doff, do_l + doff};
  RART_BQ_BARRIER()                                     // (E)
  {
    const int q = wave * 32 + l31;
    bf16x8 bq[2][4], bdo[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        uint4 vq = make_uint4(0, 0, 0, 0), vd = make_uint4(0, 0, 0, 0);
        if (q < T) {
          vq = *reinterpret_cast<const uint4*>(qb[p] + (size_t)q * ld + kb * 16 + hh * 8);
          vd = *reinterpret_cast<const uint4*>(db[p] + (size_t)q * D + kb * 16 + hh * 8);
        }
        bq[p][kb] = *reinterpret_cast<bf16x8*>(&vq);
        bdo[p][kb] = *reinterpret_cast<bf16x8*>(&vd);
      }
    // delta_q = sum_d dO[q][d] O[q][d]: a lane sums its half of the channels, one shuffle adds the halves
    float delta = 0.f;
    if (q < T) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float ov[8], dv[8];
        join8(*reinterpret_cast<const uint4*>(ob[0] + (size_t)q * D + hh * 32 + c * 8), *reinterpret_cast<const uint4*>(ob[1] + (size_t)q * D + hh * 32 + c * 8), ov);
        join8(*reinterpret_cast<const uint4*>(db[0] + (size_t)q * D + hh * 32 + c * 8), *reinterpret_cast<const uint4*>(db[1] + (size_t)q * D + hh * 32 + c * 8), dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) delta = fmaf(ov[j], dv[j], delta);
      }
    }
    delta += __shfl_xor(delta, 32, 64);
    f32x16 sacc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt == NKT - 1) { RART_BQ_BARRIER() }          // (A6) the last rows of K (and V) have landed
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sK[0][(kt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8]);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sK[1][(kt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8]);
        RART_MFMA3(sacc[kt], ah, al, bq[0][kb], bq[1][kb])
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((NKT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= T) sacc[NKT - 1][r] = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sacc[kt][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64)) * scale_log2e;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], scale_log2e, -m));
        sacc[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (hh == 0) stats[(size_t)it * TP + q] = make_float4(m, inv, delta, 0.f);
    const float sinv = scale * inv;
    f32x16 dq[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[nt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x16 dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sV[0][(kt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8]);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sV[1][(kt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8]);
        RART_MFMA3(dp, ah, al, bdo[0][kb], bdo[1][kb])
      }
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        float dsv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dsv[j] = sinv * sacc[kt][8 * kb2 + j] * (dp[8 * kb2 + j] - delta);
        bf16x8 dsh, dsl;
        patt_split8(dsv, dsh, dsl);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int t0 = kt * 32 + 16 * kb2 + 4 * hh, dch = nt * 32 + l31;
          const bf16x8 ah = patt_tr_frag(sK[0], t0, dch), al = patt_tr_frag(sK[1], t0, dch);
          RART_MFMA3(dq[nt], ah, al, dsh, dsl)
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // one key tile at a time
      if (kt == 1 || kt == 3 || kt == 5) { RART_BQ_BARRIER() }      // (S1 / S3 / S5) a 64-row slice of K and of V is free
    }
    // dq[nt][r] = dQ[query q][d = nt*32 + (r&3) + 8*(r>>2) + 4*hh]
    if (q < T) {
      const size_t ro = boff + (size_t)q * ld;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v[4] = {dq[nt][4 * g], dq[nt][4 * g + 1], dq[nt][4 * g + 2], dq[nt][4 * g + 3]};
          uint2 vh, vl;
          split4(v, vh, vl);
          RART_ATT_ST8(dq_h + ro + nt * 32 + 8 * g + 4 * hh, vh);
          RART_ATT_ST8(dq_l + ro + nt * 32 + 8 * g + 4 * hh, vl);
        }
    }
  }
  }
#undef RART_BQ_BARRIER
}

template <int NKT>
__global__ __launch_bounds__(kAttBlock, 1) void k_vit_attention_bwd_kv_pair(const uint16_t* __restrict__ qkv_h, const uint16_t* __restrict__ qkv_l,
                                                                         const uint16_t* __restrict__ do_h, const uint16_t* __restrict__ do_l,
                                                                         uint16_t* __restrict__ dq_h, uint16_t* __restrict__ dq_l,
                                                                         const float4* __restrict__ stats, int T, int H, int ld, int D, float scale,
                                                                         float scale_log2e) {
  constexpr int TP = NKT * 32;
  __shared__ __attribute__((aligned(16))) uint16_t sQ[2][TP * PATT_LDK];
  __shared__ __attribute__((aligned(16))) uint16_t sdO[2][TP * PATT_LDK];
  __shared__ __attribute__((aligned(16))) float4 sStat[TP];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
  const size_t boff = (size_t)b * T * ld + h * PATT_HD, doff = (size_t)b * T * D + h * PATT_HD;
  const uint16_t* const qb[2] = {qkv_h + boff, qkv_l + boff};
  const uint16_t* const db[2] = {do_h + doff, do_l + doff};
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    patt_load_rows(sQ[p], qb[p], T, TP, ld, tid);
    patt_load_rows(sdO[p], db[p], T, TP, D, tid);
  }
  for (int i = tid; i < TP; i += kAttBlock) sStat[i] = stats[(size_t)blockIdx.x * TP + i];
  __syncthreads();
  for (int kt = wave; kt < NKT; kt += kAttBlock / 64) {
    const int key = kt * 32 + l31;
    const bool key_ok = key < T;
    bf16x8 bk[2][4], bv[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        uint4 vk = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key_ok) {
          vk = *reinterpret_cast<const uint4*>(qb[p] + (size_t)key * ld + D + kb * 16 + hh * 8);
          vv = *reinterpret_cast<const uint4*>(qb[p] + (size_t)key * ld + 2 * D + kb * 16 + hh * 8);
        }
        bk[p][kb] = *reinterpret_cast<bf16x8*>(&vk);
        bv[p][kb] = *reinterpret_cast<bf16x8*>(&vv);
      }
    // dV^T = dO^T . P and dK^T = Q^T . dS (operands swapped): lane = key, registers = channels
    f32x16 dvv[2], dkk[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dvv[nt][r] = dkk[nt][r] = 0.f;
#pragma unroll 1
    for (int qt = 0; qt < NKT; ++qt) {
      f32x16 sa, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = dp[r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int so = (qt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8;
        const bf16x8 qh = *reinterpret_cast<const bf16x8*>(&sQ[0][so]), ql = *reinterpret_cast<const bf16x8*>(&sQ[1][so]);
        const bf16x8 dh = *reinterpret_cast<const bf16x8*>(&sdO[0][so]), dl = *reinterpret_cast<const bf16x8*>(&sdO[1][so]);
        RART_MFMA3(sa, qh, ql, bk[0][kb], bk[1][kb])
        RART_MFMA3(dp, dh, dl, bv[0][kb], bv[1][kb])
      }
      // lane = key kt*32 + l31; register r = query qt*32 + (r&3) + 8*(r>>2) + 4*hh (queries past the sequence have zero Q / dO rows:
      // whatever probability they get multiplies zeros)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 st = sStat[qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
        const float pv = key_ok ? __builtin_amdgcn_exp2f(fmaf(sa[r], scale_log2e, -st.x)) * st.y : 0.f;
        sa[r] = pv;
        dp[r] = scale * pv * (dp[r] - st.z);
      }
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        float pw[8], dw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pw[j] = sa[8 * kb2 + j];
          dw[j] = dp[8 * kb2 + j];
        }
        bf16x8 pbh, pbl, dbh, dbl;
        patt_split8(pw, pbh, pbl);
        patt_split8(dw, dbh, dbl);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int t0 = qt * 32 + 16 * kb2 + 4 * hh, dch = nt * 32 + l31;
          const bf16x8 oh = patt_tr_frag(sdO[0], t0, dch), ol = patt_tr_frag(sdO[1], t0, dch);
          const bf16x8 qh = patt_tr_frag(sQ[0], t0, dch), ql = patt_tr_frag(sQ[1], t0, dch);
          RART_MFMA3(dvv[nt], oh, ol, pbh, pbl)
          RART_MFMA3(dkk[nt], qh, ql, dbh, dbl)
        }
      }
    }
    // dkk / dvv[nt][r] = dK / dV[key][d = nt*32 + (r&3) + 8*(r>>2) + 4*hh]: 8-byte runs of the lane's own key row
    if (key_ok) {
      const size_t ro = boff + (size_t)key * ld;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float vk[4] = {dkk[nt][4 * g], dkk[nt][4 * g + 1], dkk[nt][4 * g + 2], dkk[nt][4 * g + 3]};
          const float vv[4] = {dvv[nt][4 * g], dvv[nt][4 * g + 1], dvv[nt][4 * g + 2], dvv[nt][4 * g + 3]};
          uint2 kh, kl, vh, vl;
          split4(vk, kh, kl);
          split4(vv, vh, vl);
          RART_ATT_ST8(dq_h + ro + D + nt * 32 + 8 * g + 4 * hh, kh);
          RART_ATT_ST8(dq_l + ro + D + nt * 32 + 8 * g + 4 * hh, kl);
          RART_ATT_ST8(dq_h + ro + 2 * D + nt * 32 + 8 * g + 4 * hh, vh);
          RART_ATT_ST8(dq_l + ro + 2 * D + nt * 32 + 8 * g + 4 * hh, vl);
        }
    }
  }
}
template <int NKT>
__global__ __launch_bounds__(kAttBlock, 1) void k_vit_attention_bwd_kv_pair_walk(const uint16_t* __restrict__ qkv_h, const uint16_t* __restrict__ qkv_l,
                                                                         const uint16_t* __restrict__ do_h, const uint16_t* __restrict__ do_l,
                                                                         uint16_t* __restrict__ dq_h, uint16_t* __restrict__ dq_l,
                                                                         const float4* __restrict__ stats, int T, int H, int ld, int D, float scale,
                                                                         float scale_log2e, int n_items) {
  constexpr int TP = NKT * 32;
  constexpr int NI = (TP * 9 + 63) / 64;                   // DMA instructions per plane of a padded [token][72] image
  constexpr int PLANE = NI * 512;                          // elements per plane buffer
  static_assert(NKT == 7 && NI == 32, "wave 7 is the loader; slices of 64 rows = 9 instructions, the last one 5");
  __shared__ __attribute__((aligned(16))) uint16_t sQ[2][PLANE];
  __shared__ __attribute__((aligned(16))) uint16_t sdO[2][PLANE];
  __shared__ __attribute__((aligned(16))) float4 sStat2[2][TP];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int first = blockIdx.x, step = gridDim.x;
  if (first >= n_items) return;
#define RART_BK_BARRIER()                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
  __builtin_amdgcn_s_barrier();                                                                                  \
  asm volatile("" ::: "memory");
  if (wave == 7) {
    // ---- the loader wave.  The compute waves consume Q and dO one 32-query tile at a time, in the same order, so the rows of a 64-row
    //      slice are free once every wave is past its two tiles (barriers S1 / S3 / S5 below): the NEXT item's slice goes there while the
    //      current item still works on the later ones; the last slice (rows 192 ..) follows the item boundary and has until tile 6.
    const uint32_t q_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)&sQ[0][0];
    const uint32_t d_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)&sdO[0][0];
    const rart_srd_t sq_h = rart_dma_srd(qkv_h), sq_l = rart_dma_srd(qkv_l), sd_h = rart_dma_srd(do_h), sd_l = rart_dma_srd(do_l);
#define RART_BK_SLICE(IT, I0, I1)                                                                                \
    {                                                                                                            \
      const int it_ = (IT);                                                                                      \
      const uint32_t qo_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)((((long long)(it_ / H) * T) * ld + (it_ % H) * PATT_HD) * 2)); \
      const uint32_t do_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)((((long long)(it_ / H) * T) * D + (it_ % H) * PATT_HD) * 2));  \
      _Pragma("unroll") for (int i = (I0); i < (I1); ++i) {                                                     \
        const int c_ = 64 * i + lane, t_ = c_ / 9, cc_ = c_ - 9 * t_;                                            \
        const bool ok_ = cc_ < 8 && t_ < T;                                                                      \
        const uint32_t vq_ = ok_ ? (uint32_t)((t_ * ld + cc_ * 8) * 2) : RART_DMA_OOR;                           \
        const uint32_t vd_ = ok_ ? (uint32_t)((t_ * D + cc_ * 8) * 2) : RART_DMA_OOR;                            \
        rart_dma_load16_nt(vq_, sq_h, qo_, q_lds + i * 1024);                                                    \
        rart_dma_load16_nt(vq_, sq_l, qo_, q_lds + PLANE * 2 + i * 1024);                                        \
        rart_dma_load16_nt(vd_, sd_h, do_, d_lds + i * 1024);                                                    \
        rart_dma_load16_nt(vd_, sd_l, do_, d_lds + PLANE * 2 + i * 1024);                                        \
      }                                                                                                          \
    }
#define RART_BK_STATS(IT, PAR)                                                                                   \
    for (int i = lane; i < TP; i += 64) sStat2[PAR][i] = stats[(size_t)(IT)*TP + i];
    RART_BK_SLICE(first, 0, 27)
    RART_BK_STATS(first, 0)
    int par = 0;
    for (int it = first; it < n_items; it += step) {
      const bool more = it + step < n_items;
      rart_dma_wait<0>();                               // rows 0 .. 191 of this item (and its statistics) are in LDS
      RART_BK_BARRIER()                                 // (E) the item starts
      RART_BK_SLICE(it, 27, 32)                         // rows 192 ..: free since the previous item's last tile
      if (more) RART_BK_STATS(it + step, par ^ 1)
      rart_dma_wait<0>();
      RART_BK_BARRIER()                                 // (S1) tiles 0, 1 done everywhere; the last slice is visible
      if (more) RART_BK_SLICE(it + step, 0, 9)
      RART_BK_BARRIER()                                 // (S3)
      if (more) RART_BK_SLICE(it + step, 9, 18)
      RART_BK_BARRIER()                                 // (S5)
      if (more) RART_BK_SLICE(it + step, 18, 27)
      par ^= 1;
    }
    rart_dma_wait<0>();
    return;
#undef RART_BK_STATS
#undef RART_BK_SLICE
  }
  int par = 0;
  for (int it = first; it < n_items; it += step) {
  const int b = it / H, h = it - b * H;
  const size_t boff = (size_t)b * T * ld + h * PATT_HD;
  const uint16_t* const qb[2] = {qkv_h + boff, qkv_l + boff};
  const float4* const sStat = sStat2[par];
  RART_BK_BARRIER()                                     // (E)
  {
    const int kt = wave;
    const int key = kt * 32 + l31;
    const bool key_ok = key < T;
    bf16x8 bk[2][4], bv[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        uint4 vk = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key_ok) {
          vk = *reinterpret_cast<const uint4*>(qb[p] + (size_t)key * ld + D + kb * 16 + hh * 8);
          vv = *reinterpret_cast<const uint4*>(qb[p] + (size_t)key * ld + 2 * D + kb * 16 + hh * 8);
        }
        bk[p][kb] = *reinterpret_cast<bf16x8*>(&vk);
        bv[p][kb] = *reinterpret_cast<bf16x8*>(&vv);
      }
    // dV^T = dO^T . P and dK^T = Q^T . dS (operands swapped): lane = key, registers = channels
    f32x16 dvv[2], dkk[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dvv[nt][r] = dkk[nt][r] = 0.f;
#pragma unroll 1
    for (int qt = 0; qt < NKT; ++qt) {
      f32x16 sa, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = dp[r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int so = (qt * 32 + l31) * PATT_LDK + kb * 16 + hh * 8;
        const bf16x8 qh = *reinterpret_cast<const bf16x8*>(&sQ[0][so]), ql = *reinterpret_cast<const bf16x8*>(&sQ[1][so]);
        const bf16x8 dh = *reinterpret_cast<const bf16x8*>(&sdO[0][so]), dl = *reinterpret_cast<const bf16x8*>(&sdO[1][so]);
        RART_MFMA3(sa, qh, ql, bk[0][kb], bk[1][kb])
        RART_MFMA3(dp, dh, dl, bv[0][kb], bv[1][kb])
      }
      // lane = key kt*32 + l31; register r = query qt*32 + (r&3) + 8*(r>>2) + 4*hh (queries past the sequence have zero Q / dO rows:
      // whatever probability they get multiplies zeros)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 st = sStat[qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
        const float pv = key_ok ? __builtin_amdgcn_exp2f(fmaf(sa[r], scale_log2e, -st.x)) * st.y : 0.f;
        sa[r] = pv;
        dp[r] = scale * pv * (dp[r] - st.z);
      }
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        float pw[8], dw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pw[j] = sa[8 * kb2 + j];
          dw[j] = dp[8 * kb2 + j];
        }
        bf16x8 pbh, pbl, dbh, dbl;
        patt_split8(pw, pbh, pbl);
        patt_split8(dw, dbh, dbl);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int t0 = qt * 32 + 16 * kb2 + 4 * hh, dch = nt * 32 + l31;
          const bf16x8 oh = patt_tr_frag(sdO[0], t0, dch), ol = patt_tr_frag(sdO[1], t0, dch);
          const bf16x8 qh = patt_tr_frag(sQ[0], t0, dch), ql = patt_tr_frag(sQ[1], t0, dch);
          RART_MFMA3(dvv[nt], oh, ol, pbh, pbl)
          RART_MFMA3(dkk[nt], qh, ql, dbh, dbl)
        }
      }
      if (qt == 1 || qt == 3 || qt == 5) { RART_BK_BARRIER() }       // (S1 / S3 / S5) a 64-row slice of Q and dO is free
    }
    // dkk / dvv[nt][r] = dK / dV[key][d = nt*32 + (r&3) + 8*(r>>2) + 4*hh]: 8-byte runs of the lane's own key row
    if (key_ok) {
      const size_t ro = boff + (size_t)key * ld;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float vk[4] = {dkk[nt][4 * g], dkk[nt][4 * g + 1], dkk[nt][4 * g + 2], dkk[nt][4 * g + 3]};
          const float vv[4] = {dvv[nt][4 * g], dvv[nt][4 * g + 1], dvv[nt][4 * g + 2], dvv[nt][4 * g + 3]};
          uint2 kh, kl, vh, vl;
          split4(vk, kh, kl);
          split4(vv, vh, vl);
          RART_ATT_ST8(dq_h + ro + D + nt * 32 + 8 * g + 4 * hh, kh);
          RART_ATT_ST8(dq_l + ro + D + nt * 32 + 8 * g + 4 * hh, kl);
          RART_ATT_ST8(dq_h + ro + 2 * D + nt * 32 + 8 * g + 4 * hh, vh);
          RART_ATT_ST8(dq_l + ro + 2 * D + nt * 32 + 8 * g + 4 * hh, vl);
        }
    }
  }
  par ^= 1;
  }
#undef RART_BK_BARRIER
}
#undef RART_MFMA3
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

int rart_vit_attention_pair(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, int n, int tokens, int heads, int head_dim,
                            rart_stream_t stream) {
  RART_CHECK_ARG(qkv_hi && qkv_lo && out_hi && out_lo && n > 0 && tokens > 0 && heads > 0, "rart_vit_attention_pair: bad arguments");
  RART_CHECK_ARG(head_dim == 64, "rart_vit_attention_pair: head_dim must be 64 (ViT-B/16)");
  RART_CHECK_ARG(tokens <= 224, "rart_vit_attention_pair: at most 224 tokens (197 for 224x224 / patch 16)");
  const int D = heads * head_dim;
  const float scale_log2e = (1.0f / sqrtf((float)head_dim)) * 1.4426950408889634f;
  const dim3 grid((uint32_t)(n * heads));
  hipStream_t st = (hipStream_t)stream;
  {
    // round 6: the walking form (one workgroup per CU, the next item's K / V in flight) for the seven-key-tile shapes of a full batch
    const char* we = getenv("RART_ATT_WALK");                                                // lab switch (read per call: tests flip it)
    const int walk = we ? atoi(we) : 1;
    static int cu_cache[64] = {0};                     // compute units per device (the walking kernel's grid)
    int dev = 0, cus = 0;
    if (walk && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
      if (cu_cache[dev] == 0 && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) cu_cache[dev] = cus;
      cus = cu_cache[dev];
    }
    if (walk && (tokens + 31) / 32 == 7 && cus > 0 && (long long)n * heads >= 4ll * cus &&
        (long long)n * tokens * 3 * D * 2 < (1ll << 31)) {
      hipLaunchKernelGGL(k_vit_attention_pair_walk<7>, dim3((uint32_t)cus), dim3(kAttBlock), 0, st, (const uint16_t*)qkv_hi,
                         (const uint16_t*)qkv_lo, (uint16_t*)out_hi, (uint16_t*)out_lo, tokens, heads, 3 * D, D, scale_log2e, n * heads);
      RART_CHECK_LAUNCH("rart_vit_attention_pair (walking)");
      return RART_OK;
    }
  }
#define RART_PATT_CASE(N) case N: hipLaunchKernelGGL(k_vit_attention_pair<N>, grid, dim3(kAttBlock), 0, st, (const uint16_t*)qkv_hi, \
    (const uint16_t*)qkv_lo, (uint16_t*)out_hi, (uint16_t*)out_lo, tokens, heads, 3 * D, D, scale_log2e); break;
  switch ((tokens + 31) / 32) {                   // key tiles: only the last one is partial
    RART_PATT_CASE(1) RART_PATT_CASE(2) RART_PATT_CASE(3) RART_PATT_CASE(4) RART_PATT_CASE(5) RART_PATT_CASE(6)
    default: hipLaunchKernelGGL(k_vit_attention_pair<7>, grid, dim3(kAttBlock), 0, st, (const uint16_t*)qkv_hi, (const uint16_t*)qkv_lo,
                                (uint16_t*)out_hi, (uint16_t*)out_lo, tokens, heads, 3 * D, D, scale_log2e); break;
  }
#undef RART_PATT_CASE
  RART_CHECK_LAUNCH("rart_vit_attention_pair");
  return RART_OK;
}

int rart_vit_attention_bwd_pair(const void* qkv_hi, const void* qkv_lo, const void* out_hi, const void* out_lo, const void* dout_hi,
                                const void* dout_lo, void* dqkv_hi, void* dqkv_lo, float* stats, int n, int tokens, int heads, int head_dim,
                                rart_stream_t stream) {
  RART_CHECK_ARG(qkv_hi && qkv_lo && out_hi && out_lo && dout_hi && dout_lo && dqkv_hi && dqkv_lo && stats && n > 0 && tokens > 0 && heads > 0,
                 "rart_vit_attention_bwd_pair: bad arguments");
  RART_CHECK_ARG(head_dim == 64, "rart_vit_attention_bwd_pair: head_dim must be 64 (ViT-B/16)");
  RART_CHECK_ARG(tokens <= 224, "rart_vit_attention_bwd_pair: at most 224 tokens (197 for 224x224 / patch 16)");
  const int D = heads * head_dim;
  const float scale = 1.0f / sqrtf((float)head_dim), sl2e = scale * 1.4426950408889634f;
  const dim3 grid((uint32_t)(n * heads));
  hipStream_t st = (hipStream_t)stream;
  const uint16_t *qh = (const uint16_t*)qkv_hi, *ql = (const uint16_t*)qkv_lo, *oh = (const uint16_t*)out_hi, *ol = (const uint16_t*)out_lo;
  const uint16_t *dh = (const uint16_t*)dout_hi, *dl = (const uint16_t*)dout_lo;
  uint16_t *gh = (uint16_t*)dqkv_hi, *gl = (uint16_t*)dqkv_lo;
  float4* s4 = (float4*)stats;
#define RART_PATTB_CASE(N)                                                                                                              \
  hipLaunchKernelGGL(k_vit_attention_bwd_q_pair<N>, grid, dim3(kAttBlock), 0, st, qh, ql, oh, ol, dh, dl, gh, gl, s4, tokens, heads, 3 * D, D, \
                     scale, sl2e);                                                                                                      \
  hipLaunchKernelGGL(k_vit_attention_bwd_kv_pair<N>, grid, dim3(kAttBlock), 0, st, qh, ql, dh, dl, gh, gl, (const float4*)s4, tokens, heads,  \
                     3 * D, D, scale, sl2e);
  {
    // round 6: the walking backward-q kernel (loader wave: V lands under S + soft-max) for the seven-key-tile shapes of a full batch
    const char* we = getenv("RART_ATT_WALK");
    int dev = 0, cus = 0;
    static int cu_cache_b[64] = {0};
    if ((we ? atoi(we) : 1) && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
      if (cu_cache_b[dev] == 0 && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) cu_cache_b[dev] = cus;
      cus = cu_cache_b[dev];
    }
    if ((tokens + 31) / 32 == 7 && cus > 0 && (long long)n * heads >= 4ll * cus && (long long)n * tokens * 3 * D * 2 < (1ll << 31)) {      // (dO: a third of that)
      hipLaunchKernelGGL(k_vit_attention_bwd_q_pair_walk<7>, dim3((uint32_t)cus), dim3(kAttBlock), 0, st, qh, ql, oh, ol, dh, dl, gh, gl, s4, tokens,
                         heads, 3 * D, D, scale, sl2e, n * heads);
      if ((we ? atoi(we) : 1) >= 2 || !we)      // (RART_ATT_WALK=1: backward-q alone walks; lab)
        hipLaunchKernelGGL(k_vit_attention_bwd_kv_pair_walk<7>, dim3((uint32_t)cus), dim3(kAttBlock), 0, st, qh, ql, dh, dl, gh, gl,
                           (const float4*)s4, tokens, heads, 3 * D, D, scale, sl2e, n * heads);
      else
        hipLaunchKernelGGL(k_vit_attention_bwd_kv_pair<7>, grid, dim3(kAttBlock), 0, st, qh, ql, dh, dl, gh, gl, (const float4*)s4, tokens, heads,
                           3 * D, D, scale, sl2e);
      RART_CHECK_LAUNCH("rart_vit_attention_bwd_pair (walking)");
      return RART_OK;
    }
  }
  switch ((tokens + 31) / 32) {                   // key tiles: only the last one is partial
    case 1: RART_PATTB_CASE(1) break;
    case 2: RART_PATTB_CASE(2) break;
    case 3: RART_PATTB_CASE(3) break;
    case 4: RART_PATTB_CASE(4) break;
    case 5: RART_PATTB_CASE(5) break;
    case 6: RART_PATTB_CASE(6) break;
    default: RART_PATTB_CASE(7) break;
  }
#undef RART_PATTB_CASE
  RART_CHECK_LAUNCH("rart_vit_attention_bwd_pair");
  return RART_OK;
}

}  // extern "C"

#ifdef RART_ATT_STAMPS
extern "C" int rart_debug_att_stamps(unsigned long long* out) {      // lab build only: out[4096][8][4]
  if (hipDeviceSynchronize() != hipSuccess) return RART_ERR_HIP;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_att_stamps), sizeof(unsigned long long) * 4096 * 8 * 4) == hipSuccess ? RART_OK : RART_ERR_HIP;
}
#endif
