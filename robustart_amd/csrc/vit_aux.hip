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

// out[b][p][c*ps*ps + r*ps + s] = (x[b][c][py*ps + r][px*ps + s] - mean)/std, as hi and lo bf16 planes.  A thread makes EIGHT consecutive
// s of one (patch, channel, row): two 16-byte stores (ps % 8 == 0; the first version was one element per thread with 64-bit divisions:
// 157 us per 256 images against a 25 us stream)
template <bool SRC_U8>
__global__ __launch_bounds__(kBlock) void k_patchify(const void* __restrict__ src, uint16_t* __restrict__ hi,
                                                     uint16_t* __restrict__ lo, int n, int h, int w, int ps, Norm3 nm) {
  const uint32_t gw = (uint32_t)(w / ps), gh = (uint32_t)(h / ps), kk8 = (uint32_t)(3 * ps * ps / 8), s8n = (uint32_t)(ps / 8);
  const uint32_t total = (uint32_t)n * gh * gw * kk8;                     // host: < 2^32
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
    const uint32_t k8 = i % kk8, pidx = i / kk8;
    const uint32_t px = pidx % gw, t = pidx / gw, py = t % gh, img = t / gh;
    const uint32_t s8 = k8 % s8n, cr = k8 / s8n, r = cr % (uint32_t)ps, c = cr / (uint32_t)ps;
    const uint32_t y = py * ps + r, x = px * ps + s8 * 8;
    float v01[8];
    if (SRC_U8) {
      const uint8_t* p = (const uint8_t*)src + (((size_t)img * h + y) * w + x) * 3 + c;
#pragma unroll
      for (int j = 0; j < 8; ++j) v01[j] = (float)p[3 * j] * (1.0f / 255.0f);
    } else {
      const float4* p = reinterpret_cast<const float4*>((const float*)src + (((size_t)img * 3 + c) * h + y) * w + x);   // x % 8 == 0, w % 8 == 0
      const float4 a = p[0], b = p[1];
      v01[0] = a.x; v01[1] = a.y; v01[2] = a.z; v01[3] = a.w; v01[4] = b.x; v01[5] = b.y; v01[6] = b.z; v01[7] = b.w;
    }
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v0 = (v01[2 * j] - nm.mean[c]) * nm.istd[c], v1 = (v01[2 * j + 1] - nm.mean[c]) * nm.istd[c];
      const uint16_t h0 = f2bf(v0), h1 = f2bf(v1);
      hw[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
      lw[j] = (uint32_t)f2bf(v0 - bf2f(h0)) | ((uint32_t)f2bf(v1 - bf2f(h1)) << 16);
    }
    reinterpret_cast<uint4*>(hi)[i] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    reinterpret_cast<uint4*>(lo)[i] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
  }
}

// x[b][0][:] = cls_pos0; x[b][t][:] += pos[t] (t >= 1); eight channels per thread (d % 8 == 0)
__global__ __launch_bounds__(kBlock) void k_add_pos_cls(uint16_t* __restrict__ x, const float* __restrict__ cls_pos0,
                                                        const float* __restrict__ pos, int n, int t, int d) {
  const uint32_t d8 = (uint32_t)d / 8, total = (uint32_t)n * t * d8;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
    const uint32_t c8 = i % d8, tok = (i / d8) % (uint32_t)t;
    const float* add = tok == 0 ? cls_pos0 + c8 * 8 : pos + (size_t)tok * d + c8 * 8;
    const float4 a0 = reinterpret_cast<const float4*>(add)[0], a1 = reinterpret_cast<const float4*>(add)[1];
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    uint4 xv = reinterpret_cast<uint4*>(x)[i];
    uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float b0 = tok == 0 ? 0.f : __uint_as_float(xw[j] << 16), b1 = tok == 0 ? 0.f : __uint_as_float(xw[j] & 0xFFFF0000u);
      xw[j] = (uint32_t)f2bf(b0 + av[2 * j]) | ((uint32_t)f2bf(b1 + av[2 * j + 1]) << 16);
    }
    reinterpret_cast<uint4*>(x)[i] = make_uint4(xw[0], xw[1], xw[2], xw[3]);
  }
}

// ---- row kernels: one wave per row, the row lives in registers as 16-byte vectors (lane l holds vectors l and l + 64),
//      so every tensor is read exactly once with coalesced 16-byte loads.  Rows of up to 1024 elements, multiple of 8.
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
// loads the row's vectors v (< nv) into r[2][8] (zeros elsewhere)
__device__ __forceinline__ void load_row(const uint16_t* row, int nv, int lane, float r[2][8]) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int v = lane + 64 * k;
    uint4 q = make_uint4(0, 0, 0, 0);
    if (v < nv) q = *reinterpret_cast<const uint4*>(row + (size_t)v * 8);
    unpack8(q, r[k]);
  }
}

// LayerNorm over the last dim, fp32 two-pass statistics.  (Round 5, scratch/r5/time_layernorm.py: 36.4 us per launch on [256 x 197][768] = 4.25 TB/s on
// input + output; a wave walking eight rows with gamma / beta in registers and the next row prefetched was SLOWER, 49.2 us: one row per wave
// keeps far more rows in flight, and the 6 KB of gamma / beta per row are L1 hits.)
__global__ __launch_bounds__(kBlock) void k_layernorm(const uint16_t* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ b, uint16_t* __restrict__ out,
                                                      int rows, int d, long long in_stride, long long out_stride,
                                                      float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = d / 8;
  float xr[2][8];
  load_row(x + (size_t)row * in_stride, nv, lane, xr);
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
  const float rstd = rsqrtf(rart_wave_sum(v) / (float)d + eps);
  uint16_t* o = out + (size_t)row * out_stride;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int vi = lane + 64 * k;
    if (vi < nv) {
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (xr[k][j] - mean) * rstd * g[vi * 8 + j] + b[vi * 8 + j];
      *reinterpret_cast<uint4*>(o + (size_t)vi * 8) = pack8(r);
    }
  }
}

// P[row][0..n_valid) = softmax(scale * S[row][0..n_valid)), P[row][n_valid..ld_out) = 0
__global__ __launch_bounds__(kBlock) void k_softmax_rows(const uint16_t* __restrict__ sm, uint16_t* __restrict__ pm,
                                                         long long rows, int n_valid, int ld_in, int ld_out,
                                                         float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float sr[2][8];
  load_row(sm + row * ld_in, ld_in / 8, lane, sr);
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if ((lane + 64 * k) * 8 + j < n_valid) mx = fmaxf(mx, sr[k][j] * scale);
  mx = rart_wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = (lane + 64 * k) * 8 + j < n_valid;
      sr[k][j] = ok ? __expf(sr[k][j] * scale - mx) : 0.f;
      sum += sr[k][j];
    }
  const float inv = 1.0f / rart_wave_sum(sum);
  uint16_t* pr = pm + row * ld_out;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int vi = lane + 64 * k;
    if (vi < ld_out / 8) {
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = sr[k][j] * inv;
      *reinterpret_cast<uint4*>(pr + (size_t)vi * 8) = pack8(r);
    }
  }
}

// vt[b][h][dd][t] = qkv[b][t][v_off + h*hd + dd] (t < T), 0 for T <= t < t_pad: 64 tokens x 64 dims per workgroup
// through LDS (head_dim == 64); reads are 128-byte rows, writes 128-byte runs of tokens.
__global__ __launch_bounds__(kBlock) void k_transpose_v(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ vt,
                                                        int n, int t, int heads, int hd, int ld, int v_off, int t_pad) {
  __shared__ uint16_t tile[64][72];
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * 64, bh = blockIdx.y, img = bh / heads, hh = bh - img * heads;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = tid / 8 + 32 * i, ch = tid % 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t0 + r < t) v = *reinterpret_cast<const uint4*>(qkv + ((size_t)img * t + t0 + r) * ld + v_off + hh * hd + ch * 8);
    *reinterpret_cast<uint4*>(&tile[r][ch * 8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int dd = tid / 8 + 32 * i, tc = tid % 8;
    if (t0 + tc * 8 >= t_pad) continue;
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (uint32_t)tile[tc * 8 + 2 * j][dd] | ((uint32_t)tile[tc * 8 + 2 * j + 1][dd] << 16);
    *reinterpret_cast<uint4*>(vt + ((size_t)bh * hd + dd) * t_pad + t0 + tc * 8) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}
int grid_for(size_t items) { return rart_grid_for(items, kBlock, 256 * 16); }

// ---- fused multi-head attention (head_dim 64, tokens <= NKT*32), one workgroup per (image, head) -------------
// S^T = K . Q^T on v_mfma_f32_32x32x16_bf16: in the accumulator layout lane (l & 31) is the QUERY and the 16
// registers x 2 lane halves are its keys, so the soft-max statistics of a query live in one lane pair
// (one shuffle with lane ^ 32) and the whole row block stays in registers (NKT*16 fp32 per lane; no online
// rescaling).  The normalised probabilities are packed to bf16 and fed straight back as the A operand of
// O = P . V: the MFMA's k index is only a summation label, so instead of re-laying P through LDS the V operand
// is read from LDS (stored transposed) in the key order the accumulator registers already have.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int ATT_HD = 64, ATT_LDK = ATT_HD + 8;

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {   // round to nearest even (v_cvt_pk_bf16_f32)
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
  f2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2_t));
}

// Round 5 measured the launch alone (scratch/r5/time_attention.py, B = 256, 12 heads, 197 tokens).  One workgroup of 4 waves per (image, head), two
// resident per CU: 122 us = 2.5 TB/s on q, k, v, out, 250 TFLOP/s -- the CU idle ~70 % of the launch between 75 KB staging phases; the interleaved
// [token][3 x 768] layout costs nothing (every head as its own image: 121 us).  A wave per query tile with the Q fragments requested before the
// staging, still one workgroup per item: 134 us (128 VGPRs for two resident workgroups: 18 spilled).  This kernel -- persistent, the next item's
// K / V / Q in flight into registers under the current item's work -- 98.7 us = 3.1 TB/s, 309 TFLOP/s; what is left is the soft-max's
// 112 v_exp_f32 per lane and tile (quarter rate) on 7 waves over 4 SIMDs.
template <int NKT>
__global__ __launch_bounds__(NKT * 64) void k_vit_attention(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ att,
                                                            int T, int H, int ld, int D, float scale_log2e, int items) {
  // Persistent (round 5): a workgroup of NKT waves -- one 32-query tile each -- walks (image, head) items blockIdx.x, + gridDim.x, ...; the K, V
  // and Q pieces of the NEXT item are requested into registers (12 x 16 bytes per thread) before this item is multiplied and committed to LDS
  // after it, so the 75 KB staging of an item is in flight under the matrix + soft-max work of its predecessor.
  constexpr int TP = NKT * 32, LDV = TP + 4;     // 228-element rows: conflict-free 8-byte reads across 32 lanes
  constexpr int NT = NKT * 64;                   // = TP * 8 / 4 K chunks per thread = (TP / 4) * 8 V items: four + one per thread
  __shared__ __attribute__((aligned(16))) uint16_t sK[TP * ATT_LDK];
  __shared__ __attribute__((aligned(16))) uint16_t sVt[ATT_HD * LDV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
  const int q = wave * 32 + l31;
  const int vtq = tid % (TP / 4), vc = tid / (TP / 4);
  uint4 kr[4], vr4[4], qr[4];
  auto issue = [&](int item) {
    const int b = item / H, h = item - b * H;
    const uint16_t* base = qkv + (size_t)b * T * ld + h * ATT_HD;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                     // K rows: 16-byte chunks, coalesced
      const int i = tid + NT * j, t = i >> 3, c = i & 7;
      kr[j] = make_uint4(0, 0, 0, 0);
      if (t < T) kr[j] = *reinterpret_cast<const uint4*>(base + (size_t)t * ld + D + c * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {                                     // V: 8 channels of four consecutive tokens
      vr4[u] = make_uint4(0, 0, 0, 0);
      if (vtq * 4 + u < T) vr4[u] = *reinterpret_cast<const uint4*>(base + (size_t)(vtq * 4 + u) * ld + 2 * D + vc * 8);
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      qr[kb] = make_uint4(0, 0, 0, 0);
      if (q < T) qr[kb] = *reinterpret_cast<const uint4*>(base + (size_t)q * ld + kb * 16 + hh * 8);
    }
  };
  int item = blockIdx.x;
  if (item < items) issue(item);
  while (item < items) {
    // commit the staged item: K rows, V transposed (a thread writes eight 8-byte runs, one per channel; consecutive lanes take consecutive
    // token quads, so a wave's writes to a channel row are contiguous: conflict free)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + NT * j, t = i >> 3, c = i & 7;
      *reinterpret_cast<uint4*>(sK + t * ATT_LDK + c * 8) = kr[j];
    }
    {
      const uint32_t w[4][4] = {{vr4[0].x, vr4[0].y, vr4[0].z, vr4[0].w}, {vr4[1].x, vr4[1].y, vr4[1].z, vr4[1].w},
                                {vr4[2].x, vr4[2].y, vr4[2].z, vr4[2].w}, {vr4[3].x, vr4[3].y, vr4[3].z, vr4[3].w}};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint16_t* row = sVt + (vc * 8 + 2 * j) * LDV + vtq * 4;
        *reinterpret_cast<uint2*>(row) = make_uint2(__builtin_amdgcn_perm(w[1][j], w[0][j], 0x05040100u),
                                                    __builtin_amdgcn_perm(w[3][j], w[2][j], 0x05040100u));
        *reinterpret_cast<uint2*>(row + LDV) = make_uint2(__builtin_amdgcn_perm(w[1][j], w[0][j], 0x07060302u),
                                                          __builtin_amdgcn_perm(w[3][j], w[2][j], 0x07060302u));
      }
    }
    bf16x8 bq[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) bq[kb] = *reinterpret_cast<bf16x8*>(&qr[kb]);
    const int b = item / H, h = item - b * H;
    __syncthreads();
    const int next = item + (int)gridDim.x;
    if (next < items) issue(next);
    f32x16 sacc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(sK + (kt * 32 + l31) * ATT_LDK + kb * 16 + hh * 8);
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq[kb], sacc[kt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);          // keep the K fragments of later key tiles out of the register file
    }
    // sacc[kt][r] = q . k for key kt*32 + (r&3) + 8*(r>>2) + 4*hh of query q: the soft-max statistics of a query live in one lane pair.
    // The soft-max is the VALU-bound part of the kernel (112 scores per lane), so it is kept to max (v_max3), one fma folding the
    // 1/sqrt(d) log2(e) scale into the exponent, v_exp, one add and the bf16 pack per score.
    float m = -INFINITY;
    // keys past the sequence must not win the maximum: only the LAST key tile can hold any (the host picks NKT = ceil(T / 32))
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((NKT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= T) sacc[NKT - 1][r] = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sacc[kt][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mneg = -m * scale_log2e;            // scale > 0: max(s) * scale == max(s * scale)
    // un-normalised probabilities, packed to bf16 as the B operand of O^T = V^T . P^T (the MFMA's k index is only a summation label:
    // V^T is read from LDS in the key order the accumulator registers already have); O is scaled by 1 / sum at the end
    float sum = 0.f;
    uint32_t pk[NKT][8];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float e0 = __builtin_amdgcn_exp2f(fmaf(sacc[kt][2 * j], scale_log2e, mneg));
        const float e1 = __builtin_amdgcn_exp2f(fmaf(sacc[kt][2 * j + 1], scale_log2e, mneg));
        pk[kt][j] = pack_bf16x2(e0, e1);
        sum += e0 + e1;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    __builtin_amdgcn_sched_barrier(0);
    f32x16 o[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[nt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        uint4 pv = make_uint4(pk[kt][4 * kb2], pk[kt][4 * kb2 + 1], pk[kt][4 * kb2 + 2], pk[kt][4 * kb2 + 3]);
        const bf16x8 pb = *reinterpret_cast<bf16x8*>(&pv);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const uint16_t* vr = sVt + (nt * 32 + l31) * LDV + kt * 32 + 16 * kb2 + 4 * hh;
          const uint2 lo = *reinterpret_cast<const uint2*>(vr), hi = *reinterpret_cast<const uint2*>(vr + 8);
          uint4 av = make_uint4(lo.x, lo.y, hi.x, hi.y);
          o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&av), pb, o[nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    // o[nt][r] = O[query q][d = nt*32 + (r&3) + 8*(r>>2) + 4*hh]: a lane owns runs of four channels of ITS query -> 8-byte stores
    if (q < T) {
      uint16_t* orow = att + ((size_t)b * T + q) * D + h * ATT_HD;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint2*>(orow + nt * 32 + 8 * g + 4 * hh) =
              make_uint2(pack_bf16x2(o[nt][4 * g] * inv, o[nt][4 * g + 1] * inv), pack_bf16x2(o[nt][4 * g + 2] * inv, o[nt][4 * g + 3] * inv));
    }
    __syncthreads();                                // every wave is done with this item's K / V before the next one is committed
    item = next;
  }
}
// ---- fused attention backward (head_dim 64, tokens <= NKT*32), one workgroup per (image, head) ----------------------
// dQ, dK, dV of softmax(Q K^T / sqrt(d)) V from Q, K, V, O (the forward's output) and dO; everything stays in LDS / registers (the unfused path
// writes ~2 GB of score-sized temporaries per layer at B = 256).  Two phases over the same LDS-resident operands:
//   B (a wave owns 32 QUERIES): S^T = K Q^T and dP^T = V dO^T land with lane = query, so the row statistics
//     (max, 1/sum; delta = rowsum(dO * O)) need one shuffle; dS feeds straight back as the A operand of dQ = dS K (K read
//     transposed from LDS in the accumulator's key order, the forward kernel's P.V trick); stats go to LDS.
//   A (a wave owns 32 KEYS): S = Q K^T and dP = dO V^T are recomputed with lane = key (statistics broadcast from LDS),
//     so P^T and dS^T are directly the A operands of dV = P^T dO and dK = dS^T Q (dO, Q read transposed from LDS);
//     dK / dV accumulate in registers over the query tiles.  7 small products instead of 5, no cross-wave reduction.
// NW waves per workgroup: with 8 waves each wave owns ONE query tile in phase B and ONE key tile in phase A (7 tiles),
// so a workgroup walks each phase once instead of twice (one workgroup per CU either way: 158 KB of LDS)
template <int NKT, int NW>
__global__ __launch_bounds__(NW * 64, 1) void k_vit_attention_bwd(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ oout,
                                                                 const uint16_t* __restrict__ dout, uint16_t* __restrict__ dqkv, int T, int H, int ld, int D,
                                                                 float scale, float scale_log2e) {
  constexpr int TP = NKT * 32, LDV = TP + 4;
  __shared__ __attribute__((aligned(16))) uint16_t sQ[TP * ATT_LDK];
  __shared__ __attribute__((aligned(16))) uint16_t sK[TP * ATT_LDK];     // phase A: Q^T  [64][LDV]
  __shared__ __attribute__((aligned(16))) uint16_t sV[TP * ATT_LDK];     // phase A: dO^T [64][LDV]
  __shared__ __attribute__((aligned(16))) uint16_t sdO[TP * ATT_LDK];
  __shared__ __attribute__((aligned(16))) uint16_t sKt[ATT_HD * LDV];
  __shared__ __attribute__((aligned(16))) float4 sStat[TP];              // per query: max (log2 units), 1 / sum, delta, -
  static_assert(ATT_HD * LDV <= TP * ATT_LDK, "transposed operands reuse the K / V arrays");
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
  const uint16_t* base = qkv + (size_t)b * T * ld + h * ATT_HD;
  const uint16_t* dbase = dout + (size_t)b * T * D + h * ATT_HD;
  constexpr int NT = NW * 64, KI = (NKT + NW - 1) / NW;      // threads; key tiles per wave in phase A
  for (int i = tid; i < TP * 8; i += NT) {
    const int t = i >> 3, c = i & 7;
    uint4 qv = make_uint4(0, 0, 0, 0), kv = qv, vv = qv, dv = qv;
    if (t < T) {
      qv = *reinterpret_cast<const uint4*>(base + (size_t)t * ld + c * 8);
      kv = *reinterpret_cast<const uint4*>(base + (size_t)t * ld + D + c * 8);
      vv = *reinterpret_cast<const uint4*>(base + (size_t)t * ld + 2 * D + c * 8);
      dv = *reinterpret_cast<const uint4*>(dbase + (size_t)t * D + c * 8);
    }
    *reinterpret_cast<uint4*>(sQ + t * ATT_LDK + c * 8) = qv;
    *reinterpret_cast<uint4*>(sK + t * ATT_LDK + c * 8) = kv;
    *reinterpret_cast<uint4*>(sV + t * ATT_LDK + c * 8) = vv;
    *reinterpret_cast<uint4*>(sdO + t * ATT_LDK + c * 8) = dv;
  }
  __syncthreads();
  // transposed copy [channel][token] of a token-major LDS array: a thread turns 8 channels of FOUR consecutive tokens into eight 8-byte
  // runs; consecutive lanes take consecutive token quads, so the writes to a channel row are contiguous (the forward kernel's staging)
  auto transpose_rows = [&](const uint16_t* src, uint16_t* dst) {
    for (int i = tid; i < (TP / 4) * 8; i += NT) {
      const int tq = i % (TP / 4), c = i / (TP / 4);
      uint32_t w[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint4 v = *reinterpret_cast<const uint4*>(src + (tq * 4 + u) * ATT_LDK + c * 8);
        w[u][0] = v.x; w[u][1] = v.y; w[u][2] = v.z; w[u][3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint16_t* row = dst + (c * 8 + 2 * j) * LDV + tq * 4;
        *reinterpret_cast<uint2*>(row) = make_uint2(__builtin_amdgcn_perm(w[1][j], w[0][j], 0x05040100u),
                                                    __builtin_amdgcn_perm(w[3][j], w[2][j], 0x05040100u));
        *reinterpret_cast<uint2*>(row + LDV) = make_uint2(__builtin_amdgcn_perm(w[1][j], w[0][j], 0x07060302u),
                                                          __builtin_amdgcn_perm(w[3][j], w[2][j], 0x07060302u));
      }
    }
  };
  transpose_rows(sK, sKt);
  __syncthreads();
  uint16_t* gq = dqkv + (size_t)b * T * ld + h * ATT_HD;      // dQ | dK (+D) | dV (+2D), same layout as qkv

  // ---------------- phase B: queries ----------------
  for (int qt = wave; qt < NKT; qt += NW) {
    const int q = qt * 32 + l31;
    bf16x8 bq[4], bdo[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      bq[kb] = *reinterpret_cast<const bf16x8*>(sQ + q * ATT_LDK + kb * 16 + hh * 8);
      bdo[kb] = *reinterpret_cast<const bf16x8*>(sdO + q * ATT_LDK + kb * 16 + hh * 8);
    }
    // delta_q = sum_k P dP = sum_d dO[q][d] O[q][d] (O = the forward's output row): known before any dP tile exists, so
    // the dP^T tiles can be consumed one at a time instead of holding all 7 (112 more registers) next to S^T
    float delta = 0.f;
    if (q < T) {
      const uint16_t* orow = oout + ((size_t)b * T + q) * D + h * ATT_HD + hh * 32;
      const uint16_t* drow = sdO + q * ATT_LDK + hh * 32;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint4 ov = *reinterpret_cast<const uint4*>(orow + c * 8), dv = *reinterpret_cast<const uint4*>(drow + c * 8);
        const uint32_t ow[4] = {ov.x, ov.y, ov.z, ov.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          delta += bf2f((uint16_t)(ow[j] & 0xFFFF)) * bf2f((uint16_t)(dw[j] & 0xFFFF));
          delta += bf2f((uint16_t)(ow[j] >> 16)) * bf2f((uint16_t)(dw[j] >> 16));
        }
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
        const bf16x8 ak = *reinterpret_cast<const bf16x8*>(sK + (kt * 32 + l31) * ATT_LDK + kb * 16 + hh * 8);
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, bq[kb], sacc[kt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // lane = query qt*32 + l31; register r of tile kt = key kt*32 + (r&3) + 8*(r>>2) + 4*hh.  Only the LAST key tile can hold keys past
    // the sequence (the host picks NKT = ceil(T / 32)): they must not win the maximum and get probability 0
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((NKT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= T) sacc[NKT - 1][r] = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, sacc[kt][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64)) * scale_log2e;       // scale > 0: max(s) * scale == max(s * scale)
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
    if (hh == 0) sStat[q] = make_float4(m, inv, delta, 0.f);
    const float sinv = scale * inv;
    // dQ^T = K^T . dS^T (operands swapped like the forward's O^T): a lane owns runs of four channels of ITS query
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
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(sV + (kt * 32 + l31) * ATT_LDK + kb * 16 + hh * 8);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bdo[kb], dp, 0, 0, 0);
      }
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        uint32_t pw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r0 = 8 * kb2 + 2 * j;
          pw[j] = pack_bf16x2(sinv * sacc[kt][r0] * (dp[r0] - delta), sinv * sacc[kt][r0 + 1] * (dp[r0 + 1] - delta));
        }
        uint4 pv = make_uint4(pw[0], pw[1], pw[2], pw[3]);
        const bf16x8 ds = *reinterpret_cast<bf16x8*>(&pv);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const uint16_t* kr = sKt + (nt * 32 + l31) * LDV + kt * 32 + 16 * kb2 + 4 * hh;
          const uint2 lo = *reinterpret_cast<const uint2*>(kr), hi = *reinterpret_cast<const uint2*>(kr + 8);
          uint4 av = make_uint4(lo.x, lo.y, hi.x, hi.y);
          dq[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&av), ds, dq[nt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);     // one key tile at a time: interleaving the unrolled tiles multiplies the live dP tiles
    }
    // dq[nt][r] = dQ[query q][d = nt*32 + (r&3) + 8*(r>>2) + 4*hh]
    if (q < T) {
      uint16_t* orow = gq + (size_t)q * ld;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint2*>(orow + nt * 32 + 8 * g + 4 * hh) =
              make_uint2(pack_bf16x2(dq[nt][4 * g], dq[nt][4 * g + 1]), pack_bf16x2(dq[nt][4 * g + 2], dq[nt][4 * g + 3]));
    }
  }

  // ---------------- phase A: keys ----------------
  // K / V fragments of the (up to two) key tiles this wave owns move to registers, then K's and V's LDS arrays are
  // overwritten by Q^T and dO^T
  bf16x8 bk[KI][4], bvv[KI][4];
#pragma unroll
  for (int ki = 0; ki < KI; ++ki) {
    const int kt = wave + ki * NW;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      uint4 z = make_uint4(0, 0, 0, 0);
      bk[ki][kb] = *reinterpret_cast<bf16x8*>(&z);
      bvv[ki][kb] = *reinterpret_cast<bf16x8*>(&z);
      if (kt < NKT) {
        bk[ki][kb] = *reinterpret_cast<const bf16x8*>(sK + (kt * 32 + l31) * ATT_LDK + kb * 16 + hh * 8);
        bvv[ki][kb] = *reinterpret_cast<const bf16x8*>(sV + (kt * 32 + l31) * ATT_LDK + kb * 16 + hh * 8);
      }
    }
  }
  __syncthreads();
  uint16_t* sQt = sK;
  uint16_t* sdOt = sV;
  transpose_rows(sQ, sQt);
  transpose_rows(sdO, sdOt);
  __syncthreads();
#pragma unroll
  for (int ki = 0; ki < KI; ++ki) {
    const int kt = wave + ki * NW;
    if (kt >= NKT) break;
    const int key = kt * 32 + l31;
    const bool key_ok = key < T;
    // dV^T = dO^T . P and dK^T = Q^T . dS (operands swapped): lane = key, registers = channels
    f32x16 dvv[2], dkk[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dvv[nt][r] = dkk[nt][r] = 0.f;
#pragma unroll 1
    for (int qt = 0; qt < NKT; ++qt) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const bf16x8 aq = *reinterpret_cast<const bf16x8*>(sQ + (qt * 32 + l31) * ATT_LDK + kb * 16 + hh * 8);
        const bf16x8 ad = *reinterpret_cast<const bf16x8*>(sdO + (qt * 32 + l31) * ATT_LDK + kb * 16 + hh * 8);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, bk[ki][kb], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, bvv[ki][kb], dp, 0, 0, 0);
      }
      // lane = key kt*32 + l31; register r = query qt*32 + (r&3) + 8*(r>>2) + 4*hh (queries past the sequence have zero Q / dO rows:
      // whatever probability they get multiplies zeros)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 st = sStat[qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
        const float p = key_ok ? __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -st.x)) * st.y : 0.f;
        s[r] = p;
        dp[r] = scale * p * (dp[r] - st.z);
      }
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        uint32_t pw[4], dw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r0 = 8 * kb2 + 2 * j;
          pw[j] = pack_bf16x2(s[r0], s[r0 + 1]);
          dw[j] = pack_bf16x2(dp[r0], dp[r0 + 1]);
        }
        uint4 pv = make_uint4(pw[0], pw[1], pw[2], pw[3]), dv = make_uint4(dw[0], dw[1], dw[2], dw[3]);
        const bf16x8 pb = *reinterpret_cast<bf16x8*>(&pv), db = *reinterpret_cast<bf16x8*>(&dv);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const uint16_t* orow = sdOt + (nt * 32 + l31) * LDV + qt * 32 + 16 * kb2 + 4 * hh;
          const uint16_t* qrow = sQt + (nt * 32 + l31) * LDV + qt * 32 + 16 * kb2 + 4 * hh;
          const uint2 olo = *reinterpret_cast<const uint2*>(orow), ohi = *reinterpret_cast<const uint2*>(orow + 8);
          const uint2 qlo = *reinterpret_cast<const uint2*>(qrow), qhi = *reinterpret_cast<const uint2*>(qrow + 8);
          uint4 ao = make_uint4(olo.x, olo.y, ohi.x, ohi.y), aqv = make_uint4(qlo.x, qlo.y, qhi.x, qhi.y);
          dvv[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&ao), pb, dvv[nt], 0, 0, 0);
          dkk[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&aqv), db, dkk[nt], 0, 0, 0);
        }
      }
    }
    // dkk / dvv[nt][r] = dK / dV[key][d = nt*32 + (r&3) + 8*(r>>2) + 4*hh]: 8-byte runs of the lane's own key row
    if (key_ok) {
      uint16_t* orow = gq + (size_t)key * ld;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          *reinterpret_cast<uint2*>(orow + D + nt * 32 + 8 * g + 4 * hh) =
              make_uint2(pack_bf16x2(dkk[nt][4 * g], dkk[nt][4 * g + 1]), pack_bf16x2(dkk[nt][4 * g + 2], dkk[nt][4 * g + 3]));
          *reinterpret_cast<uint2*>(orow + 2 * D + nt * 32 + 8 * g + 4 * hh) =
              make_uint2(pack_bf16x2(dvv[nt][4 * g], dvv[nt][4 * g + 1]), pack_bf16x2(dvv[nt][4 * g + 2], dvv[nt][4 * g + 3]));
        }
    }
  }
}
}  // namespace

extern "C" {

int rart_vit_patchify(const void* src, int src_is_u8, void* hi, void* lo, int n, int h, int w, int patch,
                      const float* mean_host, const float* std_host, rart_stream_t stream) {
  RART_CHECK_ARG(src && hi && lo && n > 0 && patch > 0 && h % patch == 0 && w % patch == 0, "rart_vit_patchify: bad arguments");
  RART_CHECK_ARG(patch % 8 == 0 && (size_t)n * (h / patch) * (w / patch) * 3 * patch * patch / 8 < (1ull << 32),
                 "rart_vit_patchify: the patch side must be a multiple of 8 (eight pixels per thread) and the batch below 2^32 vectors");
  Norm3 nm;
  for (int c = 0; c < 3; ++c) {
    nm.mean[c] = mean_host ? mean_host[c] : 0.f;
    nm.istd[c] = std_host ? 1.0f / std_host[c] : 1.f;
  }
  const size_t total = (size_t)n * (h / patch) * (w / patch) * 3 * patch * patch / 8;
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
  RART_CHECK_ARG(x && cls_pos0 && pos && n > 0 && tokens > 0 && dim > 0 && dim % 8 == 0 && (size_t)n * tokens * dim / 8 < (1ull << 32),
                 "rart_vit_add_pos_cls: bad arguments (dim must be a multiple of 8)");
  hipLaunchKernelGGL(k_add_pos_cls, dim3(grid_for((size_t)n * tokens * dim / 8)), dim3(kBlock), 0, (hipStream_t)stream,
                     (uint16_t*)x, cls_pos0, pos, n, tokens, dim);
  RART_CHECK_LAUNCH("rart_vit_add_pos_cls");
  return RART_OK;
}

int rart_layernorm_bf16(const void* x, const float* gamma, const float* beta, void* out, int rows, int dim,
                        int64_t in_row_stride, int64_t out_row_stride, float eps, rart_stream_t stream) {
  RART_CHECK_ARG(x && gamma && beta && out && rows > 0 && dim > 0, "rart_layernorm_bf16: bad arguments");
  RART_CHECK_ARG(dim % 8 == 0 && dim <= 1024 && in_row_stride % 8 == 0 && out_row_stride % 8 == 0,
                 "rart_layernorm_bf16: dim must be a multiple of 8, at most 1024; row strides multiples of 8");
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
  RART_CHECK_ARG(ld_in % 8 == 0 && ld_out % 8 == 0 && ld_in <= 1024 && ld_out <= 1024,
                 "rart_softmax_rows_bf16: leading dimensions must be multiples of 8, at most 1024");
  hipLaunchKernelGGL(k_softmax_rows, dim3((uint32_t)((rows + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                     (hipStream_t)stream, (const uint16_t*)scores, (uint16_t*)probs, (long long)rows, n_valid, ld_in,
                     ld_out, scale);
  RART_CHECK_LAUNCH("rart_softmax_rows_bf16");
  return RART_OK;
}

int rart_vit_attention(const void* qkv, void* out, int n, int tokens, int heads, int head_dim, rart_stream_t stream) {
  RART_CHECK_ARG(qkv && out && n > 0 && tokens > 0 && heads > 0, "rart_vit_attention: bad arguments");
  RART_CHECK_ARG(head_dim == 64, "rart_vit_attention: head_dim must be 64 (ViT-B/16)");
  RART_CHECK_ARG(tokens <= 224, "rart_vit_attention: at most 224 tokens (197 for 224x224 / patch 16)");
  const int D = heads * head_dim;
  const float scale_log2e = (1.0f / sqrtf((float)head_dim)) * 1.4426950408889634f;
  const int items = n * heads;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t pr;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
  }
  const dim3 grid((uint32_t)(items < n_cu ? items : n_cu));          // one persistent workgroup per CU (61 KB of LDS, up to 7 waves)
  hipStream_t st = (hipStream_t)stream;
  const uint16_t* q = (const uint16_t*)qkv;
  uint16_t* o = (uint16_t*)out;
#define RART_ATT_CASE(N) case N: hipLaunchKernelGGL(k_vit_attention<N>, grid, dim3(N * 64), 0, st, q, o, tokens, heads, 3 * D, D, scale_log2e, items); break;
  switch ((tokens + 31) / 32) {                   // key tiles: only the last one is partial
    RART_ATT_CASE(1) RART_ATT_CASE(2) RART_ATT_CASE(3) RART_ATT_CASE(4) RART_ATT_CASE(5) RART_ATT_CASE(6)
    default: hipLaunchKernelGGL(k_vit_attention<7>, grid, dim3(7 * 64), 0, st, q, o, tokens, heads, 3 * D, D, scale_log2e, items); break;
  }
#undef RART_ATT_CASE
  RART_CHECK_LAUNCH("rart_vit_attention");
  return RART_OK;
}

int rart_vit_attention_bwd(const void* qkv, const void* out, const void* dout, void* dqkv, int n, int tokens, int heads,
                           int head_dim, rart_stream_t stream) {
  RART_CHECK_ARG(qkv && out && dout && dqkv && n > 0 && tokens > 0 && heads > 0, "rart_vit_attention_bwd: bad arguments");
  RART_CHECK_ARG(head_dim == 64, "rart_vit_attention_bwd: head_dim must be 64 (ViT-B/16)");
  RART_CHECK_ARG(tokens <= 224, "rart_vit_attention_bwd: at most 224 tokens (197 for 224x224 / patch 16)");
  const int D = heads * head_dim;
  const float scale = 1.0f / sqrtf((float)head_dim);
  const dim3 grid((uint32_t)(n * heads));
  hipStream_t st = (hipStream_t)stream;
#define RART_ATTB_CASE(N) case N: hipLaunchKernelGGL((k_vit_attention_bwd<N, 8>), grid, dim3(512), 0, st, (const uint16_t*)qkv, (const uint16_t*)out, \
    (const uint16_t*)dout, (uint16_t*)dqkv, tokens, heads, 3 * D, D, scale, scale * 1.4426950408889634f); break;
  switch ((tokens + 31) / 32) {                   // key tiles: only the last one is partial
    RART_ATTB_CASE(1) RART_ATTB_CASE(2) RART_ATTB_CASE(3) RART_ATTB_CASE(4) RART_ATTB_CASE(5) RART_ATTB_CASE(6)
    default: hipLaunchKernelGGL((k_vit_attention_bwd<7, 8>), grid, dim3(512), 0, st, (const uint16_t*)qkv, (const uint16_t*)out,
                                (const uint16_t*)dout, (uint16_t*)dqkv, tokens, heads, 3 * D, D, scale, scale * 1.4426950408889634f); break;
  }
#undef RART_ATTB_CASE
  RART_CHECK_LAUNCH("rart_vit_attention_bwd");
  return RART_OK;
}

int rart_vit_transpose_v(const void* qkv, void* vt, int n, int tokens, int heads, int head_dim, int qkv_ld, int v_off,
                         int t_pad, rart_stream_t stream) {
  RART_CHECK_ARG(qkv && vt && n > 0 && tokens > 0 && t_pad >= tokens, "rart_vit_transpose_v: bad arguments");
  RART_CHECK_ARG(head_dim == 64 && t_pad % 8 == 0 && qkv_ld % 8 == 0 && v_off % 8 == 0,
                 "rart_vit_transpose_v: head_dim must be 64; t_pad, qkv_ld and v_off multiples of 8");
  hipLaunchKernelGGL(k_transpose_v, dim3((t_pad + 63) / 64, n * heads), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint16_t*)qkv, (uint16_t*)vt, n, tokens, heads, head_dim, qkv_ld, v_off, t_pad);
  RART_CHECK_LAUNCH("rart_vit_transpose_v");
  return RART_OK;
}

}  // extern "C"
