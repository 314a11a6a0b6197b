// Device-side pieces shared by the split-bf16 ("fp32x") GEMM kernels: csrc/gemm_pair.hip (k_gemm_pair) and csrc/conv3x3_res_pair.hip
// (k_conv3x3_res_pair, the 3x3 convolution with the input chunk resident in LDS): the device descriptor, the hi + lo helpers and the
// epilogue (per wave, 32 rows x 64 columns at a time through LDS: bias is already in the accumulators; residual PAIR, GELU forms, 1-bit
// masks and sign bits, ReLU, then hi / lo as two coalesced 16-byte stores, or fp32).  Moved out of gemm_pair.hip in round 5, unchanged.
#pragma once
#include "rart_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {
enum { GP_OUT_F32 = 2, GP_GELU = 4, GP_GELU_BWD = 8, GP_GELU_KEEP = 64, GP_RELU = 1, GP_W_INTERLEAVED = 16 };
constexpr int GP_BK = 32;
constexpr int GP_LDE = 68;                         // epilogue staging row (floats): 64 columns + 4

struct GemmPairDev {
  const uint16_t *a_hi, *a_lo, *w_hi, *w_lo;
  const float* bias;
  const uint16_t *res_hi, *res_lo;
  uint16_t *dst_hi, *dst_lo;
  uint16_t *aux_hi, *aux_lo;
  int M, N, K, lda, ldw, ldc, w_rows;
  int rpi, src_rpi, src_off, dst_rpi, dst_off, map_rows;
  int flags, z_inner;
  long long a_zo, a_zi, w_zo, w_zi, c_zo, c_zi;
  // CONV instances: row grid, source / destination geometry, taps (k_per_tap / 32 a power of two: tap = kt >> tpt_shift)
  int grid_h, grid_w, src_h, src_w, sy, sx, n_taps, tpt_shift;
  int tap_dy[16], tap_dx[16];
  int dst_h, dst_w, dst_sy, dst_sx, dst_oy, dst_ox;
  uint32_t gw_magic, gw_shift, gh_magic, gh_shift;      // exact division by multiply-shift for dividends < 2^31
  const uint8_t* mask_bits;                              // 1 bit per destination element (byte (off + col) / 8): v = 0 where clear
  uint8_t* sign_out;                                     // receives (output hi plane > 0), same indexing
  int m_begin;                                           // k_gemm_pair_pp only: the launch covers rows m_begin .. M - 1 (0 elsewhere)
  int tile_rows;                                         // k_gemm_pair_pp only: 256, or 224 = the last 32-row block of a tile is left out
  int nt;                                                // the epilogue's streams (residual / pre-activation loads, pair stores) are non-temporal
};
__device__ __forceinline__ uint32_t gp_fastdiv(uint32_t n, uint32_t magic, uint32_t shift) {
  return (uint32_t)(((uint64_t)n * magic) >> shift);
}

__device__ __forceinline__ uint32_t gp_pack_bf16x2(float lo, float hi) {   // round to nearest even (v_cvt_pk_bf16_f32)
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
  f2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2_t));
}
// exact (erf) GELU, timm's nn.GELU default, at fp32 accuracy without libm's erff.  Round 6: the per-workgroup trace of the ping-pong
// kernel (scratch/r6/vit_pp_trace.py) put the epilogue of ViT's fc1 tiles at 50 k cycles (forward, GELU) and 74 k (backward, GELU') against
// 15 k for a plain tile and 95 k for the whole K loop -- erff (and expf) of libm are ~45 instructions with divergent range branches, 128
// elements per lane, two waves per SIMD.  Here: 0.5 erfc(x) = t q(t) exp(-x^2), t = 1 / (1 + x / 2), q a degree-7 polynomial fitted over
// x in [0, 6.6] (scratch/r6/fit_gelu_erfc.py: |erfc error| <= 2.5e-10 before rounding); Phi(u) = 0.5 erfc(|x|) for u < 0 -- no
// cancellation in the tail -- and 1 - that for u >= 0.  In fp32 against fp64 over [-12, 12]: gelu max |error| 3.8e-7 (the libm form
// 4.5e-7), relative L2 1.18e-8 (1.31e-8); gelu' max |error| 2.1e-7 (1.4e-7).  The exponential is shared with gelu'.
__device__ __forceinline__ float gp_half_erfc_abs(float u, float& e) {     // 0.5 erfc(|u| / sqrt 2); e <- exp(-u^2 / 2)
  const float x = fabsf(u) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, x, 1.0f));
  float q = fmaf(-2.548219442e-02f, t, 1.698278524e-01f);
  q = fmaf(q, t, -4.384240636e-01f);
  q = fmaf(q, t, 4.869355765e-01f);
  q = fmaf(q, t, -1.740313807e-01f);
  q = fmaf(q, t, 2.156719095e-01f);
  q = fmaf(q, t, 1.229157723e-01f);
  q = fmaf(q, t, 1.425865282e-01f);
  e = __builtin_amdgcn_exp2f(x * x * -1.4426950408889634f);
  return q * t * e;
}
__device__ __forceinline__ float gp_gelu(float v) {
  float e;
  const float hc = gp_half_erfc_abs(v, e);
  return v * (v < 0.f ? hc : 1.0f - hc);
}
// d/du [u * Phi(u)] = Phi(u) + u * phi(u)
__device__ __forceinline__ float gp_gelu_grad(float u) {
  float e;
  const float hc = gp_half_erfc_abs(u, e);
  return fmaf(u * 0.3989422804014327f, e, u < 0.f ? hc : 1.0f - hc);
}
__device__ __forceinline__ void gp_split8(const float* v, uint4& hi, uint4& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = gp_pack_bf16x2(v[2 * j], v[2 * j + 1]);
    l[j] = gp_pack_bf16x2(v[2 * j] - __uint_as_float(h[j] << 16), v[2 * j + 1] - __uint_as_float(h[j] & 0xFFFF0000u));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}
// Round 6: the epilogue's residual / pre-activation loads and its pair stores are NON-TEMPORAL accesses (streams of 100-800 MB per launch that
// the producing kernel never touches again and that do not fit the 4 MB L2 of an XCD): reference-precision ResNet-50 gradient evaluation
// 19.50 -> 19.13 ms, ViT-B/16 65.73 -> 65.09 ms (same-box A/B of lab builds, together with the tail kernel's streams).
typedef __attribute__((ext_vector_type(4))) uint32_t gp_u4;
__device__ __forceinline__ uint4 gp_nt_load(const uint16_t* p) {
  const gp_u4 v = __builtin_nontemporal_load(reinterpret_cast<const gp_u4*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void gp_nt_store(uint16_t* p, const uint4& v) {
  gp_u4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<gp_u4*>(p));
}
// hi + lo is exact in fp32 (two 8-bit significands, lo below half an ulp of hi)
__device__ __forceinline__ void gp_join8(const uint4& hi, const uint4& lo, float* v) {
  const uint32_t h[4] = {hi.x, hi.y, hi.z, hi.w}, l[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[2 * j] = __uint_as_float(h[j] << 16) + __uint_as_float(l[j] << 16);
    v[2 * j + 1] = __uint_as_float(h[j] & 0xFFFF0000u) + __uint_as_float(l[j] & 0xFFFF0000u);
  }
}


// The point-wise tail of one 8-column row segment (element offset e, CONV or plain): v = the fp32 accumulator values of the segment; residual
// pair, GELU forms, 1-bit mask, ReLU, then hi / lo (two 16-byte stores) or fp32, and the sign byte.  Shared by gp_epilogue and the
// persistent kernel's deferred epilogue (csrc/gemm_pair_pp.hip) so that both produce the same bits.
template <bool CONV, bool GELU = true>
__device__ __forceinline__ void gp_finish_segment(const GemmPairDev& d, int flags, bool out_f32, long long e, float (&v)[8], const uint4& rh_q,
                                            const uint4& rl_q, const uint4& uh_q, const uint4& ul_q, uint32_t mb_q) {
  if (d.res_hi) {
    float rv[8];
    gp_join8(rh_q, rl_q, rv);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += rv[j];
  }
  if (GELU && (flags & GP_GELU)) {     // gelu of the value the PAIR of the pre-activation represents: the same result as GELU_KEEP's, so a forward-only
    uint4 ph, pl;            // evaluation and the forward of a gradient evaluation agree bit for bit
    gp_split8(v, ph, pl);
    gp_join8(ph, pl, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = gp_gelu(v[j]);
  }
  if (GELU && (flags & GP_GELU_KEEP)) {
    // two outputs: the pre-activation pair u goes to `aux` (the backward's GELU' operand); dst receives gelu of the value the
    // pair REPRESENTS, so forward and backward see the same u
    uint4 ph, pl;
    gp_split8(v, ph, pl);
    gp_nt_store(d.aux_hi + e, ph);
    gp_nt_store(d.aux_lo + e, pl);
    gp_join8(ph, pl, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = gp_gelu(v[j]);
  }
  if (GELU && (flags & GP_GELU_BWD)) {
    float u[8];
    gp_join8(uh_q, ul_q, u);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= gp_gelu_grad(u[j]);
  }
  if (CONV) {      // 1-bit ReLU mask of the destination (backward-to-input): bit k of the byte = column col + k
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!((mb_q >> j) & 1u)) v[j] = 0.f;
  }
  if (flags & GP_RELU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
  }
  if (out_f32) {
    float* o = reinterpret_cast<float*>(d.dst_hi) + e;
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint4 ph, pl;
    gp_split8(v, ph, pl);
    if (d.nt) {
      gp_nt_store(d.dst_hi + e, ph);
      gp_nt_store(d.dst_lo + e, pl);
    } else {
      *reinterpret_cast<uint4*>(d.dst_hi + e) = ph;
      *reinterpret_cast<uint4*>(d.dst_lo + e) = pl;
    }
    if (CONV && d.sign_out) {     // (hi plane > 0): a bf16 is > 0 exactly when its bits, read as int16, are > 0
      const uint32_t hw[4] = {ph.x, ph.y, ph.z, ph.w};
      uint32_t sb = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sb |= ((short)(hw[j] & 0xFFFFu) > 0 ? 1u : 0u) << (2 * j);
        sb |= ((short)(hw[j] >> 16) > 0 ? 1u : 0u) << (2 * j + 1);
      }
      d.sign_out[e >> 3] = (uint8_t)sb;
    }
  }
}

// Epilogue of a TM x TN tile whose accumulators acc[MI][2] follow k_gemm_pair's wave layout (wave = (wm, wn), a wave owns RW rows x 64
// columns).  `lds` is the workgroup's tile buffer (free after the K loop).  Must be called by every thread of the workgroup.
template <int TM, int TN, bool CONV, int MI>
__device__ __forceinline__ void gp_epilogue(const GemmPairDev& d, uint8_t* lds, f32x16 (&acc)[MI][2], int m0, int n0, long long c_off,
                                            int tile_rows = TM) {
  constexpr int NW = TM / 32;
  constexpr int WN = (TN / 64 < NW) ? TN / 64 : NW, WM = NW / WN, RW = TM / WM;
  static_assert(MI == RW / 32, "accumulator blocks per wave");
  constexpr int GP_STAGING = NW * 32 * GP_LDE * 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
  const int fr = lane & 31, h = lane >> 5;
  // ---- epilogue: per wave, 32 rows x 64 columns at a time through LDS -> 128-byte row segments per plane; residual / mask / GELU
  //      operands of a pass are requested before its transposition
  float* sE = reinterpret_cast<float*>(lds) + wave * 32 * GP_LDE;
  uint32_t* const row_dst = reinterpret_cast<uint32_t*>(lds + GP_STAGING);      // CONV: destination element offset of tile row r, ~0 past M
  if (CONV) {
    if (tid < TM) {
      const uint32_t m = (uint32_t)(m0 + tid);
      uint32_t off = 0xFFFFFFFFu;
      if (m < (uint32_t)d.M) {
        const uint32_t t = gp_fastdiv(m, d.gw_magic, d.gw_shift);
        const int ox = (int)(m - t * (uint32_t)d.grid_w);
        const int n = (int)gp_fastdiv(t, d.gh_magic, d.gh_shift);
        const int oy = (int)(t - (uint32_t)n * (uint32_t)d.grid_h);
        off = (uint32_t)(((n * d.dst_h + (oy * d.dst_sy + d.dst_oy)) * d.dst_w + (ox * d.dst_sx + d.dst_ox)) * d.ldc);
      }
      row_dst[tid] = off;
    }
    __syncthreads();
  }
  const int cw = lane & 7, rw = lane >> 3;
  const int col = n0 + wn * 64 + cw * 8;
  const bool col_ok = col < d.N;
  const int flags = d.flags;
  const bool out_f32 = flags & GP_OUT_F32;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    if (wm * RW + i * 32 >= tile_rows) continue;          // (wave-uniform) a 224-row tile of k_gemm_pair_pp: the block belongs to the next tile
    long long eo[4];
    uint4 rh[4], rl[4], uh[4], ul[4];
    uint32_t mb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rt = wm * RW + i * 32 + q * 8 + rw;       // row inside the tile
      const int m = m0 + rt;
      eo[q] = -1;
      rh[q] = rl[q] = uh[q] = ul[q] = make_uint4(0, 0, 0, 0);
      mb[q] = 0xFFu;
      if (m < d.M && col_ok) {
        long long e;
        if (CONV) {
          e = (long long)row_dst[rt] + col;
        } else {
          long long drow = m;
          if (d.map_rows) {
            const int img = m / d.rpi;
            drow = (long long)img * d.dst_rpi + (m - img * d.rpi);
          }
          e = c_off + (drow + d.dst_off) * d.ldc + col;
        }
        eo[q] = e;
        if (d.res_hi) {
          if (d.nt) {
            rh[q] = gp_nt_load(d.res_hi + e);
            rl[q] = gp_nt_load(d.res_lo + e);
          } else {
            rh[q] = *reinterpret_cast<const uint4*>(d.res_hi + e);
            rl[q] = *reinterpret_cast<const uint4*>(d.res_lo + e);
          }
        }
        if (flags & GP_GELU_BWD) {
          uh[q] = gp_nt_load(d.aux_hi + e);
          ul[q] = gp_nt_load(d.aux_lo + e);
        }
        if (CONV && d.mask_bits) mb[q] = d.mask_bits[e >> 3];
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sE[((r & 3) + 8 * (r >> 2) + 4 * h) * GP_LDE + j * 32 + fr] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = q * 8 + rw;
      const float4 v0 = *reinterpret_cast<const float4*>(sE + r * GP_LDE + cw * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(sE + r * GP_LDE + cw * 8 + 4);
      const long long e = eo[q];
      if (e >= 0) {
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        gp_finish_segment<CONV>(d, flags, out_f32, e, v, rh[q], rl[q], uh[q], ul[q], mb[q]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}
}  // namespace
