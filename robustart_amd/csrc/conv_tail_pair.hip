// 3x3 convolution + the 1x1 expansion that follows it, as ONE kernel on split-bf16 pairs (gfx950): the second half of a ResNet
// Bottleneck of the REFERENCE-PRECISION engine, forward and backward-to-input.
//
//     forward :  out = relu(W3 . relu(W2 * a1 + b2) + b3 + skip)                 (+ the 1-bit sign tensors of both results)
//     backward:  dx  = mask_x . (W1^T . (mask_a . (W2^T * d_b)) + g)
//
// Both are "3x3 (C -> C), point-wise step, 1x1 (C -> 4C), add a 4C-channel pair, point-wise step" (C = 64: layer1, 128: layer2), so one
// kernel serves both with flipped / transposed tables; the point-wise steps are bias + ReLU + sign out, or a 1-bit mask in.
//
// Why (profiles/r04_igemm_per_shape_fp32x.txt, B = 256): as two launches of k_gemm_pair the 3x3 is MFMA-bound (254 / 221 us per launch in
// layer1 / layer2) and the expansion HBM-bound (327 / 206 us: it reads the C-channel pair the 3x3 just wrote, the 4C-channel skip pair,
// and writes the 4C-channel pair).  Fused, the C-channel intermediate never exists in HBM (410 / 205 MB per block and direction) and the
// MFMA-bound main loop of one resident workgroup overlaps the HBM-bound stores of another.
//
// Structure: a workgroup owns 128 positions x all C channels; four wave64s, each 32 positions x C channels.  Main loop = k_gemm_pair's
// (csrc/gemm_pair.hip: the four operand planes of a 32-deep K step go global -> LDS with global_load_lds_dwordx4 into a lane-linear,
// XOR-swizzled image, two stages, three MFMAs per fragment pair) with the operand roles SWAPPED -- weights are the MFMA A operand,
// positions the B operand -- so a lane ends up with 4 consecutive channels of ONE position: bias / ReLU / mask / hi + lo split happen in
// registers, one v_permlane32_swap per register pair turns the results into the B-operand fragments of the 1x1 (the intermediate
// stays in registers, as in csrc/bottleneck_fused.hip), and its sign bits leave as whole bytes.  The 1x1's table comes straight from
// L2 in MFMA fragment order (1 KiB contiguous per fragment load), 64 output channels at a time; each 32 x 64 result is transposed
// through the wave's private LDS region so that the skip loads and the stores are 128-byte row segments per plane.
//
// NEXT instances also run the 1x1 REDUCTION of the neighbouring block on the tile they just produced (a 1x1 convolution is
// position-local): forward, conv1 + bias + ReLU of the NEXT block on `out`; backward, conv3^T + mask of the PREVIOUS block on dx.  Each
// 64-channel chunk of the result goes back to the wave's LDS region as a pair of bf16 planes, is read as B-operand fragments and
// multiplied into C more accumulators (table in fragment order from L2); the neighbour's own launch -- which re-read the whole
// 4C-channel pair from HBM (822 MB per block of layer1 at B = 256) to produce a C-channel one -- disappears.
//
// Reference step: Bottleneck.forward of the public ResNet-50 (RobustART/model/__init__.py:1 -> absent submodule;
// robustart_amd/model/resnet_torch.py) and its autograd inside every attack iteration, in fp32
// (RobustART/noise/utils/adv/attack.py:20-23, Attacks/autoattack/autopgd_base.py:271-289).
#include "rart_common.h"
#include "rart_lds_dma.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {
constexpr int CT_TM = 128;                       // positions per workgroup
constexpr int CT_LDE = 68;                       // epilogue staging row (floats): 64 columns + 4
constexpr int CT_PLANE_A = CT_TM * 64;           // one position plane of a stage: 128 rows x 64 B
constexpr int CT_NLD = 144;                      // NEXT: row stride (bytes) of a 32 x 64 bf16 plane of the chunk: ds_read_b128 conflict free
constexpr int CT_WREG = 2 * 32 * CT_NLD;         // a wave's private LDS region (9 216 B >= the 32 x 68 fp32 staging)
static_assert(CT_WREG >= 32 * CT_LDE * 4, "the staging must fit the wave's region");

struct ConvTailDev {
  const uint16_t *a_hi, *a_lo, *w_hi, *w_lo, *t_hi, *t_lo;
  const float *bias_mid, *bias_out;
  const uint8_t *mask_mid, *mask_out;
  uint8_t *sign_mid, *sign_out;
  const uint16_t *res_hi, *res_lo;
  uint16_t *dst_hi, *dst_lo;
  int M, H, W, ldw, relu_mid, relu_out;
  int tap_dy[9], tap_dx[9];
  uint32_t w_magic, w_shift, h_magic, h_shift;
  // NEXT instances: the neighbouring block's 1x1 reduction (4C -> C) of the tile
  const uint16_t *n_hi, *n_lo;      // fragment order: ((chunk * (C / 32) + blk) * 4 + s) * 64 + lane
  const float* bias_next;
  const uint8_t* mask_next;
  uint8_t* sign_next;
  uint16_t *dstn_hi, *dstn_lo;
  int relu_next;
};
// round 6: the skip loads and the output stores are non-temporal accesses (see rart_gemm_pair_dev.h)
typedef __attribute__((ext_vector_type(4))) uint32_t ct_u4;
__device__ __forceinline__ uint4 ct_nt_load(const uint16_t* p) {
  const ct_u4 v = __builtin_nontemporal_load(reinterpret_cast<const ct_u4*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void ct_nt_store(uint16_t* p, const uint4& v) {
  ct_u4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<ct_u4*>(p));
}
__device__ __forceinline__ uint32_t ct_fastdiv(uint32_t n, uint32_t magic, uint32_t shift) { return (uint32_t)(((uint64_t)n * magic) >> shift); }

__device__ __forceinline__ uint32_t ct_pack_bf16x2(float lo, float hi) {   // round to nearest even (v_cvt_pk_bf16_f32)
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
  f2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2_t));
}
__device__ __forceinline__ void ct_split8(const float* v, uint4& hi, uint4& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = ct_pack_bf16x2(v[2 * j], v[2 * j + 1]);
    l[j] = ct_pack_bf16x2(v[2 * j] - __uint_as_float(h[j] << 16), v[2 * j + 1] - __uint_as_float(h[j] & 0xFFFF0000u));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void ct_join8(const uint4& hi, const uint4& lo, float* v) {
  const uint32_t h[4] = {hi.x, hi.y, hi.z, hi.w}, l[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[2 * j] = __uint_as_float(h[j] << 16) + __uint_as_float(l[j] << 16);
    v[2 * j + 1] = __uint_as_float(h[j] & 0xFFFF0000u) + __uint_as_float(l[j] & 0xFFFF0000u);
  }
}
// (element > 0) of the 8 bf16 values of a 16-byte chunk as one byte: a bf16 is > 0 exactly when its bits, read as int16, are > 0
__device__ __forceinline__ uint32_t ct_sign_byte(const uint4& v) {
  const uint32_t hw[4] = {v.x, v.y, v.z, v.w};
  uint32_t sb = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sb |= ((short)(hw[j] & 0xFFFFu) > 0 ? 1u : 0u) << (2 * j);
    sb |= ((short)(hw[j] >> 16) > 0 ? 1u : 0u) << (2 * j + 1);
  }
  return sb;
}

// Point-wise step of NJ accumulator blocks in registers (acc[j][r] = channel j*32 + (r&3) + 8*(r>>2) + 4h of the lane's position): ReLU
// and / or a 1-bit mask (mw[j] = the 32 mask bits of block j), hi + lo split, then lanes l and l + 32 exchange halves so that lane half h
// owns the whole 8-channel chunk 2s + h of the 16-channel group s = 2j + g/2: MFMA B-operand fragments / 16-byte row segments
template <int NJ>
__device__ __forceinline__ void ct_pointwise_frags(const f32x16* acc, const uint32_t* mw, bool relu, int h, uint4* fh, uint4* fl) {
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int ge = 0; ge < 4; ge += 2) {
      uint32_t eh[2], el[2], oh[2], ol[2];
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        float ve[2], vo[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ve[i] = acc[j][4 * ge + 2 * w + i];
          vo[i] = acc[j][4 * ge + 4 + 2 * w + i];
          if (relu) { ve[i] = fmaxf(ve[i], 0.f); vo[i] = fmaxf(vo[i], 0.f); }
          if (!((mw[j] >> (8 * ge + 4 * h + 2 * w + i)) & 1u)) ve[i] = 0.f;
          if (!((mw[j] >> (8 * ge + 8 + 4 * h + 2 * w + i)) & 1u)) vo[i] = 0.f;
        }
        eh[w] = ct_pack_bf16x2(ve[0], ve[1]);
        el[w] = ct_pack_bf16x2(ve[0] - __uint_as_float(eh[w] << 16), ve[1] - __uint_as_float(eh[w] & 0xFFFF0000u));
        oh[w] = ct_pack_bf16x2(vo[0], vo[1]);
        ol[w] = ct_pack_bf16x2(vo[0] - __uint_as_float(oh[w] << 16), vo[1] - __uint_as_float(oh[w] & 0xFFFF0000u));
        const auto sh = __builtin_amdgcn_permlane32_swap(eh[w], oh[w], false, false);
        eh[w] = sh[0]; oh[w] = sh[1];
        const auto sl = __builtin_amdgcn_permlane32_swap(el[w], ol[w], false, false);
        el[w] = sl[0]; ol[w] = sl[1];
      }
      fh[j * 2 + ge / 2] = make_uint4(eh[0], eh[1], oh[0], oh[1]);
      fl[j * 2 + ge / 2] = make_uint4(el[0], el[1], ol[0], ol[1]);
    }
}

template <int C, bool NEXT>
__global__ __launch_bounds__(256, (C == 64 && !NEXT) ? 3 : 2) void k_conv3x3_tail_pair(const ConvTailDev d) {
  constexpr int NJ = C / 32;                     // 32-channel blocks of the 3x3's output
  constexpr int KS = C / 16;                     // 16-deep K steps of the 1x1
  constexpr int NOUT = 4 * C, NC = NOUT / 64;    // the 1x1's output channels, in chunks of 64
  constexpr int TPT_SHIFT = C == 64 ? 1 : 2;     // K steps of 32 per tap: C / 32
  constexpr int KT = 9 << TPT_SHIFT;
  constexpr int PLANE_B = C * 64, STAGE = 2 * CT_PLANE_A + 2 * PLANE_B;
  constexpr int BQ = (C / 16) / 4;               // 1 KiB pieces of a weight plane per wave
  static_assert(4 * CT_WREG <= 2 * STAGE, "the waves' epilogue regions must fit the tile buffers");
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // vertically adjacent row tiles share input rows: keep neighbours on one XCD (its L2 serves the overlap)
  uint32_t tile;
  {
    const uint32_t nb = gridDim.x, bid = blockIdx.x, xcd = bid & 7u, slot = bid >> 3, q = nb >> 3, r = nb & 7u;
    tile = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + slot;
  }
  const int m0 = (int)tile * CT_TM;

  // ---- loader (see k_gemm_pair): a plane goes to LDS in 1 KiB pieces (16 rows x 64 B); lane -> row (lane >> 2), LDS chunk (lane & 3)
  //      <- the row's chunk (lane & 3) ^ ((row >> 2) & 3).  Round 6 (rart_lds_dma.h): buffer_load ... lds with one constant per-lane byte
  //      offset per piece; the tap and the K step move the scalar offset (resource base shifted down by the most negative tap offset); a
  //      pixel outside the image for tap t has bit t of `nok` set and its offset ORed with RART_DMA_OOR (zero-filled by the range check)
  uint32_t avoff[2], nok[2] = {0u, 0u};
  int tap_min = 0;
  for (int t = 0; t < 9; ++t) tap_min = min(tap_min, (d.tap_dy[t] * d.W + d.tap_dx[t]) * (C * 2));
  const int tapreg = lane < 9 ? (d.tap_dy[lane] * d.W + d.tap_dx[lane]) * (C * 2) - tap_min : 0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = 16 * (wave + 4 * q) + (lane >> 2);
    const int cs = ((lane & 3) ^ ((r >> 2) & 3)) * 16;
    const int m = m0 + r;
    const bool ok = m < d.M;
    const uint32_t mm = ok ? (uint32_t)m : 0u;
    const uint32_t t = ct_fastdiv(mm, d.w_magic, d.w_shift);
    const int ax = (int)(mm - t * (uint32_t)d.W);
    const int n = (int)ct_fastdiv(t, d.h_magic, d.h_shift);
    const int ay = (int)(t - (uint32_t)n * (uint32_t)d.H);
    avoff[q] = (uint32_t)((n * d.H * d.W + ay * d.W + ax) * (C * 2) + cs);
    for (int t2 = 0; t2 < 9; ++t2) {
      const int iy = ay + d.tap_dy[t2], ix = ax + d.tap_dx[t2];
      if (!(ok && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W)) nok[q] |= 1u << t2;
    }
  }
  uint32_t wvoff[BQ];
#pragma unroll
  for (int q = 0; q < BQ; ++q) {
    const int r = 16 * (wave + 4 * q) + (lane >> 2);
    wvoff[q] = (uint32_t)((r * d.ldw + (((lane & 3) ^ ((r >> 2) & 3)) * 8)) * 2);
  }
  const rart_srd_t srd_ah = rart_dma_srd(reinterpret_cast<const char*>(d.a_hi) + tap_min), srd_al = rart_dma_srd(reinterpret_cast<const char*>(d.a_lo) + tap_min);
  const rart_srd_t srd_wh = rart_dma_srd(d.w_hi), srd_wl = rart_dma_srd(d.w_lo);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
#define RART_CT_ISSUE(KT_, BUF)                                                                                 \
  {                                                                                                             \
    const uint32_t st_ = lds_base + (BUF)*STAGE;                                                                \
    const int kt_ = (KT_);                                                                                      \
    const int tap_ = kt_ >> TPT_SHIFT;                                                                          \
    const uint32_t so_ = (uint32_t)(__builtin_amdgcn_readlane(tapreg, tap_) + (kt_ - (tap_ << TPT_SHIFT)) * 64); \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                             \
      const uint32_t vo_ = avoff[q] | ((uint32_t)__builtin_amdgcn_sbfe(nok[q], tap_, 1) & RART_DMA_OOR);        \
      rart_dma_load16(vo_, srd_ah, so_, st_ + (wave + 4 * q) * 1024);                                           \
      rart_dma_load16(vo_, srd_al, so_, st_ + CT_PLANE_A + (wave + 4 * q) * 1024);                              \
    }                                                                                                           \
    _Pragma("unroll") for (int q = 0; q < BQ; ++q) {                                                            \
      rart_dma_load16(wvoff[q], srd_wh, (uint32_t)kt_ * 64u, st_ + 2 * CT_PLANE_A + (wave + 4 * q) * 1024);     \
      rart_dma_load16(wvoff[q], srd_wl, (uint32_t)kt_ * 64u, st_ + 2 * CT_PLANE_A + PLANE_B + (wave + 4 * q) * 1024); \
    }                                                                                                           \
  }
  const int fr = lane & 31, h = lane >> 5;
  uint32_t xo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) xo[ks] = (uint32_t)(fr * 64 + (((2 * ks + h) ^ ((fr >> 2) & 3)) << 4));
  // ---- the 3x3: acc[j][r] = channel j*32 + (r&3) + 8*(r>>2) + 4h of position fr (of this wave's 32), starting at the channel's bias
  f32x16 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d.bias_mid) bv = *reinterpret_cast<const float4*>(d.bias_mid + j * 32 + 8 * g + 4 * h);
      acc[j][4 * g] = bv.x; acc[j][4 * g + 1] = bv.y; acc[j][4 * g + 2] = bv.z; acc[j][4 * g + 3] = bv.w;
    }
  RART_CT_ISSUE(0, 0)
  rart_dma_wait<0>();
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) RART_CT_ISSUE(kt + 1, buf ^ 1)
    const uint8_t* Pb = lds + buf * STAGE + (wave * 32) * 64;
    const uint8_t* Wb = lds + buf * STAGE + 2 * CT_PLANE_A;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 ph = *reinterpret_cast<const bf16x8*>(Pb + xo[ks]);
      const bf16x8 pl = *reinterpret_cast<const bf16x8*>(Pb + CT_PLANE_A + xo[ks]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(Wb + j * 32 * 64 + xo[ks]);
        const bf16x8 wl = *reinterpret_cast<const bf16x8*>(Wb + PLANE_B + j * 32 * 64 + xo[ks]);
        // the two small products first (x_lo . w_hi, x_hi . w_lo), the large one last, as in k_gemm_pair
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, pl, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ph, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ph, acc[j], 0, 0, 0);
      }
    }
    rart_dma_wait<0>();                     // the next stage has landed in LDS
    __syncthreads();
  }
#undef RART_CT_ISSUE

  // ---- point-wise step of the intermediate, in registers: ReLU or the 1-bit mask, hi + lo split, then lanes l and l + 32 exchange
  //      halves so that lane half h owns the whole 8-channel chunk 2s + h of K step s: the B-operand fragments of the 1x1
  const int p = m0 + wave * 32 + fr;
  const bool p_ok = p < d.M;
  bf16x8 a2h[KS], a2l[KS];
  {
    uint32_t mw[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) mw[j] = 0xFFFFFFFFu;
    if (d.mask_mid && p_ok) {
      const uint32_t* mp = reinterpret_cast<const uint32_t*>(d.mask_mid + (size_t)p * (C / 8));
#pragma unroll
      for (int j = 0; j < NJ; ++j) mw[j] = mp[j];
    }
    uint4 fh[KS], fl[KS];
    ct_pointwise_frags<NJ>(acc, mw, d.relu_mid != 0, h, fh, fl);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      a2h[s] = __builtin_bit_cast(bf16x8, fh[s]);
      a2l[s] = __builtin_bit_cast(bf16x8, fl[s]);
      if (d.sign_mid && p_ok) d.sign_mid[(size_t)p * (C / 8) + 2 * s + h] = (uint8_t)ct_sign_byte(fh[s]);
    }
  }

  // ---- the 1x1, 64 output channels at a time: table fragments straight from L2, result transposed through the wave's LDS region
  uint8_t* const wreg = lds + wave * CT_WREG;                 // the wave's private region: fp32 staging, then (NEXT) the chunk as two bf16 planes
  float* sE = reinterpret_cast<float*>(wreg);
  const int cw = lane & 7, rw = lane >> 3;
  f32x16 accn[NEXT ? NJ : 1];                                 // NEXT: the neighbour's reduction, channel j*32 + (r&3) + 8*(r>>2) + 4h of position fr
  if (NEXT) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias_next) bv = *reinterpret_cast<const float4*>(d.bias_next + j * 32 + 8 * g + 4 * h);
        accn[j][4 * g] = bv.x; accn[j][4 * g + 1] = bv.y; accn[j][4 * g + 2] = bv.z; accn[j][4 * g + 3] = bv.w;
      }
  }
  const uint4* th_base = reinterpret_cast<const uint4*>(d.t_hi) + lane;
  const uint4* tl_base = reinterpret_cast<const uint4*>(d.t_lo) + lane;
  const bool relu_out = d.relu_out != 0;
#pragma unroll 1
  for (int c = 0; c < NC; ++c) {
    int eo[4];                 // element offsets stay below 2^31 (checked on the host)
    uint4 rh[4], rl[4];
    uint32_t mb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pp = m0 + wave * 32 + q * 8 + rw;
      eo[q] = -1;
      rh[q] = rl[q] = make_uint4(0, 0, 0, 0);
      mb[q] = 0xFFu;
      if (pp < d.M) {
        const int e = pp * NOUT + c * 64 + cw * 8;
        eo[q] = e;
        if (d.res_hi) {
          rh[q] = ct_nt_load(d.res_hi + e);
          rl[q] = ct_nt_load(d.res_lo + e);
        }
        if (d.mask_out) mb[q] = d.mask_out[e >> 3];
      }
    }
    f32x16 acc2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias_out) bv = *reinterpret_cast<const float4*>(d.bias_out + c * 64 + b * 32 + 8 * g + 4 * h);
        acc2[b][4 * g] = bv.x; acc2[b][4 * g + 1] = bv.y; acc2[b][4 * g + 2] = bv.z; acc2[b][4 * g + 3] = bv.w;
      }
#pragma unroll
      for (int sg = 0; sg < KS; sg += 4) {          // four K steps of table fragments in flight (32 registers)
        uint4 th[4], tl[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          th[s] = th_base[((c * 2 + b) * KS + sg + s) * 64];
          tl[s] = tl_base[((c * 2 + b) * KS + sg + s) * 64];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 wh = __builtin_bit_cast(bf16x8, th[s]), wl = __builtin_bit_cast(bf16x8, tl[s]);
          acc2[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, a2l[sg + s], acc2[b], 0, 0, 0);
          acc2[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, a2h[sg + s], acc2[b], 0, 0, 0);
          acc2[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, a2h[sg + s], acc2[b], 0, 0, 0);
        }
      }
    }
    // lane: position fr, channels b*32 + 8g + 4h + (0..3) -> one 16-byte LDS store each
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(sE + fr * CT_LDE + b * 32 + 8 * g + 4 * h) =
            make_float4(acc2[b][4 * g], acc2[b][4 * g + 1], acc2[b][4 * g + 2], acc2[b][4 * g + 3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint4 oph[4], opl[4];              // NEXT: the chunk's results as pairs (zeros for rows past M)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = q * 8 + rw;
      const float4 v0 = *reinterpret_cast<const float4*>(sE + r * CT_LDE + cw * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(sE + r * CT_LDE + cw * 8 + 4);
      const int e = eo[q];
      oph[q] = opl[q] = make_uint4(0, 0, 0, 0);
      if (e >= 0) {
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (d.res_hi) {
          float rv[8];
          ct_join8(rh[q], rl[q], rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += rv[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (!((mb[q] >> j) & 1u)) v[j] = 0.f;
        if (relu_out) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        uint4 ph, pl;
        ct_split8(v, ph, pl);
        ct_nt_store(d.dst_hi + e, ph);
        ct_nt_store(d.dst_lo + e, pl);
        if (d.sign_out) d.sign_out[e >> 3] = (uint8_t)ct_sign_byte(ph);
        oph[q] = ph;
        opl[q] = pl;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (NEXT) {
      // the chunk as two [32 positions][64 channels] bf16 planes in the region the staging just left, then as B-operand fragments
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<uint4*>(wreg + (q * 8 + rw) * CT_NLD + cw * 16) = oph[q];
        *reinterpret_cast<uint4*>(wreg + 32 * CT_NLD + (q * 8 + rw) * CT_NLD + cw * 16) = opl[q];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      bf16x8 xh[4], xl[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        xh[s] = *reinterpret_cast<const bf16x8*>(wreg + fr * CT_NLD + s * 32 + h * 16);
        xl[s] = *reinterpret_cast<const bf16x8*>(wreg + 32 * CT_NLD + fr * CT_NLD + s * 32 + h * 16);
      }
      const uint4* nh_base = reinterpret_cast<const uint4*>(d.n_hi) + lane;
      const uint4* nl_base = reinterpret_cast<const uint4*>(d.n_lo) + lane;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        uint4 nh[4], nl[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          nh[s] = nh_base[((c * NJ + j) * 4 + s) * 64];
          nl[s] = nl_base[((c * NJ + j) * 4 + s) * 64];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 wh = __builtin_bit_cast(bf16x8, nh[s]), wl = __builtin_bit_cast(bf16x8, nl[s]);
          accn[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[s], accn[j], 0, 0, 0);
          accn[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[s], accn[j], 0, 0, 0);
          accn[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[s], accn[j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();       // the fragment reads are done before the next chunk's staging
    }
  }
  if (NEXT) {
    // ---- point-wise step of the neighbour's reduction and its stores: a lane owns 16-byte segments of its position's row
    uint32_t mw[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) mw[j] = 0xFFFFFFFFu;
    if (d.mask_next && p_ok) {
      const uint32_t* mp = reinterpret_cast<const uint32_t*>(d.mask_next + (size_t)p * (C / 8));
#pragma unroll
      for (int j = 0; j < NJ; ++j) mw[j] = mp[j];
    }
    uint4 fh[KS], fl[KS];
    ct_pointwise_frags<NJ>(accn, mw, d.relu_next != 0, h, fh, fl);
    if (p_ok) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        ct_nt_store(d.dstn_hi + (size_t)p * C + 16 * s + 8 * h, fh[s]);
        ct_nt_store(d.dstn_lo + (size_t)p * C + 16 * s + 8 * h, fl[s]);
        if (d.sign_next) d.sign_next[(size_t)p * (C / 8) + 2 * s + h] = (uint8_t)ct_sign_byte(fh[s]);
      }
    }
  }
}

void ct_magic(uint32_t dv, uint32_t& mg, uint32_t& sh) {   // exact for dividends < 2^31
  uint32_t l = 0;
  while ((1ull << l) < dv) ++l;
  sh = 31 + l;
  mg = (uint32_t)(((1ull << sh) + dv - 1) / dv);
}
}  // namespace

extern "C" int rart_conv3x3_tail_pair_supported(int c_mid) { return (c_mid == 64 || c_mid == 128) ? 1 : 0; }

extern "C" int rart_conv3x3_tail_pair(const rart_conv_tail_desc* t, rart_stream_t stream) {
  RART_CHECK_ARG(t != nullptr, "rart_conv3x3_tail_pair: null descriptor");
  RART_CHECK_ARG(t->a_hi && t->a_lo && t->w_hi && t->w_lo && t->t_hi && t->t_lo && t->dst_hi && t->dst_lo,
                 "rart_conv3x3_tail_pair: null operand plane");
  RART_CHECK_ARG(t->c_mid == 64 || t->c_mid == 128, "rart_conv3x3_tail_pair: c_mid must be 64 or 128 (layer1 / layer2)");
  RART_CHECK_ARG(t->batch > 0 && t->h > 0 && t->w > 0, "rart_conv3x3_tail_pair: empty geometry");
  RART_CHECK_ARG((t->res_hi == nullptr) == (t->res_lo == nullptr), "rart_conv3x3_tail_pair: the skip is a pair: both planes or none");
  RART_CHECK_ARG(t->ldw % 8 == 0 && t->ldw >= 9 * t->c_mid, "rart_conv3x3_tail_pair: ldw must cover 9 taps x c_mid and keep 16-byte alignment");
  const long long M = (long long)t->batch * t->h * t->w;
  RART_CHECK_ARG(M * 4 * t->c_mid < (1ll << 31), "rart_conv3x3_tail_pair: tensors must stay below 2^31 elements (split the batch)");
  ConvTailDev d;
  d.a_hi = (const uint16_t*)t->a_hi; d.a_lo = (const uint16_t*)t->a_lo; d.w_hi = (const uint16_t*)t->w_hi; d.w_lo = (const uint16_t*)t->w_lo;
  d.t_hi = (const uint16_t*)t->t_hi; d.t_lo = (const uint16_t*)t->t_lo;
  d.bias_mid = t->bias_mid; d.bias_out = t->bias_out;
  d.mask_mid = (const uint8_t*)t->mask_mid; d.mask_out = (const uint8_t*)t->mask_out;
  d.sign_mid = (uint8_t*)t->sign_mid; d.sign_out = (uint8_t*)t->sign_out;
  d.res_hi = (const uint16_t*)t->res_hi; d.res_lo = (const uint16_t*)t->res_lo;
  d.dst_hi = (uint16_t*)t->dst_hi; d.dst_lo = (uint16_t*)t->dst_lo;
  d.M = (int)M; d.H = t->h; d.W = t->w; d.ldw = t->ldw; d.relu_mid = t->relu_mid; d.relu_out = t->relu_out;
  for (int i = 0; i < 9; ++i) { d.tap_dy[i] = t->tap_dy[i]; d.tap_dx[i] = t->tap_dx[i]; }
  const bool next = t->n_hi != nullptr;
  RART_CHECK_ARG(!next || (t->n_lo && t->dstn_hi && t->dstn_lo), "rart_conv3x3_tail_pair: the neighbour's reduction needs both table planes and its destination pair");
  d.n_hi = (const uint16_t*)t->n_hi; d.n_lo = (const uint16_t*)t->n_lo; d.bias_next = t->bias_next; d.mask_next = (const uint8_t*)t->mask_next;
  d.sign_next = (uint8_t*)t->sign_next; d.dstn_hi = (uint16_t*)t->dstn_hi; d.dstn_lo = (uint16_t*)t->dstn_lo; d.relu_next = t->relu_next;
  ct_magic((uint32_t)d.W, d.w_magic, d.w_shift);
  ct_magic((uint32_t)d.H, d.h_magic, d.h_shift);
  const dim3 grid((uint32_t)((M + CT_TM - 1) / CT_TM));
  hipStream_t st = (hipStream_t)stream;
  if (t->c_mid == 64) {
    if (next) hipLaunchKernelGGL((k_conv3x3_tail_pair<64, true>), grid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL((k_conv3x3_tail_pair<64, false>), grid, dim3(256), 0, st, d);
  } else {
    if (next) hipLaunchKernelGGL((k_conv3x3_tail_pair<128, true>), grid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL((k_conv3x3_tail_pair<128, false>), grid, dim3(256), 0, st, d);
  }
  RART_CHECK_LAUNCH("rart_conv3x3_tail_pair");
  return RART_OK;
}
