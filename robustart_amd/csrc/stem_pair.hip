// Fused stem BACKWARD of the reference-precision ("fp32x") ResNet-50 engine (gfx950): max-pool backward + ReLU mask + transposed 7x7/2
// convolution to the fp32 image gradient in ONE kernel, every tensor a hi + lo pair of bf16 planes, every contraction the three MFMA
// products lo.hi + hi.lo + hi.hi (fp32 accumulation).
//
// Replaces the chain k_maxpool_bwd_pair -> patches GEMM (k_gemm_pair, N = 152, fp32 out) -> k_stem_col2im_f32, which wrote and re-read the
// 112 x 112 x 64 gradient pair (2 x 411 MB) and a 1.95 GB fp32 patch tensor per 256 images: 2.5 ms of the 21 ms of a gradient evaluation
// (profiles/r04_bench_kernel_stats.csv: 344 + 723 + 1 477 us).  Same structure as k_stem_bwd_fused (csrc/stem_fused.hip, which states the
// math): a workgroup owns 16 x 16 positions of the stem-output grid, rebuilds the 19 x 19 halo tile of the gradient at the stem output in
// LDS from the pooled gradient pair and the argmax codes (code 15 = window maximum <= 0 = ReLU dead), walks the 16 taps as an implicit
// GEMM (M = positions, K = 16 taps x 64 channels, N = 12 = 4 pixel parities x 3 colours) on v_mfma_f32_16x16x32_bf16 and writes a
// 32 x 32 x 3 fp32 tile.  The pair doubles every LDS image, so the K dimension is walked in two channel halves (32 channels = one MFMA K
// step per tap): stage -> pool backward -> 16 taps, twice, into the same accumulators -- 66 KB of LDS, two workgroups per CU whose
// phases overlap, as in the bf16 kernel.
//
// Reference step: the autograd pass of every attack iteration (RobustART/noise/utils/adv/attack.py:21-22 via foolbox;
// Attacks/autoattack/autopgd_base.py:271-289, fp32) through conv1 / bn1 / relu / maxpool of the public ResNet-50
// (RobustART/model/__init__.py:1 -> absent submodule; robustart_amd/model/resnet_torch.py states it).
#include "rart_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {
constexpr int T = 16;                     // positions per tile side
constexpr int HT = T + 3;                 // halo tile side (dp, dq in -1..2)
constexpr int NPOS_PAD = 368;             // 361 rounded up to a multiple of 16: chunk planes start on a 256-byte bank row
constexpr int PT = T / 2 + 3;             // pooled positions per side that reach the halo tile (11)
constexpr int NPOOL = PT * PT;            // 121
constexpr int OT = 2 * T;                 // image pixels per tile side (32)
constexpr int OLD = OT + 1;               // padded fp32 output row in LDS
constexpr int DZ_BYTES = 2 * 4 * NPOS_PAD * 16;       // [hi | lo][4 chunks of a channel half][NPOS_PAD] x 16 B
constexpr int DP_BYTES = 2 * NPOOL * 4 * 16;          // [hi | lo][NPOOL][4 chunks]
constexpr int ARG_BYTES = NPOOL * 4 * 8;              // [NPOOL][4 chunks] x 8 codes
static_assert(3 * OT * OLD * 4 <= DP_BYTES + ARG_BYTES, "output tile must fit the raw pooled tile");

__device__ __forceinline__ uint32_t sp_pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  f2 f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
}

// the halo positions whose stem row has parity EY and whose stem column has parity EX (see pool_bwd_class in stem_fused.hip): one
// window per even coordinate, two per odd; the gradient of a position is the fp32 sum of the pair VALUES of its matching windows
template <int EY, int EX>
__device__ __forceinline__ void pool_bwd_class_pair(uint4* sDz1, const uint4* sDp, const uint2* sArg, int tid, int a0, int b0, int qy0,
                                                    int qx0, int oh, int ow) {
  constexpr int NY = EY ? (HT + 1) / 2 : HT / 2, NX = EX ? (HT + 1) / 2 : HT / 2;
  constexpr int NYS = EY ? 2 : 1, NXS = EX ? 2 : 1;
  for (int i = tid; i < 4 * NY * NX; i += 256) {
    const int c = i / (NY * NX), j = i - c * (NY * NX);
    const int iy = j / NX, ix = j - iy * NX;
    const int hy = 2 * iy + (EY ? 0 : 1), hx = 2 * ix + (EX ? 0 : 1);
    const int py = a0 - 1 + hy, px = b0 - 1 + hx;           // stem-output coordinates
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if ((unsigned)py < (unsigned)oh && (unsigned)px < (unsigned)ow) {
#pragma unroll
      for (int ia = 0; ia < NYS; ++ia) {
        const int qy = (py >> 1) + ia;
        const uint32_t ky = (uint32_t)(py - (2 * qy - 1));
#pragma unroll
        for (int ib = 0; ib < NXS; ++ib) {
          const int qx = (px >> 1) + ib;
          const uint32_t mine = (ky * 3 + (uint32_t)(px - (2 * qx - 1))) * 0x01010101u;
          const int lp = (qy - qy0) * PT + (qx - qx0);      // out-of-grid windows hold code 15 / zeros
          const uint2 cd = sArg[lp * 4 + c];
          const uint4 dh = sDp[lp * 4 + c], dl = sDp[NPOOL * 4 + lp * 4 + c];
          const uint32_t cw[2] = {cd.x, cd.y};
          const uint32_t hw_[4] = {dh.x, dh.y, dh.z, dh.w}, lw_[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
          for (int hw = 0; hw < 2; ++hw) {
            // codes are < 16, so (code ^ mine) + 0x7F sets bit 7 of a byte exactly when the byte differs
            const uint32_t ne = ((cw[hw] ^ mine) + 0x7F7F7F7Fu) & 0x80808080u;
            const uint32_t eq = (ne ^ 0x80808080u) >> 7;              // 1 per equal byte
            const uint32_t m = (eq << 8) - eq;                         // 0xFF per equal byte
            const uint32_t m0 = __builtin_amdgcn_perm(m, m, 0x01010000u), m1 = __builtin_amdgcn_perm(m, m, 0x03030202u);
            const uint32_t h0 = hw_[2 * hw] & m0, h1 = hw_[2 * hw + 1] & m1, l0 = lw_[2 * hw] & m0, l1 = lw_[2 * hw + 1] & m1;
            g[4 * hw + 0] += __uint_as_float(h0 << 16) + __uint_as_float(l0 << 16);                   // hi + lo is exact in fp32
            g[4 * hw + 1] += __uint_as_float(h0 & 0xFFFF0000u) + __uint_as_float(l0 & 0xFFFF0000u);
            g[4 * hw + 2] += __uint_as_float(h1 << 16) + __uint_as_float(l1 << 16);
            g[4 * hw + 3] += __uint_as_float(h1 & 0xFFFF0000u) + __uint_as_float(l1 & 0xFFFF0000u);
          }
        }
      }
    }
    uint32_t hv[4], lv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hv[k] = sp_pack2(g[2 * k], g[2 * k + 1]);
      lv[k] = sp_pack2(g[2 * k] - __uint_as_float(hv[k] << 16), g[2 * k + 1] - __uint_as_float(hv[k] & 0xFFFF0000u));
    }
    sDz1[c * NPOS_PAD + hy * HT + hx] = make_uint4(hv[0], hv[1], hv[2], hv[3]);
    sDz1[(4 + c) * NPOS_PAD + hy * HT + hx] = make_uint4(lv[0], lv[1], lv[2], lv[3]);
  }
}

struct Istd3p { float v[3]; };

__global__ __launch_bounds__(256, 2) void k_stem_bwd_pair(const uint4* __restrict__ dpool_h,    // [n][oh2][ow2][64] bf16, hi plane
                                                          const uint4* __restrict__ dpool_l,    // lo plane
                                                          const uint4* __restrict__ arg,        // [n][oh2][ow2][64] u8 codes
                                                          const uint16_t* __restrict__ wt_h,    // [16][1024] bf16 (stem_fused.hip's table), hi
                                                          const uint16_t* __restrict__ wt_l,    // lo
                                                          float* __restrict__ grad,             // [n][3][h][w]
                                                          int h, int w, Istd3p istd) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[DZ_BYTES + DP_BYTES + ARG_BYTES];
  uint4* sDz1 = reinterpret_cast<uint4*>(lds);                                     // [2][4][NPOS_PAD]
  uint4* sDp = reinterpret_cast<uint4*>(lds + DZ_BYTES);                           // [2][NPOOL][4]
  uint2* sArg = reinterpret_cast<uint2*>(lds + DZ_BYTES + DP_BYTES);               // [NPOOL][4]
  float* sOut = reinterpret_cast<float*>(lds + DZ_BYTES);                          // [3][OT][OLD] (aliases the raw tile)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int oh = h >> 1, ow = w >> 1;            // stem-output grid
  const int oh2 = oh >> 1, ow2 = ow >> 1;        // pooled grid
  const int a0 = blockIdx.y * T, b0 = blockIdx.x * T, img = blockIdx.z;
  const int qy0 = (a0 >> 1) - 1, qx0 = (b0 >> 1) - 1;      // first pooled row / column that reaches the halo tile
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint4* wrow_h = reinterpret_cast<const uint4*>(wt_h + (size_t)fr * 1024 + fg * 8);   // + tap * 8 + half * 4 (uint4 units)
  const uint4* wrow_l = reinterpret_cast<const uint4*>(wt_l + (size_t)fr * 1024 + fg * 8);

#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    // (no barrier here: the raw tile's readers passed the barrier below in the first half; the halo tile is rewritten only after the next one)
    // ---- stage the pooled gradient pair and the argmax codes of this channel half (zeros / code 15 outside the grid)
    for (int i = tid; i < NPOOL * 10; i += 256) {
      const int pos = i / 10, v = i - pos * 10;
      const int qy = qy0 + pos / PT, qx = qx0 + pos % PT;
      const bool ok = (unsigned)qy < (unsigned)oh2 && (unsigned)qx < (unsigned)ow2;
      const size_t base = ((size_t)img * oh2 + qy) * ow2 + qx;
      if (v < 8) {
        const int pl = v >> 2, c = v & 3;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (ok) val = (pl ? dpool_l : dpool_h)[base * 8 + half * 4 + c];
        sDp[pl * NPOOL * 4 + pos * 4 + c] = val;
      } else {
        uint4 val = make_uint4(0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu);
        if (ok) val = arg[base * 4 + half * 2 + (v - 8)];
        reinterpret_cast<uint4*>(sArg)[pos * 2 + (v - 8)] = val;
      }
    }
    __syncthreads();
    // ---- max-pool backward into the halo tile, one pass per pixel-parity class
    pool_bwd_class_pair<0, 0>(sDz1, sDp, sArg, tid, a0, b0, qy0, qx0, oh, ow);
    pool_bwd_class_pair<0, 1>(sDz1, sDp, sArg, tid, a0, b0, qy0, qx0, oh, ow);
    pool_bwd_class_pair<1, 0>(sDz1, sDp, sArg, tid, a0, b0, qy0, qx0, oh, ow);
    pool_bwd_class_pair<1, 1>(sDz1, sDp, sArg, tid, a0, b0, qy0, qx0, oh, ow);
    __syncthreads();
    // ---- implicit GEMM over the 16 taps of this channel half: wave w owns tile rows 4w..4w+3 (one 16-position M tile each)
#pragma unroll
    for (int dpi = 0; dpi < 4; ++dpi) {
      uint4 bh[4], bl[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bh[t] = wrow_h[(dpi * 4 + t) * 8 + half * 4];
        bl[t] = wrow_l[(dpi * 4 + t) * 8 + half * 4];
      }
#pragma unroll
      for (int dqi = 0; dqi < 4; ++dqi) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, bh[dqi]), wl = __builtin_bit_cast(bf16x8, bl[dqi]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int pos = (wave * 4 + m + dpi) * HT + fr + dqi;
          const bf16x8 ah = __builtin_bit_cast(bf16x8, sDz1[fg * NPOS_PAD + pos]);
          const bf16x8 al = __builtin_bit_cast(bf16x8, sDz1[(4 + fg) * NPOS_PAD + pos]);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh, acc[m], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: D[row = fg*4 + j (position column)][col = fr = (py*2+px)*3 + c] -> sOut[c][2a+py][2b+px]
  //      (the raw pooled tile is dead: every wave passed the barrier after its last read of it)
  if (fr < 12) {
    const int pq = fr / 3, c = fr - pq * 3;
    const int py = pq >> 1, px = pq & 1;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        sOut[(c * OT + 2 * (wave * 4 + m) + py) * OLD + 2 * (fg * 4 + j) + px] = acc[m][j];
  }
  __syncthreads();
  const size_t plane = (size_t)h * w;
  for (int i = tid; i < 3 * OT * (OT / 4); i += 256) {
    const int x4 = i & 7, row = i >> 3;
    const int c = row / OT, y = row - c * OT;
    const int gy = 2 * a0 + y, gx = 2 * b0 + x4 * 4;
    if (gy < h && gx < w) {
      const float s = istd.v[c];
      const float* r = sOut + (c * OT + y) * OLD + x4 * 4;
      *reinterpret_cast<float4*>(grad + ((size_t)img * 3 + c) * plane + (size_t)gy * w + gx) =
          make_float4(r[0] * s, r[1] * s, r[2] * s, r[3] * s);
    }
  }
}
}  // namespace

extern "C" int rart_engine_stem_bwd_fused_pair(const void* dpool_hi, const void* dpool_lo, const void* argmax, const void* wtab_hi,
                                               const void* wtab_lo, float* grad, int n, int h, int w, const float* std_host,
                                               rart_stream_t stream) {
  RART_CHECK_ARG(dpool_hi && dpool_lo && argmax && wtab_hi && wtab_lo && grad && n > 0 && n <= 65535,
                 "rart_engine_stem_bwd_fused_pair: bad arguments");
  RART_CHECK_ARG(h % 4 == 0 && w % 4 == 0 && h >= 4 && w >= 4, "rart_engine_stem_bwd_fused_pair: h and w must be multiples of 4");
  Istd3p is;
  for (int c = 0; c < 3; ++c) is.v[c] = std_host ? 1.0f / std_host[c] : 1.0f;
  const int oh = h / 2, ow = w / 2;
  hipLaunchKernelGGL(k_stem_bwd_pair, dim3((ow + T - 1) / T, (oh + T - 1) / T, n), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)dpool_hi, (const uint4*)dpool_lo, (const uint4*)argmax, (const uint16_t*)wtab_hi,
                     (const uint16_t*)wtab_lo, grad, h, w, is);
  RART_CHECK_LAUNCH("rart_engine_stem_bwd_fused_pair");
  return RART_OK;
}
