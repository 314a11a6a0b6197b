// Fused stem backward of the eval-mode ResNet-50 engine (gfx950): max-pool backward + ReLU mask + transposed 7x7/2
// convolution to the fp32 image gradient in ONE kernel.  Replaces the chain k_maxpool_bwd -> patches GEMM (igemm, N = 152)
// -> k_stem_col2im, which moved 1.6 MB + 3.9 MB + 2.6 MB per image through HBM (1.36 ms of a 6.7 ms backward at
// B = 256); here a workgroup reads 11 x 11 pooled gradients + argmax codes and writes a 32 x 32 x 3 fp32 tile.
//
// Reference step: the autograd pass of every attack iteration (RobustART/noise/utils/adv/attack.py:21-22 via foolbox;
// Attacks/autoattack/autopgd_base.py:271-289) through conv1 / bn1 / relu / maxpool of the public ResNet-50
// (RobustART/model/__init__.py:1 -> absent submodule; robustart_amd/model/resnet_torch.py states it).
//
// Math.  Forward: y1[p][q][k] = sum_{c,r,s} W[k][c][r][s] * x[2p+r-3][2q+s-3][c].  Hence, with y = 2a+py, x = 2b+px,
//   dx[2a+py][2b+px][c] = sum_{dp,dq in -1..2} sum_k dz1[a+dp][b+dq][k] * W[k][c][py+3-2dp][px+3-2dq]      (taps outside 0..6: 0)
// i.e. per position (a,b) of the 112 x 112 stem-output grid a [16 taps x 64] . [1024 x 12] product: an implicit GEMM with
// M = positions, K = 1024, N = 12 (padded to 16) on v_mfma_f32_16x16x32_bf16.  The A operand is never materialised:
// a workgroup owns 16 x 16 positions, rebuilds the 19 x 19 halo tile of dz1 (max-pool backward from the pooled gradient
// and the argmax codes; code 15 = window maximum <= 0 = ReLU dead) in LDS, chunk-major ([8 x 16-byte chunk][368 positions])
// so that the 16 consecutive positions of an MFMA A fragment are 256 contiguous bytes (ds_read_b128 conflict free,
// scratch/lds_bank_model.py), and walks the 16 taps from LDS.  B fragments (32 KB table, L2 resident) come straight from
// global memory, one tap row (8 fragments) ahead.  Epilogue: the 16 x 16 x 12 results are transposed through LDS to
// [c][32][32] and leave as 128-byte rows of the NCHW fp32 gradient, scaled by 1/std[c].
#include "rart_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {
constexpr int T = 16;                     // positions per tile side
constexpr int HT = T + 3;                 // halo tile side (dp, dq in -1..2)
constexpr int NPOS = HT * HT;             // 361
constexpr int NPOS_PAD = 368;             // multiple of 16: chunk planes start on a 256-byte bank row
constexpr int PT = T / 2 + 3;             // pooled positions per side that reach the halo tile (11)
constexpr int NPOOL = PT * PT;            // 121
constexpr int OT = 2 * T;                 // image pixels per tile side (32)
constexpr int OLD = OT + 1;               // padded fp32 output row in LDS

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  f2 f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
}

struct Istd3 { float v[3]; };

__global__ __launch_bounds__(256, 2) void k_stem_bwd_fused(const uint4* __restrict__ dpool,   // [n][oh2][ow2][64] bf16
                                                           const uint2* __restrict__ arg,     // [n][oh2][ow2][64] u8
                                                           const uint16_t* __restrict__ wt,   // [16][1024] bf16
                                                           float* __restrict__ grad,          // [n][3][h][w]
                                                           int h, int w, Istd3 istd) {
  // one LDS object (a second one makes hipcc wait vmcnt(0) before LDS reads): dz1 tile | raw pooled tile (later: out tile)
  __shared__ __attribute__((aligned(16))) uint8_t lds[8 * NPOS_PAD * 16 + NPOOL * 192];
  uint4* sDz1 = reinterpret_cast<uint4*>(lds);                          // [8][NPOS_PAD]
  uint4* sDp = reinterpret_cast<uint4*>(lds + 8 * NPOS_PAD * 16);       // [NPOOL][8]  pooled gradient, 8 channels / 16 B
  uint2* sArg = reinterpret_cast<uint2*>(lds + 8 * NPOS_PAD * 16 + NPOOL * 128);   // [NPOOL][8]  argmax codes, 8 / 8 B
  float* sOut = reinterpret_cast<float*>(lds + 8 * NPOS_PAD * 16);      // [3][OT][OLD] (aliases the raw tile)
  static_assert(3 * OT * OLD * 4 <= NPOOL * 192, "output tile must fit the raw pooled tile");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int oh = h >> 1, ow = w >> 1;            // stem-output grid
  const int oh2 = oh >> 1, ow2 = ow >> 1;        // pooled grid
  const int a0 = blockIdx.y * T, b0 = blockIdx.x * T, img = blockIdx.z;
  const int qy0 = (a0 >> 1) - 1, qx0 = (b0 >> 1) - 1;      // first pooled row / column that reaches the halo tile

  // ---- stage the pooled gradient and the argmax codes (zeros / code 15 outside the grid)
  for (int i = tid; i < NPOOL * 12; i += 256) {
    const int pos = i / 12, v = i - pos * 12;
    const int qy = qy0 + pos / PT, qx = qx0 + pos % PT;
    const bool ok = (unsigned)qy < (unsigned)oh2 && (unsigned)qx < (unsigned)ow2;
    const size_t base = ((size_t)img * oh2 + qy) * ow2 + qx;
    if (v < 8) {
      uint4 val = make_uint4(0, 0, 0, 0);
      if (ok) val = dpool[base * 8 + v];
      sDp[pos * 8 + v] = val;
    } else {
      uint4 val = make_uint4(0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu);
      if (ok) val = reinterpret_cast<const uint4*>(arg)[base * 4 + (v - 8)];
      reinterpret_cast<uint4*>(sArg)[pos * 4 + (v - 8)] = val;
    }
  }
  __syncthreads();

  // ---- max-pool backward into the halo tile: dz1[p] = sum over the <= 4 windows holding p whose argmax is p
  for (int i = tid; i < 8 * NPOS; i += 256) {
    const int c = i / NPOS, pos = i - c * NPOS;
    const int hy = pos / HT, hx = pos - hy * HT;
    const int py = a0 - 1 + hy, px = b0 - 1 + hx;           // stem-output coordinates
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if ((unsigned)py < (unsigned)oh && (unsigned)px < (unsigned)ow) {
      const int nys = (py & 1) ? 2 : 1, nxs = (px & 1) ? 2 : 1;
      for (int ia = 0; ia < nys; ++ia) {
        const int qy = (py >> 1) + ia;                     // windows q with 2q-1 <= p <= 2q+1
        const uint32_t ky = (uint32_t)(py - (2 * qy - 1));
        for (int ib = 0; ib < nxs; ++ib) {
          const int qx = (px >> 1) + ib;
          const uint32_t mine = ky * 3 + (uint32_t)(px - (2 * qx - 1));
          const int lp = (qy - qy0) * PT + (qx - qx0);      // out-of-grid windows hold code 15 / zeros
          const uint2 cd = sArg[lp * 8 + c];
          const uint4 dv = sDp[lp * 8 + c];
          const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const uint32_t cj = ((j < 4 ? cd.x : cd.y) >> (8 * (j & 3))) & 0xFFu;
            const float d = (j & 1) ? __uint_as_float(dw[j >> 1] & 0xFFFF0000u) : __uint_as_float(dw[j >> 1] << 16);
            if (cj == mine) g[j] += d;
          }
        }
      }
    }
    sDz1[c * NPOS_PAD + pos] = make_uint4(pack2(g[0], g[1]), pack2(g[2], g[3]), pack2(g[4], g[5]), pack2(g[6], g[7]));
  }
  __syncthreads();

  // ---- implicit GEMM over the 16 taps: wave w owns tile rows 4w..4w+3 (one 16-position M tile each)
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint4* wrow = reinterpret_cast<const uint4*>(wt + (size_t)fr * 1024 + fg * 8);   // + tap * 8 + ks * 4 (uint4 units)
#pragma unroll
  for (int dpi = 0; dpi < 4; ++dpi) {
    uint4 bq[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) bq[t] = wrow[(dpi * 4 + (t >> 1)) * 8 + (t & 1) * 4];
#pragma unroll
    for (int dqi = 0; dqi < 4; ++dqi) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 b = __builtin_bit_cast(bf16x8, bq[dqi * 2 + ks]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int pos = (wave * 4 + m + dpi) * HT + fr + dqi;
          const bf16x8 a = __builtin_bit_cast(bf16x8, sDz1[(ks * 4 + fg) * NPOS_PAD + pos]);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m], 0, 0, 0);
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

extern "C" int rart_engine_stem_bwd_fused(const void* dpool, const void* argmax, const void* wtab, float* grad, int n, int h,
                                          int w, const float* std_host, rart_stream_t stream) {
  RART_CHECK_ARG(dpool && argmax && wtab && grad && n > 0 && n <= 65535, "rart_engine_stem_bwd_fused: bad arguments");
  RART_CHECK_ARG(h % 4 == 0 && w % 4 == 0 && h >= 4 && w >= 4, "rart_engine_stem_bwd_fused: h and w must be multiples of 4");
  Istd3 is;
  for (int c = 0; c < 3; ++c) is.v[c] = std_host ? 1.0f / std_host[c] : 1.0f;
  const int oh = h / 2, ow = w / 2;
  hipLaunchKernelGGL(k_stem_bwd_fused, dim3((ow + T - 1) / T, (oh + T - 1) / T, n), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)dpool, (const uint2*)argmax, (const uint16_t*)wtab, grad, h, w, is);
  RART_CHECK_LAUNCH("rart_engine_stem_bwd_fused");
  return RART_OK;
}
