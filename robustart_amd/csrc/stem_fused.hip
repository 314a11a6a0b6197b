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


// the halo positions whose stem row has parity EY and whose stem column has parity EX (tile origins are multiples of 16, so
// halo row hy has stem-row parity (hy + 1) & 1): windows q with 2q-1 <= p <= 2q+1 -> one window per even coordinate, two per odd
template <int EY, int EX>
__device__ __forceinline__ void pool_bwd_class(uint4* sDz1, const uint4* sDp, const uint2* sArg,
                                               int tid, int a0, int b0, int qy0, int qx0, int oh, int ow) {
  constexpr int NY = EY ? (HT + 1) / 2 : HT / 2, NX = EX ? (HT + 1) / 2 : HT / 2;      // 10 / 9 halo rows (columns) of the class
  constexpr int NYS = EY ? 2 : 1, NXS = EX ? 2 : 1;
  for (int i = tid; i < 8 * NY * NX; i += 256) {
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
          const uint2 cd = sArg[c * NPOOL + lp];            // chunk-major: the lanes of a wave walk consecutive windows
          const uint4 dv = sDp[c * NPOOL + lp];
          const uint32_t cw[2] = {cd.x, cd.y};
          const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
          for (int hw = 0; hw < 2; ++hw) {
            // codes are < 16, so (code ^ mine) + 0x7F sets bit 7 of a byte exactly when the byte differs
            const uint32_t ne = ((cw[hw] ^ mine) + 0x7F7F7F7Fu) & 0x80808080u;
            const uint32_t eq = (ne ^ 0x80808080u) >> 7;              // 1 per equal byte
            const uint32_t m = (eq << 8) - eq;                         // 0xFF per equal byte
            const uint32_t w0 = dw[2 * hw] & __builtin_amdgcn_perm(m, m, 0x01010000u);
            const uint32_t w1 = dw[2 * hw + 1] & __builtin_amdgcn_perm(m, m, 0x03030202u);
            g[4 * hw + 0] += __uint_as_float(w0 << 16);
            g[4 * hw + 1] += __uint_as_float(w0 & 0xFFFF0000u);
            g[4 * hw + 2] += __uint_as_float(w1 << 16);
            g[4 * hw + 3] += __uint_as_float(w1 & 0xFFFF0000u);
          }
        }
      }
    }
    sDz1[c * NPOS_PAD + hy * HT + hx] = make_uint4(pack2(g[0], g[1]), pack2(g[2], g[3]), pack2(g[4], g[5]), pack2(g[6], g[7]));
  }
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
  uint4* sDp = reinterpret_cast<uint4*>(lds + 8 * NPOS_PAD * 16);       // [8][NPOOL]  pooled gradient, 8 channels / 16 B (chunk-major)
  uint2* sArg = reinterpret_cast<uint2*>(lds + 8 * NPOS_PAD * 16 + NPOOL * 128);   // [8][NPOOL]  argmax codes, 8 / 8 B
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
      sDp[v * NPOOL + pos] = val;
    } else {
      uint4 val = make_uint4(0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu);
      if (ok) val = reinterpret_cast<const uint4*>(arg)[base * 4 + (v - 8)];
      sArg[(2 * (v - 8)) * NPOOL + pos] = make_uint2(val.x, val.y);
      sArg[(2 * (v - 8) + 1) * NPOOL + pos] = make_uint2(val.z, val.w);
    }
  }
  __syncthreads();

  // ---- max-pool backward into the halo tile: dz1[p] = sum over the <= 4 windows holding p whose argmax is p.
  //      One pass per pixel-parity class (even / odd row x even / odd column of the stem grid: 1, 2, 2 or 4 windows) so that
  //      the lanes of a wave walk the same number of windows (mixed parities made every wave pay for 4), and the eight
  //      argmax codes of a 16-byte chunk are compared at once (SWAR: equal bytes -> 0xFF) instead of byte by byte.
  pool_bwd_class<0, 0>(sDz1, sDp, sArg, tid, a0, b0, qy0, qx0, oh, ow);
  pool_bwd_class<0, 1>(sDz1, sDp, sArg, tid, a0, b0, qy0, qx0, oh, ow);
  pool_bwd_class<1, 0>(sDz1, sDp, sArg, tid, a0, b0, qy0, qx0, oh, ow);
  pool_bwd_class<1, 1>(sDz1, sDp, sArg, tid, a0, b0, qy0, qx0, oh, ow);
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

// =====================================================================================================================
// Fused stem FORWARD: input normalisation (hi + lo bf16 split) + 7x7/2 convolution + folded BatchNorm bias + ReLU + 3x3/2
// max pool in ONE persistent kernel.  Replaces k_prep_input -> implicit GEMM (K = 448) -> k_maxpool_fwd, which wrote and
// re-read the padded hi/lo image (110 MB) and the 112 x 112 x 64 stem output (411 MB) per 256-image forward; here the stem
// output never leaves the chip (the backward needs only the pool's argmax codes, code 15 = dead ReLU, and 1-bit signs).
//
// A workgroup loops over 8 x 8 tiles of POOLED positions.  Per tile: the 39 x 39 input patch behind the 17 x 17 stem
// outputs the tile's windows touch is staged in LDS as two [39][40 px][4 ch] bf16 planes (hi, lo; zeros outside the
// image); the convolution is the same row-tap implicit GEMM as before (a tap = one filter row of 8 px x 4 ch, hi taps then
// lo taps, same K order => same fp32 sums), with M = the 289 stem positions of the tile, read straight from the patch --
// every fragment address is patch base + an immediate -- and the 64 x 224 weight matrix resident in LDS for the whole
// launch; operands are swapped so a lane owns 4 consecutive channels of one position and the ReLU'd bf16 tile goes to LDS
// with 8-byte writes; the pool reads 16-byte channel groups from it and writes p1, the argmax codes and the sign bits.
// =====================================================================================================================
namespace {
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int SF_PT = 8;                        // pooled tile side
constexpr int SF_R = 2 * SF_PT + 1;             // stem-output region side (17)
constexpr int SF_NPOS = SF_R * SF_R;            // 289
constexpr int SF_PH = 2 * (SF_R - 1) + 7;       // patch rows (39)
constexpr int SF_PW = 40;                       // patch row stride in pixels (39 used + 1: a K slice reads 8 px from 2*16)
constexpr int SF_PLANE = SF_PH * SF_PW * 8;     // bytes per hi / lo plane (12 480)
constexpr int SF_WROW = 464;                    // weight row stride in LDS (224 k x 2 B + 16 pad: conflict-free b128 reads)
constexpr int SF_YROW = 144;                    // stem-output tile row stride (64 ch x 2 B + 16 pad)
constexpr int SF_W_BYTES = 64 * SF_WROW;        // 29 696
constexpr int SF_T_BYTES = SF_NPOS * SF_YROW;   // 41 616 (>= 2 * SF_PLANE: the tile aliases the patch)
static_assert(SF_T_BYTES >= 2 * SF_PLANE, "the stem-output tile must cover the patch it aliases");

struct StemNorm { float mean[3], istd[3]; };
// the normalisation must round exactly like k_prep_input (engine_aux.hip): multiply, subtract, multiply -- no contraction
#pragma clang fp contract(off)

__device__ __forceinline__ uint16_t sf_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <bool SRC_U8>
__global__ __launch_bounds__(256, 2) void k_stem_fwd_fused(const void* __restrict__ src, const uint16_t* __restrict__ wgt,
                                                           int wgt_row_stride, const float* __restrict__ bias,
                                                           uint4* __restrict__ p1, uint2* __restrict__ arg,
                                                           uint8_t* __restrict__ sign, int n, int h, int w, StemNorm nm) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[SF_W_BYTES + SF_T_BYTES];
  uint8_t* sW = lds;
  uint8_t* sP = lds + SF_W_BYTES;               // patch (hi plane, lo plane) / stem-output tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int kq = lane >> 5;
  const int oh = h >> 1, ow = w >> 1, oh2 = oh >> 1, ow2 = ow >> 1;
  const int tiles_x = (ow2 + SF_PT - 1) / SF_PT, tiles_y = (oh2 + SF_PT - 1) / SF_PT;
  const int n_tiles = n * tiles_y * tiles_x;

  // weights: [64][224] of the hi half (the lo taps use the same values), once per workgroup
  for (int i = tid; i < 64 * 28; i += 256) {
    const int row = i / 28, ch = i - row * 28;
    *reinterpret_cast<uint4*>(sW + row * SF_WROW + ch * 16) =
        *reinterpret_cast<const uint4*>(wgt + (size_t)row * wgt_row_stride + ch * 8);
  }
  // per-lane constants: the five 32-position M tiles of this wave half, bias of the lane's 16 channels
  uint32_t a_off[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    int p = wm * 160 + i * 32 + (lane & 31);
    p = p < SF_NPOS ? p : SF_NPOS - 1;          // rows past the region recompute the last position; never stored
    const int py = p / SF_R, px = p - py * SF_R;
    a_off[i] = (uint32_t)(((2 * py) * SF_PW + 2 * px) * 8 + kq * 16);
  }
  const uint32_t w_off = (uint32_t)((wn * 32 + (lane & 31)) * SF_WROW + kq * 16);
  float bz[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bz[r] = bias ? bias[wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq] : 0.f;

  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int img = t / (tiles_y * tiles_x);
    const int tr = t - img * (tiles_y * tiles_x);
    const int q0y = (tr / tiles_x) * SF_PT, q0x = (tr % tiles_x) * SF_PT;
    const int in_y0 = 4 * q0y - 5, in_x0 = 4 * q0x - 5;          // input pixel of patch (0, 0)
    __syncthreads();                                               // previous tile's pool is done with the LDS tile
    // ---- stage the patch: (x - mean) / std as hi + lo bf16, zeros outside the image and in the 4th channel.
    //      All of a thread's loads (7 pixels x 3 channels) are issued before the first conversion / LDS store.
    {
      constexpr int U = (SF_PH * SF_PW + 255) / 256;               // 7
      float v01[U][3];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = u * 256 + tid;
        const int pr = i / SF_PW, pc = i - pr * SF_PW;
        const int y = in_y0 + pr, x = in_x0 + pc;
        ok[u] = i < SF_PH * SF_PW && (unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w && pc < SF_PH;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          v01[u][c] = 0.f;
          if (ok[u]) {
            if (SRC_U8) v01[u][c] = (float)((const uint8_t*)src)[(((size_t)img * h + y) * w + x) * 3 + c];
            else v01[u][c] = ((const float*)src)[(((size_t)img * 3 + c) * h + y) * w + x];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = u * 256 + tid;
        if (i >= SF_PH * SF_PW) continue;
        uint16_t hv[3] = {0, 0, 0}, lv[3] = {0, 0, 0};
        if (ok[u]) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float x01 = SRC_U8 ? v01[u][c] * (1.0f / 255.0f) : v01[u][c];
            const float v = (x01 - nm.mean[c]) * nm.istd[c];
            hv[c] = sf_f2bf(v);
            lv[c] = sf_f2bf(v - __uint_as_float((uint32_t)hv[c] << 16));
          }
        }
        *reinterpret_cast<uint2*>(sP + i * 8) = make_uint2(hv[0] | ((uint32_t)hv[1] << 16), hv[2]);
        *reinterpret_cast<uint2*>(sP + SF_PLANE + i * 8) = make_uint2(lv[0] | ((uint32_t)lv[1] << 16), lv[2]);
      }
    }
    __syncthreads();
    // ---- implicit GEMM: 2 planes x 7 row taps x 2 k-steps, D^T accumulators (register -> channel, lane -> position)
    f32x16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = bz[r];
#pragma unroll
    for (int plane = 0; plane < 2; ++plane)
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sW + w_off + r * 64 + ks * 32);
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(sP + a_off[i] + plane * SF_PLANE + r * (SF_PW * 8) + ks * 32);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, acc[i], 0, 0, 0);
          }
        }
    __syncthreads();                                               // every wave is done reading the patch
    // ---- ReLU, bf16, into the LDS tile [position][64 ch]
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int p = wm * 160 + i * 32 + (lane & 31);
      if (p < SF_NPOS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t lo2 = pack2(fmaxf(acc[i][4 * q], 0.f), fmaxf(acc[i][4 * q + 1], 0.f));
          const uint32_t hi2 = pack2(fmaxf(acc[i][4 * q + 2], 0.f), fmaxf(acc[i][4 * q + 3], 0.f));
          *reinterpret_cast<uint2*>(sP + p * SF_YROW + (wn * 32 + 8 * q + 4 * kq) * 2) = make_uint2(lo2, hi2);
        }
      }
    }
    __syncthreads();
    // ---- 3x3/2 max pool (pad 1): first maximum in scan order, code 15 when the maximum is <= 0
    for (int i = tid; i < SF_PT * SF_PT * 8; i += 256) {
      const int c = i & 7, q = i >> 3;
      const int qy = q / SF_PT, qx = q - qy * SF_PT;
      const int gy = q0y + qy, gx = q0x + qx;
      if (gy >= oh2 || gx >= ow2) continue;
      float m[8];
      uint32_t code[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; code[j] = 0; }
      for (int ky = 0; ky < 3; ++ky) {
        const int y1 = 2 * gy - 1 + ky;
        if ((unsigned)y1 >= (unsigned)oh) continue;
        for (int kx = 0; kx < 3; ++kx) {
          const int x1 = 2 * gx - 1 + kx;
          if ((unsigned)x1 >= (unsigned)ow) continue;
          const uint4 v = *reinterpret_cast<const uint4*>(sP + ((2 * qy + ky) * SF_R + 2 * qx + kx) * SF_YROW + c * 16);
          const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float f = (j & 1) ? __uint_as_float(wv[j >> 1] & 0xFFFF0000u) : __uint_as_float(wv[j >> 1] << 16);
            if (f > m[j]) { m[j] = f; code[j] = (uint32_t)(ky * 3 + kx); }
          }
        }
      }
      const size_t o = (((size_t)img * oh2 + gy) * ow2 + gx) * 8 + c;
      p1[o] = make_uint4(pack2(m[0], m[1]), pack2(m[2], m[3]), pack2(m[4], m[5]), pack2(m[6], m[7]));
      uint32_t sb = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (m[j] > 0.f) sb |= 1u << j; else code[j] = 15u;
      }
      if (sign) sign[o] = (uint8_t)sb;
      if (arg) arg[o] = make_uint2(code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24),
                                   code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24));
    }
  }
}
#pragma clang fp contract(fast)
}  // namespace

extern "C" int rart_engine_stem_fwd_fused(const void* src, int src_is_u8, const void* wgt, int wgt_row_stride, const float* bias,
                                          void* p1, void* argmax_out, void* sign_out, int n, int h, int w,
                                          const float* mean_host, const float* std_host, rart_stream_t stream) {
  RART_CHECK_ARG(src && wgt && p1 && n > 0 && h % 4 == 0 && w % 4 == 0 && h >= 4 && w >= 4 && wgt_row_stride >= 224 &&
                 wgt_row_stride % 8 == 0, "rart_engine_stem_fwd_fused: bad arguments (h, w multiples of 4)");
  StemNorm nm;
  for (int c = 0; c < 3; ++c) {
    nm.mean[c] = mean_host ? mean_host[c] : 0.f;
    nm.istd[c] = std_host ? 1.0f / std_host[c] : 1.f;
  }
  const long long tiles = (long long)n * ((h / 4 + SF_PT - 1) / SF_PT) * ((w / 4 + SF_PT - 1) / SF_PT);
  const int grid = (int)(tiles < 512 ? tiles : 512);              // persistent: 2 workgroups per CU
  if (src_is_u8)
    hipLaunchKernelGGL(k_stem_fwd_fused<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (const uint16_t*)wgt,
                       wgt_row_stride, bias, (uint4*)p1, (uint2*)argmax_out, (uint8_t*)sign_out, n, h, w, nm);
  else
    hipLaunchKernelGGL(k_stem_fwd_fused<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (const uint16_t*)wgt,
                       wgt_row_stride, bias, (uint4*)p1, (uint2*)argmax_out, (uint8_t*)sign_out, n, h, w, nm);
  RART_CHECK_LAUNCH("rart_engine_stem_fwd_fused");
  return RART_OK;
}
