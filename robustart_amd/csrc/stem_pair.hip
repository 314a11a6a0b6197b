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
          const uint2 cd = sArg[c * NPOOL + lp];            // chunk-major: the lanes of a wave walk consecutive windows
          const uint4 dh = sDp[c * NPOOL + lp], dl = sDp[(4 + c) * NPOOL + lp];
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
  uint4* sDp = reinterpret_cast<uint4*>(lds + DZ_BYTES);                           // [2][4][NPOOL] (chunk-major)
  uint2* sArg = reinterpret_cast<uint2*>(lds + DZ_BYTES + DP_BYTES);               // [4][NPOOL]
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
        sDp[(pl * 4 + c) * NPOOL + pos] = val;
      } else {
        uint4 val = make_uint4(0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu);
        if (ok) val = arg[base * 4 + half * 2 + (v - 8)];
        sArg[(2 * (v - 8)) * NPOOL + pos] = make_uint2(val.x, val.y);
        sArg[(2 * (v - 8) + 1) * NPOOL + pos] = make_uint2(val.z, val.w);
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

// =====================================================================================================================
// Fused stem FORWARD on pairs: input normalisation (hi + lo bf16 split) + 7x7/2 convolution (x_lo.w_hi + x_hi.w_lo + x_hi.w_hi per K
// step, fp32 accumulation) + folded BatchNorm bias + ReLU + hi + lo split + 3x3/2 max pool of the pair VALUES in ONE persistent kernel.
// Replaces rart_engine_prep_input -> rart_gemm_pair_bf16 (row taps, K = 224) -> rart_engine_maxpool_pair, which wrote and re-read the
// padded hi / lo image and the 112 x 112 x 64 stem-output pair (2 x 411 MB per 256 images): 0.77 ms of a 10.1 ms forward.
//
// Same structure as k_stem_fwd_fused (csrc/stem_fused.hip): a workgroup loops over 8 x 8 tiles of POOLED positions; the 39 x 39 input
// patch behind the 17 x 17 stem outputs the tile's windows touch is staged in LDS as two [39][40 px][4 ch] bf16 planes; the convolution
// is the row-tap implicit GEMM (a tap = one filter row of 8 px x 4 ch) read straight from the patch, operands swapped so a lane owns 4
// consecutive channels of one position.  What the pair changes: both weight planes are LDS resident (2 x 29.7 KB), and the stem-output
// tile is kept as fp32 -- the value hi + lo of the pair the unfused chain would have written, so the pool's comparisons, argmax codes and
// outputs are bit-identical to it -- 78.6 KB: one workgroup of 512 threads per CU (138 KB of LDS).
// =====================================================================================================================
namespace {
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int SP_PT = 8;                        // pooled tile side
constexpr int SP_R = 2 * SP_PT + 1;             // stem-output region side (17)
constexpr int SP_NPOS = SP_R * SP_R;            // 289
constexpr int SP_PH = 2 * (SP_R - 1) + 7;       // patch rows (39)
constexpr int SP_PW = 40;                       // patch row stride in pixels
constexpr int SP_PLANE = SP_PH * SP_PW * 8;     // bytes per hi / lo plane (12 480)
constexpr int SP_WROW = 464;                    // weight row stride in LDS (224 k x 2 B + 16 pad: conflict-free b128 reads)
constexpr int SP_YROW = 272;                    // stem-output tile row stride: 64 fp32 + 16 B pad
constexpr int SP_W_BYTES = 64 * SP_WROW;        // 29 696 per plane
constexpr int SP_T_BYTES = SP_NPOS * SP_YROW;   // 78 608 (>= 2 * SP_PLANE: the tile aliases the patch)
constexpr int SP_NT = 3;                        // 32-position M tiles per wave row (4 wave rows x 3 >= 10 tiles)
static_assert(SP_T_BYTES >= 2 * SP_PLANE, "the stem-output tile must cover the patch it aliases");

struct StemNormP { float mean[3], istd[3]; };
// the normalisation must round exactly like k_prep_input (engine_aux.hip): multiply, subtract, multiply -- no contraction
#pragma clang fp contract(off)

template <bool SRC_U8>
__global__ __launch_bounds__(512, 1) void k_stem_fwd_pair(const void* __restrict__ src, const uint16_t* __restrict__ w_h,
                                                          const uint16_t* __restrict__ w_l, const float* __restrict__ bias,
                                                          uint4* __restrict__ p1_h, uint4* __restrict__ p1_l, uint2* __restrict__ arg,
                                                          uint8_t* __restrict__ sign, int n, int h, int w, StemNormP nm) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * SP_W_BYTES + SP_T_BYTES];
  uint8_t* sW = lds;                            // [hi | lo][64][SP_WROW]
  uint8_t* sP = lds + 2 * SP_W_BYTES;           // patch (hi plane, lo plane) / stem-output tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int kq = lane >> 5;
  const int oh = h >> 1, ow = w >> 1, oh2 = oh >> 1, ow2 = ow >> 1;
  const int tiles_x = (ow2 + SP_PT - 1) / SP_PT, tiles_y = (oh2 + SP_PT - 1) / SP_PT;
  const int n_tiles = n * tiles_y * tiles_x;

  // weights: [64][224] per plane, once per workgroup
  for (int i = tid; i < 2 * 64 * 28; i += 512) {
    const int pl = i / (64 * 28), j = i - pl * (64 * 28), row = j / 28, ch = j - row * 28;
    *reinterpret_cast<uint4*>(sW + pl * SP_W_BYTES + row * SP_WROW + ch * 16) =
        *reinterpret_cast<const uint4*>((pl ? w_l : w_h) + (size_t)row * 224 + ch * 8);
  }
  uint32_t a_off[SP_NT];
#pragma unroll
  for (int i = 0; i < SP_NT; ++i) {
    int p = (wm * SP_NT + i) * 32 + (lane & 31);
    p = p < SP_NPOS ? p : SP_NPOS - 1;          // rows past the region recompute the last position; never stored
    const int py = p / SP_R, px = p - py * SP_R;
    a_off[i] = (uint32_t)(((2 * py) * SP_PW + 2 * px) * 8 + kq * 16);
  }
  const uint32_t w_off = (uint32_t)((wn * 32 + (lane & 31)) * SP_WROW + kq * 16);
  float bz[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bz[r] = bias ? bias[wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq] : 0.f;

  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int img = t / (tiles_y * tiles_x);
    const int tr = t - img * (tiles_y * tiles_x);
    const int q0y = (tr / tiles_x) * SP_PT, q0x = (tr % tiles_x) * SP_PT;
    const int in_y0 = 4 * q0y - 5, in_x0 = 4 * q0x - 5;          // input pixel of patch (0, 0)
    __syncthreads();                                               // previous tile's pool is done with the LDS tile
    // ---- stage the patch: (x - mean) / std as hi + lo bf16, zeros outside the image and in the 4th channel
    {
      constexpr int U = (SP_PH * SP_PW + 511) / 512;               // 4
      float v01[U][3];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = u * 512 + tid;
        const int pr = i / SP_PW, pc = i - pr * SP_PW;
        const int y = in_y0 + pr, x = in_x0 + pc;
        ok[u] = i < SP_PH * SP_PW && (unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w && pc < SP_PH;
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
        const int i = u * 512 + tid;
        if (i >= SP_PH * SP_PW) continue;
        uint32_t hv[3] = {0, 0, 0}, lv[3] = {0, 0, 0};
        if (ok[u]) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float x01 = SRC_U8 ? v01[u][c] * (1.0f / 255.0f) : v01[u][c];
            const float v = (x01 - nm.mean[c]) * nm.istd[c];
            hv[c] = sp_pack2(v, 0.f) & 0xFFFFu;
            lv[c] = sp_pack2(v - __uint_as_float(hv[c] << 16), 0.f) & 0xFFFFu;
          }
        }
        *reinterpret_cast<uint2*>(sP + i * 8) = make_uint2(hv[0] | (hv[1] << 16), hv[2]);
        *reinterpret_cast<uint2*>(sP + SP_PLANE + i * 8) = make_uint2(lv[0] | (lv[1] << 16), lv[2]);
      }
    }
    __syncthreads();
    // ---- implicit GEMM: 7 row taps x 2 k-steps x 3 products, D^T accumulators (register -> channel, lane -> position)
    f32x16 acc[SP_NT];
#pragma unroll
    for (int i = 0; i < SP_NT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = bz[r];
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(sW + w_off + r * 64 + ks * 32);
        const bf16x8 wl = *reinterpret_cast<const bf16x8*>(sW + SP_W_BYTES + w_off + r * 64 + ks * 32);
#pragma unroll
        for (int i = 0; i < SP_NT; ++i) {
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sP + a_off[i] + r * (SP_PW * 8) + ks * 32);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(sP + a_off[i] + SP_PLANE + r * (SP_PW * 8) + ks * 32);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc[i], 0, 0, 0);
        }
      }
    __syncthreads();                                               // every wave is done reading the patch
    // ---- ReLU, the value of the hi + lo pair (what the unfused chain stores), into the LDS tile [position][64 ch] fp32
#pragma unroll
    for (int i = 0; i < SP_NT; ++i) {
      const int p = (wm * SP_NT + i) * 32 + (lane & 31);
      if (p < SP_NPOS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = fmaxf(acc[i][4 * q + k], 0.f);
          const uint32_t h0 = sp_pack2(v[0], v[1]), h1 = sp_pack2(v[2], v[3]);
          const uint32_t l0 = sp_pack2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xFFFF0000u));
          const uint32_t l1 = sp_pack2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xFFFF0000u));
          *reinterpret_cast<float4*>(sP + p * SP_YROW + (wn * 32 + 8 * q + 4 * kq) * 4) =
              make_float4(__uint_as_float(h0 << 16) + __uint_as_float(l0 << 16), __uint_as_float(h0 & 0xFFFF0000u) + __uint_as_float(l0 & 0xFFFF0000u),
                          __uint_as_float(h1 << 16) + __uint_as_float(l1 << 16), __uint_as_float(h1 & 0xFFFF0000u) + __uint_as_float(l1 & 0xFFFF0000u));
        }
      }
    }
    __syncthreads();
    // ---- 3x3/2 max pool (pad 1) of the pair values: first maximum in scan order, code 15 when the maximum is <= 0
    {
      const int c = tid & 7, q = tid >> 3;          // 64 pooled positions x 8 channel groups = 512 threads
      const int qy = q / SP_PT, qx = q - qy * SP_PT;
      const int gy = q0y + qy, gx = q0x + qx;
      if (gy < oh2 && gx < ow2) {
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
            const float* tp = reinterpret_cast<const float*>(sP + ((2 * qy + ky) * SP_R + 2 * qx + kx) * SP_YROW) + c * 8;
            const float4 v0 = *reinterpret_cast<const float4*>(tp), v1 = *reinterpret_cast<const float4*>(tp + 4);
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (f[j] > m[j]) { m[j] = f[j]; code[j] = (uint32_t)(ky * 3 + kx); }
          }
        }
        const size_t o = (((size_t)img * oh2 + gy) * ow2 + gx) * 8 + c;
        uint32_t hv[4], lv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          hv[k] = sp_pack2(m[2 * k], m[2 * k + 1]);
          lv[k] = sp_pack2(m[2 * k] - __uint_as_float(hv[k] << 16), m[2 * k + 1] - __uint_as_float(hv[k] & 0xFFFF0000u));
        }
        p1_h[o] = make_uint4(hv[0], hv[1], hv[2], hv[3]);
        p1_l[o] = make_uint4(lv[0], lv[1], lv[2], lv[3]);
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
}
#pragma clang fp contract(fast)
}  // namespace

extern "C" int rart_engine_stem_fwd_fused_pair(const void* src, int src_is_u8, const void* wgt_hi, const void* wgt_lo, const float* bias,
                                               void* p1_hi, void* p1_lo, void* argmax_out, void* sign_out, int n, int h, int w,
                                               const float* mean_host, const float* std_host, rart_stream_t stream) {
  RART_CHECK_ARG(src && wgt_hi && wgt_lo && p1_hi && p1_lo && n > 0 && h % 4 == 0 && w % 4 == 0 && h >= 4 && w >= 4,
                 "rart_engine_stem_fwd_fused_pair: bad arguments (h, w multiples of 4)");
  StemNormP nm;
  for (int c = 0; c < 3; ++c) {
    nm.mean[c] = mean_host ? mean_host[c] : 0.f;
    nm.istd[c] = std_host ? 1.0f / std_host[c] : 1.f;
  }
  const long long tiles = (long long)n * ((h / 4 + SP_PT - 1) / SP_PT) * ((w / 4 + SP_PT - 1) / SP_PT);
  const int grid = (int)(tiles < 256 ? tiles : 256);              // persistent: one workgroup per CU
  if (src_is_u8)
    hipLaunchKernelGGL(k_stem_fwd_pair<true>, dim3(grid), dim3(512), 0, (hipStream_t)stream, src, (const uint16_t*)wgt_hi,
                       (const uint16_t*)wgt_lo, bias, (uint4*)p1_hi, (uint4*)p1_lo, (uint2*)argmax_out, (uint8_t*)sign_out, n, h, w, nm);
  else
    hipLaunchKernelGGL(k_stem_fwd_pair<false>, dim3(grid), dim3(512), 0, (hipStream_t)stream, src, (const uint16_t*)wgt_hi,
                       (const uint16_t*)wgt_lo, bias, (uint4*)p1_hi, (uint4*)p1_lo, (uint2*)argmax_out, (uint8_t*)sign_out, n, h, w, nm);
  RART_CHECK_LAUNCH("rart_engine_stem_fwd_fused_pair");
  return RART_OK;
}
