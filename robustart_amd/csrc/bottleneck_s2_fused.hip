// The FIRST Bottleneck of ResNet-50's layer2 / layer3 -- the stride-2 blocks with a projection shortcut -- forward, as ONE kernel
// (gfx950):      out = relu( W3 . relu( W2 *s2 relu(W1 . x + b1) + b2 ) + Wd . x[::2, ::2] + (b3 + bd) )
//   layer2 block 0:  x 56 x 56 x 256 -> a1 56 x 56 x 128 -> a2 28 x 28 x 128 -> out 28 x 28 x 512
//   layer3 block 0:  x 28 x 28 x 512 -> a1 28 x 28 x 256 -> a2 14 x 14 x 256 -> out 14 x 14 x 1024
// As four implicit-GEMM launches (1x1, 3x3 / 2, projection 1x1 / 2, 1x1 + residual) these blocks took 548 / 407 us per forward at
// B = 256 (profiles/r02_igemm_per_shape.txt): a1, a2 and the projection output all made a round trip through HBM.  Same structure
// as bottleneck28_fused.hip (read that file first):
//   * a workgroup (CM / 32 waves) owns a 7 x 7 tile of OUTPUT positions.  The 3x3 / 2 needs a1 on the 15 x 15 input positions
//     (2 oy0 - 1 .. 2 oy0 + 13)^2 around it: stage A computes a1 = relu(W1 . x + b1) on the 16 x 16 grid that contains them (8
//     position tiles; positions outside the image are the 3x3's zero padding), x streaming through LDS in 64-channel slices
//     (global_load_lds_dwordx4, two slices in flight inside the memory of the later image), a1 written as the chunk-major image
//     of k_bottleneck28 (CM / 8 planes x 257 slots x 16 B);
//   * stage B: the 9 taps x CM / 64 K steps over TWO 32-slot position tiles (4 output rows x 8 slots, 7 valid): slot (r, c) reads the
//     image at grid position (2 r + 1 + dy, 2 c + 1 + dx) -- the stride lives in the slot -> address map, the taps are the same
//     immediate offsets as in the stride-1 kernels; a2 goes back into LDS (64 slots per plane, over the dead a1 image);
//   * stage C: 4 rounds of CM output channels, all 4 x 2 accumulator tiles live at once (128 VGPRs): first the projection shortcut
//     (K = c_in: x at the 49 tile centres straight from L2 into the positions operand, read ONCE per wave for all four rounds),
//     then W3 . a2 from LDS, then the wave-private transposition to 64-byte row segments, ReLU, sign bits.
//   * a wave owns 32 channels x all position tiles in every stage; all four weight tables in fragment order from L2.
// With only two position tiles behind every weight fragment the launch is bound by the weight stream from L2 (0.74 / 2.9 MB per
// tile), not by HBM; the 14 x 14 -> 7 x 7 block of layer4 (a1 image 263 KB) does not fit and stays on the implicit GEMM.
//
// LDS: layer2 65 792 B image + 4 x 4 032 B staging = 80 KiB -> 2 workgroups per CU; layer3 131 584 + 8 x 4 032 = 160 KiB.
//
// Reference step: Bottleneck.forward with a downsample branch of the public ResNet-50 v1.5 (RobustART/model/__init__.py:1 ->
// absent submodule; robustart_amd/model/resnet_torch.py:27-35) inside every forward of the attacks / evaluations
// (RobustART/noise/utils/adv/attack.py:21-22, Attacks/autoattack/autopgd_base.py:271-289).
#include "rart_common.h"
#include "rart_bf16_helpers.h"

struct RartBneckS2Desc {
  const uint16_t* x;        // [n][hin][hin][c_in] bf16
  const uint16_t* w1;       // [c_mid][c_in]     fragment order (rart_pack_frag_bf16(rows c_mid, k c_in))
  const uint16_t* w2;       // [c_mid][9*c_mid]  fragment order, k = (r*3+s)*c_mid + c, taps (r-1, s-1)
  const uint16_t* w3;       // [c_out][c_mid]    fragment order
  const uint16_t* wd;       // [c_out][c_in]     fragment order: the projection shortcut
  const float* b1;
  const float* b2;
  const float* b3;          // conv3 bias + shortcut bias, or null
  uint8_t* m1;              // sign bits of a1 [n][hin][hin][c_mid/8], of a2 [n][hout][hout][c_mid/8], of out [..][c_out/8]; nullable
  uint8_t* m2;
  uint8_t* m3;
  uint16_t* out;            // [n][hout][hout][c_out]
};

namespace {
using namespace rart_bf16;
__device__ __attribute__((aligned(16))) const uint32_t g_s2_zero16[4] = {0u, 0u, 0u, 0u};   // source of grid slots outside the image

constexpr int S2_PLANE = (256 + 1) * 16;                 // 16 x 16 grid positions + 1 slot: 16 mod 256 bytes (as k_bottleneck28)
constexpr int S2_SLICE = 256 * 128;                      // one 64-channel slice of x: 256 slots x 128 B
constexpr int S2_LDE = 36;                               // staging row: 32 floats + 4
constexpr int S2_STG = 28 * S2_LDE * 4;                  // 4 032 B per wave
constexpr int S2_T = 7;                                  // output tile side

template <int CIN, int CM, int COUT, int HIN>
__global__ __launch_bounds__(CM * 2, CM == 128 ? 2 : 1) void k_bottleneck_s2(const RartBneckS2Desc d) {
  constexpr int NW = CM / 32;                            // waves per workgroup
  constexpr int NT = NW * 64;
  constexpr int KA = CIN / 64;                           // x slices of stage A / K steps of the shortcut
  constexpr int NPL = CM / 8;                            // planes of the a1 / a2 images
  constexpr int IMG = NPL * S2_PLANE;
  constexpr int HOUT = HIN / 2, TPS = HOUT / S2_T;       // tiles per image side
  constexpr int RT = COUT / 32;                          // row tiles of the W3 / Wd tables
  static_assert(2 * S2_SLICE <= IMG, "two x slices must fit the image memory");
  static_assert(COUT == 4 * CM && HOUT % S2_T == 0 && (32 % NW) == 0, "geometry");
  static_assert(IMG + NW * S2_STG <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) uint8_t lds[IMG + NW * S2_STG];
  uint8_t* const sImg = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p32 = lane & 31, h = lane >> 5;
  const int lb = (int)rart_xcd_block(blockIdx.x, gridDim.x);          // the tiles of an image on one XCD (shared halo in its L2)
  const int img = lb / (TPS * TPS), tile = lb - img * (TPS * TPS);
  const int oy0 = (tile / TPS) * S2_T, ox0 = (tile % TPS) * S2_T;      // first output position of the tile
  const int gy0 = 2 * oy0 - 1, gx0 = 2 * ox0 - 1;                       // input position of grid slot (0, 0)
  const long long ipos0 = (long long)img * HIN * HIN;                   // raster index of the image's first input position
  const long long opos0 = (long long)img * HOUT * HOUT;

  // ================================ stage A: a1 = x . W1^T on the 16 x 16 grid (8 position tiles) ================================
  {
    f32x16 acc[8];
    constexpr int QN = 32 / NW;                          // direct-load instructions of this wave per slice (8 slots each)
    const char* xsrc[QN];
    uint32_t xdst[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int j = wave + NW * q, slot = 8 * j + (lane >> 3);           // slot = grid row * 16 + grid column
      const int gy = gy0 + (slot >> 4), gx = gx0 + (slot & 15);
      const int chunk = (lane & 7) ^ ((slot >> 1) & 7);
      const bool in = (unsigned)gy < (unsigned)HIN && (unsigned)gx < (unsigned)HIN;
      xsrc[q] = in ? reinterpret_cast<const char*>(d.x + (ipos0 + gy * HIN + gx) * CIN + chunk * 8) : nullptr;
      xdst[q] = (uint32_t)__builtin_amdgcn_readfirstlane(8 * j) * 128u;
    }
#define RART_S2_ISSUE(S, BUF)                                                                                   \
  _Pragma("unroll") for (int q = 0; q < QN; ++q) {                                                              \
    const char* s_ = xsrc[q] ? xsrc[q] + (S)*128 : reinterpret_cast<const char*>(g_s2_zero16);                  \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s_,                         \
                                     (__attribute__((address_space(3))) void*)(sImg + (BUF)*S2_SLICE + xdst[q]), 16, 0, 0); \
  }
    // weights of this wave: a1 channels 32 wave .. +31; fragment (K step st, row tile wave, ks) of the [CM][CIN] table
    const uint16_t* wp = d.w1 + (size_t)wave * 2048 + lane * 8;
    bf16x8 wq[2][4];
#define RART_S2_LOADW(S, SET)                                                                                   \
  _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                                 \
    wq[SET][f] = *reinterpret_cast<const bf16x8*>(wp + (size_t)((S)*NW) * 2048 + f * 512);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (d.b1) bv = *reinterpret_cast<const f32x4*>(d.b1 + wave * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = bv[i];
    }
    RART_S2_ISSUE(0, 0)
    RART_S2_LOADW(0, 0)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    // fragment read of tile t: slot r = t * 32 + p32, logical chunk 2 ks + h at position (2 ks + h) ^ ((r >> 1) & 7)
    const uint32_t xrow = (uint32_t)(p32 * 128), xsw = (uint32_t)((p32 >> 1) & 7);
#pragma unroll
    for (int s = 0; s < KA; ++s) {
      const int buf = s & 1;
      if (s + 1 < KA) {
        RART_S2_ISSUE(s + 1, buf ^ 1)
        if (buf) { RART_S2_LOADW(s + 1, 0) } else { RART_S2_LOADW(s + 1, 1) }
      }
      __builtin_amdgcn_sched_barrier(0);
      const uint8_t* xb = sImg + buf * S2_SLICE + xrow;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint32_t co = ((uint32_t)(2 * ks + h) ^ xsw) << 4;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const bf16x8 pf = *reinterpret_cast<const bf16x8*>(xb + t * 32 * 128 + co);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[buf][ks], pf, acc[t], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_waitcnt(0);        // the next slice (and the next weights) have landed
      __syncthreads();
    }
#undef RART_S2_ISSUE
#undef RART_S2_LOADW
    // every wave is past its last slice read: the memory becomes the a1 image.  lane: grid position (2t + prow, px) of tile t,
    // channels wave*32 + 8g + 4h + (0..3) -> 8 bytes of chunk wave*4 + g; positions outside the image are zero
    const int px = p32 & 15, prow = p32 >> 4;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int ry = 2 * t + prow, gy = gy0 + ry, gx = gx0 + px;
      const bool in = (unsigned)gy < (unsigned)HIN && (unsigned)gx < (unsigned)HIN;
      uint8_t* dst = sImg + (ry * 16 + px) * 16 + (wave * 4) * S2_PLANE + h * 8;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint32_t w0 = relu_bf16x2(pack_bf16x2(acc[t][4 * g], acc[t][4 * g + 1]));
        uint32_t w1 = relu_bf16x2(pack_bf16x2(acc[t][4 * g + 2], acc[t][4 * g + 3]));
        if (!in) w0 = w1 = 0u;
        *reinterpret_cast<uint2*>(dst + g * S2_PLANE) = make_uint2(w0, w1);
      }
    }
  }
  __syncthreads();
  // sign bits of a1 (for the backward pass): every input position belongs to exactly one tile -- grid rows / columns 1..14
  if (d.m1) {
    for (int i = tid; i < 14 * 14 * NPL; i += NT) {
      const int p = i / NPL, chunk = i - p * NPL, yy = p / 14, xx = p - yy * 14;
      const uint4 v = *reinterpret_cast<const uint4*>(sImg + chunk * S2_PLANE + ((yy + 1) * 16 + xx + 1) * 16);
      d.m1[(ipos0 + (gy0 + 1 + yy) * HIN + gx0 + 1 + xx) * NPL + chunk] = (uint8_t)sign_byte(v);
    }
  }

  // lane geometry of an OUTPUT tile t (stages B and C): tile-local row 4t + (p32 >> 3), column p32 & 7; row 7 / column 7 are padding
  const int ocol = p32 & 7, orow_in = p32 >> 3;
  // ================================ stage B: a2 = 3x3 / 2 over the a1 image, 2 output tiles x 32 channels per wave ==============
  {
    f32x16 acc[2];
    uint32_t abase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = 4 * t + orow_in, rc = r < S2_T ? r : S2_T - 1, cc = ocol < S2_T ? ocol : S2_T - 1;   // padding slots read a valid position
      abase[t] = (uint32_t)(((2 * rc + 1) * 16 + 2 * cc + 1) * 16 + h * S2_PLANE);
    }
    const uint16_t* wp = d.w2 + (size_t)wave * 2048 + lane * 8;              // fragment (st, wave, ks): (st * NW + wave) * 4 + ks
    // three register sets: the fragments of step st + 2 are requested before the MFMAs of step st (a step is only 8 MFMAs per
    // wave -- one step of cover is shorter than an L2 round trip)
    bf16x8 bq[3][4];
    constexpr int KH = CM / 64, NSB = 9 * KH;
#pragma unroll
    for (int s0 = 0; s0 < 2; ++s0)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bq[s0][ks] = *reinterpret_cast<const bf16x8*>(wp + (size_t)s0 * NW * 2048 + ks * 512);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (d.b2) bv = *reinterpret_cast<const f32x4*>(d.b2 + wave * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = bv[i];
    }
#pragma unroll
    for (int st = 0; st < NSB; ++st) {
      const int tap = st / KH, kh = st - tap * KH;
      if (st + 2 < NSB) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          bq[(st + 2) % 3][ks] = *reinterpret_cast<const bf16x8*>(wp + (size_t)(st + 2) * NW * 2048 + ks * 512);
      }
      __builtin_amdgcn_sched_barrier(0);                    // keep the prefetch ahead of this step's MFMAs
      const int toff = ((tap / 3 - 1) * 16 + (tap % 3 - 1)) * 16;        // tap (dy, dx) = (tap / 3 - 1, tap % 3 - 1)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 pf = *reinterpret_cast<const bf16x8*>(sImg + (int)abase[t] + toff + (kh * 8 + ks * 2) * S2_PLANE);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[st % 3][ks], pf, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();          // every wave is done reading a1: slots 0..63 of every plane become a2 (slot = t * 32 + p32)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      uint8_t* dst = sImg + (t * 32 + p32) * 16 + (wave * 4) * S2_PLANE + h * 8;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t w0 = relu_bf16x2(pack_bf16x2(acc[t][4 * g], acc[t][4 * g + 1]));
        const uint32_t w1 = relu_bf16x2(pack_bf16x2(acc[t][4 * g + 2], acc[t][4 * g + 3]));
        *reinterpret_cast<uint2*>(dst + g * S2_PLANE) = make_uint2(w0, w1);
      }
    }
  }
  __syncthreads();
  if (d.m2) {
    for (int i = tid; i < S2_T * S2_T * NPL; i += NT) {
      const int p = i / NPL, chunk = i - p * NPL, r = p / S2_T, c = p - r * S2_T;
      const uint4 v = *reinterpret_cast<const uint4*>(sImg + chunk * S2_PLANE + ((r >> 2) * 32 + (r & 3) * 8 + c) * 16);
      d.m2[(opos0 + (oy0 + r) * HOUT + ox0 + c) * NPL + chunk] = (uint8_t)sign_byte(v);
    }
  }

  // ================================ stage C: out = Wd . x[centres] + W3 . a2 + b, 4 rounds x 2 tiles of accumulators ============
  f32x16 acc[4][2];
#pragma unroll
  for (int rd = 0; rd < 4; ++rd)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (d.b3) bv = *reinterpret_cast<const f32x4*>(d.b3 + rd * CM + wave * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[rd][t][4 * g + i] = bv[i];
    }
  {
    // ---- the projection shortcut: positions operand straight from global memory (the tile's 49 centres were just streamed
    //      through stage A, so these are L2 hits); one group = (K step st, ks): 2 position fragments + 4 weight fragments, 8 MFMAs
    const uint16_t* xc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = 4 * t + orow_in, rc = r < S2_T ? r : S2_T - 1, cc = ocol < S2_T ? ocol : S2_T - 1;
      xc[t] = d.x + (ipos0 + (long long)(2 * (oy0 + rc)) * HIN + 2 * (ox0 + cc)) * CIN + h * 8;
    }
    const uint16_t* wp = d.wd + (size_t)wave * 2048 + lane * 8;              // fragment (st, rd * NW + wave, ks)
    bf16x8 xq[3][2], wq[3][4];                                               // three sets: two groups (16 MFMAs) of cover
#define RART_S2_LOADG(G, SET)                                                                                   \
  {                                                                                                             \
    const int st_ = (G) >> 2, ks_ = (G)&3;                                                                      \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                               \
      xq[SET][t] = *reinterpret_cast<const bf16x8*>(xc[t] + st_ * 64 + ks_ * 16);                               \
    _Pragma("unroll") for (int rd = 0; rd < 4; ++rd)                                                            \
      wq[SET][rd] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)st_ * RT + rd * NW) * 2048 + ks_ * 512);     \
  }
    RART_S2_LOADG(0, 0)
    RART_S2_LOADG(1, 1)
#pragma unroll
    for (int gidx = 0; gidx < KA * 4; ++gidx) {
      if (gidx + 2 < KA * 4) {
        if ((gidx + 2) % 3 == 0) { RART_S2_LOADG(gidx + 2, 0) } else if ((gidx + 2) % 3 == 1) { RART_S2_LOADG(gidx + 2, 1) } else { RART_S2_LOADG(gidx + 2, 2) }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rd = 0; rd < 4; ++rd)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[rd][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[gidx % 3][rd], xq[gidx % 3][t], acc[rd][t], 0, 0, 0);
    }
#undef RART_S2_LOADG
  }
  {
    // ---- W3 . a2: positions operand from the a2 planes in LDS (slot t * 32 + p32)
    const uint16_t* wp = d.w3 + (size_t)wave * 2048 + lane * 8;
    const uint32_t abase = (uint32_t)(p32 * 16 + h * S2_PLANE);
    bf16x8 wq[3][4];
#define RART_S2_LOADW3(G, SET)                                                                                  \
  {                                                                                                             \
    const int st_ = (G) >> 2, ks_ = (G)&3;                                                                      \
    _Pragma("unroll") for (int rd = 0; rd < 4; ++rd)                                                            \
      wq[SET][rd] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)st_ * RT + rd * NW) * 2048 + ks_ * 512);    \
  }
    RART_S2_LOADW3(0, 0)
    RART_S2_LOADW3(1, 1)
    constexpr int NG = (CM / 64) * 4;
#pragma unroll
    for (int gidx = 0; gidx < NG; ++gidx) {
      if (gidx + 2 < NG) {
        if ((gidx + 2) % 3 == 0) { RART_S2_LOADW3(gidx + 2, 0) } else if ((gidx + 2) % 3 == 1) { RART_S2_LOADW3(gidx + 2, 1) } else { RART_S2_LOADW3(gidx + 2, 2) }
      }
      __builtin_amdgcn_sched_barrier(0);
      const int st = gidx >> 2, ks = gidx & 3;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 pf = *reinterpret_cast<const bf16x8*>(sImg + abase + t * 32 * 16 + (st * 8 + ks * 2) * S2_PLANE);
#pragma unroll
        for (int rd = 0; rd < 4; ++rd)
          acc[rd][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[gidx % 3][rd], pf, acc[rd][t], 0, 0, 0);
      }
    }
#undef RART_S2_LOADW3
  }
  // ---- epilogue: per (round, tile) the 28 slots of 4 rows x 7 columns x 32 channels go through the wave's staging slice; then 4 lanes
  //      per position read 8 channels each: 64-byte row segments, two passes (16 + 12 slots)
  float* const sE = reinterpret_cast<float*>(lds + IMG + wave * S2_STG);
  const bool pvalid = ocol < S2_T;
  const int vp = orow_in * S2_T + ocol;                    // compact index of a slot inside its tile (0..27)
  const int cw = lane & 3, rw = lane >> 2;
#pragma unroll
  for (int rd = 0; rd < 4; ++rd) {
    const int ch0 = rd * CM + wave * 32;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (pvalid) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {acc[rd][t][4 * g], acc[rd][t][4 * g + 1], acc[rd][t][4 * g + 2], acc[rd][t][4 * g + 3]};
          *reinterpret_cast<f32x4*>(sE + vp * S2_LDE + 8 * g + 4 * h) = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int v = q * 16 + rw;                          // slot index 0..27 inside the tile: row v / 7, column v % 7
        const int vy = v / S2_T, vx = v - vy * S2_T, r = 4 * t + vy;
        if (v < 28 && r < S2_T) {
          const long long eoff = (opos0 + (oy0 + r) * HOUT + ox0 + vx) * COUT + ch0 + cw * 8;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sE + v * S2_LDE + cw * 8);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(sE + v * S2_LDE + cw * 8 + 4);
          const uint4 o = make_uint4(relu_bf16x2(pack_bf16x2(v0[0], v0[1])), relu_bf16x2(pack_bf16x2(v0[2], v0[3])),
                                     relu_bf16x2(pack_bf16x2(v1[0], v1[1])), relu_bf16x2(pack_bf16x2(v1[2], v1[3])));
          RART_LAB_STORE16(d.out + eoff, o);
          if (d.m3) d.m3[eoff >> 3] = (uint8_t)sign_byte(o);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ================================================================================================================================
// Backward-to-input of the same blocks as ONE kernel:
//     dx = m0 . ( W1^T . ( m1 . ( W2^T *s2 ( m2 . ( W3^T . g ) ) ) )  +  Wd^T . g  [only at even / even input positions] )
// g = gradient at the block output (already masked by that output's ReLU), m2 / m1 / m0 = sign bits of a2 / a1 / the block input.
// As seven implicit-GEMM launches (conv3^T, four parity classes of the 3x3 / 2, conv1^T, projection^T with a read-modify-write of
// dx) these took 622 / 450 us per backward at B = 256.  A workgroup owns a 14 x 14 tile of INPUT positions (the footprint of a 7 x 7
// output tile):
//   * stage A': d_a2 = m2 . (W3^T . g) on the 8 x 8 output positions (oy0 .. oy0 + 7)^2 the tile's 3x3^T reaches (positions past the
//     image are zero), two 32-slot tiles; g streams through LDS in 256-channel super-slices (four 64-channel sub-slices of 64
//     slots x 128 B, the forward kernel's swizzle), d_a2 lands in the region the epilogue later uses for staging (CM / 8 planes x 64
//     slots x 16 B);
//   * stage B': the transposed 3x3 / 2 by input-parity class (ph, pw): class positions (2i + ph, 2j + pw), i, j < 7 = two 32-slot
//     tiles; its taps are the filter taps r with (ph + 1 - r) even, reading d_a2 at grid (i + (ph + 1 - r) / 2, j + ...): 1 / 2 / 2 / 4
//     taps, every weight fragment of W2^T used by exactly one class.  d_a1 = m1 . (...) is written CLASS-major into the image
//     (slot = class * 64 + tile * 32 + lane position), so that
//   * stage C' runs per class too: dx[class positions][c_in] = W1^T . d_a1 from LDS and, for class (0, 0) only, + Wd^T . g as extra K
//     (g at (oy0 + i, ox0 + j) straight from L2) in the same accumulators; mask m0, bf16, 64-byte row segments.
// LDS: image without the padding slot (no sign read-back here: b128 fragment reads are conflict-free per quarter wave at any plane
// stride) 64 / 128 KiB + 16 / 32 KiB for d_a2 = staging: 80 / 160 KiB, as the forward kernel.
struct RartBneckS2BwdDesc {
  const uint16_t* g;        // [n][hout][hout][c_out] bf16
  const uint16_t* w3t;      // [c_mid][c_out] fragment order: conv3's transposed table
  const uint16_t* w2t;      // four parity-class tables [c_mid][ntaps * c_mid], each in fragment order, concatenated in class order
                            //   (0,0) (0,1) (1,0) (1,1): element offsets 0, 1, 3, 5 times c_mid * c_mid (k = tap * c_mid + c_out_of_conv2)
  const uint16_t* w1t;      // [c_in][c_mid] fragment order: conv1's transposed table
  const uint16_t* wdt;      // [c_in][c_out] fragment order: the projection shortcut's transposed table
  const uint8_t* m2;        // sign bits of a2 [n][hout][hout][c_mid/8]
  const uint8_t* m1;        // sign bits of a1 [n][hin][hin][c_mid/8]
  const uint8_t* m0;        // sign bits of the block input [n][hin][hin][c_in/8], or null
  uint16_t* dx;             // [n][hin][hin][c_in]
};

template <int CIN, int CM, int COUT, int HIN>
__global__ __launch_bounds__(CM * 2, CM == 128 ? 2 : 1) void k_bottleneck_s2_bwd(const RartBneckS2BwdDesc d) {
  constexpr int NW = CM / 32;
  constexpr int NPL = CM / 8;                            // planes of the d_a1 / d_a2 images
  constexpr int PL1 = 256 * 16;                          // d_a1 image plane: 4 classes x 64 slots, unpadded
  constexpr int PL2 = 64 * 16;                           // d_a2 plane: 8 x 8 grid slots
  constexpr int IMG = NPL * PL1;
  constexpr int R2 = NPL * PL2;                          // d_a2 image, later the epilogue staging
  constexpr int HOUT = HIN / 2, TPS = HOUT / S2_T;
  constexpr int SS = COUT / 256;                         // super-slices of g in stage A'
  constexpr int RT1 = CIN / 32;                          // row tiles of the W1^T / Wd^T tables
  constexpr int RD = CIN / CM;                           // rounds of CM output channels in stage C' (2)
  static_assert(2 * S2_SLICE <= IMG && NW * S2_STG <= R2 && IMG + R2 <= 160 * 1024 && CIN == 2 * CM, "LDS budget / geometry");
  __shared__ __attribute__((aligned(16))) uint8_t lds[IMG + R2];
  uint8_t* const sImg = lds;
  uint8_t* const sD2 = lds + IMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p32 = lane & 31, h = lane >> 5;
  const int lb = (int)rart_xcd_block(blockIdx.x, gridDim.x);          // the tiles of an image on one XCD (shared halo in its L2)
  const int img = lb / (TPS * TPS), tile = lb - img * (TPS * TPS);
  const int oy0 = (tile / TPS) * S2_T, ox0 = (tile % TPS) * S2_T;      // first output position the tile's centre class reads
  const int iy0 = 2 * oy0, ix0 = 2 * ox0;                               // first input position of the tile
  const long long ipos0 = (long long)img * HIN * HIN, opos0 = (long long)img * HOUT * HOUT;
  // lane geometry of a 32-slot tile t: row 4t + (p32 >> 3), column p32 & 7 (8 x 8 grid in stage A', 7 x 7 class positions later)
  const int lrow = p32 >> 3, lcol = p32 & 7;

  // ================================ stage A': d_a2 = m2 . (W3^T . g) on the 8 x 8 output grid ===================================
  {
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    constexpr int QN = 32 / NW;                          // direct-load instructions of this wave per super-slice
    const char* gsrc[QN];
    uint32_t gdst[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int jj = wave + NW * q, sub = jj >> 3, j8 = jj & 7, slot = 8 * j8 + (lane >> 3);
      const int oy = oy0 + (slot >> 3), ox = ox0 + (slot & 7);
      const int chunk = (lane & 7) ^ ((slot >> 1) & 7);
      const bool in = oy < HOUT && ox < HOUT;
      gsrc[q] = in ? reinterpret_cast<const char*>(d.g + (opos0 + oy * HOUT + ox) * COUT + sub * 64 + chunk * 8) : nullptr;
      gdst[q] = (uint32_t)__builtin_amdgcn_readfirstlane(sub * 8192 + j8 * 1024);
    }
#define RART_S2B_ISSUE(S, BUF)                                                                                  \
  _Pragma("unroll") for (int q = 0; q < QN; ++q) {                                                              \
    const char* s_ = gsrc[q] ? gsrc[q] + (S)*512 : reinterpret_cast<const char*>(g_s2_zero16);                  \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s_,                         \
                                     (__attribute__((address_space(3))) void*)(sImg + (BUF)*S2_SLICE + gdst[q]), 16, 0, 0); \
  }
    const uint16_t* wp = d.w3t + (size_t)wave * 2048 + lane * 8;             // fragment (K step st, row tile wave, ks)
    bf16x8 wq[2][4];
#define RART_S2B_LOADW(ST, SET)                                                                                 \
  _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                                 \
    wq[SET][f] = *reinterpret_cast<const bf16x8*>(wp + (size_t)((ST)*NW) * 2048 + f * 512);
    RART_S2B_ISSUE(0, 0)
    RART_S2B_LOADW(0, 0)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const uint32_t xrow = (uint32_t)(p32 * 128), xsw = (uint32_t)((p32 >> 1) & 7);
#pragma unroll
    for (int s = 0; s < SS; ++s) {
      const int buf = s & 1;
      if (s + 1 < SS) RART_S2B_ISSUE(s + 1, buf ^ 1)
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int st = 4 * s + sub;                        // K step of 64 channels of g; its fragments sit in set st & 1
        if (st + 1 < 4 * SS) {
          if (st & 1) { RART_S2B_LOADW(st + 1, 0) } else { RART_S2B_LOADW(st + 1, 1) }
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint8_t* xb = sImg + buf * S2_SLICE + sub * 8192 + xrow;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint32_t co = ((uint32_t)(2 * ks + h) ^ xsw) << 4;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = *reinterpret_cast<const bf16x8*>(xb + t * 32 * 128 + co);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[st & 1][ks], pf, acc[t], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0);        // the next super-slice has landed
      __syncthreads();
    }
#undef RART_S2B_ISSUE
#undef RART_S2B_LOADW
    // d_a2 -> its planes (slot = t * 32 + p32 = grid row * 8 + column), masked by the sign of a2; positions past the image are zero
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int oy = oy0 + 4 * t + lrow, ox = ox0 + lcol;
      const bool in = oy < HOUT && ox < HOUT;
      uint32_t mbits = 0u;
      if (in) mbits = *reinterpret_cast<const uint32_t*>(d.m2 + (opos0 + oy * HOUT + ox) * NPL + wave * 4);
      uint8_t* dst = sD2 + (wave * 4) * PL2 + (t * 32 + p32) * 16 + h * 8;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t byte = (mbits >> (8 * g)) & 0xFFu;
        const uint32_t w0 = pack_bf16x2(acc[t][4 * g], acc[t][4 * g + 1]) & halves_from_bits(byte, 2 * h);
        const uint32_t w1 = pack_bf16x2(acc[t][4 * g + 2], acc[t][4 * g + 3]) & halves_from_bits(byte, 2 * h + 1);
        *reinterpret_cast<uint2*>(dst + g * PL2) = make_uint2(w0, w1);
      }
    }
  }
  __syncthreads();

  // class-position geometry: slot (i, j) = (4t + lrow, lcol), valid for i, j < 7; padding slots read / compute a valid neighbour
  const bool pvalid = lcol < S2_T;
  const int jc = lcol < S2_T ? lcol : S2_T - 1;
  // ================================ stage B': d_a1 = m1 . (W2^T *s2 d_a2), one input-parity class at a time ======================
  {
    constexpr int KH = CM / 64;
    size_t wbase = 0;                                      // element offset of the class table inside w2t
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      const int nty = ph ? 2 : 1, ntx = pw ? 2 : 1, ntap = nty * ntx;
      f32x16 acc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      uint32_t abase[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = 4 * t + lrow, ic = i < S2_T ? i : S2_T - 1;
        abase[t] = (uint32_t)((ic * 8 + jc) * 16 + h * PL2);
      }
      const uint16_t* wp = d.w2t + wbase + (size_t)wave * 2048 + lane * 8;   // fragment (st, wave, ks) of this class's table
      bf16x8 bq[3][4];
      const int nst = ntap * KH;
#pragma unroll
      for (int s0 = 0; s0 < 2; ++s0)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bq[s0][ks] = *reinterpret_cast<const bf16x8*>(wp + (size_t)(s0 < nst ? s0 : 0) * NW * 2048 + ks * 512);
#pragma unroll
      for (int st = 0; st < nst; ++st) {
        const int tap = st / KH, kh = st - tap * KH;
        if (st + 2 < nst) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            bq[(st + 2) % 3][ks] = *reinterpret_cast<const bf16x8*>(wp + (size_t)(st + 2) * NW * 2048 + ks * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
        // taps in the engine's order: filter rows r ascending with (ph + 1 - r) even -> dy = (ph + 1 - r) / 2: ph 0: {0}; ph 1: {1, 0}
        const int ty_ = tap / ntx, tx_ = tap - ty_ * ntx;
        const int dy = ph ? 1 - ty_ : 0, dx = pw ? 1 - tx_ : 0;
        const int toff = (dy * 8 + dx) * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = *reinterpret_cast<const bf16x8*>(sD2 + (int)abase[t] + toff + (kh * 8 + ks * 2) * PL2);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[st % 3][ks], pf, acc[t], 0, 0, 0);
          }
        }
      }
      // masked by the sign of a1 at input position (iy0 + 2i + ph, ix0 + 2j + pw); class-major slot cls * 64 + t * 32 + p32
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = 4 * t + lrow;
        const bool ok = pvalid && i < S2_T;
        uint32_t mbits = 0u;
        if (ok) mbits = *reinterpret_cast<const uint32_t*>(d.m1 + (ipos0 + (iy0 + 2 * i + ph) * HIN + ix0 + 2 * lcol + pw) * NPL + wave * 4);
        uint8_t* dst = sImg + (wave * 4) * PL1 + (cls * 64 + t * 32 + p32) * 16 + h * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint32_t byte = (mbits >> (8 * g)) & 0xFFu;
          const uint32_t w0 = pack_bf16x2(acc[t][4 * g], acc[t][4 * g + 1]) & halves_from_bits(byte, 2 * h);
          const uint32_t w1 = pack_bf16x2(acc[t][4 * g + 2], acc[t][4 * g + 3]) & halves_from_bits(byte, 2 * h + 1);
          *reinterpret_cast<uint2*>(dst + g * PL1) = make_uint2(w0, w1);
        }
      }
      wbase += (size_t)ntap * CM * CM;
    }
  }
  __syncthreads();          // every wave's d_a1 planes are in the image; d_a2 is dead: its memory becomes the epilogue staging

  // ================================ stage C': dx = m0 . (W1^T . d_a1 [+ Wd^T . g for class (0, 0)]), per class ===================
  float* const sE = reinterpret_cast<float*>(sD2 + wave * S2_STG);
  const int vp = lrow * S2_T + lcol;                       // compact index of a slot inside its tile (0..27)
  const int cw = lane & 3, rw = lane >> 2;
#pragma unroll 1
  for (int cls = 0; cls < 4; ++cls) {
    const int ph = cls >> 1, pw = cls & 1;
    // hipcc hoists every lane-derived address of the unrolled K loops (hundreds of fragment pointers) out of this class loop and then
    // spills them; laundering the lane id once per iteration keeps the address arithmetic inside (the same pitfall as
    // in the persistent Bottleneck variant)
    int lane_l = lane;
    asm volatile("" : "+v"(lane_l));
    const int p32 = lane_l & 31, h = lane_l >> 5, lrow = p32 >> 3, lcol = p32 & 7;
    const int jc = lcol < S2_T ? lcol : S2_T - 1;
    const int lane = lane_l;
    f32x16 acc[RD][2];
#pragma unroll
    for (int rd = 0; rd < RD; ++rd)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rd][t][r] = 0.f;
    if (cls == 0) {
      // the projection shortcut's gradient: positions (oy0 + i, ox0 + j) of g straight from L2, K = c_out
      const uint16_t* gc[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = 4 * t + lrow, ic = i < S2_T ? i : S2_T - 1;
        gc[t] = d.g + (opos0 + (long long)(oy0 + ic) * HOUT + ox0 + jc) * COUT + h * 8;
      }
      const uint16_t* wp = d.wdt + (size_t)wave * 2048 + lane * 8;           // fragment (st, rd * NW + wave, ks) of [c_in][c_out]
      bf16x8 xq[3][2], wq[3][RD];
#define RART_S2B_LOADG(G, SET)                                                                                  \
  {                                                                                                             \
    const int st_ = (G) >> 2, ks_ = (G)&3;                                                                      \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                               \
      xq[SET][t] = *reinterpret_cast<const bf16x8*>(gc[t] + st_ * 64 + ks_ * 16);                               \
    _Pragma("unroll") for (int rd = 0; rd < RD; ++rd)                                                           \
      wq[SET][rd] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)st_ * RT1 + rd * NW) * 2048 + ks_ * 512);    \
  }
      constexpr int NG = (COUT / 64) * 4;
      RART_S2B_LOADG(0, 0)
      RART_S2B_LOADG(1, 1)
#pragma unroll
      for (int gidx = 0; gidx < NG; ++gidx) {
        if (gidx + 2 < NG) {
          if ((gidx + 2) % 3 == 0) { RART_S2B_LOADG(gidx + 2, 0) } else if ((gidx + 2) % 3 == 1) { RART_S2B_LOADG(gidx + 2, 1) } else { RART_S2B_LOADG(gidx + 2, 2) }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rd = 0; rd < RD; ++rd)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[rd][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[gidx % 3][rd], xq[gidx % 3][t], acc[rd][t], 0, 0, 0);
      }
#undef RART_S2B_LOADG
    }
    {
      // W1^T . d_a1: positions operand from the class's 64 slots of the image
      const uint16_t* wp = d.w1t + (size_t)wave * 2048 + lane * 8;           // fragment (st, rd * NW + wave, ks) of [c_in][c_mid]
      const uint32_t abase = (uint32_t)((cls * 64 + p32) * 16 + h * PL1);
      constexpr int NG = (CM / 64) * 4;
      bf16x8 wq[2][RD];
#pragma unroll
      for (int rd = 0; rd < RD; ++rd) wq[0][rd] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)rd * NW) * 2048);
#pragma unroll
      for (int gidx = 0; gidx < NG; ++gidx) {
        if (gidx + 1 < NG) {
          const int st_ = (gidx + 1) >> 2, ks_ = (gidx + 1) & 3;
#pragma unroll
          for (int rd = 0; rd < RD; ++rd)
            wq[(gidx + 1) & 1][rd] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)st_ * RT1 + rd * NW) * 2048 + ks_ * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int st = gidx >> 2, ks = gidx & 3;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 pf = *reinterpret_cast<const bf16x8*>(sImg + abase + t * 32 * 16 + (st * 8 + ks * 2) * PL1);
#pragma unroll
          for (int rd = 0; rd < RD; ++rd)
            acc[rd][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[gidx & 1][rd], pf, acc[rd][t], 0, 0, 0);
        }
      }
    }
    // epilogue: per (round, tile) through the wave's staging slice, then 64-byte row segments at the class's input positions
#pragma unroll
    for (int rd = 0; rd < RD; ++rd) {
      const int ch0 = rd * CM + wave * 32;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (pvalid) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = {acc[rd][t][4 * g], acc[rd][t][4 * g + 1], acc[rd][t][4 * g + 2], acc[rd][t][4 * g + 3]};
            *reinterpret_cast<f32x4*>(sE + vp * S2_LDE + 8 * g + 4 * h) = v;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int v = q * 16 + rw;                        // slot 0..27 of the tile: row v / 7, column v % 7
          const int vy = v / S2_T, vx = v - vy * S2_T, i = 4 * t + vy;
          if (v < 28 && i < S2_T) {
            const long long eoff = (ipos0 + (long long)(iy0 + 2 * i + ph) * HIN + ix0 + 2 * vx + pw) * CIN + ch0 + cw * 8;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sE + v * S2_LDE + cw * 8);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(sE + v * S2_LDE + cw * 8 + 4);
            const uint32_t mb = d.m0 ? (uint32_t)d.m0[eoff >> 3] : 0xFFu;
            const uint4 o = make_uint4(pack_bf16x2(v0[0], v0[1]) & halves_from_bits(mb, 0), pack_bf16x2(v0[2], v0[3]) & halves_from_bits(mb, 1),
                                       pack_bf16x2(v1[0], v1[1]) & halves_from_bits(mb, 2), pack_bf16x2(v1[2], v1[3]) & halves_from_bits(mb, 3));
            RART_LAB_STORE16(d.dx + eoff, o);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}
}  // namespace

// 1 if rart_bottleneck_s2_fwd_bf16 runs this block geometry
extern "C" int rart_bottleneck_s2_fwd_supported(int c_in, int c_mid, int c_out, int h, int w) {
  if (h != w) return 0;
  return ((c_in == 256 && c_mid == 128 && c_out == 512 && h == 56) || (c_in == 512 && c_mid == 256 && c_out == 1024 && h == 28)) ? 1 : 0;
}

extern "C" int rart_bottleneck_s2_fwd_bf16(const void* x, const void* w1, const void* w2, const void* w3, const void* wd,
                                           const float* b1, const float* b2, const float* b3, void* m1, void* m2, void* m3,
                                           void* out, int n, int h, int w, int c_in, int c_mid, int c_out, rart_stream_t stream) {
  RART_CHECK_ARG(x && w1 && w2 && w3 && wd && out && n > 0, "rart_bottleneck_s2_fwd_bf16: bad arguments");
  RART_CHECK_ARG(rart_bottleneck_s2_fwd_supported(c_in, c_mid, c_out, h, w),
                 "rart_bottleneck_s2_fwd_bf16: unsupported geometry (256 -> 128 -> 512 at 56 x 56 or 512 -> 256 -> 1024 at 28 x 28)");
  RART_CHECK_ARG(x != out, "rart_bottleneck_s2_fwd_bf16: out must not alias x");
  RART_CHECK_ARG((long long)n * h * w * c_in < (1ll << 31), "rart_bottleneck_s2_fwd_bf16: tensor must stay below 2^31 elements");
  RartBneckS2Desc d;
  d.x = (const uint16_t*)x; d.w1 = (const uint16_t*)w1; d.w2 = (const uint16_t*)w2; d.w3 = (const uint16_t*)w3; d.wd = (const uint16_t*)wd;
  d.b1 = b1; d.b2 = b2; d.b3 = b3;
  d.m1 = (uint8_t*)m1; d.m2 = (uint8_t*)m2; d.m3 = (uint8_t*)m3;
  d.out = (uint16_t*)out;
  const int tiles = (h / 2 / S2_T) * (h / 2 / S2_T);
  if (c_in == 256)
    hipLaunchKernelGGL((k_bottleneck_s2<256, 128, 512, 56>), dim3((uint32_t)n * tiles), dim3(256), 0, (hipStream_t)stream, d);
  else
    hipLaunchKernelGGL((k_bottleneck_s2<512, 256, 1024, 28>), dim3((uint32_t)n * tiles), dim3(512), 0, (hipStream_t)stream, d);
  RART_CHECK_LAUNCH("rart_bottleneck_s2_fwd_bf16");
  return RART_OK;
}

extern "C" int rart_bottleneck_s2_bwd_bf16(const void* g, const void* w3t, const void* w2t, const void* w1t, const void* wdt,
                                           const void* m2, const void* m1, const void* m0, void* dx, int n, int h, int w, int c_in,
                                           int c_mid, int c_out, rart_stream_t stream) {
  RART_CHECK_ARG(g && w3t && w2t && w1t && wdt && m2 && m1 && dx && n > 0, "rart_bottleneck_s2_bwd_bf16: bad arguments");
  RART_CHECK_ARG(rart_bottleneck_s2_fwd_supported(c_in, c_mid, c_out, h, w),
                 "rart_bottleneck_s2_bwd_bf16: unsupported geometry (256 -> 128 -> 512 at 56 x 56 or 512 -> 256 -> 1024 at 28 x 28)");
  RART_CHECK_ARG(g != dx, "rart_bottleneck_s2_bwd_bf16: dx must not alias g");
  RART_CHECK_ARG((long long)n * h * w * c_in < (1ll << 31), "rart_bottleneck_s2_bwd_bf16: tensor must stay below 2^31 elements");
  RartBneckS2BwdDesc d;
  d.g = (const uint16_t*)g; d.w3t = (const uint16_t*)w3t; d.w2t = (const uint16_t*)w2t; d.w1t = (const uint16_t*)w1t;
  d.wdt = (const uint16_t*)wdt; d.m2 = (const uint8_t*)m2; d.m1 = (const uint8_t*)m1; d.m0 = (const uint8_t*)m0;
  d.dx = (uint16_t*)dx;
  const int tiles = (h / 2 / S2_T) * (h / 2 / S2_T);
  if (c_in == 256)
    hipLaunchKernelGGL((k_bottleneck_s2_bwd<256, 128, 512, 56>), dim3((uint32_t)n * tiles), dim3(256), 0, (hipStream_t)stream, d);
  else
    hipLaunchKernelGGL((k_bottleneck_s2_bwd<512, 256, 1024, 28>), dim3((uint32_t)n * tiles), dim3(512), 0, (hipStream_t)stream, d);
  RART_CHECK_LAUNCH("rart_bottleneck_s2_bwd_bf16");
  return RART_OK;
}
