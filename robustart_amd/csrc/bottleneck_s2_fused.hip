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
  const int img = blockIdx.x / (TPS * TPS), tile = blockIdx.x - img * (TPS * TPS);
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
          *reinterpret_cast<uint4*>(d.out + eoff) = o;
          if (d.m3) d.m3[eoff >> 3] = (uint8_t)sign_byte(o);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
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
