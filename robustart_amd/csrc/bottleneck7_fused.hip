// One identity Bottleneck of ResNet-50's layer4 (2048 -> 512 -> 512 -> 2048 channels at 7 x 7) as ONE kernel (gfx950), forward and
// backward-to-input, one image per workgroup.  Same three-stage structure as bottleneck14_fused.hip (read that file first); what
// changes with the geometry:
//   * 49 positions = two position tiles of 32 slots (4 image rows x 8 slots, 7 valid per row; the second tile has 3 rows), so a weight
//     fragment serves only two MFMAs: a wave owns TWO 32-channel blocks (64 channels) x both tiles in every stage, and the launch is
//     bound by the weight stream from L2 (8.7 MB per image, shared by the 32 workgroups of an XCD that walk it in step);
//   * the 512-channel image is 64 planes x 97 slots x 16 B around a 9 x 9 ring grid (row stride 9);
//   * x streams through LDS in eight 256-channel slices (64 slots x 512 B, chunk s of slot r at position s ^ (r & 31)).
// As three launches the block took 47 + 83 + 60 us (profiles/r02_igemm_per_shape.txt), tile-quantised (98-392 tiles of 128 x 128).
//
// Reference step: Bottleneck.forward of the public ResNet-50 (RobustART/model/__init__.py:1 -> absent submodule;
// robustart_amd/model/resnet_torch.py) and its autograd inside every attack iteration
// (RobustART/noise/utils/adv/attack.py:21-22, Attacks/autoattack/autopgd_base.py:271-289).
#include "rart_common.h"
#include "rart_bf16_helpers.h"

struct RartBneck7Desc {
  const uint16_t* x;        // [n][7][7][2048] bf16: block input (forward) / masked gradient at the block output (backward)
  const uint16_t* w1;       // [512][2048] in fragment order (rart_pack_frag_bf16(rows 512, k 2048))
  const uint16_t* w2;       // [512][9*512] in fragment order (rart_pack_frag_bf16(rows 512, k 4608))
  const uint16_t* w3;       // [2048][512] in fragment order (rart_pack_frag_bf16(rows 2048, k 512))
  const float* b1;
  const float* b2;
  const float* b3;          // fp32 biases or null
  uint8_t* m1;              // 1 bit per element of the stage-A result ([P][64] bytes): forward = sign out (or null), backward = mask in
  uint8_t* m2;              // same for the stage-B result
  uint8_t* m3;              // [P][256] bytes for the output
  uint16_t* out;
  int tap_off[9];           // (dy * 9 + dx) * 16: byte offset of a tap inside an image plane
};

namespace {
using namespace rart_bf16;
__device__ __attribute__((aligned(16))) const uint32_t g_b7_zero16[4] = {0u, 0u, 0u, 0u};   // source of the padding slots of an x slice

constexpr int B7_HW = 7, B7_NP = 49;
constexpr int B7_CM = 512, B7_CIO = 2048;
constexpr int B7_PLANE = 97 * 16;                       // 9 x 9 ring grid (81 slots) padded to 97: 16 mod 256 bytes
static_assert(B7_PLANE % 256 == 16, "plane stride must be 16 mod 256");
constexpr int B7_IMG = 64 * B7_PLANE;                   // 99 328 B: the 512-channel image, chunk-major
constexpr int B7_SLICE = 64 * 512;                      // one 256-channel slice of x: 64 slots x 512 B
static_assert(2 * B7_SLICE <= B7_IMG, "two x slices must fit the image memory");
constexpr int B7_LDE = 36;
constexpr int B7_STG = 28 * B7_LDE * 4;                 // 4 032 B per wave

template <bool BWD>
__global__ __launch_bounds__(512, 1) void k_bottleneck7(const RartBneck7Desc d) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[B7_IMG + 8 * B7_STG];
  uint8_t* const sImg = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p32 = lane & 31, h = lane >> 5;
  const long long pos0 = (long long)blockIdx.x * B7_NP;
  // lane geometry of position tile t (0, 1): image row 4t + (p32 >> 3), column p32 & 7 (column 7 and row 7 are padding slots)
  const int px = p32 & 7, prow = p32 >> 3;
  f32x16 acc[2][2];                                        // [tile][block]: channels (2 wave + block) * 32 + 8g + 4h + i

  // ================================ stage A: a1[pos][512] = x[pos][2048] . W1^T =================================================
  {
    // a wave-wide direct load covers 2 slots x 512 B: lane l -> slot 2 j + (l >> 5), chunk position l & 31, fetching chunk
    // (l & 31) ^ (slot & 31); instruction j of a slice = wave + 8 q, q = 0..3
    const char* xsrc[4];
    uint32_t xdst[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = wave + 8 * q, slot = 2 * j + (lane >> 5);
      const int sy = 4 * (slot >> 5) + ((slot >> 3) & 3), sx = slot & 7;
      const int chunk = (lane & 31) ^ (slot & 31);
      xsrc[q] = (sx < B7_HW && sy < B7_HW) ? reinterpret_cast<const char*>(d.x + (pos0 + sy * B7_HW + sx) * B7_CIO + chunk * 8) : nullptr;
      xdst[q] = (uint32_t)__builtin_amdgcn_readfirstlane(2 * j) * 512u;
    }
#define RART_B7_ISSUE(S, BUF)                                                                                   \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                               \
    const char* s_ = xsrc[q] ? xsrc[q] + (S)*512 : reinterpret_cast<const char*>(g_b7_zero16);                  \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s_,                         \
                                     (__attribute__((address_space(3))) void*)(sImg + (BUF)*B7_SLICE + xdst[q]), 16, 0, 0); \
  }
    // weights: fragment (K step st of 64, row tile wn, ks) of the [512][2048] table = (st * 16 + wn) * 4 + ks; this wave: wn = 2 wave + b
    const uint16_t* wp = d.w1 + (size_t)(2 * wave) * 2048 + lane * 8;
    bf16x8 wq[2][2][4];
#define RART_B7_LOADW(ST, SET)                                                                                  \
  _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                            \
      wq[SET][b][ks] = *reinterpret_cast<const bf16x8*>(wp + (size_t)((ST)*16 + b) * 2048 + ks * 512);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (!BWD && d.b1) bv = *reinterpret_cast<const f32x4*>(d.b1 + (2 * wave + b) * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t][b][4 * g + i] = bv[i];
      }
    RART_B7_ISSUE(0, 0)
    RART_B7_LOADW(0, 0)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const uint32_t xrow = (uint32_t)(p32 * 512), xsw = (uint32_t)p32;
#pragma unroll 1
    for (int s = 0; s < 8; ++s) {                         // slices of 256 channels = 4 K steps of 64
      const int buf = s & 1;
      if (s + 1 < 8) RART_B7_ISSUE(s + 1, buf ^ 1)
      const uint8_t* xb = sImg + buf * B7_SLICE + xrow;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int st = s * 4 + k4;
        if (st + 1 < 32) {
          if (k4 & 1) { RART_B7_LOADW(st + 1, 0) } else { RART_B7_LOADW(st + 1, 1) }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint32_t co = ((uint32_t)(k4 * 8 + 2 * ks + h) ^ xsw) << 4;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = *reinterpret_cast<const bf16x8*>(xb + t * 32 * 512 + co);
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[k4 & 1][b][ks], pf, acc[t][b], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
    }
#undef RART_B7_ISSUE
#undef RART_B7_LOADW
  }
  // the memory becomes the zero-ringed image: every non-interior slot of every plane is zeroed, the interior comes from the accumulators
  for (int i = tid; i < 64 * 97; i += 512) {
    const int plane = i / 97, slot = i - plane * 97, r = slot / 9, c = slot - r * 9;
    if (!(slot < 81 && r >= 1 && r <= 7 && c >= 1 && c <= 7)) *reinterpret_cast<uint4*>(sImg + plane * B7_PLANE + slot * 16) = make_uint4(0, 0, 0, 0);
  }
#define RART_B7_STORE_IMG(MASKPTR)                                                                              \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                               \
    const int y_ = 4 * t + prow;                                                                                \
    const bool ok_ = px < B7_HW && y_ < B7_HW;                                                                  \
    _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                             \
      uint32_t mbits_ = 0xFFFFFFFFu;                                                                            \
      if (BWD && ok_) mbits_ = *reinterpret_cast<const uint32_t*>((MASKPTR) + (pos0 + y_ * B7_HW + px) * 64 + (2 * wave + b) * 4); \
      uint8_t* dst_ = sImg + ((y_ + 1) * 9 + px + 1) * 16 + ((2 * wave + b) * 4) * B7_PLANE + h * 8;            \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                           \
        uint32_t w0 = pack_bf16x2(acc[t][b][4 * g], acc[t][b][4 * g + 1]), w1 = pack_bf16x2(acc[t][b][4 * g + 2], acc[t][b][4 * g + 3]); \
        if (BWD) {                                                                                              \
          const uint32_t byte = (mbits_ >> (8 * g)) & 0xFFu;                                                    \
          w0 &= halves_from_bits(byte, 2 * h);                                                                  \
          w1 &= halves_from_bits(byte, 2 * h + 1);                                                              \
        } else {                                                                                                \
          w0 = relu_bf16x2(w0);                                                                                 \
          w1 = relu_bf16x2(w1);                                                                                 \
        }                                                                                                       \
        if (ok_) *reinterpret_cast<uint2*>(dst_ + g * B7_PLANE) = make_uint2(w0, w1);                           \
      }                                                                                                         \
    }                                                                                                           \
  }
  RART_B7_STORE_IMG(d.m1)
  __syncthreads();
#define RART_B7_SIGN_IMG(PTR)                                                                                   \
  if (!BWD && (PTR)) {                                                                                          \
    for (int i = tid; i < B7_NP * 64; i += 512) {                                                               \
      const int p = i >> 6, chunk = i & 63, y = p / B7_HW, xx = p - y * B7_HW;                                  \
      const uint4 v = *reinterpret_cast<const uint4*>(sImg + chunk * B7_PLANE + ((y + 1) * 9 + xx + 1) * 16);   \
      (PTR)[pos0 * 64 + i] = (uint8_t)sign_byte(v);                                                             \
    }                                                                                                           \
  }
  RART_B7_SIGN_IMG(d.m1)

  // ================================ stage B: a2 = 3x3 over the a1 image: 72 K steps ==============================================
  uint32_t abase[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int y = 4 * t + prow, yc = y < B7_HW ? y : B7_HW - 1, xc = px < B7_HW ? px : B7_HW - 1;   // padding slots read a valid position
    abase[t] = (uint32_t)(((yc + 1) * 9 + xc + 1) * 16 + h * B7_PLANE);
  }
  {
    const uint16_t* wp = d.w2 + (size_t)(2 * wave) * 2048 + lane * 8;      // fragment (st, wn, ks): (st * 16 + wn) * 4 + ks
    bf16x8 bq[2][2][4];
#define RART_B7_LOADW2(ST, SET)                                                                                 \
  _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                            \
      bq[SET][b][ks] = *reinterpret_cast<const bf16x8*>(wp + (size_t)((ST)*16 + b) * 2048 + ks * 512);
    RART_B7_LOADW2(0, 0)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (!BWD && d.b2) bv = *reinterpret_cast<const f32x4*>(d.b2 + (2 * wave + b) * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t][b][4 * g + i] = bv[i];
      }
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = d.tap_off[tap];
#pragma unroll
      for (int kh = 0; kh < 8; ++kh) {
        const int st = tap * 8 + kh;
        if (st + 1 < 72) {
          if (kh & 1) { RART_B7_LOADW2(st + 1, 0) } else { RART_B7_LOADW2(st + 1, 1) }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = *reinterpret_cast<const bf16x8*>(sImg + (int)abase[t] + toff + (kh * 8 + ks * 2) * B7_PLANE);
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[kh & 1][b][ks], pf, acc[t][b], 0, 0, 0);
          }
        }
      }
    }
#undef RART_B7_LOADW2
  }
  __syncthreads();
  RART_B7_STORE_IMG(d.m2)
  __syncthreads();
  RART_B7_SIGN_IMG(d.m2)
#undef RART_B7_STORE_IMG
#undef RART_B7_SIGN_IMG

  // ================================ stage C: out[pos][2048] = a2[pos][512] . W3^T + x, 512 channels per round ===================
  float* const sE = reinterpret_cast<float*>(lds + B7_IMG + wave * B7_STG);
  const int vp = prow * B7_HW + px;
#pragma unroll 1
  for (int rd = 0; rd < 4; ++rd) {
    const uint16_t* wp = d.w3 + (size_t)(rd * 16 + 2 * wave) * 2048 + lane * 8;   // fragment (st, wn, ks): (st * 64 + wn) * 4 + ks
    bf16x8 bq[2][2][4];
#define RART_B7_LOADW3(ST, SET)                                                                                 \
  _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                            \
      bq[SET][b][ks] = *reinterpret_cast<const bf16x8*>(wp + (size_t)((ST)*64 + b) * 2048 + ks * 512);
    RART_B7_LOADW3(0, 0)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (!BWD && d.b3) bv = *reinterpret_cast<const f32x4*>(d.b3 + rd * 512 + (2 * wave + b) * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[t][b][4 * g + i] = bv[i];
      }
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      if (st + 1 < 8) {
        if (st & 1) { RART_B7_LOADW3(st + 1, 0) } else { RART_B7_LOADW3(st + 1, 1) }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 pf = *reinterpret_cast<const bf16x8*>(sImg + (int)abase[t] + (st * 8 + ks * 2) * B7_PLANE);
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[st & 1][b][ks], pf, acc[t][b], 0, 0, 0);
        }
      }
    }
#undef RART_B7_LOADW3
    // epilogue: per (tile, block) the valid positions (28 / 21) x 32 channels go through the wave's staging slice
    const int cw = lane & 3, rw = lane >> 2;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ch0 = rd * 512 + (2 * wave + b) * 32;
        const int nvalid = t == 0 ? 28 : 21;
        u32x4 rv[2];
        uint32_t mb[2];
        long long eoff[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int v = q * 16 + rw;
          const int vy = v / B7_HW, vx = v - vy * B7_HW;
          eoff[q] = v < nvalid ? (pos0 + (4 * t + vy) * B7_HW + vx) * B7_CIO + ch0 + cw * 8 : -1;
          rv[q] = (u32x4){0u, 0u, 0u, 0u};
          mb[q] = 0xFFu;
          if (eoff[q] >= 0) {
            rv[q] = *reinterpret_cast<const u32x4*>(d.x + eoff[q]);
            if (BWD && d.m3) mb[q] = d.m3[eoff[q] >> 3];
          }
        }
        if (px < B7_HW && 4 * t + prow < B7_HW) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = {acc[t][b][4 * g], acc[t][b][4 * g + 1], acc[t][b][4 * g + 2], acc[t][b][4 * g + 3]};
            *reinterpret_cast<f32x4*>(sE + vp * B7_LDE + 8 * g + 4 * h) = v;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (eoff[q] >= 0) {
            const int v = q * 16 + rw;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sE + v * B7_LDE + cw * 8);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(sE + v * B7_LDE + cw * 8 + 4);
            float vv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              vv[2 * j] += __uint_as_float(rv[q][j] << 16);
              vv[2 * j + 1] += __uint_as_float(rv[q][j] & 0xFFFF0000u);
              o[j] = pack_bf16x2(vv[2 * j], vv[2 * j + 1]);
              if (BWD) o[j] &= halves_from_bits(mb[q], j);
              else o[j] = relu_bf16x2(o[j]);
            }
            RART_LAB_STORE16(d.out + eoff[q], make_uint4(o[0], o[1], o[2], o[3]));
            if (!BWD && d.m3) d.m3[eoff[q] >> 3] = (uint8_t)sign_byte(make_uint4(o[0], o[1], o[2], o[3]));
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
  }
}
}  // namespace

// 1 if rart_bottleneck7_fused_bf16 runs this block geometry
extern "C" int rart_bottleneck7_fused_supported(int c_io, int c_mid, int h, int w) {
  return (c_io == B7_CIO && c_mid == B7_CM && h == B7_HW && w == B7_HW) ? 1 : 0;
}

extern "C" int rart_bottleneck7_fused_bf16(const void* x, const void* w1, const void* w2, const void* w3, const float* b1,
                                           const float* b2, const float* b3, void* m1, void* m2, void* m3, void* out, int n,
                                           int h, int w, int c_io, int c_mid, const int* tap_dy, const int* tap_dx, int backward,
                                           rart_stream_t stream) {
  RART_CHECK_ARG(x && w1 && w2 && w3 && out && tap_dy && tap_dx && n > 0, "rart_bottleneck7_fused_bf16: bad arguments");
  RART_CHECK_ARG(rart_bottleneck7_fused_supported(c_io, c_mid, h, w),
                 "rart_bottleneck7_fused_bf16: unsupported geometry (2048 -> 512 -> 2048 channels at 7 x 7 only)");
  RART_CHECK_ARG(x != out, "rart_bottleneck7_fused_bf16: out must not alias x");
  RART_CHECK_ARG(!backward || (m1 && m2), "rart_bottleneck7_fused_bf16: the backward pass needs both inner masks");
  RART_CHECK_ARG((long long)n * B7_NP * B7_CIO < (1ll << 31), "rart_bottleneck7_fused_bf16: tensor must stay below 2^31 elements");
  RartBneck7Desc d;
  d.x = (const uint16_t*)x; d.w1 = (const uint16_t*)w1; d.w2 = (const uint16_t*)w2; d.w3 = (const uint16_t*)w3;
  d.b1 = b1; d.b2 = b2; d.b3 = b3;
  d.m1 = (uint8_t*)m1; d.m2 = (uint8_t*)m2; d.m3 = (uint8_t*)m3;
  d.out = (uint16_t*)out;
  for (int t = 0; t < 9; ++t) {
    RART_CHECK_ARG(tap_dy[t] >= -1 && tap_dy[t] <= 1 && tap_dx[t] >= -1 && tap_dx[t] <= 1,
                   "rart_bottleneck7_fused_bf16: taps must lie in -1..1");
    d.tap_off[t] = (tap_dy[t] * 9 + tap_dx[t]) * 16;
  }
  if (backward) hipLaunchKernelGGL(k_bottleneck7<true>, dim3((uint32_t)n), dim3(512), 0, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(k_bottleneck7<false>, dim3((uint32_t)n), dim3(512), 0, (hipStream_t)stream, d);
  RART_CHECK_LAUNCH("rart_bottleneck7_fused_bf16");
  return RART_OK;
}
