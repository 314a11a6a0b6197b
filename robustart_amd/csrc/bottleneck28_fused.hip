// One identity Bottleneck of ResNet-50's layer2 (512 -> 128 -> 128 -> 512 channels at 28 x 28) as ONE kernel (gfx950), forward and
// backward-to-input.  Same three-stage structure as bottleneck14_fused.hip (read that file first), re-sized so that TWO workgroups
// fit a CU and their memory and matrix phases overlap:
//
//   * a workgroup (4 waves) owns a 14 x 14 QUARTER of an image; its a1 halo is the 16 x 16 grid around it, so stage A computes
//     a1 = relu(W1 . x + b1) on all 256 grid positions (8 position tiles, 31 % recompute; grid positions outside the image are the
//     3x3's zero padding and are written as zeros), x streaming through LDS in eight 64-channel slices (256 slots x 128 B,
//     global_load_lds_dwordx4, chunk s of slot r at position s ^ ((r >> 1) & 7)) inside the memory of the later image;
//   * the image is 16 planes x 257 slots x 16 B (chunk-major, as k_conv3x3_image256 / k_bottleneck14); stage B walks the 18 K steps of
//     the 3x3 over the 7 centre tiles, stage C (4 rounds of 128 output channels) adds the residual and writes 64-byte row segments;
//   * a wave owns 32 output channels x all position tiles in every stage; all three weight tables in fragment order from L2.
//
// LDS: 65 792 B image + 4 x 4 032 B staging = 80 KiB exactly -> 2 workgroups (8 waves) per CU.
//
// Reference step: Bottleneck.forward of the public ResNet-50 (RobustART/model/__init__.py:1 -> absent submodule;
// robustart_amd/model/resnet_torch.py) and its autograd inside every attack iteration
// (RobustART/noise/utils/adv/attack.py:21-22, Attacks/autoattack/autopgd_base.py:271-289).
#include "rart_common.h"
#include "rart_bf16_helpers.h"

struct RartBneck28Desc {
  const uint16_t* x;        // [n][28][28][512] bf16: block input (forward) / masked gradient at the block output (backward)
  const uint16_t* w1;       // [128][512] in fragment order (rart_pack_frag_bf16(rows 128, k 512))
  const uint16_t* w2;       // [128][9*128] in fragment order (rart_conv3x3_pack_frag_bf16)
  const uint16_t* w3;       // [512][128] in fragment order (rart_pack_frag_bf16(rows 512, k 128))
  const float* b1;
  const float* b2;
  const float* b3;          // fp32 biases or null
  uint8_t* m1;              // 1 bit per element of the stage-A result ([P][16] bytes): forward = sign out (or null), backward = mask in
  uint8_t* m2;              // same for the stage-B result
  uint8_t* m3;              // [P][64] bytes for the output
  uint16_t* out;
  int tap_off[9];           // (dy * 16 + dx) * 16: byte offset of a tap inside an image plane
};

namespace {
using namespace rart_bf16;
__device__ __attribute__((aligned(16))) const uint32_t g_b28_zero16[4] = {0u, 0u, 0u, 0u};   // source of grid slots outside the image

constexpr int B28_HW = 28, B28_T = 14;                   // image side, tile side
constexpr int B28_CM = 128, B28_CIO = 512;
constexpr int B28_PLANE = (256 + 1) * 16;                // 16 x 16 grid positions + 1: 16 mod 256 bytes
constexpr int B28_IMG = 16 * B28_PLANE;                  // 65 792 B: the 128-channel halo image, chunk-major
constexpr int B28_SLICE = 256 * 128;                     // one 64-channel slice of x: 256 slots x 128 B
static_assert(2 * B28_SLICE <= B28_IMG, "two x slices must fit the image memory");
constexpr int B28_LDE = 36;                              // staging row: 32 floats + 4
constexpr int B28_STG = 28 * B28_LDE * 4;                // 4 032 B per wave
static_assert(B28_IMG + 4 * B28_STG == 80 * 1024, "LDS budget: two workgroups per CU");

template <bool BWD>
__global__ __launch_bounds__(256, 2) void k_bottleneck28(const RartBneck28Desc d) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[B28_IMG + 4 * B28_STG];
  uint8_t* const sImg = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p32 = lane & 31, h = lane >> 5;
  const int px = p32 & 15, prow = p32 >> 4;
  // workgroup -> (image, tile row, tile column); the four tiles of an image are consecutive blocks
  // (remapping them onto one XCD with rart_xcd_block measured 0.7 % SLOWER: the shared halo already comes from the Infinity Cache)
  const int img = blockIdx.x >> 2, y0 = ((blockIdx.x >> 1) & 1) * B28_T, x0 = (blockIdx.x & 1) * B28_T;
  const long long ipos0 = (long long)img * B28_HW * B28_HW;          // raster index of the image's first position

  // ================================ stage A: a1 = x . W1^T on the 16 x 16 grid around the tile (8 position tiles) ================
  {
    f32x16 acc[8];
    // a wave-wide direct load covers 8 slots x 128 B: lane l -> slot 8 j + (l >> 3), chunk position l & 7, fetching chunk
    // (l & 7) ^ ((slot >> 1) & 7); instruction j of a slice = wave + 4 q, q = 0..7
    const char* xsrc[8];
    uint32_t xdst[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = wave + 4 * q, slot = 8 * j + (lane >> 3);            // slot = grid row * 16 + grid column
      const int gy = y0 - 1 + (slot >> 4), gx = x0 - 1 + (slot & 15);
      const int chunk = (lane & 7) ^ ((slot >> 1) & 7);
      const bool in = (unsigned)gy < (unsigned)B28_HW && (unsigned)gx < (unsigned)B28_HW;
      xsrc[q] = in ? reinterpret_cast<const char*>(d.x + (ipos0 + gy * B28_HW + gx) * B28_CIO + chunk * 8) : nullptr;
      xdst[q] = (uint32_t)__builtin_amdgcn_readfirstlane(8 * j) * 128u;
    }
#define RART_B28_ISSUE(S, BUF)                                                                                  \
  _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                               \
    const char* s_ = xsrc[q] ? xsrc[q] + (S)*128 : reinterpret_cast<const char*>(g_b28_zero16);                 \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s_,                         \
                                     (__attribute__((address_space(3))) void*)(sImg + (BUF)*B28_SLICE + xdst[q]), 16, 0, 0); \
  }
    // weights of this wave: output channels 32 wave .. +31; fragment (K step st of 64, row tile wave, ks) of the [128][512] table
    const uint16_t* wp = d.w1 + (size_t)wave * 2048 + lane * 8;
    bf16x8 wq[2][4];
#define RART_B28_LOADW(S, SET)                                                                                  \
  _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                                 \
    wq[SET][f] = *reinterpret_cast<const bf16x8*>(wp + (size_t)((S)*4) * 2048 + f * 512);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (!BWD && d.b1) bv = *reinterpret_cast<const f32x4*>(d.b1 + wave * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = bv[i];
    }
    RART_B28_ISSUE(0, 0)
    RART_B28_LOADW(0, 0)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    // fragment read of tile t: slot r = t * 32 + p32, logical chunk 2 ks + h at position (2 ks + h) ^ ((r >> 1) & 7)
    const uint32_t xrow = (uint32_t)(p32 * 128), xsw = (uint32_t)((p32 >> 1) & 7);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int buf = s & 1;
      if (s + 1 < 8) {
        RART_B28_ISSUE(s + 1, buf ^ 1)
        if (buf) { RART_B28_LOADW(s + 1, 0) } else { RART_B28_LOADW(s + 1, 1) }
      }
      __builtin_amdgcn_sched_barrier(0);
      const uint8_t* xb = sImg + buf * B28_SLICE + xrow;
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
#undef RART_B28_ISSUE
#undef RART_B28_LOADW
    // every wave is past its last slice read: the memory becomes the a1 halo image.  lane: grid position (2t + prow, px) of tile
    // t, channels wave*32 + 8g + 4h + (0..3) -> 8 bytes of chunk wave*4 + g; positions outside the image are zero
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int ry = 2 * t + prow, gy = y0 - 1 + ry, gx = x0 - 1 + px;
      const bool in = (unsigned)gy < (unsigned)B28_HW && (unsigned)gx < (unsigned)B28_HW;
      uint32_t mbits = 0xFFFFFFFFu;
      if (BWD && in) mbits = *reinterpret_cast<const uint32_t*>(d.m1 + (ipos0 + gy * B28_HW + gx) * 16 + wave * 4);
      uint8_t* dst = sImg + (ry * 16 + px) * 16 + (wave * 4) * B28_PLANE + h * 8;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint32_t w0 = pack_bf16x2(acc[t][4 * g], acc[t][4 * g + 1]), w1 = pack_bf16x2(acc[t][4 * g + 2], acc[t][4 * g + 3]);
        if (BWD) {
          const uint32_t byte = (mbits >> (8 * g)) & 0xFFu;
          w0 &= halves_from_bits(byte, 2 * h);
          w1 &= halves_from_bits(byte, 2 * h + 1);
        } else {
          w0 = relu_bf16x2(w0);
          w1 = relu_bf16x2(w1);
        }
        if (!in) w0 = w1 = 0u;
        *reinterpret_cast<uint2*>(dst + g * B28_PLANE) = make_uint2(w0, w1);
      }
    }
  }
  __syncthreads();

  // lane geometry of a CENTRE tile t (stages B and C): tile row 2t + prow, tile column px (14, 15 are padding slots)
  const bool pvalid = px < B28_T;
  // sign bits of the centre of the image (forward, when the backward pass will follow): byte pos * 16 + chunk
#define RART_B28_SIGN_IMG(PTR)                                                                                  \
  if (!BWD && (PTR)) {                                                                                          \
    for (int i = tid; i < B28_T * B28_T * 16; i += 256) {                                                       \
      const int p = i >> 4, chunk = i & 15, yy = p / B28_T, xx = p - yy * B28_T;                                \
      const uint4 v = *reinterpret_cast<const uint4*>(sImg + chunk * B28_PLANE + ((yy + 1) * 16 + xx + 1) * 16); \
      (PTR)[(ipos0 + (y0 + yy) * B28_HW + x0 + xx) * 16 + chunk] = (uint8_t)sign_byte(v);                       \
    }                                                                                                           \
  }
  RART_B28_SIGN_IMG(d.m1)

  // ================================ stage B: a2 = 3x3 over the a1 image, 7 centre tiles x 32 channels per wave ==================
  f32x16 acc[7];
  uint32_t abase[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int yy = 2 * t + prow, xc = px < B28_T ? px : B28_T - 1;       // padding slots read a valid position; never stored
    abase[t] = (uint32_t)(((yy + 1) * 16 + xc + 1) * 16 + h * B28_PLANE);
  }
  {
    const uint16_t* wp = d.w2 + wave * 2048 + lane * 8;                    // fragment (st, wave, ks): (st * 4 + wave) * 4 + ks
    bf16x8 bq[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bq[0][ks] = *reinterpret_cast<const bf16x8*>(wp + ks * 512);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (!BWD && d.b2) bv = *reinterpret_cast<const f32x4*>(d.b2 + wave * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = bv[i];
    }
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      const int tap = st >> 1, kh = st & 1;
      if (st + 1 < 18) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bq[(st + 1) & 1][ks] = *reinterpret_cast<const bf16x8*>(wp + (size_t)(st + 1) * 4 * 2048 + ks * 512);
      }
      __builtin_amdgcn_sched_barrier(0);                    // keep the prefetch ahead of this step's MFMAs
      const int toff = d.tap_off[tap];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
          const bf16x8 pf = *reinterpret_cast<const bf16x8*>(sImg + (int)abase[t] + toff + (kh * 8 + ks * 2) * B28_PLANE);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[st & 1][ks], pf, acc[t], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();          // every wave is done reading a1: the centre of the image is overwritten by a2
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int yy = 2 * t + prow;
    uint32_t mbits = 0xFFFFFFFFu;
    if (BWD && pvalid) mbits = *reinterpret_cast<const uint32_t*>(d.m2 + (ipos0 + (y0 + yy) * B28_HW + x0 + px) * 16 + wave * 4);
    uint8_t* dst = sImg + ((yy + 1) * 16 + px + 1) * 16 + (wave * 4) * B28_PLANE + h * 8;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint32_t w0 = pack_bf16x2(acc[t][4 * g], acc[t][4 * g + 1]), w1 = pack_bf16x2(acc[t][4 * g + 2], acc[t][4 * g + 3]);
      if (BWD) {
        const uint32_t byte = (mbits >> (8 * g)) & 0xFFu;
        w0 &= halves_from_bits(byte, 2 * h);
        w1 &= halves_from_bits(byte, 2 * h + 1);
      } else {
        w0 = relu_bf16x2(w0);
        w1 = relu_bf16x2(w1);
      }
      if (pvalid) *reinterpret_cast<uint2*>(dst + g * B28_PLANE) = make_uint2(w0, w1);
    }
  }
  __syncthreads();
  RART_B28_SIGN_IMG(d.m2)
#undef RART_B28_SIGN_IMG

  // ================================ stage C: out[pos][512] = a2[pos][128] . W3^T + x, 128 channels per round ====================
  float* const sE = reinterpret_cast<float*>(lds + B28_IMG + wave * B28_STG);
  const int vp = prow * B28_T + px;                         // compact index of a valid slot inside its tile (0..27)
#pragma unroll 1
  for (int rd = 0; rd < 4; ++rd) {
    const int ch0 = rd * 128 + wave * 32;                   // first output channel of this wave in this round
    const uint16_t* wp = d.w3 + (size_t)(rd * 4 + wave) * 2048 + lane * 8;     // fragment (st, row tile rd*4 + wave, ks), 16 row tiles
    bf16x8 bq[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bq[0][ks] = *reinterpret_cast<const bf16x8*>(wp + ks * 512);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (!BWD && d.b3) bv = *reinterpret_cast<const f32x4*>(d.b3 + ch0 + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = bv[i];
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      if (st + 1 < 2) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bq[(st + 1) & 1][ks] = *reinterpret_cast<const bf16x8*>(wp + (size_t)(st + 1) * 16 * 2048 + ks * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
          const bf16x8 pf = *reinterpret_cast<const bf16x8*>(sImg + (int)abase[t] + (st * 8 + ks * 2) * B28_PLANE);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[st & 1][ks], pf, acc[t], 0, 0, 0);
        }
      }
    }
    // epilogue: per tile the 28 valid positions x 32 channels go through the wave's staging slice; then 4 lanes per position read
    // 8 channels each: 64-byte row segments, two passes (16 + 12 positions)
    const int cw = lane & 3, rw = lane >> 2;
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      u32x4 rv[2];
      uint32_t mb[2];
      long long eoff[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int v = q * 16 + rw;                          // valid-slot index 0..27
        const int vy = v >= B28_T ? 1 : 0, vx = v - vy * B28_T;
        eoff[q] = v < 28 ? (ipos0 + (y0 + 2 * t + vy) * B28_HW + x0 + vx) * B28_CIO + ch0 + cw * 8 : -1;
        rv[q] = (u32x4){0u, 0u, 0u, 0u};
        mb[q] = 0xFFu;
        if (eoff[q] >= 0) {
          rv[q] = *reinterpret_cast<const u32x4*>(d.x + eoff[q]);
          if (BWD && d.m3) mb[q] = d.m3[eoff[q] >> 3];
        }
      }
      if (pvalid) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
          *reinterpret_cast<f32x4*>(sE + vp * B28_LDE + 8 * g + 4 * h) = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (eoff[q] >= 0) {
          const int v = q * 16 + rw;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sE + v * B28_LDE + cw * 8);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(sE + v * B28_LDE + cw * 8 + 4);
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

// 1 if rart_bottleneck28_fused_bf16 runs this block geometry
extern "C" int rart_bottleneck28_fused_supported(int c_io, int c_mid, int h, int w) {
  return (c_io == B28_CIO && c_mid == B28_CM && h == B28_HW && w == B28_HW) ? 1 : 0;
}

extern "C" int rart_bottleneck28_fused_bf16(const void* x, const void* w1, const void* w2, const void* w3, const float* b1,
                                            const float* b2, const float* b3, void* m1, void* m2, void* m3, void* out, int n,
                                            int h, int w, int c_io, int c_mid, const int* tap_dy, const int* tap_dx, int backward,
                                            rart_stream_t stream) {
  RART_CHECK_ARG(x && w1 && w2 && w3 && out && tap_dy && tap_dx && n > 0, "rart_bottleneck28_fused_bf16: bad arguments");
  RART_CHECK_ARG(rart_bottleneck28_fused_supported(c_io, c_mid, h, w),
                 "rart_bottleneck28_fused_bf16: unsupported geometry (512 -> 128 -> 512 channels at 28 x 28 only)");
  RART_CHECK_ARG(x != out, "rart_bottleneck28_fused_bf16: out must not alias x");
  RART_CHECK_ARG(!backward || (m1 && m2), "rart_bottleneck28_fused_bf16: the backward pass needs both inner masks");
  RART_CHECK_ARG((long long)n * B28_HW * B28_HW * B28_CIO < (1ll << 31), "rart_bottleneck28_fused_bf16: tensor must stay below 2^31 elements");
  RartBneck28Desc d;
  d.x = (const uint16_t*)x; d.w1 = (const uint16_t*)w1; d.w2 = (const uint16_t*)w2; d.w3 = (const uint16_t*)w3;
  d.b1 = b1; d.b2 = b2; d.b3 = b3;
  d.m1 = (uint8_t*)m1; d.m2 = (uint8_t*)m2; d.m3 = (uint8_t*)m3;
  d.out = (uint16_t*)out;
  for (int t = 0; t < 9; ++t) {
    RART_CHECK_ARG(tap_dy[t] >= -1 && tap_dy[t] <= 1 && tap_dx[t] >= -1 && tap_dx[t] <= 1,
                   "rart_bottleneck28_fused_bf16: taps must lie in -1..1");
    d.tap_off[t] = (tap_dy[t] * 16 + tap_dx[t]) * 16;
  }
  if (backward) hipLaunchKernelGGL(k_bottleneck28<true>, dim3((uint32_t)n * 4), dim3(256), 0, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(k_bottleneck28<false>, dim3((uint32_t)n * 4), dim3(256), 0, (hipStream_t)stream, d);
  RART_CHECK_LAUNCH("rart_bottleneck28_fused_bf16");
  return RART_OK;
}
