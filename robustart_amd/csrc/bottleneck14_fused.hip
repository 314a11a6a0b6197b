// One identity Bottleneck of ResNet-50's layer3 (1024 -> 256 -> 256 -> 1024 channels at 14 x 14) as ONE kernel (gfx950), forward and
// backward-to-input, ONE IMAGE PER WORKGROUP (at B = 256: one workgroup per CU):
//
//     forward :  out = relu(W3 . relu(W2 * relu(W1 . x + b1) + b2) + b3 + x)          (+ the three 1-bit ReLU sign tensors)
//     backward:  dx  = mask_prev . (W1^T . (mask_a . (W2^T * (mask_b . (W3^T . g)))) + g)     (same three-stage shape, see
//                bottleneck_fused.hip)
//
// Why: as three launches the block runs 53 + 62 + 79 us (profiles/r02_igemm_per_shape.txt): the two 1x1 layers are far below the
// matrix rate because their tiles stream activations through HBM / L2 (784 and 3 136 tiles of 128 x 128 with 4-16 K steps each), only
// the LDS-resident 3x3 (k_conv3x3_image256, 944 TFLOP/s) is near what the 1 KiB-of-LDS-per-MFMA wave tiling allows.  Here all three
// stages use that kernel's inner loop -- a wave owns 32 output channels x all 7 position tiles (two image rows x 16 slots each),
// streams its weight fragments from L2 in fragment order and reads the positions operand from LDS -- and the two 256-channel
// intermediates never leave the CU:
//
//   stage A  a1 = W1 . x: x streams through LDS in 128-channel slices (224 slots x 256 B, global_load_lds_dwordx4, chunk s of row r in
//            slot s ^ (r & 15), two slices in flight inside the memory that later holds the image); operands are SWAPPED (weights = A) so
//            a lane ends up with 4 consecutive channels of one position and writes its bias / ReLU'd bf16 results with 8-byte ds_writes
//            into the zero-ringed, chunk-major a1 image (the k_conv3x3_image256 layout: 32 planes x 257 slots x 16 B)
//   stage B  the 36 K steps of the 3x3 from the a1 image (no barrier); results are packed to bf16 in registers and, after a barrier,
//            overwrite the a1 image as a2
//   stage C  out = W3 . a2 + x, 256 output channels per round (4 rounds), positions operand = the a2 image; results transposed through
//            LDS (28 valid positions x 32 channels per wave) so that residual loads and stores are 64-byte row segments
//
// LDS: image 131 584 B + staging 32 256 B = 160 KiB exactly.  HBM per block: x (103 MB at B = 256), the residual re-read, out.
//
// Reference step: Bottleneck.forward of the public ResNet-50 (RobustART/model/__init__.py:1 -> absent submodule;
// robustart_amd/model/resnet_torch.py) and its autograd inside every attack iteration
// (RobustART/noise/utils/adv/attack.py:21-22, Attacks/autoattack/autopgd_base.py:271-289).
#include "rart_common.h"
#include "rart_bf16_helpers.h"

struct RartBneck14Desc {
  const uint16_t* x;        // [n][14][14][1024] bf16: block input (forward) / masked gradient at the block output (backward)
  const uint16_t* w1;       // [256][1024] in fragment order (rart_pack_frag_bf16(rows 256, k 1024))
  const uint16_t* w2;       // [256][9*256] in fragment order (rart_conv3x3_pack_frag_bf16)
  const uint16_t* w3;       // [1024][256] in fragment order (rart_pack_frag_bf16(rows 1024, k 256))
  const float* b1;
  const float* b2;
  const float* b3;          // fp32 biases or null
  uint8_t* m1;              // 1 bit per element of the stage-A result ([P][32] bytes): forward = sign out (or null), backward = mask in
  uint8_t* m2;              // same for the stage-B result
  uint8_t* m3;              // [P][128] bytes for the output
  uint16_t* out;
  int tap_off[9];           // (dy * 16 + dx) * 16: byte offset of a tap inside an image plane
};

namespace {
using namespace rart_bf16;
__device__ __attribute__((aligned(16))) const uint32_t g_b14_zero16[4] = {0u, 0u, 0u, 0u};   // source of the padding slots of an x slice

constexpr int B14_HW = 14, B14_NP = 196;                 // image side, positions
constexpr int B14_CM = 256, B14_CIO = 1024;
constexpr int B14_MT = 7;                                // position tiles: two image rows x 16 slots (14 valid) each
constexpr int B14_PLANE = (256 + 1) * 16;                // 16 x 16 ring positions + 1: 16 mod 256 bytes
constexpr int B14_IMG = 32 * B14_PLANE;                  // 131 584 B: the 256-channel image, chunk-major
constexpr int B14_SLICE = 224 * 256;                     // one 128-channel slice of x: 224 slots x 256 B
static_assert(2 * B14_SLICE <= B14_IMG, "two x slices must fit the image memory");
constexpr int B14_LDE = 36;                              // staging row: 32 floats + 4
constexpr int B14_STG = 28 * B14_LDE * 4;                // 4 032 B per wave: 28 valid positions of a tile x 32 channels
static_assert(B14_IMG + 8 * B14_STG == 160 * 1024, "LDS budget");

template <bool BWD>
__global__ __launch_bounds__(512, 1) void k_bottleneck14(const RartBneck14Desc d) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[B14_IMG + 8 * B14_STG];
  uint8_t* const sImg = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p32 = lane & 31, h = lane >> 5;
  const long long pos0 = (long long)blockIdx.x * B14_NP;            // raster index of the image's first position
  // lane geometry of a position tile t: image row 2t + (p32 >> 4), column p32 & 15 (columns 14, 15 are padding slots)
  const int px = p32 & 15, prow = p32 >> 4;
  const bool pvalid = px < B14_HW;

  // ================================ stage A: a1[pos][256] = x[pos][1024] . W1^T ===============================================
  f32x16 acc[B14_MT];
  {
    // a wave-wide direct load covers 4 slots x 256 B: lane l -> slot 4 j + (l >> 4), chunk position l & 15, fetching chunk
    // (l & 15) ^ (slot & 15); instruction j of a slice = wave + 8 q, q = 0..6
    uint32_t xsrc[7];                                                    // byte offset from d.x (tensor < 2^31 elements), ~0u = zero page
    uint32_t xdst[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int j = wave + 8 * q, slot = 4 * j + (lane >> 4);          // slot = tile * 32 + row-in-tile * 16 + column
      const int t = slot >> 5, sy = 2 * t + ((slot >> 4) & 1), sx = slot & 15;
      const int chunk = (lane & 15) ^ (slot & 15);
      xsrc[q] = sx < B14_HW ? (uint32_t)(((pos0 + sy * B14_HW + sx) * B14_CIO + chunk * 8) * 2) : 0xFFFFFFFFu;
      xdst[q] = (uint32_t)__builtin_amdgcn_readfirstlane(4 * j) * 256u;
    }
#define RART_B14_ISSUE(S, BUF)                                                                                  \
  _Pragma("unroll") for (int q = 0; q < 7; ++q) {                                                               \
    const char* s_ = xsrc[q] != 0xFFFFFFFFu ? reinterpret_cast<const char*>(d.x) + (size_t)(xsrc[q] + (uint32_t)((S)*256))  \
                                            : reinterpret_cast<const char*>(g_b14_zero16);                        \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s_,                         \
                                     (__attribute__((address_space(3))) void*)(sImg + (BUF)*B14_SLICE + xdst[q]), 16, 0, 0); \
  }
    // weights of this wave: output channels 32 wave .. +31; fragment (K step st of 64, row tile wave, ks) of the [256][1024] table
    const uint16_t* wp = d.w1 + (size_t)wave * 2048 + lane * 8;
    // weight fragments are pipelined per K STEP of 64 (two per 128-channel slice of x), one step ahead: four fragments per set.
    // (Round 2 prefetched a whole slice -- eight fragments per set, 64 VGPRs -- which put the stage at the 256-register cap with
    // 6 / 10 spilled registers; half a slice, 28 MFMAs, still covers the L2 round trip.)
    bf16x8 wq[2][4];
#define RART_B14_LOADW(A, SET)                                                                                  \
  _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                                 \
    wq[SET][f] = *reinterpret_cast<const bf16x8*>(wp + (size_t)((A) * 8) * 2048 + f * 512);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (!BWD && d.b1) bv = *reinterpret_cast<const f32x4*>(d.b1 + wave * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < B14_MT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = bv[i];
    }
    RART_B14_ISSUE(0, 0)
    RART_B14_LOADW(0, 0)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const uint32_t xrow = (uint32_t)(p32 * 256), xsw = (uint32_t)(p32 & 15);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int buf = s & 1;
      if (s + 1 < 8) RART_B14_ISSUE(s + 1, buf ^ 1)
      const uint8_t* xb = sImg + buf * B14_SLICE + xrow;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int a = 2 * s + half;                        // K step of 64 channels; its fragments sit in set a & 1
        if (a + 1 < 16) {
          if (a & 1) { RART_B14_LOADW(a + 1, 0) } else { RART_B14_LOADW(a + 1, 1) }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint32_t co = ((uint32_t)(2 * (4 * half + ks) + h) ^ xsw) << 4;
#pragma unroll
          for (int t = 0; t < B14_MT; ++t) {
            const bf16x8 pf = *reinterpret_cast<const bf16x8*>(xb + t * 32 * 256 + co);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[a & 1][ks], pf, acc[t], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0);        // the next slice (and the next weights) have landed
      __syncthreads();
    }
#undef RART_B14_ISSUE
#undef RART_B14_LOADW
  }
  // every wave is past its last slice read: the memory becomes the zero-ringed a1 image.  Ring slots first (row 0, row 15,
  // column 0, column 15 of the 16 x 16 grid, + slot 256), then the interior from the accumulators.
  for (int i = tid; i < 32 * 61; i += 512) {
    const int plane = i / 61, e = i - plane * 61;
    const int slot = e < 16 ? e : (e < 32 ? 240 + (e - 16) : (e < 46 ? (e - 32 + 1) * 16 : (e < 60 ? (e - 46 + 1) * 16 + 15 : 256)));
    *reinterpret_cast<uint4*>(sImg + plane * B14_PLANE + slot * 16) = make_uint4(0, 0, 0, 0);
  }
  // lane: position p32 of tile t, channels wave*32 + 8g + 4h + (0..3) -> 8 bytes of chunk wave*4 + g
#define RART_B14_STORE_IMG(MASKPTR, BIASED)                                                                     \
  _Pragma("unroll") for (int t = 0; t < B14_MT; ++t) {                                                          \
    const int y_ = 2 * t + prow;                                                                                \
    uint32_t mbits_ = 0xFFFFFFFFu;                                                                              \
    if (BWD && pvalid) mbits_ = *reinterpret_cast<const uint32_t*>((MASKPTR) + (pos0 + y_ * B14_HW + px) * 32 + wave * 4); \
    uint8_t* dst_ = sImg + ((y_ + 1) * 16 + px + 1) * 16 + (wave * 4) * B14_PLANE + h * 8;                      \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                             \
      uint32_t w0 = pack_bf16x2(acc[t][4 * g], acc[t][4 * g + 1]), w1 = pack_bf16x2(acc[t][4 * g + 2], acc[t][4 * g + 3]); \
      if (BWD) {                                                                                                \
        const uint32_t byte = (mbits_ >> (8 * g)) & 0xFFu;                                                      \
        w0 &= halves_from_bits(byte, 2 * h);                                                                    \
        w1 &= halves_from_bits(byte, 2 * h + 1);                                                                \
      } else {                                                                                                  \
        w0 = relu_bf16x2(w0);                                                                                   \
        w1 = relu_bf16x2(w1);                                                                                   \
      }                                                                                                         \
      if (pvalid) *reinterpret_cast<uint2*>(dst_ + g * B14_PLANE) = make_uint2(w0, w1);                         \
    }                                                                                                           \
  }
  RART_B14_STORE_IMG(d.m1, 1)
  __syncthreads();

  // sign bits of an image (forward, when the backward pass will follow): byte (pos * 256 + ch) >> 3 = pos * 32 + chunk
#define RART_B14_SIGN_IMG(PTR)                                                                                  \
  if (!BWD && (PTR)) {                                                                                          \
    for (int i = tid; i < B14_NP * 32; i += 512) {                                                              \
      const int p = i >> 5, chunk = i & 31, y = p / B14_HW, xx = p - y * B14_HW;                                \
      const uint4 v = *reinterpret_cast<const uint4*>(sImg + chunk * B14_PLANE + ((y + 1) * 16 + xx + 1) * 16); \
      (PTR)[pos0 * 32 + i] = (uint8_t)sign_byte(v);                                                             \
    }                                                                                                           \
  }
  RART_B14_SIGN_IMG(d.m1)

  // ================================ stage B: a2 = 3x3 over the a1 image (k_conv3x3_image256's loop, operands swapped) ==========
  // position tile t reads the image at abase0 + t * 512 (two image rows = 32 slots of 16 B further down): one register, the tile
  // offset is an immediate (an array of seven bases cost six more VGPRs at the 256-register cap)
  const uint32_t abase0 = (uint32_t)(((prow + 1) * 16 + (px < B14_HW ? px : B14_HW - 1) + 1) * 16 + h * B14_PLANE);   // padding slots read a valid position; never stored
  {
    const uint16_t* wp = d.w2 + wave * 2048 + lane * 8;                    // fragment (st, wave, ks): (st * 8 + wave) * 4 + ks
    bf16x8 bq[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bq[0][ks] = *reinterpret_cast<const bf16x8*>(wp + ks * 512);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (!BWD && d.b2) bv = *reinterpret_cast<const f32x4*>(d.b2 + wave * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < B14_MT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = bv[i];
    }
#pragma unroll
    for (int st = 0; st < 36; ++st) {
      const int tap = st >> 2, kh = st & 3;
      if (st + 1 < 36) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bq[(st + 1) & 1][ks] = *reinterpret_cast<const bf16x8*>(wp + (size_t)(st + 1) * 8 * 2048 + ks * 512);
      }
      __builtin_amdgcn_sched_barrier(0);                    // keep the prefetch ahead of this step's MFMAs
      const int toff = d.tap_off[tap];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int t = 0; t < B14_MT; ++t) {
          const bf16x8 pf = *reinterpret_cast<const bf16x8*>(sImg + (int)abase0 + t * 512 + toff + (kh * 8 + ks * 2) * B14_PLANE);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[st & 1][ks], pf, acc[t], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();          // every wave is done reading a1: its interior is overwritten by a2 (the ring stays zero, unused by stage C)
  RART_B14_STORE_IMG(d.m2, 1)
  __syncthreads();
  RART_B14_SIGN_IMG(d.m2)
#undef RART_B14_STORE_IMG
#undef RART_B14_SIGN_IMG

  // ================================ stage C: out[pos][1024] = a2[pos][256] . W3^T + x, 256 channels per round =================
  float* const sE = reinterpret_cast<float*>(lds + B14_IMG + wave * B14_STG);
  const int vp = prow * B14_HW + px;                        // compact index of a valid slot inside its tile (0..27)
#pragma unroll 1
  for (int rd = 0; rd < 4; ++rd) {
    const int ch0 = rd * 256 + wave * 32;                   // first output channel of this wave in this round
    const uint16_t* wp = d.w3 + (size_t)(rd * 8 + wave) * 2048 + lane * 8;     // fragment (st, row tile rd*8 + wave, ks), 32 row tiles
    bf16x8 bq[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bq[0][ks] = *reinterpret_cast<const bf16x8*>(wp + ks * 512);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (!BWD && d.b3) bv = *reinterpret_cast<const f32x4*>(d.b3 + ch0 + 8 * g + 4 * h);
#pragma unroll
      for (int t = 0; t < B14_MT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][4 * g + i] = bv[i];
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      if (st + 1 < 4) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bq[(st + 1) & 1][ks] = *reinterpret_cast<const bf16x8*>(wp + (size_t)(st + 1) * 32 * 2048 + ks * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int t = 0; t < B14_MT; ++t) {
          const bf16x8 pf = *reinterpret_cast<const bf16x8*>(sImg + (int)abase0 + t * 512 + (st * 8 + ks * 2) * B14_PLANE);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[st & 1][ks], pf, acc[t], 0, 0, 0);
        }
      }
    }
    // epilogue: per tile the 28 valid positions x 32 channels go through the wave's staging slice; then 4 lanes per position read
    // 8 channels each: 64-byte row segments, two passes (16 + 12 positions)
    const int cw = lane & 3, rw = lane >> 2;
#pragma unroll
    for (int t = 0; t < B14_MT; ++t) {
      u32x4 rv[2];
      uint32_t mb[2];
      int eoff[2];                                          // element offsets: the host guarantees the tensor stays below 2^31 elements
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int v = q * 16 + rw;                          // valid-slot index 0..27
        const int vy = v >= B14_HW ? 1 : 0, vx = v - vy * B14_HW;
        eoff[q] = v < 28 ? (int)((pos0 + (2 * t + vy) * B14_HW + vx) * B14_CIO) + ch0 + cw * 8 : -1;
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
          *reinterpret_cast<f32x4*>(sE + vp * B14_LDE + 8 * g + 4 * h) = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (eoff[q] >= 0) {
          const int v = q * 16 + rw;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sE + v * B14_LDE + cw * 8);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(sE + v * B14_LDE + cw * 8 + 4);
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

// 1 if rart_bottleneck14_fused_bf16 runs this block geometry
extern "C" int rart_bottleneck14_fused_supported(int c_io, int c_mid, int h, int w) {
  return (c_io == B14_CIO && c_mid == B14_CM && h == B14_HW && w == B14_HW) ? 1 : 0;
}

extern "C" int rart_bottleneck14_fused_bf16(const void* x, const void* w1, const void* w2, const void* w3, const float* b1,
                                            const float* b2, const float* b3, void* m1, void* m2, void* m3, void* out, int n,
                                            int h, int w, int c_io, int c_mid, const int* tap_dy, const int* tap_dx, int backward,
                                            rart_stream_t stream) {
  RART_CHECK_ARG(x && w1 && w2 && w3 && out && tap_dy && tap_dx && n > 0, "rart_bottleneck14_fused_bf16: bad arguments");
  RART_CHECK_ARG(rart_bottleneck14_fused_supported(c_io, c_mid, h, w),
                 "rart_bottleneck14_fused_bf16: unsupported geometry (1024 -> 256 -> 1024 channels at 14 x 14 only)");
  RART_CHECK_ARG(x != out, "rart_bottleneck14_fused_bf16: out must not alias x");
  RART_CHECK_ARG(!backward || (m1 && m2), "rart_bottleneck14_fused_bf16: the backward pass needs both inner masks");
  RART_CHECK_ARG((long long)n * B14_NP * B14_CIO < (1ll << 31), "rart_bottleneck14_fused_bf16: tensor must stay below 2^31 elements");
  RartBneck14Desc d;
  d.x = (const uint16_t*)x; d.w1 = (const uint16_t*)w1; d.w2 = (const uint16_t*)w2; d.w3 = (const uint16_t*)w3;
  d.b1 = b1; d.b2 = b2; d.b3 = b3;
  d.m1 = (uint8_t*)m1; d.m2 = (uint8_t*)m2; d.m3 = (uint8_t*)m3;
  d.out = (uint16_t*)out;
  for (int t = 0; t < 9; ++t) {
    RART_CHECK_ARG(tap_dy[t] >= -1 && tap_dy[t] <= 1 && tap_dx[t] >= -1 && tap_dx[t] <= 1,
                   "rart_bottleneck14_fused_bf16: taps must lie in -1..1");
    d.tap_off[t] = (tap_dy[t] * 16 + tap_dx[t]) * 16;
  }
  if (backward) hipLaunchKernelGGL(k_bottleneck14<true>, dim3((uint32_t)n), dim3(512), 0, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(k_bottleneck14<false>, dim3((uint32_t)n), dim3(512), 0, (hipStream_t)stream, d);
  RART_CHECK_LAUNCH("rart_bottleneck14_fused_bf16");
  return RART_OK;
}
