// One identity Bottleneck of ResNet-50's layer1 (256 -> 64 -> 64 -> 256 channels at 56 x 56) as ONE kernel (gfx950), forward
// and backward-to-input:
//
//     forward :  out = relu(W3 . relu(W2 * relu(W1 . x + b1) + b2) + b3 + x)             (+ the three 1-bit ReLU sign tensors)
//     backward:  dx  = mask_prev . (W1^T . (mask_a . (W2^T * (mask_b . (W3^T . g)))) + g)
//
// Both are "1x1 reduce (256 -> 64), 3x3 (64 -> 64), 1x1 expand (64 -> 256), add the kernel's own input", so one kernel serves
// both with transposed / flipped weight tables; only the point-wise steps differ (bias + ReLU + sign out  vs  mask in).
//
// Why: at batch 256 the three layers of such a block are HBM-bound (profiles/r02_igemm_per_shape.txt: 128 + 100 + 179 us
// against a 1.6 GB stream of x, a1, a2, x again and out).  The two 64-channel intermediates never need to exist in HBM -- the
// backward pass only wants their signs -- so a workgroup that owns 4 image rows keeps them on chip:
//
//   stage A  x (6 halo rows x 56 positions x 256 ch, straight from global memory into the MFMA B operand: 4 lanes of a
//            16x16x32 MFMA read 64 contiguous bytes of one position) . W1 (LDS resident) -> a1, bf16, into the LDS halo
//            tile T1 (chunk-major planes as in conv3x3_halo.hip, zero columns left / right, zero rows outside the image)
//   stage B  the 9 taps from T1, weights straight from L2 one tap ahead (no barrier in the tap loop); operands are swapped
//            (weights = A, positions = B) so a lane ends up with 4 consecutive channels of one position, and one
//            v_permlane32_swap turns the bias / ReLU'd bf16 results into the B-operand fragments of stage C: a2 stays in
//            registers
//   stage C  a2 . W3 (LDS resident, loaded while stage B runs) 64 output channels at a time, transposed through LDS (the T1
//            tile is dead by then) so that residual loads and stores are 128-byte row segments
//
// HBM traffic per block: x once (+ 50 % halo rows, mostly L2 / Infinity Cache hits), the residual re-read (same rows, L2 /
// Infinity Cache) and out: ~0.9-1.1 GB instead of 1.6 GB, and one launch instead of three.
//
// Reference step: Bottleneck.forward of the public ResNet-50 (RobustART/model/__init__.py:1 -> absent submodule;
// robustart_amd/model/resnet_torch.py) and its autograd inside every attack iteration
// (RobustART/noise/utils/adv/attack.py:21-22, Attacks/autoattack/autopgd_base.py:271-289).
#include "rart_common.h"
#include "rart_bf16_helpers.h"

struct RartBneckDesc {
  const uint16_t* x;        // [n][56][56][256] bf16: the block input (forward) / the masked gradient at the block output (backward)
  const uint16_t* w1;       // [64][256]   rows = stage-A output channel
  const uint16_t* w2;       // fragment-major 3x3 table (rart_conv3x3_pack_frag_bf16): [tap][blk][s][lane][8]
  const uint16_t* w3;       // [256][64]   rows = stage-C output channel ([64][256] in the first block's backward)
  const uint16_t* w4;       // first block only: the projection-shortcut table (see rart_bottleneck_first_bf16)
  const float* b1;
  const float* b2;
  const float* b3;          // fp32 biases or null
  uint8_t* m1;              // 1 bit per element of the stage-A result ([P][8] bytes): forward = sign out (or null), backward = mask in
  uint8_t* m2;              // same for the stage-B result
  uint8_t* m3;              // [P][32] bytes for the output
  uint16_t* out;
  uint32_t tiles;           // n * 14
  int tap_off[9];           // (dy * 58 + dx) * 16: byte offset of a tap inside a T1 plane
#ifdef RART_BNECK_TS
  unsigned long long* ts;   // scratch/exp/bneck_ts.hip: cycle stamps per wave and phase
#endif
};

namespace {
using namespace rart_bf16;

constexpr int BF_W = 56, BF_H = 56, BF_R = 4;          // image size; output rows per workgroup
constexpr int BF_TPI = BF_H / BF_R;                     // 14 tiles per image
constexpr int BF_HR = BF_R + 2;                         // halo rows
constexpr int BF_SW = BF_W + 2;                         // slots per halo row (zero column on either side)
constexpr int BF_NQ = BF_HR * BF_W;                     // 336 halo positions stage A computes
constexpr int BF_NP = BF_R * BF_W;                      // 224 output positions
constexpr int BF_PLANE = (BF_HR * BF_SW + 5) * 16;      // 353 slots: 16 mod 256 bytes (see conv3x3_halo.hip)
static_assert(BF_PLANE % 256 == 16, "plane stride must be 16 mod 256");
constexpr int BF_T1 = 8 * BF_PLANE;                     // 45 184 B: a1 halo tile, later the epilogue staging
constexpr int BF_WB = 32768;                            // W1 (64 x 256), later W3 (256 x 64), XOR-swizzled 16-byte chunks
constexpr int BF_BIAS = (64 + 64 + 256) * 4;            // the three bias vectors
constexpr int BF_LDE = 68;                              // staging row: 64 floats + 4
static_assert(4 * 32 * BF_LDE * 4 <= BF_T1, "epilogue staging must fit the halo tile");

// FIRST = the layer's first block (64 input channels, projection shortcut W4 instead of the identity): stage A reads 64
// channels (forward) and stage C folds the shortcut in as extra K (forward: [a2 | x] . [W3 | W4]^T; backward: 64 output
// channels = d_a1 . W1^T + g . W4^T)
template <bool BWD, bool FIRST>
__global__ __launch_bounds__(256, 2) void k_bottleneck56(const RartBneckDesc d) {
  constexpr bool FF = FIRST && !BWD, FB = FIRST && BWD;
  constexpr int CIN = FF ? 64 : 256;          // channels of d.x
  constexpr int NJ = CIN / 32;                // 32-channel K steps of stage A
  __shared__ __attribute__((aligned(16))) uint8_t lds[BF_T1 + BF_WB + BF_BIAS];
  uint8_t* const sT = lds;
  uint8_t* const sW = lds + BF_T1;
  float* const sBias = reinterpret_cast<float*>(lds + BF_T1 + BF_WB);      // b1[64] | b2[64] | b3[256] (zeros when absent)
#ifdef RART_BNECK_TS
  unsigned long long ts_[8];
  ts_[0] = __builtin_readcyclecounter();
#define RART_STAMP(I) ts_[I] = __builtin_readcyclecounter();
#else
#define RART_STAMP(I)
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // vertically adjacent tiles share halo rows: keep neighbours on one XCD (its L2 serves the overlap)
  uint32_t tile;
  {
    const uint32_t nb = gridDim.x, bid = blockIdx.x, xcd = bid & 7u, slot = bid >> 3, q = nb >> 3, r = nb & 7u;
    tile = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + slot;
  }
  const uint32_t img = tile / BF_TPI;
  const int y0 = (int)(tile - img * BF_TPI) * BF_R;
  const long long pos0 = ((long long)img * BF_H + y0) * BF_W;      // raster index of the first output position

  // ---- prologue: W1 -> LDS (chunk c of row r at chunk c ^ (r & 31)), zero columns of T1
  if constexpr (FF) {      // W1 is 64 x 64: 128-byte rows, chunk c of row r at chunk c ^ ((r >> 1) & 7)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = u * 256 + tid, row = idx >> 3, chunk = idx & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(d.w1 + idx * 8);
      *reinterpret_cast<uint4*>(sW + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)) = v;
    }
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = u * 256 + tid, row = idx >> 5, chunk = idx & 31;
      const uint4 v = *reinterpret_cast<const uint4*>(d.w1 + idx * 8);
      *reinterpret_cast<uint4*>(sW + row * 512 + ((chunk ^ (row & 31)) << 4)) = v;
    }
  }
  {
    float b = 0.f;                                   // tid: 0..63 b1, 64..127 b2; b3 below
    if (!BWD && tid < 64 && d.b1) b = d.b1[tid];
    if (!BWD && tid >= 64 && tid < 128 && d.b2) b = d.b2[tid - 64];
    if (tid < 128) sBias[tid] = b;
    sBias[128 + tid] = (!BWD && d.b3) ? d.b3[tid] : 0.f;
  }
  if (tid < 8 * 2 * BF_HR) {
    const int plane = tid & 7, e = tid >> 3, hr = e >> 1, side = e & 1;
    *reinterpret_cast<uint4*>(sT + plane * BF_PLANE + (hr * BF_SW + side * (BF_SW - 1)) * 16) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  RART_STAMP(1)

  // ---- stage A: a1[q][64] = x[q][256] . W1^T for the 336 halo positions, 32 positions (two 16-wide MFMA tiles) per pass;
  //      a wave's passes are u = wave, wave + 4, wave + 8 (< 11); the loads of pass i + 1 are in flight while pass i computes
  {
    const int c16 = lane & 15, kq = lane >> 4;
    constexpr int NU = (BF_NQ + 31) / 32;
    bf16x8 xf[2][2][NJ];
    bool valid[2][2];
    int q[2][2];
    unsigned long long mbits[2][2];
#define RART_BN_LOAD_A(SET, U)                                                                                  \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                               \
    q[SET][t] = (U) * 32 + t * 16 + c16;                                                                        \
    const int hr_ = q[SET][t] / BF_W;                                                                           \
    valid[SET][t] = q[SET][t] < BF_NQ && (unsigned)(y0 - 1 + hr_) < (unsigned)BF_H;                             \
    const uint16_t* src_ = d.x + (pos0 - BF_W + q[SET][t]) * CIN + kq * 8;                                      \
    mbits[SET][t] = 0;                                                                                          \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                             \
      uint4 v_ = make_uint4(0, 0, 0, 0);                                                                        \
      if (valid[SET][t]) v_ = *reinterpret_cast<const uint4*>(src_ + j * 32);                                   \
      xf[SET][t][j] = __builtin_bit_cast(bf16x8, v_);                                                           \
    }                                                                                                           \
    if (BWD && valid[SET][t])                                                                                   \
      mbits[SET][t] = *reinterpret_cast<const unsigned long long*>(d.m1 + (pos0 - BF_W + q[SET][t]) * 8);       \
  }
    RART_BN_LOAD_A(0, wave)
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int u = wave + 4 * it;
      if (u >= NU) continue;
      const int cur = it & 1;
      if (it + 1 < 3 && u + 4 < NU) {
        if (cur == 0) { RART_BN_LOAD_A(1, u + 4) } else { RART_BN_LOAD_A(0, u + 4) }
      }
      __builtin_amdgcn_sched_barrier(0);       // keep the next pass's loads ahead of this pass's MFMAs
      f32x4 acc[2][4];
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + blk * 16 + 4 * kq);
        acc[0][blk] = bv;
        acc[1][blk] = bv;
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
          const int row = blk * 16 + c16;
          const bf16x8 wf = FF ? *reinterpret_cast<const bf16x8*>(sW + row * 128 + (((j * 4 + kq) ^ ((row >> 1) & 7)) << 4))
                               : *reinterpret_cast<const bf16x8*>(sW + row * 512 + (((j * 4 + kq) ^ (row & 31)) << 4));
          acc[0][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[cur][0][j], acc[0][blk], 0, 0, 0);
          acc[1][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[cur][1][j], acc[1][blk], 0, 0, 0);
        }
      }
      // lane: position c16 of tile t, channels blk*16 + 4*kq + (0..3) -> 8 bytes of chunk blk*2 + (kq >> 1)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (q[cur][t] < BF_NQ) {
          const int hr = q[cur][t] / BF_W, col = q[cur][t] - hr * BF_W;
          uint8_t* dst = sT + (hr * BF_SW + col + 1) * 16 + (kq >> 1) * BF_PLANE + (kq & 1) * 8;
#pragma unroll
          for (int blk = 0; blk < 4; ++blk) {
            uint32_t w0 = pack_bf16x2(acc[t][blk][0], acc[t][blk][1]), w1 = pack_bf16x2(acc[t][blk][2], acc[t][blk][3]);
            if (BWD) {
              const uint32_t byte = (uint32_t)(mbits[cur][t] >> (8 * (blk * 2 + (kq >> 1)))) & 0xFFu;
              w0 &= halves_from_bits(byte, (kq & 1) * 2);
              w1 &= halves_from_bits(byte, (kq & 1) * 2 + 1);
            } else {
              w0 = relu_bf16x2(w0);
              w1 = relu_bf16x2(w1);
            }
            if (!valid[cur][t]) w0 = w1 = 0u;       // rows outside the image are the 3x3's zero padding, not conv1(0)
            *reinterpret_cast<uint2*>(dst + blk * 2 * BF_PLANE) = make_uint2(w0, w1);
          }
        }
      }
    }
#undef RART_BN_LOAD_A
  }
  RART_STAMP(2)
  __syncthreads();          // T1 complete; W1 is dead
  RART_STAMP(3)

  // ---- W3 -> registers now (in flight during stage B), LDS after the tap loop
  // (explicit scalars: a uint4 array of 8 lands in scratch with this compiler)
#define RART_W3_LOAD(U) const uint4 w3r##U = *reinterpret_cast<const uint4*>(d.w3 + ((U) * 256 + tid) * 8);
  RART_W3_LOAD(0) RART_W3_LOAD(1) RART_W3_LOAD(2) RART_W3_LOAD(3) RART_W3_LOAD(4) RART_W3_LOAD(5) RART_W3_LOAD(6) RART_W3_LOAD(7)
#undef RART_W3_LOAD

  if (!BWD && d.m1) {       // sign bits of a1 for the backward pass: the centre rows of T1
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int i = u * 256 + tid, p = i >> 3, chunk = i & 7, r = p / BF_W, c = p - r * BF_W;
      const uint4 v = *reinterpret_cast<const uint4*>(sT + chunk * BF_PLANE + ((r + 1) * BF_SW + c + 1) * 16);
      d.m1[pos0 * 8 + i] = (uint8_t)sign_byte(v);
    }
  }

  // ---- stage B: a2[p][64] = 3x3 over T1; wave: position tiles {wave, wave + 4} x all 64 channels
  const int p32 = lane & 31, h = lane >> 5;
  int pt[2];
  pt[0] = wave;
  pt[1] = wave + 4 < BF_NP / 32 ? wave + 4 : wave;     // wave 3 has one tile: its second slot repeats the first, results unused
  const bool two = wave + 4 < BF_NP / 32;
  bf16x8 a2f[2][4];
  {
    uint32_t abase[2];
    unsigned long long mbits[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int p = pt[t] * 32 + p32, r = p / BF_W, c = p - r * BF_W;
      abase[t] = (uint32_t)(((r + 1) * BF_SW + c + 1) * 16 + h * BF_PLANE);
      mbits[t] = 0;
      if (BWD) mbits[t] = *reinterpret_cast<const unsigned long long*>(d.m2 + (pos0 + p) * 8);
    }
    // fragment-major table: the 64 lanes of a fragment load read 1 KiB contiguous (row-major weights put every lane on its own
    // cache line: the tap loop was bound by the texture-address unit, 20 k of the kernel's 97 k cycles per wave)
    const uint16_t* wp = d.w2 + lane * 8;
    bf16x8 wq[3][2][4];       // weights of three taps: loads run TWO taps ahead (one tap of MFMAs is shorter than an L2 round trip)
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int s = 0; s < 4; ++s) wq[tp][blk][s] = *reinterpret_cast<const bf16x8*>(wp + ((tp * 2 + blk) * 4 + s) * 512);
    f32x16 acc[2][2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + 64 + blk * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0][blk][4 * g + i] = acc[1][blk][4 * g + i] = bv[i];
      }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 2 < 9) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            wq[(tap + 2) % 3][blk][s] = *reinterpret_cast<const bf16x8*>(wp + (((tap + 2) * 2 + blk) * 4 + s) * 512);
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the prefetch at the top of the tap (see conv3x3_halo.hip)
      const int off = d.tap_off[tap];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 pf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) pf[t] = *reinterpret_cast<const bf16x8*>(sT + abase[t] + off + 2 * s * BF_PLANE);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
            acc[t][blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[tap % 3][blk][s], pf[t], acc[t][blk], 0, 0, 0);
      }
    }
    // lane: position p32 of tile t, channels blk*32 + 8g + 4h + (0..3).  Pack, ReLU / mask, and exchange halves between
    // lane l and l + 32 so that lane half h owns the whole 8-channel chunk 2s + h of k-step s = blk*2 + g/2
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int ge = 0; ge < 4; ge += 2) {
          uint32_t e[2], o[2];
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            e[w] = pack_bf16x2(acc[t][blk][4 * ge + 2 * w], acc[t][blk][4 * ge + 2 * w + 1]);
            o[w] = pack_bf16x2(acc[t][blk][4 * ge + 4 + 2 * w], acc[t][blk][4 * ge + 4 + 2 * w + 1]);
            if (BWD) {
              const uint32_t be = (uint32_t)(mbits[t] >> (8 * (blk * 4 + ge))) & 0xFFu;
              const uint32_t bo = (uint32_t)(mbits[t] >> (8 * (blk * 4 + ge + 1))) & 0xFFu;
              e[w] &= halves_from_bits(be, 2 * h + w);
              o[w] &= halves_from_bits(bo, 2 * h + w);
            } else {
              e[w] = relu_bf16x2(e[w]);
              o[w] = relu_bf16x2(o[w]);
            }
            const auto sw = __builtin_amdgcn_permlane32_swap(e[w], o[w], false, false);
            e[w] = sw[0];
            o[w] = sw[1];
          }
          const uint4 frag = make_uint4(e[0], e[1], o[0], o[1]);
          a2f[t][blk * 2 + ge / 2] = __builtin_bit_cast(bf16x8, frag);
          if (!BWD && d.m2 && (t == 0 || two))
            d.m2[(pos0 + pt[t] * 32 + p32) * 8 + blk * 4 + ge + h] = (uint8_t)sign_byte(frag);
        }
  }
  RART_STAMP(4)
  // W3 -> LDS: chunk c of row r at chunk c ^ ((r >> 1) & 7) (stage A's readers of this region passed the barrier above)
#define RART_W3_STORE(U)                                                                          \
  if constexpr (FB) {      /* 64 x 256 shortcut table: the stage-A layout (512-byte rows) */       \
    const int idx = (U) * 256 + tid, row = idx >> 5, chunk = idx & 31;                            \
    *reinterpret_cast<uint4*>(sW + row * 512 + ((chunk ^ (row & 31)) << 4)) = w3r##U;             \
  } else {                                                                                        \
    const int idx = (U) * 256 + tid, row = idx >> 3, chunk = idx & 7;                             \
    *reinterpret_cast<uint4*>(sW + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)) = w3r##U;       \
  }
  RART_W3_STORE(0) RART_W3_STORE(1) RART_W3_STORE(2) RART_W3_STORE(3) RART_W3_STORE(4) RART_W3_STORE(5) RART_W3_STORE(6) RART_W3_STORE(7)
#undef RART_W3_STORE
  __syncthreads();          // every wave is done with T1 (it becomes the staging area) and W3 is in place
  RART_STAMP(5)

  // ---- stage C: out[p][256] = a2[p][64] . W3^T + x, 64 channels per round
  float* const sE = reinterpret_cast<float*>(sT) + wave * 32 * BF_LDE;
  const int cw = lane & 7, rw = lane >> 3;
  uint32_t woff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) woff[s] = (uint32_t)(p32 * 128 + (((2 * s + h) ^ ((p32 >> 1) & 7)) << 4));
  if constexpr (FF) {
    // ---- first block, forward: out[p][256] = [a2 | x][p][128] . [W3 | W4]^T: W3 from LDS, the shortcut table W4 in fragment
    //      order from L2 one round ahead, x fragments straight from global memory; rounds outer, tiles inner (W4 fragments
    //      serve both tiles)
    bf16x8 xcf[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        xcf[t][s] = *reinterpret_cast<const bf16x8*>(d.x + (pos0 + pt[t] * 32 + p32) * 64 + (2 * s + h) * 8);
    const uint16_t* w4p = d.w4 + lane * 8;
    bf16x8 wd[2][2][4];
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
      for (int s = 0; s < 4; ++s) wd[0][b2][s] = *reinterpret_cast<const bf16x8*>(w4p + (b2 * 4 + s) * 512);
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      if (rd + 1 < 4) {
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            wd[(rd + 1) & 1][b2][s] = *reinterpret_cast<const bf16x8*>(w4p + (((rd + 1) * 2 + b2) * 4 + s) * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && !two) continue;
        const long long pb = pos0 + pt[t] * 32;
        f32x16 acc[2];
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + 128 + (rd * 2 + b2) * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[b2][4 * g + i] = bv[i];
          }
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sW + (rd * 2 + b2) * 4096 + woff[s]);
            acc[b2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, a2f[t][s], acc[b2], 0, 0, 0);
          }
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            acc[b2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wd[rd & 1][b2][s], xcf[t][s], acc[b2], 0, 0, 0);
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = {acc[b2][4 * g], acc[b2][4 * g + 1], acc[b2][4 * g + 2], acc[b2][4 * g + 3]};
            *reinterpret_cast<f32x4*>(sE + p32 * BF_LDE + b2 * 32 + 8 * g + 4 * h) = v;
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int r = qd * 8 + rw;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sE + r * BF_LDE + cw * 8);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(sE + r * BF_LDE + cw * 8 + 4);
          const uint4 o = make_uint4(relu_bf16x2(pack_bf16x2(v0[0], v0[1])), relu_bf16x2(pack_bf16x2(v0[2], v0[3])),
                                     relu_bf16x2(pack_bf16x2(v1[0], v1[1])), relu_bf16x2(pack_bf16x2(v1[2], v1[3])));
          const long long e = (pb + r) * 256 + rd * 64 + cw * 8;
          RART_LAB_STORE16(d.out + e, o);
          if (d.m3) d.m3[e >> 3] = (uint8_t)sign_byte(o);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else if constexpr (FB) {
    // ---- first block, backward: dx[p][64] = d_a1[p][64] . W1^T + g[p][256] . W4^T: W1^T fragments (d.w4, 64 x 64 row-major) in
    //      registers, W4^T (64 x 256) from LDS, g fragments of the centre rows straight from global memory
    bf16x8 w1t[2][4];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        w1t[blk][s] = *reinterpret_cast<const bf16x8*>(d.w4 + (blk * 32 + p32) * 64 + (2 * s + h) * 8);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 1 && !two) continue;
      const long long pb = pos0 + pt[t] * 32;
      bf16x8 gf[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) gf[s] = *reinterpret_cast<const bf16x8*>(d.x + (pb + p32) * 256 + (2 * s + h) * 8);
      uint32_t mbq[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        mbq[qd] = 0xFFu;
        if (d.m3) mbq[qd] = d.m3[((pb + qd * 8 + rw) * 64 + cw * 8) >> 3];
      }
      f32x16 acc[2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[blk][i] = 0.f;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1t[blk][s], a2f[t][s], acc[blk], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sW + (blk * 32 + p32) * 512 + (((2 * s + h) ^ p32) << 4));
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, gf[s], acc[blk], 0, 0, 0);
        }
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {acc[blk][4 * g], acc[blk][4 * g + 1], acc[blk][4 * g + 2], acc[blk][4 * g + 3]};
          *reinterpret_cast<f32x4*>(sE + p32 * BF_LDE + blk * 32 + 8 * g + 4 * h) = v;
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int r = qd * 8 + rw;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sE + r * BF_LDE + cw * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(sE + r * BF_LDE + cw * 8 + 4);
        const uint4 o = make_uint4(pack_bf16x2(v0[0], v0[1]) & halves_from_bits(mbq[qd], 0),
                                   pack_bf16x2(v0[2], v0[3]) & halves_from_bits(mbq[qd], 1),
                                   pack_bf16x2(v1[0], v1[1]) & halves_from_bits(mbq[qd], 2),
                                   pack_bf16x2(v1[2], v1[3]) & halves_from_bits(mbq[qd], 3));
        RART_LAB_STORE16(d.out + (pb + r) * 64 + cw * 8, o);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    // rounds r8 = t * 4 + rd: the residual / mask loads of round r8 + 1 are in flight during round r8's MFMAs and epilogue
    u32x4 rv[2][4];
    uint32_t mb[2][4];
#define RART_BN_LOAD_C(SET, R8)                                                                   \
    _Pragma("unroll") for (int qd = 0; qd < 4; ++qd) {                                              \
      const long long e_ = (pos0 + pt[(R8) >> 2] * 32 + qd * 8 + rw) * 256 + ((R8)&3) * 64 + cw * 8; \
      rv[SET][qd] = *reinterpret_cast<const u32x4*>(d.x + e_);                                      \
      mb[SET][qd] = 0xFFu;                                                                          \
      if (BWD && d.m3) mb[SET][qd] = d.m3[e_ >> 3];                                                 \
    }
    const int n_rounds = two ? 8 : 4;
    RART_BN_LOAD_C(0, 0)
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
      if (r8 >= 4 && !two) continue;             // (a `break` on a run-time bound keeps hipcc from unrolling: a2f[t] would go to scratch)
      const int t = r8 >> 2, rd = r8 & 3, cur = r8 & 1;
      const long long pb = pos0 + pt[t] * 32;
      if (r8 + 1 < n_rounds) {
        if (cur == 0) { RART_BN_LOAD_C(1, r8 + 1) } else { RART_BN_LOAD_C(0, r8 + 1) }
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        f32x16 acc[2];
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + 128 + (rd * 2 + b2) * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[b2][4 * g + i] = bv[i];
          }
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sW + (rd * 2 + b2) * 4096 + woff[s]);
            acc[b2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, a2f[t][s], acc[b2], 0, 0, 0);
          }
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = {acc[b2][4 * g], acc[b2][4 * g + 1], acc[b2][4 * g + 2], acc[b2][4 * g + 3]};
            *reinterpret_cast<f32x4*>(sE + p32 * BF_LDE + b2 * 32 + 8 * g + 4 * h) = v;
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int r = qd * 8 + rw;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sE + r * BF_LDE + cw * 8);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(sE + r * BF_LDE + cw * 8 + 4);
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          const uint32_t rr[4] = {rv[cur][qd][0], rv[cur][qd][1], rv[cur][qd][2], rv[cur][qd][3]};
          uint32_t o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[2 * j] += __uint_as_float(rr[j] << 16);
            v[2 * j + 1] += __uint_as_float(rr[j] & 0xFFFF0000u);
            o[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
            if (BWD) o[j] &= halves_from_bits(mb[cur][qd], j);
            else o[j] = relu_bf16x2(o[j]);
          }
          const long long e = (pb + r) * 256 + rd * 64 + cw * 8;
          RART_LAB_STORE16(d.out + e, make_uint4(o[0], o[1], o[2], o[3]));
          if (!BWD && d.m3) d.m3[e >> 3] = (uint8_t)sign_byte(make_uint4(o[0], o[1], o[2], o[3]));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
#undef RART_BN_LOAD_C
  }
#ifdef RART_BNECK_TS
  RART_STAMP(6)
  if (d.ts && lane == 0)
    for (int k = 0; k < 7; ++k) d.ts[((size_t)blockIdx.x * 4 + wave) * 7 + k] = ts_[k];
#endif
#undef RART_STAMP
}
}  // namespace

// 1 if rart_bottleneck_fused_bf16 runs this block geometry
extern "C" int rart_bottleneck_fused_supported(int c_io, int c_mid, int h, int w) {
  return (c_io == 256 && c_mid == 64 && h == BF_H && w == BF_W) ? 1 : 0;
}
// 1 if rart_bottleneck_first_bf16 runs this block geometry
extern "C" int rart_bottleneck_first_supported(int c_in, int c_mid, int c_out, int h, int w) {
  return (c_in == 64 && c_mid == 64 && c_out == 256 && h == BF_H && w == BF_W) ? 1 : 0;
}

static int bneck_launch(const void* x, const void* w1, const void* w2, const void* w3, const void* w4, const float* b1,
                        const float* b2, const float* b3, void* m1, void* m2, void* m3, void* out, int n, const int* tap_dy,
                        const int* tap_dx, int backward, int first, rart_stream_t stream, const char* who) {
  RART_CHECK_ARG(x && w1 && w2 && w3 && out && tap_dy && tap_dx && n > 0 && (!first || w4), "%s: bad arguments", who);
  RART_CHECK_ARG(x != out, "%s: out must not alias x (neighbouring tiles read each other's halo rows)", who);
  RART_CHECK_ARG(!backward || (m1 && m2), "%s: the backward pass needs both inner masks", who);
  RART_CHECK_ARG((long long)n * BF_H * BF_W * 256 < (1ll << 31), "%s: tensor must stay below 2^31 elements", who);
  RartBneckDesc d;
  d.x = (const uint16_t*)x; d.w1 = (const uint16_t*)w1; d.w2 = (const uint16_t*)w2; d.w3 = (const uint16_t*)w3;
  d.w4 = (const uint16_t*)w4;
  d.b1 = b1; d.b2 = b2; d.b3 = b3;
  d.m1 = (uint8_t*)m1; d.m2 = (uint8_t*)m2; d.m3 = (uint8_t*)m3;
  d.out = (uint16_t*)out;
  d.tiles = (uint32_t)n * BF_TPI;
#ifdef RART_BNECK_TS
  d.ts = nullptr;
#endif
  for (int t = 0; t < 9; ++t) {
    RART_CHECK_ARG(tap_dy[t] >= -1 && tap_dy[t] <= 1 && tap_dx[t] >= -1 && tap_dx[t] <= 1, "%s: taps must lie in -1..1", who);
    d.tap_off[t] = (tap_dy[t] * BF_SW + tap_dx[t]) * 16;
  }
  const dim3 grid(d.tiles), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (first && backward) hipLaunchKernelGGL((k_bottleneck56<true, true>), grid, block, 0, st, d);
  else if (first) hipLaunchKernelGGL((k_bottleneck56<false, true>), grid, block, 0, st, d);
  else if (backward) hipLaunchKernelGGL((k_bottleneck56<true, false>), grid, block, 0, st, d);
  else hipLaunchKernelGGL((k_bottleneck56<false, false>), grid, block, 0, st, d);
  RART_CHECK_LAUNCH(who);
  return RART_OK;
}

extern "C" int rart_bottleneck_fused_bf16(const void* x, const void* w1, const void* w2, const void* w3, const float* b1,
                                          const float* b2, const float* b3, void* m1, void* m2, void* m3, void* out, int n,
                                          int h, int w, int c_io, int c_mid, const int* tap_dy, const int* tap_dx, int backward,
                                          rart_stream_t stream) {
  RART_CHECK_ARG(rart_bottleneck_fused_supported(c_io, c_mid, h, w),
                 "rart_bottleneck_fused_bf16: unsupported geometry (256 -> 64 -> 256 channels at 56 x 56 only)");
  return bneck_launch(x, w1, w2, w3, nullptr, b1, b2, b3, m1, m2, m3, out, n, tap_dy, tap_dx, backward, 0, stream,
                      "rart_bottleneck_fused_bf16");
}

extern "C" int rart_bottleneck_first_bf16(const void* x, const void* w1, const void* w2, const void* w3, const void* w4,
                                          const float* b1, const float* b2, const float* b3, void* m1, void* m2, void* m3,
                                          void* out, int n, int h, int w, int c_in, int c_mid, int c_out, const int* tap_dy,
                                          const int* tap_dx, int backward, rart_stream_t stream) {
  RART_CHECK_ARG(rart_bottleneck_first_supported(c_in, c_mid, c_out, h, w),
                 "rart_bottleneck_first_bf16: unsupported geometry (64 -> 64 -> 256 channels at 56 x 56 only)");
  return bneck_launch(x, w1, w2, w3, w4, b1, b2, b3, m1, m2, m3, out, n, tap_dy, tap_dx, backward, 1, stream,
                      "rart_bottleneck_first_bf16");
}
