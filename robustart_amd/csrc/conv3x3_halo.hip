// 3x3 stride-1 "same" convolution with the input tile RESIDENT in LDS (gfx950) -- the conv2 layers of ResNet-50's
// layer1 / layer2 (C = 64 at 56x56, C = 128 at 28x28), forward and backward-to-input (flipped taps, transposed weights).
//
// Why a second conv kernel: the implicit GEMM (conv_igemm.hip) gathers the A operand from global memory once per tap, so
// a 3x3 layer with few channels issues 9x the load / ds_write instructions its unique bytes need and sits at 18-24 % of
// its roofline floor (profiles/r02_igemm_per_shape.txt: M = 802 816, K = 576, N = 64: 141 us vs a 26 us floor).  Here a
// workgroup owns POS consecutive output positions of the batch viewed as ONE tall image ([n*h] rows x w, raster order, so
// there are no partial tiles whatever h and w are), stages the (rows + 2) x (w + 2) halo of the input once -- position
// major, 16-byte chunks XOR-swizzled by the position so that the MFMA fragment reads of 16 consecutive positions are
// bank-conflict free -- and walks the 9 taps x C/64 K-steps from LDS; only the small weight slice of a step (C x 64)
// streams through LDS.  Rows of a neighbouring image inside the halo and the top / bottom padding are handled per lane
// (the fragment read is redirected to a zero slot), the left / right padding by zero columns of the halo.
//
// Reference step: conv2 of every Bottleneck of the public ResNet-50 (RobustART/model/__init__.py:1 -> absent submodule;
// robustart_amd/model/resnet_torch.py) inside the forward / autograd of each attack iteration
// (RobustART/noise/utils/adv/attack.py:21-22, Attacks/autoattack/autopgd_base.py:271-289).
#include "rart_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct RartHaloDesc {
  const uint16_t* src;
  const uint16_t* wgt;      // fragment-major table of the [C][9 * C] weights (rart_conv3x3_pack_frag_bf16)
  const float* bias;        // fp32 [C] or null
  const uint8_t* mask_bits; // 1 bit per output element or null
  uint8_t* sign_out;        // 1 bit per output element or null
  uint16_t* dst;
  int rows_total, h, w;     // rows_total = n * h
  int rows_per_block;       // whole image rows a workgroup owns (rows_per_block * w <= POS)
  int tap_dy[9], tap_dx[9];
  int relu;
  uint32_t w_magic, w_shift, h_magic, h_shift, w2_magic, w2_shift;
#ifdef RART_HALO_TS
  unsigned long long* ts;   // scratch/exp/halo_ts.hip: cycle stamps per workgroup phase
#endif
};

namespace {
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, uint32_t magic, uint32_t shift) {
  return (uint32_t)(((uint64_t)n * magic) >> shift);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) short i16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t w) {
  const i16x2_t z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w), z));
}
__device__ __forceinline__ uint32_t halves_from_bits(uint32_t byte, int pair) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe(byte, 2 * pair, 1);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe(byte, 2 * pair + 1, 1);
  return __builtin_amdgcn_perm(hi, lo, 0x07060100u);
}
__device__ __forceinline__ uint32_t bits_from_halves(uint32_t w) {
  const i16x2_t z = {0, 0}, one = {1, 1};
  const uint32_t t = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w), z), one));
  return (t | (t >> 15)) & 3u;
}

template <int C>
struct HaloCfg {
  static constexpr int POS = C == 64 ? 256 : 128;        // output positions per workgroup
  static constexpr int CP = C / 8;                        // 16-byte chunks per position
  static constexpr int CPB = C * 2;                       // bytes per position
  static constexpr int MAXHALO = C == 64 ? 416 : 208;     // halo positions the LDS tile can hold (52 KB: 3 workgroups / CU)
  static constexpr int WAVES_N = C / 32;                  // waves along the channel dimension (32 columns each): 2 / 4
  static constexpr int WAVES_M = 4 / WAVES_N;             // waves along the positions (128 each): 2 / 1
  // LDS tile, chunk major: [CP planes][PLANE bytes], plane c holds 16-byte chunk c of every halo position.  A fragment read
  // (32 consecutive positions, one chunk) is a contiguous run: conflict free, and the chunk becomes an IMMEDIATE offset of
  // ds_read_b128 (no per-read address arithmetic: the first version's XOR swizzle made the K loop VALU-issue bound).
  // PLANE = 16 mod 256 so the 8 chunks a staging wave-quarter writes for one position fall in 8 different bank slots.
  static constexpr int PLANE = (MAXHALO + 1) * 16;
  static_assert(PLANE % 256 == 16, "plane stride must be 16 mod 256");
  static constexpr int HALO_BYTES = CP * PLANE;
  static constexpr int LDS_BYTES = HALO_BYTES;
  static constexpr int KSTEPS = C / 64;                   // 64-channel K steps per tap
  static constexpr int STEPS = 9 * KSTEPS;
};

// Wave tile: 128 positions (4 M tiles of 32) x 32 output channels (one N tile), so a wave needs only a 32 x 64 weight slice
// per step: 4 fragments, loaded straight from global memory (L1 / L2 resident, 73 / 295 KB per layer) one step ahead.
// With the activations resident in LDS there is NO barrier in the K loop: waves run the 9 taps independently.
template <int C>
__global__ __launch_bounds__(256, 3) void k_conv3x3_halo(const RartHaloDesc d) {
  using Cf = HaloCfg<C>;
  __shared__ __attribute__((aligned(16))) uint8_t lds[Cf::LDS_BYTES];
#ifdef RART_HALO_TS
  unsigned long long ts_[6];
  ts_[0] = __builtin_readcyclecounter();
#define RART_STAMP(I) ts_[I] = __builtin_readcyclecounter();
#else
#define RART_STAMP(I)
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cf::WAVES_N, wn = wave % Cf::WAVES_N;
  const uint32_t W = (uint32_t)d.w, W2 = W + 2u;
  // consecutive runs of positions share halo rows: keep neighbours on one XCD (its L2 serves the overlap)
  uint32_t blk;
  {
    const uint32_t nb = gridDim.x, bid = blockIdx.x, xcd = bid & 7u, slot = bid >> 3, q = nb >> 3, r = nb & 7u;
    blk = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + slot;
  }
  // a workgroup owns rows_per_block whole rows of the tall image: halo rows row0-1 .. row0+rows
  const uint32_t row0 = blk * (uint32_t)d.rows_per_block;
  const uint32_t n_rows = min((uint32_t)d.rows_per_block, (uint32_t)d.rows_total - row0);
  const uint32_t g0 = row0 * W;
  const uint32_t g_end = g0 + n_rows * W;
  const uint32_t n_hrows = n_rows + 2u;

  // ---- weight fragments of step 0 in flight first: lane -> output channel wn*32 + (lane & 31), k half (lane >> 5)
  //      (fragment-major table, rart_conv3x3_pack_frag_bf16: the 64 lanes of a fragment load read 1 KiB contiguous; with
  //      row-major weights every lane sat on its own cache line and the K loop was bound by the texture-address unit)
  const uint16_t* wp = d.wgt + wn * 2048 + lane * 8;
  constexpr int WST = Cf::WAVES_N * 2048;         // elements per K step: WAVES_N x 4 fragments x 512
  bf16x8 bq[2][4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) bq[0][ks] = *reinterpret_cast<const bf16x8*>(wp + ks * 512);

  {
    // halo tile: every load of the thread (<= 13 chunks of 16 bytes) in flight before the first LDS store
    constexpr int U = (Cf::MAXHALO * Cf::CP + 255) / 256;
    const uint32_t n_items = n_hrows * W2 * Cf::CP;
    uint4 v[U];
    uint32_t dstoff[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t idx = u * 256 + tid;
      const uint32_t chunk = idx & (Cf::CP - 1), hp = idx / Cf::CP;
      const uint32_t hr = fastdiv(hp, d.w2_magic, d.w2_shift);
      const int x = (int)(hp - hr * W2) - 1;
      const int R = (int)(row0 + hr) - 1;
      v[u] = make_uint4(0, 0, 0, 0);
      dstoff[u] = idx < n_items ? chunk * Cf::PLANE + hp * 16 : 0xFFFFFFFFu;
      if (idx < n_items && (unsigned)x < W && (unsigned)R < (unsigned)d.rows_total)
        v[u] = *reinterpret_cast<const uint4*>(d.src + ((size_t)R * W + x) * C + chunk * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (dstoff[u] != 0xFFFFFFFFu) *reinterpret_cast<uint4*>(lds + dstoff[u]) = v[u];
  }
  RART_STAMP(1)
  // ---- per-lane geometry of the wave's four 32-position M tiles
  uint32_t hp_base[4];      // halo position of the output position itself
  int ypos[4];              // y inside its image, or a large negative value for positions past the end
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t g = g0 + wm * 128 + i * 32 + (lane & 31);
    const uint32_t gg = g < g_end ? g : (g_end - 1u);
    const uint32_t R = fastdiv(gg, d.w_magic, d.w_shift);
    const uint32_t x = gg - R * W;
    const uint32_t img = fastdiv(R, d.h_magic, d.h_shift);
    ypos[i] = g < g_end ? (int)(R - img * (uint32_t)d.h) : -(1 << 20);
    hp_base[i] = (R - row0 + 1u) * W2 + x + 1u;
  }
  f32x16 acc[4];
  {
    const float bv = d.bias ? d.bias[wn * 32 + (lane & 31)] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = bv;
  }
  __syncthreads();
  RART_STAMP(2)

  const uint32_t kq = (uint32_t)(lane >> 5);          // which 8-k half (= which chunk of a pair) of a 16-k MFMA step this lane supplies
#pragma unroll
  for (int st = 0; st < Cf::STEPS; ++st) {
    const int tap = st / Cf::KSTEPS, kh = st % Cf::KSTEPS;
    if (st + 1 < Cf::STEPS) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bq[(st + 1) & 1][ks] = *reinterpret_cast<const bf16x8*>(wp + (st + 1) * WST + ks * 512);
    }
    // pin the prefetch at the top of the step: left alone, hipcc sinks these loads behind the step's MFMAs and then waits
    // vmcnt(0) at the next step's first MFMA -- an exposed L2 round trip per tap (measured: 138 cycles per MFMA)
    __builtin_amdgcn_sched_barrier(0);
    const int dy = d.tap_dy[tap], dx = d.tap_dx[tap];
    // halo position 0 is the top-left padding corner (x = -1): always zero, so rows of another image / beyond the batch
    // are redirected there and need no masking
    uint32_t abase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = (unsigned)(ypos[i] + dy) < (unsigned)d.h;
      const uint32_t hp = hp_base[i] + (uint32_t)(dy * (int)W2 + dx);
      abase[i] = (ok ? hp : 0u) * 16u + kq * (uint32_t)Cf::PLANE;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(lds + abase[i] + (kh * 8 + ks * 2) * Cf::PLANE);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bq[st & 1][ks], acc[i], 0, 0, 0);
    }
  }
  RART_STAMP(3)
  __syncthreads();        // every wave is done reading the halo tile: its memory becomes the epilogue staging

  // ---- epilogue: each wave transposes its 128 x 32 sub-tile through a private LDS slice, 32 rows at a time, and stores
  //      64-byte row segments (16 rows per wave-wide access)
  constexpr int LDW = 32 + 4;
  static_assert(4 * 32 * LDW * 4 <= Cf::HALO_BYTES, "epilogue staging must fit the halo tile");
  float* sE = reinterpret_cast<float*>(lds) + wave * 32 * LDW;
  const int cw = lane & 3, rw0 = lane >> 2;
  const int col = wn * 32 + cw * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t mb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t g = g0 + wm * 128 + i * 32 + q * 16 + rw0;
      mb[q] = 0xFFu;
      if (d.mask_bits && g < g_end) mb[q] = d.mask_bits[(g * (uint32_t)C + (uint32_t)col) >> 3];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      sE[row * LDW + (lane & 31)] = acc[i][r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = q * 16 + rw0;
      const uint32_t g = g0 + wm * 128 + i * 32 + r;
      const float4 v0 = *reinterpret_cast<const float4*>(sE + r * LDW + cw * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(sE + r * LDW + cw * 8 + 4);
      if (g < g_end) {
        uint32_t o[4] = {pack_bf16x2(v0.x, v0.y), pack_bf16x2(v0.z, v0.w), pack_bf16x2(v1.x, v1.y), pack_bf16x2(v1.z, v1.w)};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (d.mask_bits) o[t] &= halves_from_bits(mb[q], t);
          if (d.relu) o[t] = relu_bf16x2(o[t]);
        }
        const uint32_t e = g * (uint32_t)C + (uint32_t)col;
        *reinterpret_cast<uint4*>(d.dst + e) = make_uint4(o[0], o[1], o[2], o[3]);
        if (d.sign_out)
          d.sign_out[e >> 3] = (uint8_t)(bits_from_halves(o[0]) | (bits_from_halves(o[1]) << 2) | (bits_from_halves(o[2]) << 4) |
                                         (bits_from_halves(o[3]) << 6));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#ifdef RART_HALO_TS
  RART_STAMP(4)
  if (d.ts && lane == 0)
    for (int k = 0; k < 5; ++k) d.ts[((size_t)blockIdx.x * 4 + wave) * 5 + k] = ts_[k];
#endif
#undef RART_STAMP
}

// ---- layer3-shaped variant: C = 256, images of at most 224 positions (14 x 14): ONE IMAGE PER WORKGROUP --------------------
// The whole zero-ringed image (16 x 16 halo positions x 512 B = 131 KB, chunk-major planes, row stride 16) is resident in LDS,
// an M tile is two image rows x 16 column slots (so the 32 fragment reads of a tile are two contiguous 256-byte runs: bank
// conflict free -- 32 consecutive raster positions of a 14-wide image were 2-way conflicted, SQ_LDS_BANK_CONFLICT 44 %), no tap
// needs masking and at B = 256 the grid is exactly one workgroup per CU (the implicit GEMM runs these launches as 784 tiles on 512
// slots: a 53 %-full second round, 668 TFLOP/s).  8 waves: wave n owns ALL 7 M tiles x 32 output channels (112 accumulator
// registers), streams its 32 x 64 weight slice per step from L2 one step ahead and never meets a barrier in the 36-step K loop.
constexpr int IM_C = 256, IM_CP = 32, IM_MAXPOS = 256, IM_MT = 7, IM_T = 512;
constexpr int IM_PLANE = (IM_MAXPOS + 1) * 16;
constexpr int IM_LDS = IM_CP * IM_PLANE;
static_assert(IM_PLANE % 256 == 16, "plane stride must be 16 mod 256");

__global__ __launch_bounds__(IM_T, 1) void k_conv3x3_image256(const RartHaloDesc d) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[IM_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int H = d.h, W = d.w, NP = H * W;
  constexpr int W2 = 16;                                   // halo row stride (w <= 14)
  const size_t img_base = (size_t)blockIdx.x * NP;
  const uint16_t* wp = d.wgt + wn * 2048 + lane * 8;      // fragment-major table (see k_conv3x3_halo)
  constexpr int WST = (IM_C / 32) * 2048;
  bf16x8 bq[2][4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) bq[0][ks] = *reinterpret_cast<const bf16x8*>(wp + ks * 512);
  {
    // the (h + 2) x (w + 2) ring-padded image, all 16 loads of a thread in flight before the first LDS store
    constexpr int U = IM_MAXPOS * IM_CP / IM_T;          // 16
    const int n_items = (H + 2) * W2 * IM_CP;              // halo rows 0 .. H+1, 16 slots each
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = u * IM_T + tid;
      const int chunk = idx & (IM_CP - 1), hp = idx / IM_CP;
      const int hy = hp >> 4, hx = hp & 15;
      v[u] = make_uint4(0, 0, 0, 0);
      if (idx < n_items && hy >= 1 && hy <= H && hx >= 1 && hx <= W)
        v[u] = *reinterpret_cast<const uint4*>(d.src + (img_base + (size_t)(hy - 1) * W + (hx - 1)) * IM_C + chunk * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = u * IM_T + tid;
      if (idx < n_items) *reinterpret_cast<uint4*>(lds + (idx & (IM_CP - 1)) * IM_PLANE + (idx / IM_CP) * 16) = v[u];
    }
  }
  const uint32_t kq = (uint32_t)(lane >> 5);
  uint32_t abase[IM_MT];
#pragma unroll
  for (int i = 0; i < IM_MT; ++i) {
    const int y = min(2 * i + ((lane >> 4) & 1), H - 1), x = min(lane & 15, W - 1);   // slots past the image: clamped, never stored
    abase[i] = (uint32_t)(((y + 1) * W2 + x + 1) * 16) + kq * (uint32_t)IM_PLANE;
  }
  f32x16 acc[IM_MT];
  {
    const float bv = d.bias ? d.bias[wn * 32 + (lane & 31)] : 0.f;
#pragma unroll
    for (int i = 0; i < IM_MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = bv;
  }
  __syncthreads();
#pragma unroll
  for (int st = 0; st < 36; ++st) {
    const int tap = st >> 2, kh = st & 3;
    if (st + 1 < 36) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bq[(st + 1) & 1][ks] = *reinterpret_cast<const bf16x8*>(wp + (st + 1) * WST + ks * 512);
    }
    __builtin_amdgcn_sched_barrier(0);                    // keep the prefetch ahead of this step's MFMAs (see k_conv3x3_halo)
    const int toff = (d.tap_dy[tap] * W2 + d.tap_dx[tap]) * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[IM_MT];
#pragma unroll
      for (int i = 0; i < IM_MT; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(lds + (int)abase[i] + toff + (kh * 8 + ks * 2) * IM_PLANE);
#pragma unroll
      for (int i = 0; i < IM_MT; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bq[st & 1][ks], acc[i], 0, 0, 0);
    }
  }
  __syncthreads();        // the image is dead: its memory becomes the epilogue staging (8 waves x 32 rows x 36 floats)
  constexpr int LDW = 32 + 4;
  static_assert(8 * 32 * LDW * 4 <= IM_LDS, "epilogue staging must fit");
  float* sE = reinterpret_cast<float*>(lds) + wn * 32 * LDW;
  const int cw = lane & 3, rw0 = lane >> 2;
  const int col = wn * 32 + cw * 8;
#pragma unroll
  for (int i = 0; i < IM_MT; ++i) {
    if (2 * i >= H) break;                                // block-uniform
    uint32_t mb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int y = 2 * i + q, x = rw0;                   // staged row r = q * 16 + rw0 <-> (image row 2i + q, column rw0)
      mb[q] = 0xFFu;
      if (d.mask_bits && y < H && x < W) mb[q] = d.mask_bits[((img_base + (size_t)y * W + x) * IM_C + col) >> 3];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      sE[row * LDW + (lane & 31)] = acc[i][r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = q * 16 + rw0;
      const int y = 2 * i + q, x = rw0;
      const float4 v0 = *reinterpret_cast<const float4*>(sE + r * LDW + cw * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(sE + r * LDW + cw * 8 + 4);
      if (y < H && x < W) {
        uint32_t o[4] = {pack_bf16x2(v0.x, v0.y), pack_bf16x2(v0.z, v0.w), pack_bf16x2(v1.x, v1.y), pack_bf16x2(v1.z, v1.w)};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (d.mask_bits) o[t] &= halves_from_bits(mb[q], t);
          if (d.relu) o[t] = relu_bf16x2(o[t]);
        }
        const size_t e = (img_base + (size_t)y * W + x) * IM_C + col;
        *reinterpret_cast<uint4*>(d.dst + e) = make_uint4(o[0], o[1], o[2], o[3]);
        if (d.sign_out)
          d.sign_out[e >> 3] = (uint8_t)(bits_from_halves(o[0]) | (bits_from_halves(o[1]) << 2) | (bits_from_halves(o[2]) << 4) |
                                         (bits_from_halves(o[3]) << 6));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// w [rows][k] row-major -> fragment-major: element e of lane l of fragment (K step st of 64, row tile wn, ks) =
// w[wn*32 + (l & 31)][st*64 + ks*16 + (l >> 5)*8 + e]; fragment index (st * rows/32 + wn) * 4 + ks, 512 elements each
__global__ void k_pack_frag(const uint16_t* __restrict__ w, uint16_t* __restrict__ o, int rows, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // one 16-byte chunk per thread
  if (i >= rows * k / 8) return;
  const int nwn = rows / 32;
  const int l = i & 63, ks = (i >> 6) & 3, f = i >> 8, wn = f % nwn, st = f / nwn;
  *reinterpret_cast<uint4*>(o + (size_t)i * 8) =
      *reinterpret_cast<const uint4*>(w + (size_t)(wn * 32 + (l & 31)) * k + st * 64 + ks * 16 + (l >> 5) * 8);
}

void magic_for(uint32_t dv, uint32_t& mg, uint32_t& sh) {      // exact for dividends < 2^31
  uint32_t l = 0;
  while ((1ull << l) < dv) ++l;
  sh = 31 + l;
  mg = (uint32_t)(((1ull << sh) + dv - 1) / dv);
}
}  // namespace

// whole rows per workgroup: as many as fit both the position budget and the LDS halo tile; 0 = geometry unsupported
static int halo_rows_per_block(int channels, int w) {
  if (channels != 64 && channels != 128) return 0;
  const int pos = channels == 64 ? 256 : 128, maxhalo = channels == 64 ? 416 : 208;
  int rpb = pos / w;
  while (rpb > 0 && (rpb + 2) * (w + 2) > maxhalo) --rpb;
  return rpb;
}

// 1 if rart_conv3x3_halo_bf16 can run this geometry (at least one image row per workgroup fits its LDS tile)
extern "C" int rart_conv3x3_halo_supported(int channels, int h, int w) {
  if (h < 1 || w < 1) return 0;
  if (channels == IM_C) return (h <= 2 * IM_MT && w <= 14) ? 1 : 0;    // one image per workgroup, halo row stride 16
  return halo_rows_per_block(channels, w) > 0 ? 1 : 0;
}

extern "C" int rart_pack_frag_bf16(const void* w_rows, void* w_frag, int rows, int k, rart_stream_t stream) {
  RART_CHECK_ARG(w_rows && w_frag && w_rows != w_frag, "rart_pack_frag_bf16: bad arguments");
  RART_CHECK_ARG(rows > 0 && rows % 32 == 0 && k > 0 && k % 64 == 0, "rart_pack_frag_bf16: rows must be a multiple of 32, k of 64");
  const int chunks = rows * k / 8;
  hipLaunchKernelGGL(k_pack_frag, dim3((chunks + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)w_rows,
                     (uint16_t*)w_frag, rows, k);
  RART_CHECK_LAUNCH("rart_pack_frag_bf16");
  return RART_OK;
}

extern "C" int rart_conv3x3_pack_frag_bf16(const void* w_rows, void* w_frag, int channels, rart_stream_t stream) {
  RART_CHECK_ARG(channels == 64 || channels == 128 || channels == 256, "rart_conv3x3_pack_frag_bf16: channels must be 64, 128 or 256");
  return rart_pack_frag_bf16(w_rows, w_frag, channels, 9 * channels, stream);
}

extern "C" int rart_conv3x3_halo_bf16(const void* src, const void* wgt, const float* bias, const void* mask_bits, void* sign_out,
                                      void* dst, int n, int h, int w, int channels, const int* tap_dy, const int* tap_dx,
                                      int relu, rart_stream_t stream) {
  RART_CHECK_ARG(src && wgt && dst && tap_dy && tap_dx && n > 0, "rart_conv3x3_halo_bf16: bad arguments");
  RART_CHECK_ARG(src != dst, "rart_conv3x3_halo_bf16: dst must not alias src (workgroups read each other's halo rows)");
  RART_CHECK_ARG(rart_conv3x3_halo_supported(channels, h, w), "rart_conv3x3_halo_bf16: unsupported geometry (channels 64 / 128 / 256, the halo tile must fit LDS)");
  const long long G = (long long)n * h * w;
  RART_CHECK_ARG(G * channels < (1ll << 31), "rart_conv3x3_halo_bf16: tensor must stay below 2^31 elements");
  RartHaloDesc d;
  d.src = (const uint16_t*)src; d.wgt = (const uint16_t*)wgt; d.bias = bias; d.mask_bits = (const uint8_t*)mask_bits;
  d.sign_out = (uint8_t*)sign_out; d.dst = (uint16_t*)dst;
  d.rows_total = n * h; d.h = h; d.w = w; d.relu = relu;
  d.rows_per_block = channels == IM_C ? h : halo_rows_per_block(channels, w);
  for (int t = 0; t < 9; ++t) {
    RART_CHECK_ARG(tap_dy[t] >= -1 && tap_dy[t] <= 1 && tap_dx[t] >= -1 && tap_dx[t] <= 1, "rart_conv3x3_halo_bf16: taps must lie in -1..1");
    d.tap_dy[t] = tap_dy[t]; d.tap_dx[t] = tap_dx[t];
  }
  magic_for((uint32_t)w, d.w_magic, d.w_shift);
  magic_for((uint32_t)h, d.h_magic, d.h_shift);
  magic_for((uint32_t)(w + 2), d.w2_magic, d.w2_shift);
  const dim3 grid((uint32_t)((d.rows_total + d.rows_per_block - 1) / d.rows_per_block));
  if (channels == IM_C) {
    hipLaunchKernelGGL(k_conv3x3_image256, dim3(n), dim3(IM_T), 0, (hipStream_t)stream, d);
  } else if (channels == 64) hipLaunchKernelGGL(k_conv3x3_halo<64>, grid, dim3(256), 0, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(k_conv3x3_halo<128>, grid, dim3(256), 0, (hipStream_t)stream, d);
  RART_CHECK_LAUNCH("rart_conv3x3_halo_bf16");
  return RART_OK;
}
