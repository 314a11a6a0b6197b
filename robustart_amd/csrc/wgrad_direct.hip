// Weight gradient of a convolution straight from the NHWC activations (gfx950): no transposed copies, no materialised im2col.
//
//   partial[z][t * C + c][n] = sum over the output positions m of K split z of  x[pixel(m) + tap t][c] * dz[m][n]        (fp32)
//
// A GEMM whose K dimension is the POSITION: both operands are stored position-major (a row = one pixel's channels), while an MFMA
// operand lane wants 8 consecutive K values of ONE channel.  Round 1-3 made K contiguous with two transposing passes
// (rart_transpose_gather_bf16: dz^T, and the im2col of x transposed -- 9 x the activation for a 3x3) and ran the implicit-GEMM kernel on
// the copies: 8.7 % of an adversarial-training step in the transposes alone (profiles/r03_adv_train_kernel_stats.csv).  Here the tiles
// go global -> LDS as they are ([32 positions][128 channels], global_load_lds_dwordx4, the tap shift and the zero padding in the source
// address) and the fragments come out of LDS TRANSPOSED with gfx950's ds_read_b64_tr_b16: a 16-lane group reads a [4 positions][16
// channels] block (lane a supplies the address of 4 consecutive channels of position a / 4) and lane l receives channel l of the four
// positions -- half of a v_mfma_f32_32x32x16_bf16 operand (scratch/r4/probe_tr16.hip prints the mapping).  The four positions of a block
// are 256 B (or 128 B) apart = the same banks, so the 16-byte chunk c of tile row r is stored at chunk c ^ 4 (r & 3) (256-byte rows;
// c ^ 4 ((r >> 1) & 1) for 128-byte rows): the eight 32-byte segments a 32-lane pass touches are distinct (pre-swizzled SOURCE chunk,
// the same XOR in the read address).
//
// Tile: 128 (tap, channel) rows x 128 (or 64) output channels, four wave64s, two-stage pipeline over 32-position K steps; C = 64 layers
// put two taps in one 128-row tile.  Split-K over blockIdx.z; the fp32 partial sums keep the layout rart_wgrad_reduce_f32 folds.
//
// Reference step: loss.backward() of the adversarial-training loop (RobustART/training/cls_solver.py:183-215 -> torch autograd through
// every convolution; exprs/nips_benchmark/pgd_adv_train/resnet50/config.yaml:1-33).
#include "rart_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {
struct WgradDev {
  const uint16_t* x;      // [batch][ih][iw][C]
  const uint16_t* dz;     // [batch][gh][gw][ldz]
  float* part;            // [splits][kp][ld_n]
  int C, ldz, ld_n, kp, n_valid;
  int ih, iw, gh, gw, sy, sx, n_taps;
  int tap_dy[49], tap_dx[49];
  int M, chunk;           // positions, positions per K split (a multiple of 32)
  uint32_t gw_magic, gw_shift, gh_magic, gh_shift;
};
__device__ __forceinline__ uint32_t wg_fastdiv(uint32_t n, uint32_t magic, uint32_t shift) { return (uint32_t)(((uint64_t)n * magic) >> shift); }
__device__ __attribute__((aligned(16))) const uint32_t g_wg_zero16[4] = {0u, 0u, 0u, 0u};

// transposed fragment: 8 consecutive K (positions k0 .. k0 + 7) of column `col0 + (lane & 15)` of a [32][ROWB bytes] tile
template <int ROWB>
__device__ __forceinline__ bf16x8 wg_frag(const uint8_t* tile, int k0, int col0, int a) {
  constexpr int XS = ROWB >= 256 ? 0 : 1;          // swizzle key: (row >> XS) & (ROWB >= 256 ? 3 : 1)
  constexpr int XM = ROWB >= 256 ? 3 : 1;
  s16x4 v[2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int row = k0 + 4 * rd + (a >> 2);
    const int colb = (col0 + 4 * (a & 3)) * 2;     // byte column of the lane's 4 channels
    const int phys = (((colb >> 4) ^ (4 * ((row >> XS) & XM))) << 4) + (colb & 15);
    v[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + row * ROWB + phys));
  }
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 r = {v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3]};
  return __builtin_bit_cast(bf16x8, r);
}

// STEM: x has 4 channels per pixel (the padded hi plane of the 7x7/2 stem, 3 valid): a tile row = 32 taps x 4 channels, every 8-byte piece of
// it from another pixel -- the x tile is loaded with 4-byte direct loads, one wave-load per position row
template <int TN, int TMR, bool STEM>
__global__ __launch_bounds__(TMR * 2, TMR == 128 ? 5 : 2) void k_wgrad_direct(const WgradDev d) {
  constexpr int NW = TMR / 32;                      // waves: 4 (128 rows) / 8 (256 rows)
  constexpr int WN = TN / 64, WM = NW / WN, RW = TMR / WM, MI = RW / 32;
  constexpr int ROWA = TMR * 2, ROWB = TN * 2;      // bytes of a position's row in the x / dz tile
  constexpr int A_BYTES = 32 * ROWA, B_BYTES = 32 * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024;     // 1 KiB wave-loads per tile
  constexpr int AI = A_PIECES / NW, BI = (B_PIECES + NW - 1) / NW;       // ... per wave (2; 2 / 1)
  constexpr int CPRA = ROWA / 16, RPLA = 64 / CPRA;                       // 16-byte chunks per x row; rows per wave-load
  constexpr int CPRB = ROWB / 16, RPLB = 64 / CPRB;
  static_assert(AI == 2, "two x pieces per wave");
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
  const int rt = blockIdx.x, n0 = blockIdx.y * TN, z = blockIdx.z;
  // the TMR rows of this tile: one tap x TMR channels, or (C < TMR) TMR / C taps x C channels
  const int per_tap = d.C >= TMR ? d.C / TMR : 1;
  const int tpt = d.C >= TMR ? 1 : TMR / d.C;       // taps per tile
  const int t0 = d.C >= TMR ? rt / per_tap : rt * tpt;
  const int c0 = d.C >= TMR ? (rt - t0 * per_tap) * TMR : 0;
  const int t0s = rt * (TMR / 4);                   // STEM: 32 taps per 128-row tile
  const int cpt = STEM ? 1 : (d.C >= TMR ? TMR : d.C) / 8;     // 16-byte chunks of one tap inside a tile row
  const int m_begin = z * d.chunk, m_end = min(m_begin + d.chunk, d.M);
  const int steps = (m_end - m_begin + 31) / 32;

  // ---- loader: wave w brings 1 KiB pieces AI*w .. of the x tile (RPLA rows each) and BI pieces of the dz tile; the lane's LDS slot holds
  //      the row's chunk (slot ^ 4 (row & 3))
  int a_tap[AI], a_coff[AI], a_row[AI];
#pragma unroll
  for (int q = 0; q < AI; ++q) {
    a_row[q] = RPLA * (AI * wave + q) + lane / CPRA;
    const int logical = (lane % CPRA) ^ (4 * (a_row[q] & 3));
    a_tap[q] = t0 + logical / cpt;
    a_coff[q] = c0 + (logical % cpt) * 8;
  }
  int b_row[BI], b_coff[BI];
#pragma unroll
  for (int q = 0; q < BI; ++q) {
    b_row[q] = RPLB * (BI * wave + q) + lane / CPRB;
    b_coff[q] = ((lane % CPRB) ^ (4 * (ROWB >= 256 ? (b_row[q] & 3) : ((b_row[q] >> 1) & 1)))) * 8;
  }
  const char* const zsrc = reinterpret_cast<const char*>(g_wg_zero16);
#define RART_WG_DL(SRC, DST)                                                                                    \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),                        \
                                   (__attribute__((address_space(3))) void*)(DST), 16, 0, 0);
#define RART_WG_ISSUE(STEP, BUF)                                                                                \
  {                                                                                                             \
    uint8_t* const st_ = lds + (BUF)*STAGE;                                                                     \
    const int mb_ = m_begin + (STEP)*32;                                                                        \
    if (STEM) {                                                                                                 \
      _Pragma("unroll") for (int q = 0; q < 32 / NW; ++q) {                                                     \
        const int row_ = (32 / NW) * wave + q;                                                                  \
        const int m_ = mb_ + row_;                                                                              \
        const int ld_ = ((((lane >> 2) ^ (4 * (row_ & 3))) << 2) | (lane & 3));       /* logical dword of the lane's slot */ \
        const int tap_ = t0s + (ld_ >> 1);                                                                       \
        const char* src_ = zsrc;                                                                                \
        if (m_ < m_end && tap_ < d.n_taps) {                                                                    \
          const uint32_t t_ = wg_fastdiv((uint32_t)m_, d.gw_magic, d.gw_shift);                                 \
          const int ox_ = (int)((uint32_t)m_ - t_ * (uint32_t)d.gw);                                            \
          const int im_ = (int)wg_fastdiv(t_, d.gh_magic, d.gh_shift);                                          \
          const int oy_ = (int)(t_ - (uint32_t)im_ * (uint32_t)d.gh);                                           \
          const int iy_ = oy_ * d.sy + d.tap_dy[tap_], ix_ = ox_ * d.sx + d.tap_dx[tap_];                       \
          if ((unsigned)iy_ < (unsigned)d.ih && (unsigned)ix_ < (unsigned)d.iw)                                 \
            src_ = reinterpret_cast<const char*>(d.x + ((size_t)(im_ * d.ih + iy_) * d.iw + ix_) * 4 + (ld_ & 1) * 2); \
        }                                                                                                       \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_),                 \
                                         (__attribute__((address_space(3))) void*)(st_ + row_ * ROWA), 4, 0, 0); \
      }                                                                                                         \
    } else                                                                                                      \
    _Pragma("unroll") for (int q = 0; q < AI; ++q) {                                                            \
      const int m_ = mb_ + a_row[q];                                                                            \
      const char* src_ = zsrc;                                                                                  \
      if (m_ < m_end && a_tap[q] < d.n_taps) {                                                                  \
        const uint32_t t_ = wg_fastdiv((uint32_t)m_, d.gw_magic, d.gw_shift);                                   \
        const int ox_ = (int)((uint32_t)m_ - t_ * (uint32_t)d.gw);                                              \
        const int im_ = (int)wg_fastdiv(t_, d.gh_magic, d.gh_shift);                                            \
        const int oy_ = (int)(t_ - (uint32_t)im_ * (uint32_t)d.gh);                                             \
        const int iy_ = oy_ * d.sy + d.tap_dy[a_tap[q]], ix_ = ox_ * d.sx + d.tap_dx[a_tap[q]];                 \
        if ((unsigned)iy_ < (unsigned)d.ih && (unsigned)ix_ < (unsigned)d.iw)                                   \
          src_ = reinterpret_cast<const char*>(d.x + ((size_t)(im_ * d.ih + iy_) * d.iw + ix_) * d.C + a_coff[q]); \
      }                                                                                                         \
      RART_WG_DL(src_, st_ + (AI * wave + q) * 1024)                                                             \
    }                                                                                                           \
    _Pragma("unroll") for (int q = 0; q < BI; ++q) {                                                            \
      if (BI * wave + q < B_PIECES) {                                                                           \
        const int m_ = mb_ + b_row[q];                                                                          \
        const char* src_ = zsrc;                                                                                \
        if (m_ < m_end && n0 + b_coff[q] < d.ldz) src_ = reinterpret_cast<const char*>(d.dz + (size_t)m_ * d.ldz + n0 + b_coff[q]); \
        RART_WG_DL(src_, st_ + A_BYTES + (BI * wave + q) * 1024)                                                 \
      }                                                                                                         \
    }                                                                                                           \
  }
  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int g = lane >> 4, a = lane & 15;
  if (steps > 0) {
    RART_WG_ISSUE(0, 0)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
      const int buf = s & 1;
      if (s + 1 < steps) RART_WG_ISSUE(s + 1, buf ^ 1)
      const uint8_t* At = lds + buf * STAGE;
      const uint8_t* Bt = At + A_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int k0 = ks * 16 + 8 * (g >> 1);
        bf16x8 af[MI], bfr[2];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = wg_frag<ROWA>(At, k0, wm * RW + i * 32 + 16 * (g & 1), a);
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[j] = wg_frag<ROWB>(Bt, k0, wn * 64 + j * 32 + 16 * (g & 1), a);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
    }
  }
#undef RART_WG_ISSUE
#undef RART_WG_DL
  // ---- fp32 partial sums: acc[i][j][r] = row wm*RW + i*32 + (r&3) + 8*(r>>2) + 4h, column wn*64 + j*32 + (lane & 31)
  const int fr = lane & 31, h = lane >> 5;
  const int row_base = STEM ? t0s * 4 : t0 * d.C + c0;
  float* const pz = d.part + (size_t)z * d.kp * d.ld_n;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + fr;
      if (col >= d.ld_n) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_base + wm * RW + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < d.kp) pz[(size_t)row * d.ld_n + col] = acc[i][j][r];
      }
    }
}

void wg_magic(uint32_t dv, uint32_t& mg, uint32_t& sh) {   // exact for dividends < 2^31
  uint32_t l = 0;
  while ((1ull << l) < dv) ++l;
  sh = 31 + l;
  mg = (uint32_t)(((1ull << sh) + dv - 1) / dv);
}
}  // namespace

extern "C" int rart_wgrad_direct_supported(int channels, int n_cols, int n_taps) {
  if (channels == 4) return (n_cols == 64 && n_taps >= 1 && n_taps <= 49) ? 1 : 0;          // the stem: 4-channel pixels (3 valid), 64 filters
  return ((channels == 64 || (channels >= 128 && channels % 128 == 0)) && n_cols % 64 == 0 && n_cols >= 64 && n_taps >= 1 && n_taps <= 9) ? 1 : 0;
}

extern "C" int rart_wgrad_direct_bf16(const void* x, const void* dz, float* partial, int batch, int in_h, int in_w, int channels, int grid_h,
                                      int grid_w, int dz_cols, int stride_y, int stride_x, int n_taps, const int* tap_dy, const int* tap_dx,
                                      int splits, int chunk, int ld_n, rart_stream_t stream) {
  RART_CHECK_ARG(x && dz && partial && batch > 0 && in_h > 0 && in_w > 0 && grid_h > 0 && grid_w > 0 && tap_dy && tap_dx,
                 "rart_wgrad_direct_bf16: bad arguments");
  RART_CHECK_ARG(rart_wgrad_direct_supported(channels, dz_cols, n_taps),
                 "rart_wgrad_direct_bf16: channels must be 64 or a multiple of 128 (dz_cols a multiple of 64, 1..9 taps), or 4 (dz_cols 64, 1..49 taps)");
  const long long M = (long long)batch * grid_h * grid_w;
  RART_CHECK_ARG(M < (1ll << 31) && (long long)batch * in_h * in_w * channels < (1ll << 31) && M * dz_cols < (1ll << 31),
                 "rart_wgrad_direct_bf16: tensors must stay below 2^31 elements");
  RART_CHECK_ARG(splits >= 1 && splits <= 65535 && chunk >= 32 && chunk % 32 == 0 && (long long)splits * chunk >= M,
                 "rart_wgrad_direct_bf16: chunk must be a multiple of 32 and splits * chunk must cover the positions");
  RART_CHECK_ARG(ld_n >= 8 && ld_n % 8 == 0 && ld_n <= dz_cols + 7, "rart_wgrad_direct_bf16: ld_n = the output columns rounded up to 8");
  WgradDev d;
  d.x = (const uint16_t*)x; d.dz = (const uint16_t*)dz; d.part = partial;
  d.C = channels; d.ldz = dz_cols; d.ld_n = ld_n; d.kp = n_taps * channels; d.n_valid = ld_n;
  d.ih = in_h; d.iw = in_w; d.gh = grid_h; d.gw = grid_w; d.sy = stride_y; d.sx = stride_x; d.n_taps = n_taps;
  for (int i = 0; i < 49; ++i) { d.tap_dy[i] = i < n_taps ? tap_dy[i] : 0; d.tap_dx[i] = i < n_taps ? tap_dx[i] : 0; }
  d.M = (int)M; d.chunk = chunk;
  wg_magic((uint32_t)grid_w, d.gw_magic, d.gw_shift);
  wg_magic((uint32_t)grid_h, d.gh_magic, d.gh_shift);
  // Tile height: the kernel is written for TMR = 128 or 256 rows.  256-row tiles (8 waves, 0.375 KB of LDS traffic per MFMA instead of 0.5,
  // half the split-K partials at equal workgroup count) were measured on every conv shape of ResNet-50 at B = 256 and LOSE: 7.04 vs 6.14 ms of
  // weight gradients per step (adv_train 4.82 vs 4.90 k images/s): three 8-wave workgroups per CU hide the K step's load latency worse than
  // five 4-wave ones.  128 rows it is.
  constexpr int tmr = 128;
  hipStream_t st = (hipStream_t)stream;
  if (channels == 4) {
    hipLaunchKernelGGL((k_wgrad_direct<64, 128, true>), dim3((n_taps + 31) / 32, 1, splits), dim3(256), 0, st, d);
    RART_CHECK_LAUNCH("rart_wgrad_direct_bf16");
    return RART_OK;
  }
  const int row_tiles = channels >= tmr ? n_taps * (channels / tmr) : (n_taps + tmr / channels - 1) / (tmr / channels);
  if (dz_cols % 128 == 0) hipLaunchKernelGGL((k_wgrad_direct<128, 128, false>), dim3(row_tiles, dz_cols / 128, splits), dim3(256), 0, st, d);
  else hipLaunchKernelGGL((k_wgrad_direct<64, 128, false>), dim3(row_tiles, dz_cols / 64, splits), dim3(256), 0, st, d);
  RART_CHECK_LAUNCH("rart_wgrad_direct_bf16");
  return RART_OK;
}
