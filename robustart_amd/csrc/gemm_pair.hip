// Split-bf16 ("fp32x" / "bf16x3") GEMM for gfx950 (CDNA4): the dense contraction of the REFERENCE-PRECISION engines.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]      A = A_hi + A_lo, W = W_hi + W_lo (each a plane of bf16 values)
//   A.W ~= A_lo.W_hi + A_hi.W_lo + A_hi.W_hi   (the dropped lo.lo term is 2^-16 of a product), fp32 accumulation
//
// The reference computes fp32 everywhere (RobustART/noise/utils/adv/attack.py:20-23 wraps an fp32 f_model;
// Attacks/autoattack/autopgd_base.py:271-289 takes fp32 logits and gradients; exprs/exp/imagenet_c_loop_mini/config_vit_base.yaml:1-9
// has no precision key) and the north star asks for logits within 1e-4 of it.  gfx950's fp32 MFMA peaks at 157 TFLOP/s, its bf16
// MFMA at 2 500: three bf16 products per algorithmic product are 5 x faster than fp32 operands.
//
// Round 3 ran this scheme on the implicit-GEMM kernel by listing the three products as 3 x the taps: every product stages its own
// A and W tiles (x_hi is fetched and written to LDS twice).  This kernel stages the FOUR operand planes of a K step once and issues
// the three MFMAs per fragment pair from them: 2/3 of the global -> LDS traffic and of the ds_read traffic per MFMA.
//
// Design for MI355X: 256 x 256 tile, 8 wave64s each owning 128 x 64 (4 x 2 fragments of v_mfma_f32_32x32x16_bf16, 128 accumulator
// registers), K step 32: a stage = A_hi | A_lo | W_hi | W_lo, 256 rows x 64 B each = 64 KB, double buffered (128 of the 160 KB);
// every plane goes global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write); the LDS image is lane-linear, so the
// 16-byte chunk c of row r holds the row's chunk c ^ ((r >> 2) & 3) (pre-swizzled SOURCE address, the same involution on the read
// side): ds_read_b128 fragment reads are bank-conflict free in the four 16-lane service groups of scratch/lds_bank_model.py;
// 48 MFMAs per wave between two barriers (the bf16 256 x 256 kernel has 32).  Row tiles of one column tile share an XCD's L2.
// Epilogue per wave through a private LDS region: bias (folded into the accumulator start), residual PAIR, exact GELU / GELU with the
// pre-activation kept / GELU' of a kept pre-activation, then either hi = bf16(v), lo = bf16(v - hi) as two coalesced 16-byte
// stores or fp32.  Rows may be re-based per image on either side (the class-token slot of ViT's patch embedding) and the problem
// may be batched over blockIdx.y with operand planes anywhere in memory (attention products: the "weights" are K / V activations).
// CONV instances gather the A planes on the fly like rart_conv_igemm_bf16 does (row m = (image, oy, ox) of a row grid, k = tap * k_per_tap
// + c, source pixel (oy sy + dy[tap], ox sx + dx[tap]), the zero page outside the image) -- every convolution of the reference-precision
// ResNet-50 (forward taps, flipped taps, the parity classes of a stride-2 backward, the stem's row taps) -- with 1-bit ReLU masks and
// sign bits in the epilogue; column tiles of 64 / 128 / 256 so that the 64- and 128-channel layers do not pad their MFMA work.
#include "rart_common.h"
#include <stdlib.h>

#include "rart_gemm_pair_dev.h"
#include "rart_lds_dma.h"

namespace {
// TM x TN: the tile (rows 256 / 128, columns 256 / 128 / 64), TM / 32 waves (8 / 4) in a (NW / WN) x WN grid, WN = TN / 64; a wave owns
// (TM / WM) rows x 64 columns.  The 256-row tiles are for the K-deep, MFMA-bound products (one workgroup per CU, 48 MFMAs per wave between
// two barriers at TN = 256); the 128-row tiles (48-96 KB of LDS: two or three workgroups per CU whose load / multiply / store phases
// overlap) for the short-K, HBM-bound 1x1 layers and the small-M layers of ResNet-50 whose 256-row tiling would not fill 256 CUs.
template <int TM, int TN, bool CONV>
__global__ __launch_bounds__(TM * 2, 1) void k_gemm_pair(const GemmPairDev d) {
  constexpr int NW = TM / 32;
  constexpr int WN = (TN / 64 < NW) ? TN / 64 : NW, WM = NW / WN, RW = TM / WM, MI = RW / 32, NJ = TN / (32 * WN);
  static_assert(NJ == 2, "a wave owns 64 columns");
  constexpr int GP_PLANE_A = TM * GP_BK * 2;                        // one A plane of a stage: TM rows x 64 B
  constexpr int PLANE_B = TN * GP_BK * 2, STAGE = 2 * GP_PLANE_A + 2 * PLANE_B;
  constexpr int A_PIECES = TM / 16, AQ = A_PIECES / NW;             // 1 KiB pieces (16 rows x 64 B) of an A plane; per wave (2)
  constexpr int B_PIECES = TN / 16, BQ = (B_PIECES + NW - 1) / NW;  // ... of a W plane
  constexpr int GP_STAGING = NW * 32 * GP_LDE * 4;                  // NW waves x 32 rows of the epilogue transposition
  static_assert(AQ == 2, "two A pieces per wave");
  static_assert(GP_STAGING + TM * 4 <= 2 * STAGE, "epilogue staging + row table must fit the tile buffers");
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
  // batched problems: block-uniform base shifts (element offsets)
  const uint16_t *a_hi = d.a_hi, *a_lo = d.a_lo, *w_hi = d.w_hi, *w_lo = d.w_lo;
  long long c_off = 0;
  if (!CONV && gridDim.y > 1) {
    const int z = blockIdx.y, zo = z / d.z_inner, zi = z - zo * d.z_inner;
    const long long ao = zo * d.a_zo + zi * d.a_zi, wo = zo * d.w_zo + zi * d.w_zi;
    a_hi += ao; a_lo += ao; w_hi += wo; w_lo += wo;
    c_off = zo * d.c_zo + zi * d.c_zi;
  }
  const int TMV = d.tile_rows;                    // TM, or TM - 32 (round 6): a tile steps TMV rows, its last 32-row block is left out
  const int n_tiles = (d.N + TN - 1) / TN, m_tiles = (d.M + TMV - 1) / TMV;
  int m_tile, n_tile;
  if (m_tiles >= 16) {
    // all column tiles of a row tile on one XCD (the A tile is re-read from its L2)
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    m_tile = (slot / n_tiles) * 8 + xcd;
    n_tile = slot % n_tiles;
    if (m_tile >= m_tiles) return;
  } else {
    m_tile = blockIdx.x / n_tiles;
    n_tile = blockIdx.x - m_tile * n_tiles;
  }
  const int m0 = m_tile * TMV, n0 = n_tile * TN;

  // ---- loader: a plane goes to LDS in 1 KiB pieces (16 rows x 64 B); wave w brings pieces w and w + 8; lane -> row (lane >> 2),
  //      LDS chunk (lane & 3) <- the row's chunk (lane & 3) ^ ((row >> 2) & 3).
  //      Round 6 (rart_lds_dma.h): the loads are buffer_load ... lds -- a lane's byte offset inside its plane is ONE constant (avoff /
  //      wvoff; RART_DMA_OOR for rows past M / past the table: the resource's range check zero-fills them, no zero page), the K step and
  //      the tap move the SCALAR offset; with taps the resource base is shifted down by the most negative tap offset and a pixel
  //      outside the image for tap t (bit t of `nok`) gets its offset ORed with RART_DMA_OOR: two vector instructions per piece and
  //      step where round 5's form had ~15 (which competed with the co-resident workgroups' MFMAs for the vector issue).
  uint32_t avoff[2], nok[2] = {0u, 0u};
  int tapreg = 0, tap_min = 0;
  const bool one_tap = !CONV || d.n_taps == 1;
  if (CONV) {
    for (int t = 0; t < d.n_taps; ++t) tap_min = min(tap_min, (d.tap_dy[t] * d.src_w + d.tap_dx[t]) * d.lda * 2);
    if (lane < d.n_taps) tapreg = (d.tap_dy[lane] * d.src_w + d.tap_dx[lane]) * d.lda * 2 - tap_min;
  }
  const uint32_t tap0 = CONV ? (uint32_t)__builtin_amdgcn_readfirstlane((d.tap_dy[0] * d.src_w + d.tap_dx[0]) * d.lda * 2 - tap_min) : 0u;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = 16 * (wave + NW * q) + (lane >> 2);
    const int csrc = (lane & 3) ^ ((r >> 2) & 3);
    const int m = m0 + r;
    const bool ok = m < d.M;
    if (CONV) {
      const uint32_t mm = ok ? (uint32_t)m : 0u;
      const uint32_t t = gp_fastdiv(mm, d.gw_magic, d.gw_shift);
      const int ox = (int)(mm - t * (uint32_t)d.grid_w);
      const int n = (int)gp_fastdiv(t, d.gh_magic, d.gh_shift);
      const int oy = (int)(t - (uint32_t)n * (uint32_t)d.grid_h);
      const int by = oy * d.sy, bx = ox * d.sx;
      avoff[q] = (uint32_t)((n * d.src_h * d.src_w + by * d.src_w + bx) * d.lda * 2 + csrc * 16);
      for (int t2 = 0; t2 < d.n_taps; ++t2) {
        const int iy = by + d.tap_dy[t2], ix = bx + d.tap_dx[t2];
        if (!(ok && (unsigned)iy < (unsigned)d.src_h && (unsigned)ix < (unsigned)d.src_w)) nok[q] |= 1u << t2;
      }
      if (one_tap && (nok[q] & 1u)) avoff[q] = RART_DMA_OOR;
    } else {
      long long srow = m;
      if (d.map_rows) {
        const int img = m / d.rpi;
        srow = (long long)img * d.src_rpi + (m - img * d.rpi);
      }
      avoff[q] = ok ? (uint32_t)(((srow + d.src_off) * d.lda + csrc * 8) * 2) : RART_DMA_OOR;
    }
  }
  uint32_t wvoff[BQ];
#pragma unroll
  for (int q = 0; q < BQ; ++q) {
    const int r = 16 * (wave + NW * q) + (lane >> 2);
    const int csrc = (lane & 3) ^ ((r >> 2) & 3);
    const int n = n0 + r;
    wvoff[q] = (r < TN && n < d.w_rows) ? (uint32_t)((n * d.ldw + csrc * 8) * 2) : RART_DMA_OOR;
  }
  const rart_srd_t srd_ah = rart_dma_srd(reinterpret_cast<const char*>(a_hi) + tap_min), srd_al = rart_dma_srd(reinterpret_cast<const char*>(a_lo) + tap_min);
  const rart_srd_t srd_wh = rart_dma_srd(w_hi), srd_wl = rart_dma_srd(w_lo);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  // GP_W_INTERLEAVED: the weight table holds, per row and K step, the 64 bytes of the hi plane followed by the 64 bytes of the lo plane
  // (one 128-byte line per row and step; w_lo = w_hi + 32 elements) instead of two planes K apart
  const int w_step = (d.flags & GP_W_INTERLEAVED) ? 128 : 64;
#define RART_GP_ISSUE(KT, BUF)                                                                                  \
  {                                                                                                             \
    const uint32_t st_ = lds_base + (BUF)*STAGE;                                                                \
    const int kt_ = (KT);                                                                                       \
    if (one_tap) {                                                                                              \
      const uint32_t so_ = tap0 + (uint32_t)kt_ * 64u;                                                          \
      _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                           \
        rart_dma_load16(avoff[q], srd_ah, so_, st_ + (wave + NW * q) * 1024);                                   \
        rart_dma_load16(avoff[q], srd_al, so_, st_ + GP_PLANE_A + (wave + NW * q) * 1024);                      \
      }                                                                                                         \
    } else {                                                                                                    \
      const int tap_ = kt_ >> d.tpt_shift;                                                                      \
      const uint32_t so_ = (uint32_t)(__builtin_amdgcn_readlane(tapreg, tap_) + (kt_ - (tap_ << d.tpt_shift)) * 64); \
      _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                           \
        const uint32_t vo_ = avoff[q] | ((uint32_t)__builtin_amdgcn_sbfe(nok[q], tap_, 1) & RART_DMA_OOR);      \
        rart_dma_load16(vo_, srd_ah, so_, st_ + (wave + NW * q) * 1024);                                        \
        rart_dma_load16(vo_, srd_al, so_, st_ + GP_PLANE_A + (wave + NW * q) * 1024);                           \
      }                                                                                                         \
    }                                                                                                           \
    {                                                                                                           \
      const uint32_t so_ = (uint32_t)(kt_ * w_step);                                                            \
      _Pragma("unroll") for (int q = 0; q < BQ; ++q) {                                                          \
        if (NW * q + NW <= B_PIECES || wave + NW * q < B_PIECES) {   /* (first half: compile time, no branch) */  \
          rart_dma_load16(wvoff[q], srd_wh, so_, st_ + 2 * GP_PLANE_A + (wave + NW * q) * 1024);                \
          rart_dma_load16(wvoff[q], srd_wl, so_, st_ + 2 * GP_PLANE_A + PLANE_B + (wave + NW * q) * 1024);      \
        }                                                                                                       \
      }                                                                                                         \
    }                                                                                                           \
  }
  const int fr = lane & 31, h = lane >> 5;
  uint32_t xo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) xo[ks] = (uint32_t)(fr * 64 + (((2 * ks + h) ^ ((fr >> 2) & 3)) << 4));
  // accumulators start at the bias of their column (lane & 31 is the column of a 32 x 32 MFMA tile)
  f32x16 acc[MI][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int bc = n0 + wn * 64 + j * 32 + fr;
    const float bv = (d.bias && bc < d.N) ? d.bias[bc] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
  }
  const int KT = d.K / GP_BK;
  const bool skip_last = wm * RW + MI * 32 > TMV;      // (wave-uniform) this wave's last row block lies past the tile's rows
  RART_GP_ISSUE(0, 0)
  rart_dma_wait<0>();
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) RART_GP_ISSUE(kt + 1, buf ^ 1)
    const uint8_t* Ah = lds + buf * STAGE + (wm * RW) * 64;
    const uint8_t* Bh = lds + buf * STAGE + 2 * GP_PLANE_A + (wn * 64) * 64;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[MI], al[MI], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (i == MI - 1 && skip_last) continue;
        ah[i] = *reinterpret_cast<const bf16x8*>(Ah + i * 32 * 64 + xo[ks]);
        al[i] = *reinterpret_cast<const bf16x8*>(Ah + GP_PLANE_A + i * 32 * 64 + xo[ks]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(Bh + j * 32 * 64 + xo[ks]);
        bl[j] = *reinterpret_cast<const bf16x8*>(Bh + PLANE_B + j * 32 * 64 + xo[ks]);
      }
      // the two small products first, the large one last: all three land in the same fp32 accumulator
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (i == MI - 1 && skip_last) continue;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    rart_dma_wait<0>();                     // the next stage has landed in LDS
    __syncthreads();
  }
#undef RART_GP_ISSUE
  gp_epilogue<TM, TN, CONV, MI>(d, lds, acc, m0, n0, c_off, TMV);
}

// 1 (default): the 256-row tiles with 128 / 256 columns run the ping-pong schedule of csrc/gemm_pair_pp.hip; 2 (opt-in, measured slower): as 1,
// and one-tap problems with at least two 256 x 128 tiles per CU on the PERSISTENT kernel with the deferred epilogue; 0: this file's two-stage loop everywhere (the
// A/B switch of tests/test_engine_x3_gpu.py and scratch/r6/).  Bit-identical outputs under all three.  RART_PAIR_SCHEDULE in the environment
// sets the initial value.
int g_pair_schedule = -1;
int gp_schedule() {
  if (g_pair_schedule < 0) {
    const char* e = getenv("RART_PAIR_SCHEDULE");
    g_pair_schedule = e ? atoi(e) : 1;
  }
  return g_pair_schedule;
}

int gp_rows224() {          // lab switch of the 224-row tiles below (default on)
  const char* e = getenv("RART_PAIR_ROWS224");
  return e ? atoi(e) : 1;
}
int gp_split() {            // lab switch of the remainder split below (default on)
  const char* e = getenv("RART_PAIR_SPLIT");
  return e ? atoi(e) : 1;
}

void gp_magic(uint32_t dv, uint32_t& mg, uint32_t& sh) {   // exact for dividends < 2^31
  uint32_t l = 0;
  while ((1ull << l) < dv) ++l;
  sh = 31 + l;
  mg = (uint32_t)(((1ull << sh) + dv - 1) / dv);
}
}  // namespace

bool rart_gemm_pair_pp_launch(const void* dev_desc, int tn, bool conv, unsigned grid_x, unsigned grid_y, hipStream_t st, unsigned walk_wgs);   // gemm_pair_pp.hip
void rart_gemm_pair_ps_launch(const void* dev_desc, bool conv, unsigned wgs, hipStream_t st);

namespace {
int gp_cu_count() {      // compute units of the current device (the persistent kernel's grid), per device
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus[dev];
}
}  // namespace

extern "C" int rart_gemm_pair_set_schedule(int mode) {
  RART_CHECK_ARG(mode >= 0 && mode <= 3,
                 "rart_gemm_pair_set_schedule: mode must be 0 (two-stage loop), 1 (ping-pong), 2 (ping-pong + deferred-epilogue persistent) or 3 (ping-pong, tiles walked)");
  g_pair_schedule = mode;
  return RART_OK;
}
extern "C" int rart_gemm_pair_get_schedule(void) { return gp_schedule(); }

extern "C" int rart_gemm_pair_bf16(const rart_gemm_pair_desc* h, rart_stream_t stream) {
  RART_CHECK_ARG(h != nullptr, "rart_gemm_pair_bf16: null descriptor");
  RART_CHECK_ARG(h->a_hi && h->a_lo && h->w_hi && h->w_lo && h->dst_hi, "rart_gemm_pair_bf16: null operand plane");
  const bool conv = h->conv != 0;
  GemmPairDev d;
  d.M = h->M; d.K = h->K;
  if (conv) {
    RART_CHECK_ARG(h->batch > 0 && h->grid_h > 0 && h->grid_w > 0 && h->src_h > 0 && h->src_w > 0 && h->dst_h > 0 && h->dst_w > 0,
                   "rart_gemm_pair_bf16 (conv): empty geometry");
    RART_CHECK_ARG(h->n_taps >= 1 && h->n_taps <= 16, "rart_gemm_pair_bf16 (conv): n_taps must be 1..16");
    RART_CHECK_ARG(h->k_per_tap >= 32 && (h->k_per_tap & (h->k_per_tap - 1)) == 0,
                   "rart_gemm_pair_bf16 (conv): k_per_tap must be a power of two >= 32");
    RART_CHECK_ARG(h->n_batched <= 1 && h->rows_per_image == 0 && h->src_row_off == 0 && h->dst_row_off == 0,
                   "rart_gemm_pair_bf16 (conv): no batching / row re-basing in conv mode");
    const long long M = (long long)h->batch * h->grid_h * h->grid_w;
    RART_CHECK_ARG(M < (1ll << 31), "rart_gemm_pair_bf16 (conv): row grid must stay below 2^31 rows");
    RART_CHECK_ARG((long long)h->batch * h->src_h * h->src_w * h->lda < (1ll << 30) && (long long)h->batch * h->dst_h * h->dst_w * h->ldc < (1ll << 31),
                   "rart_gemm_pair_bf16 (conv): a source plane must stay below 2 GiB (32-bit byte offsets of the stage loads) and the destination "
                   "below 2^31 elements (split the batch)");
    RART_CHECK_ARG(h->lda % 8 == 0 || h->lda == 4, "rart_gemm_pair_bf16 (conv): source pixels must keep 16-byte alignment of the K chunks");
    d.M = (int)M;
    d.K = h->k_per_tap * h->n_taps;
    d.grid_h = h->grid_h; d.grid_w = h->grid_w; d.src_h = h->src_h; d.src_w = h->src_w; d.sy = h->sy; d.sx = h->sx; d.n_taps = h->n_taps;
    int sh = 0;
    while ((32 << sh) < h->k_per_tap) ++sh;
    d.tpt_shift = sh;
    for (int i = 0; i < 16; ++i) { d.tap_dy[i] = i < h->n_taps ? h->tap_dy[i] : 0; d.tap_dx[i] = i < h->n_taps ? h->tap_dx[i] : 0; }
    d.dst_h = h->dst_h; d.dst_w = h->dst_w; d.dst_sy = h->dst_sy; d.dst_sx = h->dst_sx; d.dst_oy = h->dst_oy; d.dst_ox = h->dst_ox;
    gp_magic((uint32_t)d.grid_w, d.gw_magic, d.gw_shift);
    gp_magic((uint32_t)d.grid_h, d.gh_magic, d.gh_shift);
    d.mask_bits = (const uint8_t*)h->mask_bits; d.sign_out = (uint8_t*)h->sign_out;
    RART_CHECK_ARG(!(d.sign_out && (h->flags & GP_OUT_F32)), "rart_gemm_pair_bf16 (conv): sign_out needs a pair destination");
  } else {
    RART_CHECK_ARG(!h->mask_bits && !h->sign_out, "rart_gemm_pair_bf16: mask_bits / sign_out are served by conv mode only");
    RART_CHECK_ARG(h->lda % 8 == 0 && h->lda >= h->K, "rart_gemm_pair_bf16: lda must cover the row and keep 16-byte alignment");
    d.grid_h = d.grid_w = d.src_h = d.src_w = d.sy = d.sx = d.n_taps = 1; d.tpt_shift = 0;
    for (int i = 0; i < 16; ++i) d.tap_dy[i] = d.tap_dx[i] = 0;
    d.dst_h = d.dst_w = d.dst_sy = d.dst_sx = 1; d.dst_oy = d.dst_ox = 0;
    d.gw_magic = d.gw_shift = d.gh_magic = d.gh_shift = 0;
    d.mask_bits = nullptr; d.sign_out = nullptr;
  }
  d.N = h->N;
  RART_CHECK_ARG(d.M > 0 && d.N > 0 && d.N % 8 == 0, "rart_gemm_pair_bf16: M > 0, N a positive multiple of 8");
  RART_CHECK_ARG(d.K > 0 && d.K % GP_BK == 0, "rart_gemm_pair_bf16: K must be a positive multiple of 32");
  RART_CHECK_ARG(h->ldw % 8 == 0 && h->ldc % 8 == 0 && h->ldw >= d.K && h->ldc >= d.N,
                 "rart_gemm_pair_bf16: leading dimensions must cover the row and keep 16-byte alignment");
  RART_CHECK_ARG((long long)(h->w_rows > 0 ? h->w_rows : h->N) * h->ldw < (1ll << 30), "rart_gemm_pair_bf16: a weight plane must stay below 2 GiB");
  if (!conv) {
    const long long imgs = (h->rows_per_image > 0 && h->rows_per_image < d.M) ? (d.M + h->rows_per_image - 1) / h->rows_per_image : 1;
    const long long src_rows = imgs > 1 ? imgs * (h->src_rows_per_image > 0 ? h->src_rows_per_image : h->rows_per_image) : d.M;
    RART_CHECK_ARG((src_rows + h->src_row_off) * h->lda < (1ll << 30), "rart_gemm_pair_bf16: a source plane must stay below 2 GiB (per batched problem)");
  }
  const bool out_f32 = (h->flags & GP_OUT_F32) != 0;
  RART_CHECK_ARG(out_f32 || h->dst_lo, "rart_gemm_pair_bf16: a pair destination needs its lo plane");
  RART_CHECK_ARG((h->res_hi == nullptr) == (h->res_lo == nullptr), "rart_gemm_pair_bf16: the residual is a pair: both planes or none");
  RART_CHECK_ARG(!(h->flags & (GP_GELU_KEEP | GP_GELU_BWD)) || (h->aux_hi && h->aux_lo),
                 "rart_gemm_pair_bf16: GELU_KEEP / GELU_BWD need the pre-activation pair (aux)");
  RART_CHECK_ARG(!(h->flags & ~(GP_OUT_F32 | GP_GELU | GP_GELU_BWD | GP_GELU_KEEP | GP_RELU | GP_W_INTERLEAVED)), "rart_gemm_pair_bf16: unknown flag");
  RART_CHECK_ARG(!(h->flags & GP_W_INTERLEAVED) || h->ldw >= 2 * d.K, "rart_gemm_pair_bf16: an interleaved weight table has rows of 2 K elements");
  {
    const int g = (h->flags & GP_GELU ? 1 : 0) + (h->flags & GP_GELU_BWD ? 1 : 0) + (h->flags & GP_GELU_KEEP ? 1 : 0);
    RART_CHECK_ARG(g <= 1 && !((h->flags & GP_GELU_KEEP) && out_f32), "rart_gemm_pair_bf16: at most one GELU mode; GELU_KEEP writes pairs");
  }
  d.a_hi = (const uint16_t*)h->a_hi; d.a_lo = (const uint16_t*)h->a_lo; d.w_hi = (const uint16_t*)h->w_hi; d.w_lo = (const uint16_t*)h->w_lo;
  d.bias = h->bias; d.res_hi = (const uint16_t*)h->res_hi; d.res_lo = (const uint16_t*)h->res_lo;
  d.dst_hi = (uint16_t*)h->dst_hi; d.dst_lo = (uint16_t*)h->dst_lo; d.aux_hi = (uint16_t*)h->aux_hi; d.aux_lo = (uint16_t*)h->aux_lo;
  d.lda = h->lda; d.ldw = h->ldw; d.ldc = h->ldc;
  d.w_rows = h->w_rows > 0 ? h->w_rows : h->N;
  d.map_rows = (!conv && h->rows_per_image > 0 && h->rows_per_image < d.M) ? 1 : 0;
  d.rpi = d.map_rows ? h->rows_per_image : d.M;
  d.src_rpi = h->src_rows_per_image > 0 ? h->src_rows_per_image : d.rpi;
  d.dst_rpi = h->dst_rows_per_image > 0 ? h->dst_rows_per_image : d.rpi;
  d.src_off = h->src_row_off; d.dst_off = h->dst_row_off;
  RART_CHECK_ARG(d.src_off >= 0 && d.dst_off >= 0, "rart_gemm_pair_bf16: row offsets must be >= 0");
  d.flags = h->flags;
  {
    // non-temporal epilogue streams for outputs that do not stay cached anyway (lab switch: RART_PAIR_NT_MIN_MB, the output pair's size)
    const char* e = getenv("RART_PAIR_NT_MIN_MB");
    const long long min_mb = e ? atoll(e) : 0;
    d.nt = (long long)d.M * d.N * 4 >= min_mb * (1ll << 20) ? 1 : 0;
  }
  const int nz = h->n_batched > 1 ? h->n_batched : 1;
  RART_CHECK_ARG(nz <= 65535, "rart_gemm_pair_bf16: n_batched must be <= 65535");
  d.z_inner = h->z_inner > 0 ? h->z_inner : 1;
  d.a_zo = h->a_z_outer; d.a_zi = h->a_z_inner; d.w_zo = h->w_z_outer; d.w_zi = h->w_z_inner; d.c_zo = h->c_z_outer; d.c_zi = h->c_z_inner;
  // Tile policy, from the per-shape sweep of every launch of a ResNet-50 gradient evaluation at B = 256 (scratch/r4/sweep_pair_tiles.py,
  // profiles/r04_pair_tile_sweep.txt).  Plain products (the transformer GEMMs): 256 rows x the widest of 64 / 128 / 256 columns that does not
  // pad the MFMA work.  Convolutions: K <= 256 (the HBM-bound 1x1 layers: two K steps do not hide a tile's load and store phases) 256 x 64,
  // 80 KB of LDS = two workgroups per CU; K < 1024 128 x {128, 64} (64 / 48 KB: two or three per CU); K-deep wide layers 256 x 256 unless
  // that leaves fewer than 128 workgroups (layer4: 256 x 128), K-deep narrow layers (N <= 128) 128 x N.  tile_m / tile_n override.
  int tn = d.N <= 64 ? 64 : (d.N <= 128 ? 128 : 256), tm = 256;
  if (conv) {
    if (gp_schedule() >= 1 && d.N >= 256 && (d.K >= 512 || (d.K >= 256 && d.N <= 512))) {
      // round 6 (scratch/r6/time_pair_pp.py --tiles, profiles/r06_pair_pp.json): with the ping-pong schedule the 256-row tiles win wherever
      // the layer is at least 256 wide and 256 deep -- 256 x 256 unless that leaves fewer than 128 workgroups (layer4: 256 x 128); the one
      // exception measured: K = 256 into >= 1024 columns (layer3's expansion: epilogue-bound, 155-165 us on 256 x 64 vs 166-176).  128-column
      // layers (layer2's 3 x 3 once its tail kernel is off) stay on the two-stage 128 x 128 tiles: 19.97 vs 20.14 ms on 256 x 128 ping-pong
      tm = 256;
      tn = (long long)((d.M + 255) / 256) * ((d.N + 255) / 256) < 128 ? 128 : 256;
    } else if (d.K <= 256) { tm = 256; tn = 64; }
    else if (d.K < 1024 || d.N <= 128) { tm = 128; tn = d.N <= 64 ? 64 : 128; }
    else if ((long long)((d.M + 255) / 256) * ((d.N + 255) / 256) < 128) { tm = 256; tn = 128; }
    if (d.M <= 256) { tm = 128; tn = 64; }
  }
  if (h->tile_n == 64 || h->tile_n == 128 || h->tile_n == 256) tn = h->tile_n;
  if (h->tile_m == 128 || h->tile_m == 256) tm = h->tile_m;
  if (h->tile_m == 224) tm = 256;            // the ping-pong kernel's 224-row form (below)
  const int m_tiles = (d.M + tm - 1) / tm, n_tiles = (d.N + tn - 1) / tn;
  const int m_enum = m_tiles >= 16 ? (m_tiles + 7) / 8 * 8 : m_tiles;     // the XCD remap enumerates row tiles in groups of 8
  const long long blocks = (long long)m_enum * n_tiles;
  RART_CHECK_ARG(blocks < (1ll << 31), "rart_gemm_pair_bf16: grid too large");
  const dim3 grid((uint32_t)blocks, nz);
  hipStream_t st = (hipStream_t)stream;
  // the persistent kernel (256 x 128 tiles, deferred epilogue): one tap, no GELU epilogue, no row re-basing / batching, at least two tiles per
  // CU and a K loop with room for the 12 micro-steps of a tile's epilogue
  if (gp_schedule() == 2 && tm == 256 && tn >= 128 && h->tile_n == 0 && h->tile_m == 0 && nz == 1 && !d.map_rows && d.src_off == 0 &&
      (conv ? d.n_taps == 1 : true) && !(h->flags & (GP_GELU | GP_GELU_BWD | GP_GELU_KEEP)) && d.K / GP_BK >= 6) {
    const int cus = gp_cu_count();
    const long long tiles = (long long)((d.M + 255) / 256) * ((d.N + 127) / 128);
    if (tiles >= 2ll * cus) {
      rart_gemm_pair_ps_launch(&d, conv, (unsigned)cus, st);
      RART_CHECK_LAUNCH("rart_gemm_pair_bf16 (persistent)");
      return RART_OK;
    }
  }
  d.m_begin = 0;
  d.tile_rows = 256;
  if (tm == 256 && tn >= 128 && gp_schedule() >= 1 && nz == 1 && (h->tile_m == 224 || (h->tile_m == 0 && gp_rows224()))) {
    // Round 6: 224-row tiles.  A launch runs in passes of one tile per CU and XCD (above); ResNet-50's 50176 / 12544 rows are 196 / 49 tiles
    // of 256 rows = 25 / 7 of an XCD's 32 CUs in the fullest XCD.  A tile that steps 224 rows (the kernel leaves its last 32-row block out;
    // loads, waits and barriers unchanged) makes 224 / 56 row tiles = 28 / 7 per XCD: the same passes, 7/8 of the matrix work per tile.
    auto passes = [&](int mt, int nt) { const int cx = gp_cu_count() / 8; return ((mt >= 16 ? (mt + 7) / 8 * nt : (mt * nt + 7) / 8) + cx - 1) / cx; };
    const int mt224 = (d.M + 223) / 224;
    if (h->tile_m == 224 || passes(mt224, n_tiles) * 0.92 < passes(m_tiles, n_tiles) - 0.01) {
      d.tile_rows = 224;
      const int me = mt224 >= 16 ? (mt224 + 7) / 8 * 8 : mt224;
      if (rart_gemm_pair_pp_launch(&d, tn, conv, (unsigned)(me * n_tiles), 1, st, 0)) {
        RART_CHECK_LAUNCH("rart_gemm_pair_bf16 (ping-pong, 224-row tiles)");
        return RART_OK;
      }
      d.tile_rows = 256;
    }
  }
  if (!conv && tm == 256 && tn == 256 && gp_schedule() >= 1 && gp_split() && nz == 1 && h->tile_n == 0 && h->tile_m == 0) {
    // Round 6 (scratch/r6/time_pair_rounds.py, GRBM_GUI_ACTIVE per launch): a 256-row launch is quantised in passes of ONE TILE PER CU PER XCD
    // -- workgroup i goes to XCD i % 8, and the row-tile enumeration gives XCD x the row tiles x, x + 8, ... -- so ViT-B/16's 50432 x 768
    // outputs (197 x 3 tiles, 75 per XCD of 32 CUs) take THREE tile times for 2.3 passes of work.  Such a launch is issued as two: the row
    // tiles of the whole passes on 256 x 256 tiles, the remaining rows (m_begin) on 256 x 128 tiles, which take about half a tile time.
    // Same products in the same order per output element: same bits.  Plain products only: on the convolutions of ResNet-50 (short K, epilogue-
    // heavy tiles) the second launch costs what it saves (same-box A/B 19.53 vs 19.60 ms per gradient evaluation); ViT-B/16 66.9 -> 65.9 ms.
    const int cx = gp_cu_count() / 8;                      // CUs per XCD
    auto passes = [&](int mt, int nt) { return ((mt >= 16 ? (mt + 7) / 8 * nt : (mt * nt + 7) / 8) + cx - 1) / cx; };
    const double half_cost = 0.8;                          // measured: the second launch costs ~0.8 tile times (it starts when the first has drained)
    const int n128 = (d.N + 127) / 128;
    double best = passes(m_tiles, n_tiles);
    int best_ma = 0;
    for (int pa = 1; pa < passes(m_tiles, n_tiles); ++pa) {
      const int ma = pa * cx / n_tiles * 8;                // row tiles of the first launch: pa passes exactly on every XCD
      if (ma < 16 || ma >= m_tiles) continue;
      const double c = passes(ma, n_tiles) + half_cost * passes(m_tiles - ma, n128) + 0.05;
      if (c < best - 0.1) { best = c; best_ma = ma; }
    }
    if (best_ma > 0) {
      GemmPairDev da = d, db = d;
      da.M = best_ma * 256;
      db.m_begin = best_ma * 256;
      const int mb = m_tiles - best_ma, mb_enum = mb >= 16 ? (mb + 7) / 8 * 8 : mb;
      if (rart_gemm_pair_pp_launch(&da, 256, conv, (unsigned)(best_ma * n_tiles), 1, st, 0) &&
          rart_gemm_pair_pp_launch(&db, 128, conv, (unsigned)(mb_enum * n128), 1, st, 0)) {
        RART_CHECK_LAUNCH("rart_gemm_pair_bf16 (ping-pong, remainder split)");
        return RART_OK;
      }
    }
  }
  // schedule 3: launches of more than one tile per CU are WALKED by one workgroup per CU (k_gemm_pair_pp, OPT bit 1)
  const unsigned walk_wgs = (gp_schedule() == 3 && nz == 1 && blocks > (long long)gp_cu_count()) ? (unsigned)(gp_cu_count() & ~7) : 0u;
  if (tm == 256 && tn >= 128 && gp_schedule() >= 1 && rart_gemm_pair_pp_launch(&d, tn, conv, grid.x, grid.y, st, walk_wgs)) {
    RART_CHECK_LAUNCH("rart_gemm_pair_bf16 (ping-pong)");
    return RART_OK;
  }
  // the two-stage kernel: tiles that step tm - 32 rows where that saves work per pass (resident workgroups per CU by LDS: 2 x STAGE bytes each)
  d.tile_rows = tm;
  dim3 grid2 = grid;
  if ((gp_rows224() == 2 || (gp_rows224() == 3 && d.K >= 1024)) && nz == 1 && h->tile_m == 0 && h->tile_n == 0 && gp_schedule() >= 1) {   // (3: K-deep only)
    const int lds = 2 * (2 * tm * 64 + 2 * tn * 64), res = lds > 80 * 1024 ? 1 : (lds > 53 * 1024 ? 2 : 3), cx = gp_cu_count() / 8 * res;
    auto passes = [&](int mt) { return ((mt >= 16 ? (mt + 7) / 8 * n_tiles : (mt * n_tiles + 7) / 8) + cx - 1) / cx; };
    const int rows = tm - 32, mt2 = (d.M + rows - 1) / rows;
    if (passes(mt2) * (double)rows / tm * 1.03 < passes(m_tiles)) {
      d.tile_rows = rows;
      grid2 = dim3((uint32_t)((mt2 >= 16 ? (mt2 + 7) / 8 * 8 : mt2) * n_tiles), nz);
    }
  }
#define RART_GP_LAUNCH(TM_, TN_, CONV_) hipLaunchKernelGGL((k_gemm_pair<TM_, TN_, CONV_>), grid2, dim3(TM_ * 2), 0, st, d)
#define RART_GP_BY_TN(TM_, CONV_)                                          \
  do {                                                                     \
    if (tn == 64) RART_GP_LAUNCH(TM_, 64, CONV_);                          \
    else if (tn == 128) RART_GP_LAUNCH(TM_, 128, CONV_);                   \
    else RART_GP_LAUNCH(TM_, 256, CONV_);                                  \
  } while (0)
  if (conv) {
    if (tm == 128) RART_GP_BY_TN(128, true);
    else RART_GP_BY_TN(256, true);
  } else {
    if (tm == 128) RART_GP_BY_TN(128, false);
    else RART_GP_BY_TN(256, false);
  }
#undef RART_GP_BY_TN
#undef RART_GP_LAUNCH
  RART_CHECK_LAUNCH("rart_gemm_pair_bf16");
  return RART_OK;
}
