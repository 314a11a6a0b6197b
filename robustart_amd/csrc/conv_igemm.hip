// bf16 implicit-GEMM convolution for gfx950 (CDNA4): the one dense-contraction kernel of the engines -- ResNet-50
// eval forward / backward-to-input, its train-mode forward / backward / weight gradients, and every GEMM of ViT-B/16
// forward and backward-to-input (K22 in SURVEY.md 2.1).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]        A gathered on the fly from an NHWC bf16 tensor
//
// m enumerates (image, oy, ox) over a row grid; k = tap * k_per_tap + c; for tap t the source
// pixel is (oy*sy + dy[t], ox*sx + dx[t]) (zeros outside the image).  The same kernel therefore
// runs: forward convs (taps = filter taps), backward-to-input of stride-1 convs (taps flipped,
// weights re-laid on the host), each input-parity class of a stride-2 conv's backward (sub-grid +
// strided destination), the 7x7 stem on a pre-padded 4-channel image (a "tap" = one filter row of
// 8 pixels x 4 channels, hi/lo bf16 split of the fp32 pixels as extra taps), and plain GEMMs (fc).
//
// Design for MI355X: 128 x {128,64} x {32,64} block tile, 4 wave64s each owning a 64 x {64,32} sub-tile of
// v_mfma_f32_32x32x16_bf16 fragments (fp32 accumulate, started at the column's bias); A/B staged
// global -> VGPR -> LDS as 16-byte vectors, double-buffered, one barrier per K step (BK 32: two register sets, loads two
// K steps ahead; BK 64 for K >= 1024: one set, half the barriers per FLOP); LDS rows padded to 80 / 144 B so
// ds_read_b128 fragment reads are bank-conflict free; in the epilogue each wave transposes its own sub-tile through a
// private LDS region (no block barrier after the last MFMA), adds the residual, packs with v_cvt_pk_bf16_f32, applies
// ReLU / the ReLU mask on packed pairs (v_pk_max_i16) and stores coalesced 16-byte rows; the pixel index is decoded with
// host-computed multiply-shift constants; block ids are remapped so the column tiles of one row tile run on the same XCD
// (shared L2 for the re-read A tile) unless there are fewer than 16 row tiles (short-M GEMMs: split-K weight gradients,
// per-head attention products), which keep plain order so all XCDs work.  Batched problems (blockIdx.y) shift the
// operand bases per z: attention products and split-K partial sums run as one launch.
// Most ResNet-50 layers are HBM-bound at bf16 (K = 64..512), so the kernel is built around wide coalesced traffic first
// and MFMA issue second (DESIGN.md 4.3 has the measured breakdown).
#include "rart_common.h"
#include "rart_lds_dma.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct RartConvDescDev {
  const uint16_t* src;
  const uint16_t* wgt;
  const float* bias;
  const uint16_t* res;
  const uint16_t* mask;
  void* dst;
  int batch, grid_h, grid_w;
  int src_h, src_w, src_pix_stride;
  int k_per_tap, n_taps;
  int sy, sx;
  int tap_dy[32], tap_dx[32];
  long long tap_src_off[32];
  int n_cols;
  int dst_h, dst_w, dst_sy, dst_sx, dst_oy, dst_ox, dst_pix_stride;
  int flags;
  // batched problems (blockIdx.y = z): z -> (zo = z / z_inner, zi = z % z_inner); element offsets added to
  // src / wgt / dst(+res, mask); wgt_row_stride = elements between consecutive weight rows (0 = K)
  int z_inner, wgt_row_stride;
  long long src_zo, src_zi, wgt_zo, wgt_zi, dst_zo, dst_zi;
  // exact division by multiply-shift for dividends < 2^31 (row index -> (image, oy, ox); K step -> tap):
  // q = (n * magic) >> shift.  A hardware 64-bit division costs ~300 instructions and every workgroup needed six.
  uint32_t gw_magic, gw_shift, gh_magic, gh_shift, tpt_magic, tpt_shift;
  // 1-bit-per-element side tensors, indexed like dst (byte (off + col) / 8, bit col % 8): sign_out receives (output > 0)
  // of a bf16 store (the ReLU mask the backward pass will need); with F_MASK_BITS `mask` is such a tensor instead of bf16
  uint8_t* sign_out;
  // split-bf16 ("bf16x3") tensors: a value is the pair hi + lo of two bf16 planes; the lo plane of dst / res sits this many
  // ELEMENTS after the hi plane (PAIR kernel instances only; src planes are reached through tap_src_off)
  long long dst_pair_off, res_pair_off;
  // train-mode forward: per row tile, the column sums and sums of squares of the bf16 OUTPUT ([m_tiles][2][n_cols] fp32): the batch
  // statistics BatchNorm needs, taken from the accumulators instead of a pass over the stored tensor (plain bf16 conv only)
  float* stats_out;
};

namespace {
constexpr int BM = 128;
constexpr int kThreads = 256;
enum { F_RELU = 1, F_OUT_F32 = 2, F_GELU = 4, F_GELU_BWD = 8, F_MASK_BITS = 16, F_PAIR = 32, F_GELU_KEEP = 64, F_MASK_RES = 128 };

__device__ __forceinline__ uint32_t fastdiv(uint32_t n, uint32_t magic, uint32_t shift) {
  return (uint32_t)(((uint64_t)n * magic) >> shift);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) short i16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {   // round to nearest even
  f32x2_t f = {lo, hi};
  bf16x2_t b = __builtin_convertvector(f, bf16x2_t);
  return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t w) {           // sign bit set -> 0 (also -0.0)
  const i16x2_t z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w), z));
}
// two mask bits (bit 0 -> low half, bit 1 -> high half) -> 0xFFFF per selected bf16 half
__device__ __forceinline__ uint32_t halves_from_bits(uint32_t byte, int pair) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe(byte, 2 * pair, 1);        // 0 or 0xFFFFFFFF
  const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe(byte, 2 * pair + 1, 1);
  return __builtin_amdgcn_perm(hi, lo, 0x07060100u);                              // {hi.b3, hi.b2, lo.b1, lo.b0}
}
// packed pair of non-negative-or-zero bf16 (after ReLU) or any bf16 -> 2 bits (half > 0)
__device__ __forceinline__ uint32_t bits_from_halves(uint32_t w) {
  const i16x2_t z = {0, 0}, one = {1, 1};
  const uint32_t t = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w), z), one));
  return (t | (t >> 15)) & 3u;
}
__device__ __forceinline__ uint32_t positive_lanes_i16(uint32_t w) {    // 0xFFFF per half whose int16 is > 0
  const i16x2_t z = {0, 0}, one = {1, 1}, full = {-1, -1};
  const i16x2_t t = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w), z), one);
  return __builtin_bit_cast(uint32_t, (i16x2_t)(t * full));
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the result): branch-free, ~20 VALU
// instructions instead of libm erff's ~45 with divergent range branches -- the exact-GELU epilogue of the ViT fc1 GEMM
// spent more cycles there than in its K loop
__device__ __forceinline__ float gelu_erf(float v) {
  const float x = fabsf(v) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float erf_abs = 1.0f - p * t * __expf(-x * x);
  return 0.5f * v * (1.0f + copysignf(erf_abs, v));
}
// d/du [u * Phi(u)] = Phi(u) + u * phi(u), sharing the exponential with the erf approximation above
__device__ __forceinline__ float gelu_grad_erf(float u) {
  const float x = fabsf(u) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-x * x);                         // exp(-u^2 / 2)
  const float erf_abs = 1.0f - p * t * e;
  return fmaf(u * 0.3989422804014327f, e, 0.5f * (1.0f + copysignf(erf_abs, u)));
}
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even (inputs are finite)
  return (uint16_t)(u >> 16);
}

// Per-thread row of the A tile for the buffer loads of k_conv_igemm_bf16: vo = byte offset of the row's pixel at tap (0, 0) inside a source
// plane (+ the thread's 16-byte chunk), nok = one bit per tap whose source pixel lies outside the image (all taps for a row past M).
template <class D>
__device__ __forceinline__ void row_setup(uint32_t m, uint32_t M, const D& d, int chunk_bytes, uint32_t& vo, uint32_t& nok) {
  const bool ok = m < M;
  const uint32_t mm = ok ? m : 0u;
  const uint32_t t = fastdiv(mm, d.gw_magic, d.gw_shift);
  const int ox = (int)(mm - t * (uint32_t)d.grid_w);
  const int n = (int)fastdiv(t, d.gh_magic, d.gh_shift);
  const int oy = (int)(t - (uint32_t)n * (uint32_t)d.grid_h);
  const int by = oy * d.sy, bx = ox * d.sx;
  vo = (uint32_t)(((n * d.src_h * d.src_w + by * d.src_w + bx) * d.src_pix_stride) * 2 + chunk_bytes);
  nok = 0u;
  for (int t2 = 0; t2 < d.n_taps; ++t2) {
    const int iy = by + d.tap_dy[t2], ix = bx + d.tap_dx[t2];
    if (!(ok && (unsigned)iy < (unsigned)d.src_h && (unsigned)ix < (unsigned)d.src_w)) nok |= 1u << t2;
  }
}
// The buffer resource of tap `tap`: the source plane shifted by the tap's pixel offset and its plane offset (64-bit, scalar; it may point
// before the tensor: only lanes whose pixel is inside the image are in range)
template <class D>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tap_rsrc(const uint16_t* p_src, const D& d, int tap) {
  const uint16_t* base = p_src + d.tap_src_off[tap] + (long long)(d.tap_dy[tap] * d.src_w + d.tap_dx[tap]) * d.src_pix_stride;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
typedef unsigned int igemm_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  const igemm_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// BK = 32: two register sets, loads two K steps ahead (memory-latency-bound 1x1 layers, 3 blocks/CU).
// BK = 64: one register set, loads one (twice as long) K step ahead, half the barriers per FLOP
//          (compute-bound 3x3 layers and transformer GEMMs; 73 KB of LDS -> 2 blocks/CU).
// PAIR: the split-bf16 reference-precision mode (flag 32).  The K loop is unchanged -- the host lists every product of the
// scheme (x_hi.w_hi, x_hi.w_lo, x_lo.w_hi) as extra taps whose tap_src_off selects the operand plane, so all three
// accumulate in the same fp32 registers -- and the epilogue reads a residual pair and writes hi = bf16(v), lo = bf16(v - hi).
template <int BN, int BK, bool PAIR>
__global__ __launch_bounds__(kThreads, BK == 32 ? 3 : 2) void k_conv_igemm_bf16(const RartConvDescDev d) {
  constexpr int LDK = BK + 8;  // LDS row: 80 B (BK 32) / 144 B (BK 64): ds_read_b128 fragment reads are conflict free
  // batched problems: block-uniform base shifts, kept in scalars (copying the descriptor would move its tap
  // tables from the kernarg segment into scratch)
  const uint16_t* p_src = d.src;
  const uint16_t* p_wgt = d.wgt;
  const uint16_t* p_res = d.res;
  const uint16_t* p_mask = d.mask;
  long long p_dst_off = 0;
  if (gridDim.y > 1) {
    const int z = blockIdx.y, zo = z / d.z_inner, zi = z - zo * d.z_inner;
    p_src += zo * d.src_zo + zi * d.src_zi;
    p_wgt += zo * d.wgt_zo + zi * d.wgt_zi;
    p_dst_off = zo * d.dst_zo + zi * d.dst_zi;
    if (p_res) p_res += p_dst_off;
    if (p_mask) p_mask += p_dst_off;
  }
  constexpr int WN = BN / 2;        // wave sub-tile columns
  constexpr int TN = WN / 32;       // 32-wide MFMA tiles per wave along n
  constexpr int B_CHUNKS = BN * 4 / kThreads;  // 16-byte chunks of the W tile per thread (2 or 1)
  constexpr int kLdsBytes = 2 * (BM + BN) * LDK * 2;
  __shared__ __attribute__((aligned(16))) uint8_t lds_raw[kLdsBytes + BM * 4];
  uint16_t* sA = reinterpret_cast<uint16_t*>(lds_raw);               // [2][BM][LDK]
  uint16_t* sB = sA + 2 * BM * LDK;                                   // [2][BN][LDK]
  uint32_t* row_dst = reinterpret_cast<uint32_t*>(lds_raw + kLdsBytes);  // [BM] dst element offset or ~0u

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware tile mapping: all column tiles of a row tile on one XCD ----
  const int n_tiles = (d.n_cols + BN - 1) / BN;
  const uint32_t M = (uint32_t)(d.batch * d.grid_h * d.grid_w);   // host guarantees < 2^31
  const int m_tiles = (int)((M + BM - 1) / BM);
  int m_tile, n_tile;
  if (m_tiles >= 16) {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    m_tile = (slot / n_tiles) * 8 + xcd;
    n_tile = slot % n_tiles;
    if (m_tile >= m_tiles) return;
  } else {
    // few row tiles (split-K weight gradients, per-head attention products): the XCD remap would leave most XCDs
    // idle (row tile i lives on XCD i & 7); plain order spreads the column tiles and the z batches over all of them
    m_tile = blockIdx.x / n_tiles;
    n_tile = blockIdx.x - m_tile * n_tiles;
  }
  const uint32_t m0 = (uint32_t)m_tile * BM;
  const int n0 = n_tile * BN;

  // ---- per-thread gather rows (2 rows of the A tile) ----
  // Round 6 (rart_lds_dma.h has the measurement): the tile loads are buffer loads -- a row's byte offset inside the source plane is ONE
  // constant per thread (voffset), the tap moves the resource BASE (scalar), the K slice the scalar offset; a pixel outside the image
  // for tap t has bit t of the row's `nok` mask set and its offset ORed with RART_DMA_OOR, which the range check zero-fills.  Rounds 1-5
  // formed a 64-bit address per row and K step with ~12 vector instructions (bounds compares, multiply-adds, select): with two or three
  // workgroups per CU they competed with the other workgroups' MFMAs for the SIMD's vector issue.
  const int chunk = tid & 3;  // which 16 B of the 64 B K-slice
  uint32_t a_vo[2], a_nok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) row_setup((uint32_t)(m0 + (tid >> 2) + 64 * i), M, d, chunk * 16, a_vo[i], a_nok[i]);
  if (tid < BM) {
    const uint32_t m = m0 + tid;
    uint32_t off = 0xFFFFFFFFu;
    if (m < M) {
      const uint32_t t = fastdiv(m, d.gw_magic, d.gw_shift);
      const int ox = (int)(m - t * (uint32_t)d.grid_w);
      const int n = (int)fastdiv(t, d.gh_magic, d.gh_shift);
      const int oy = (int)(t - (uint32_t)n * (uint32_t)d.grid_h);
      off = (uint32_t)(((n * d.dst_h + (oy * d.dst_sy + d.dst_oy)) * d.dst_w + (ox * d.dst_sx + d.dst_ox)) *
                       d.dst_pix_stride);
    }
    row_dst[tid] = off;
  }

  const int K = d.k_per_tap * d.n_taps;
  const int WRS = d.wgt_row_stride > 0 ? d.wgt_row_stride : K;   // weight row stride in elements
  const int KT = K / BK;
  const int tiles_per_tap = d.k_per_tap / BK;
  const __amdgpu_buffer_rsrc_t w_srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p_wgt), 0, 0x7FFFFFFF, 0x00020000);

  // accumulators start at the bias of their column (lane & 31 is the column of a 32x32 MFMA tile): the
  // epilogue then has no bias pass
  f32x16 acc[2][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int bc = n0 + wn * WN + j * 32 + (lane & 31);
    const float bv = (d.bias && bc < d.n_cols) ? d.bias[bc] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
  }

  const int frag_row = lane & 31, frag_k = (lane >> 5) * 8;
#define RART_COMPUTE(BUF)                                                                                       \
  {                                                                                                             \
    const uint16_t* A = sA + (BUF)*BM * LDK + (wm * 64 + frag_row) * LDK + frag_k;                              \
    const uint16_t* B = sB + (BUF)*BN * LDK + (wn * WN + frag_row) * LDK + frag_k;                              \
    _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                                    \
      bf16x8 af[2], bfr[TN];                                                                                    \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(A + i * 32 * LDK + ks * 16); \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(B + j * 32 * LDK + ks * 16); \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                          \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);               \
    }                                                                                                           \
  }

  if constexpr (BK == 32) {
  // Two register sets: global loads run TWO K steps ahead of the MFMAs (the K step of a 1x1 layer is far
  // shorter than an HBM round trip, so one step of cover leaves the kernel latency-bound at 3 blocks/CU).
  // Written as macros over named register arrays so every index is static (no scratch).
  uint32_t w_vo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + (tid >> 2) + 64 * i;
    w_vo[i] = n < d.n_cols ? (uint32_t)((n * WRS + chunk * 8) * 2) : RART_DMA_OOR;        // rows past the table: zeros (they only reach columns never stored)
  }
  uint4 ra0_0, ra0_1, ra1_0, ra1_1, rb0_0, rb0_1, rb1_0, rb1_1;   // set{0,1} x chunk{0,1}; explicit scalars
  rb0_1 = rb1_1 = make_uint4(0, 0, 0, 0);
  ra1_0 = ra1_1 = rb1_0 = make_uint4(0, 0, 0, 0);
#define RART_LOAD_A(I, DST) DST = buf_load16(srd_, a_vo[I] | ((uint32_t)__builtin_amdgcn_sbfe(a_nok[I], tap, 1) & RART_DMA_OOR), kc_);
#define RART_LOAD_B(I, DST) DST = buf_load16(w_srd, w_vo[I], (uint32_t)(kt_ * BK * 2));
#define RART_LOAD_TILE(KT_, SET)                                                                                \
  {                                                                                                             \
    const int kt_ = (KT_);                                                                                      \
    const int tap = (int)fastdiv((uint32_t)kt_, d.tpt_magic, d.tpt_shift);                                      \
    const uint32_t kc_ = (uint32_t)((kt_ - tap * tiles_per_tap) * BK * 2);                                      \
    const __amdgpu_buffer_rsrc_t srd_ = tap_rsrc(p_src, d, tap);                                                \
    RART_LOAD_A(0, ra##SET##_0)                                                                                 \
    RART_LOAD_A(1, ra##SET##_1)                                                                                 \
    RART_LOAD_B(0, rb##SET##_0)                                                                                 \
    if constexpr (B_CHUNKS > 1) RART_LOAD_B(1, rb##SET##_1)                                                     \
  }
#define RART_STORE_TILE(BUF, SET)                                                                               \
  {                                                                                                             \
    *reinterpret_cast<uint4*>(sA + (((BUF)*BM + (tid >> 2)) * LDK + chunk * 8)) = ra##SET##_0;                  \
    *reinterpret_cast<uint4*>(sA + (((BUF)*BM + (tid >> 2) + 64) * LDK + chunk * 8)) = ra##SET##_1;             \
    *reinterpret_cast<uint4*>(sB + (((BUF)*BN + (tid >> 2)) * LDK + chunk * 8)) = rb##SET##_0;                  \
    if constexpr (B_CHUNKS > 1)                                                                                 \
      *reinterpret_cast<uint4*>(sB + (((BUF)*BN + (tid >> 2) + 64) * LDK + chunk * 8)) = rb##SET##_1;           \
  }

  RART_LOAD_TILE(0, 0);
  if (KT > 1) RART_LOAD_TILE(1, 1);
  RART_STORE_TILE(0, 0);
  __syncthreads();
  // step kt on LDS buffer kt&1: prefetch tile kt+2 into the register set whose tile (kt) is already in LDS,
  // multiply, publish tile kt+1 from the other set
  for (int kt = 0; kt < KT; kt += 2) {
    if (kt + 2 < KT) RART_LOAD_TILE(kt + 2, 0);
    RART_COMPUTE(0);
    if (kt + 1 < KT) RART_STORE_TILE(1, 1);
    __syncthreads();
    if (kt + 1 < KT) {
      if (kt + 3 < KT) RART_LOAD_TILE(kt + 3, 1);
      RART_COMPUTE(1);
      if (kt + 2 < KT) RART_STORE_TILE(0, 0);
      __syncthreads();
    }
  }

  } else {
    // ---- BK = 64 pipeline: 8 chunks per row, 32 rows per pass ----
    constexpr int NA = BM / 32, NB = BN / 32;
    const int chunk8 = tid & 7, prow = tid >> 3;
    uint32_t b_vo[NA], b_nok[NA], w_vo[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) row_setup((uint32_t)(m0 + prow + 32 * i), M, d, chunk8 * 16, b_vo[i], b_nok[i]);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = n0 + prow + 32 * i;
      w_vo[i] = n < d.n_cols ? (uint32_t)((n * WRS + chunk8 * 8) * 2) : RART_DMA_OOR;
    }
    uint4 qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;   // explicit scalars: uint4 arrays of 4 end up in scratch here
    qb2 = qb3 = make_uint4(0, 0, 0, 0);
#define RART_LD64_A(I, DST) DST = buf_load16(srd_, b_vo[I] | ((uint32_t)__builtin_amdgcn_sbfe(b_nok[I], tap, 1) & RART_DMA_OOR), kc_);
#define RART_LD64_B(I, DST) DST = buf_load16(w_srd, w_vo[I], (uint32_t)(kt_ * BK * 2));
#define RART_LOAD64(KT_)                                                                                        \
  {                                                                                                             \
    const int kt_ = (KT_);                                                                                      \
    const int tap = (int)fastdiv((uint32_t)kt_, d.tpt_magic, d.tpt_shift);                                      \
    const uint32_t kc_ = (uint32_t)((kt_ - tap * tiles_per_tap) * BK * 2);                                      \
    const __amdgpu_buffer_rsrc_t srd_ = tap_rsrc(p_src, d, tap);                                                \
    RART_LD64_A(0, qa0) RART_LD64_A(1, qa1) RART_LD64_A(2, qa2) RART_LD64_A(3, qa3)                             \
    RART_LD64_B(0, qb0) RART_LD64_B(1, qb1)                                                                     \
    if constexpr (NB > 2) { RART_LD64_B(2, qb2) RART_LD64_B(3, qb3) }                                           \
  }
#define RART_ST64(PTR, ROWS, I, SRC) *reinterpret_cast<uint4*>(PTR + (((BUF_)*ROWS + prow + 32 * (I)) * LDK + chunk8 * 8)) = SRC;
#define RART_STORE64(BUF)                                                                                       \
  {                                                                                                             \
    const int BUF_ = (BUF);                                                                                     \
    RART_ST64(sA, BM, 0, qa0) RART_ST64(sA, BM, 1, qa1) RART_ST64(sA, BM, 2, qa2) RART_ST64(sA, BM, 3, qa3)     \
    RART_ST64(sB, BN, 0, qb0) RART_ST64(sB, BN, 1, qb1)                                                         \
    if constexpr (NB > 2) { RART_ST64(sB, BN, 2, qb2) RART_ST64(sB, BN, 3, qb3) }                               \
  }
    RART_LOAD64(0);
    RART_STORE64(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < KT) RART_LOAD64(kt + 1);
      RART_COMPUTE(buf);
      if (kt + 1 < KT) RART_STORE64(buf ^ 1);
      __syncthreads();
    }
#undef RART_LOAD64
#undef RART_STORE64
#undef RART_LD64_A
#undef RART_LD64_B
#undef RART_ST64
  }
#undef RART_LOAD_TILE
#undef RART_LOAD_A
#undef RART_LOAD_B
#undef RART_STORE_TILE
#undef RART_COMPUTE

  // ---- BatchNorm batch statistics of the tile (train-mode forward): sum and sum of squares of the values AS STORED (bf16, round to
  //      nearest even) per column; rows past M hold exact zeros (no bias on this path).  lane = column of a 32 x 32 MFMA tile, the 16
  //      registers x 2 passes are 32 of the wave's 64 rows, the other lane half holds the other 32: one shuffle; the two row waves of a
  //      column meet in LDS after the epilogue (fixed order: deterministic).
  constexpr int STAT_OFF = 4 * 32 * (BN / 2 + 4) * 4;
  static_assert(STAT_OFF + 4 * BN * 4 <= kLdsBytes, "the statistics slots must fit behind the epilogue staging");
  float* const sStat = reinterpret_cast<float*>(lds_raw + STAT_OFF);           // [wm][sum | sumsq][BN]
  if (!PAIR && d.stats_out) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float cs = 0.f, cq = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const uint32_t pk = pack_bf16x2(acc[i][j][r], acc[i][j][r + 1]);
          const float a = __uint_as_float(pk << 16), b = __uint_as_float(pk & 0xFFFF0000u);
          cs += a + b;
          cq = fmaf(a, a, cq);
          cq = fmaf(b, b, cq);
        }
      cs += __shfl_xor(cs, 32, 64);
      cq += __shfl_xor(cq, 32, 64);
      if (lane < 32) {
        sStat[(wm * 2 + 0) * BN + wn * WN + j * 32 + lane] = cs;
        sStat[(wm * 2 + 1) * BN + wn * WN + j * 32 + lane] = cq;
      }
    }
  }

  // ---- epilogue: each wave transposes its own 64 x WN sub-tile through a private LDS region, 32 rows at a
  //      time, so the four waves drain concurrently and no block barrier sits between MFMAs and stores (the K
  //      loop's last barrier already fenced the tile buffers).  fp32 staging -> bias / residual / ReLU mask /
  //      activation on 16-byte rows -> coalesced 16 B stores (128 B or 64 B per row per instruction).
  constexpr int LDW = WN + 4;                    // staging row (floats)
  constexpr int CW = WN / 8;                     // 8-column chunks per sub-tile row (8 / 4)
  constexpr int RPI = 64 / CW;                   // rows covered by one wave-wide access (8 / 16)
  constexpr int NQ = 32 / RPI;                   // accesses per 32-row pass (4 / 2)
  static_assert(4 * 32 * LDW * 4 <= kLdsBytes, "epilogue staging must fit the tile buffers");
  float* sW = reinterpret_cast<float*>(lds_raw) + wave * 32 * LDW;
  const int cw = lane % CW, rw0 = lane / CW;
  const int col = n0 + wn * WN + cw * 8;
  const bool col_ok = col < d.n_cols;
  const bool relu = d.flags & F_RELU, out_f32 = d.flags & F_OUT_F32, mask_bits = d.flags & F_MASK_BITS;
  uint8_t* const p_sign = d.sign_out;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // issue the residual / mask reads of this pass first: their HBM latency overlaps the LDS transposition
    uint32_t offs[NQ];
    uint4 rv[NQ], mv[NQ], rl[PAIR ? NQ : 1];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const uint32_t off = col_ok ? row_dst[wm * 64 + i * 32 + q * RPI + rw0] : 0xFFFFFFFFu;
      offs[q] = off;
      rv[q] = make_uint4(0, 0, 0, 0);
      if constexpr (PAIR) rl[q] = make_uint4(0, 0, 0, 0);
      mv[q] = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);  // bf16 1.0 pairs: mask passes
      if (off != 0xFFFFFFFFu) {
        const size_t bo = (size_t)((off + (uint32_t)col) * 2u);
        if (p_res) {
          rv[q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p_res) + bo);
          if constexpr (PAIR) rl[q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p_res + d.res_pair_off) + bo);
        }
        if (p_mask) {
          if (mask_bits) mv[q].x = reinterpret_cast<const uint8_t*>(p_mask)[(off + (uint32_t)col) >> 3];
          else mv[q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p_mask) + bo);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        sW[row * LDW + j * 32 + (lane & 31)] = acc[i][j][r];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const uint32_t off = offs[q];
      const int r = q * RPI + rw0;
      const float4 v0 = *reinterpret_cast<const float4*>(sW + r * LDW + cw * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(sW + r * LDW + cw * 8 + 4);
      if (off != 0xFFFFFFFFu) {
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const uint32_t rw[4] = {rv[q].x, rv[q].y, rv[q].z, rv[q].w};
        const uint32_t mw[4] = {mv[q].x, mv[q].y, mv[q].z, mv[q].w};
        if (p_res) {
          if constexpr (PAIR) {   // hi + lo is exact in fp32 (two 8-bit significands, lo below half an ulp of hi)
            const uint32_t rlw[4] = {rl[q].x, rl[q].y, rl[q].z, rl[q].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[2 * j] += __uint_as_float(rw[j] << 16) + __uint_as_float(rlw[j] << 16);
              v[2 * j + 1] += __uint_as_float(rw[j] & 0xFFFF0000u) + __uint_as_float(rlw[j] & 0xFFFF0000u);
            }
          } else {
            const bool mres = d.flags & F_MASK_RES;     // the 1-bit mask applies to the RESIDUAL (train engine: the skip gradient g = d_out . [out > 0])
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t rj = mres ? (rw[j] & halves_from_bits(mw[0], j)) : rw[j];
              v[2 * j] += __uint_as_float(rj << 16);
              v[2 * j + 1] += __uint_as_float(rj & 0xFFFF0000u);
            }
          }
        }
        if (d.flags & F_GELU) {   // exact (erf) GELU, timm's nn.GELU default
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_erf(v[j]);
        }
        if (d.flags & F_GELU_BWD) {   // `mask` holds the GELU's pre-activation u: v *= gelu'(u) (ViT fc2 backward-to-input)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[2 * j] *= gelu_grad_erf(__uint_as_float(mw[j] << 16));
            v[2 * j + 1] *= gelu_grad_erf(__uint_as_float(mw[j] & 0xFFFF0000u));
          }
        }
        if (out_f32) {
          if (p_mask && !(d.flags & F_GELU_BWD)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (!((short)(mw[j] & 0xFFFF) > 0)) v[2 * j] = 0.f;
              if (!((short)(mw[j] >> 16) > 0)) v[2 * j + 1] = 0.f;
            }
          }
          if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          float* o = reinterpret_cast<float*>(d.dst) + p_dst_off + off + col;
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else if constexpr (PAIR) {
          // split-bf16 output: 1-bit mask and ReLU on the fp32 value, then hi = bf16(v), lo = bf16(v - hi): 16 significand bits
          uint32_t oh[4], ol[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (p_mask) {
              if (!((mw[0] >> (2 * j)) & 1u)) v[2 * j] = 0.f;
              if (!((mw[0] >> (2 * j + 1)) & 1u)) v[2 * j + 1] = 0.f;
            }
            if (relu) {
              v[2 * j] = fmaxf(v[2 * j], 0.f);
              v[2 * j + 1] = fmaxf(v[2 * j + 1], 0.f);
            }
            oh[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
            ol[j] = pack_bf16x2(v[2 * j] - __uint_as_float(oh[j] << 16), v[2 * j + 1] - __uint_as_float(oh[j] & 0xFFFF0000u));
          }
          if (p_sign) {
            const uint32_t sb = bits_from_halves(oh[0]) | (bits_from_halves(oh[1]) << 2) | (bits_from_halves(oh[2]) << 4) |
                                (bits_from_halves(oh[3]) << 6);
            p_sign[(off + (uint32_t)col) >> 3] = (uint8_t)sb;
          }
          uint16_t* const o16 = reinterpret_cast<uint16_t*>(d.dst) + p_dst_off + off + col;
          *reinterpret_cast<uint4*>(o16) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
          *reinterpret_cast<uint4*>(o16 + d.dst_pair_off) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
        } else {
          // bf16 output: hardware RNE pack (v_cvt_pk_bf16_f32), then mask and ReLU on the packed pairs with
          // 16-bit integer ops (a bf16 is > 0 exactly when its bits, read as int16, are > 0)
          uint32_t o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
            if (p_mask && !(d.flags & (F_GELU_BWD | F_MASK_RES))) o[j] &= mask_bits ? halves_from_bits(mw[0], j) : positive_lanes_i16(mw[j]);
            if (relu) o[j] = relu_bf16x2(o[j]);
          }
          if (p_sign) {
            const uint32_t sb = bits_from_halves(o[0]) | (bits_from_halves(o[1]) << 2) | (bits_from_halves(o[2]) << 4) |
                                (bits_from_halves(o[3]) << 6);
            p_sign[(off + (uint32_t)col) >> 3] = (uint8_t)sb;
          }
          {                      // (round 6) a non-temporal store: +1.5 % on adv_train, see csrc/train_convbn.hip
            typedef __attribute__((ext_vector_type(4))) uint32_t ig_u4;
            ig_u4 w_ = {o[0], o[1], o[2], o[3]};
            __builtin_nontemporal_store(w_, reinterpret_cast<ig_u4*>(reinterpret_cast<uint16_t*>(d.dst) + p_dst_off + off + col));
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (!PAIR && d.stats_out) {          // block-uniform
    __syncthreads();
    if (tid < BN && n0 + tid < d.n_cols) {
      float* o = d.stats_out + (size_t)m_tile * 2 * d.n_cols + n0 + tid;
      o[0] = sStat[(0 * 2 + 0) * BN + tid] + sStat[(1 * 2 + 0) * BN + tid];
      o[d.n_cols] = sStat[(0 * 2 + 1) * BN + tid] + sStat[(1 * 2 + 1) * BN + tid];
    }
  }
}

// ---- 256 x 256 x 64 GEMM for the transformer layers ------------------------------------------------------------------------
// C[M][N] = A[M][K] . W[N][K]^T (+ bias, + residual, GELU / GELU') for PLAIN products with enough tiles to fill the chip (ViT-B/16
// at B = 256: M = 50 432, N = 768 ... 3 072: 600-2 400 tiles).  The 128 x 128 kernel above gives each wave a 64 x 64 sub-tile:
// 1 KiB of LDS fragment reads per MFMA, which at 8 waves per CU is exactly the LDS bandwidth (128 B / clk) the matrix pipe would
// need at full rate -- K-deep launches stall at ~700 TFLOP/s.  Here 8 waves (2 x 4) own 128 x 64 sub-tiles (0.75 KiB per MFMA),
// and the tiles arrive by global_load_lds_dwordx4: no VGPR staging, no ds_write.  A wave-wide direct load writes 8 rows x 128 B
// contiguously (lane l -> row l >> 3, slot l & 7), so rows cannot be padded; slot s of row r holds chunk s ^ ((r >> 1) & 7), the
// swizzle being applied on the GLOBAL side (each lane fetches the chunk its slot must hold: same 128-byte row, coalescing
// unchanged), which puts the 16 rows of a quarter-wave fragment read in 16 different 16-byte bank groups.  Two LDS stages of
// 64 KiB: one workgroup per CU.  Measured (scratch/exp/gemm256.hip): 880-900 TFLOP/s on the ViT shapes, 1 160 on 8192^3
// (128 x 128 kernel: 791).
struct RartGemm256Desc {
  const uint16_t* a;      // [M][lda]
  const uint16_t* w;      // [N][K]
  const float* bias;      // [N] or null
  const uint16_t* res;    // [M][ldc] or null
  const uint16_t* mask;   // F_GELU_BWD: the GELU's pre-activation [M][ldc]
  uint16_t* c;            // [M][ldc]
  int M, N, K, lda, ldc, flags;
  int tile_rows;          // k_gemm256_pp: 256, or 224 = a tile steps 224 rows and leaves its last 32-row block out (csrc/gemm_pair.hip, round 6)
};
__device__ __attribute__((aligned(16))) const uint32_t g_gemm_zero16[4] = {0u, 0u, 0u, 0u};   // source of rows past M
constexpr int G2_TM = 256, G2_TN = 256, G2_STAGE = (G2_TM + G2_TN) * 128, G2_LDE = 68;
static_assert(8 * 32 * G2_LDE * 4 <= 2 * G2_STAGE, "epilogue staging must fit the tile buffers");

// Epilogue of a 256 x 256 tile of the transformer GEMMs (k_gemm256_bf16 and its ping-pong form): per wave, 32 rows x 64 columns at a time
// through LDS -> 128-byte row segments; residual / GELU operands of a pass are requested before its transposition.  `lds`: the tile buffers
// (free after the K loop); must be called by every thread.
// round 6: the 256 x 256 GEMM's outputs (77-310 MB per launch at ViT-B/16's shapes) are non-temporal stores: ViT-B/16 bf16 forward + backward
// 28.19 -> 27.59 ms, forward 12.35 -> 12.04 ms (same-box A/B of lab builds)
#define RART_G2_ST(P, V) rart_nt_store16((P), (V))
#ifdef RART_G2_NT_LD     // lab build: the residual / pre-activation loads too
#define RART_G2_LD(P) rart_nt_load16(P)
#else
#define RART_G2_LD(P) (*reinterpret_cast<const uint4*>(P))
#endif
__device__ __forceinline__ void g2_epilogue(const RartGemm256Desc& d, uint8_t* lds, f32x16 (&acc)[4][2], int m0, int n0, int tile_rows = 256) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 31, h = lane >> 5;
  float* sE = reinterpret_cast<float*>(lds) + wave * 32 * G2_LDE;
  const int cw = lane & 7, rw = lane >> 3;
  const int col = n0 + wn * 64 + cw * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (wm * 128 + i * 32 >= tile_rows) continue;         // (wave-uniform) a 224-row tile: the block belongs to the next tile
    uint4 rv[4], mv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = m0 + wm * 128 + i * 32 + q * 8 + rw;
      rv[q] = make_uint4(0, 0, 0, 0);
      mv[q] = make_uint4(0, 0, 0, 0);
      if (row < d.M) {
        const size_t e = (size_t)row * d.ldc + col;
        if (d.res) rv[q] = RART_G2_LD(d.res + e);
        if (d.flags & F_GELU_BWD) mv[q] = RART_G2_LD(d.mask + e);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sE[((r & 3) + 8 * (r >> 2) + 4 * h) * G2_LDE + j * 32 + fr] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = q * 8 + rw, row = m0 + wm * 128 + i * 32 + r;
      const float4 v0 = *reinterpret_cast<const float4*>(sE + r * G2_LDE + cw * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(sE + r * G2_LDE + cw * 8 + 4);
      if (row < d.M) {
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const uint32_t rr[4] = {rv[q].x, rv[q].y, rv[q].z, rv[q].w};
        const uint32_t mw[4] = {mv[q].x, mv[q].y, mv[q].z, mv[q].w};
        if (d.res) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[2 * j] += __uint_as_float(rr[j] << 16);
            v[2 * j + 1] += __uint_as_float(rr[j] & 0xFFFF0000u);
          }
        }
        if (d.flags & F_GELU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_erf(v[j]);
        }
        if (d.flags & F_GELU_KEEP) {
          // two outputs: the bf16 pre-activation u (the backward's GELU' operand) goes to `mask`, dst receives gelu(u) of the ROUNDED u --
          // bit-identical to writing u and running k_gelu over it, without the second pass over the hidden tensor
          const uint4 up = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
          RART_G2_ST(const_cast<uint16_t*>(d.mask) + (size_t)row * d.ldc + col, up);
          const uint32_t uw[4] = {up.x, up.y, up.z, up.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[2 * j] = gelu_erf(__uint_as_float(uw[j] << 16));
            v[2 * j + 1] = gelu_erf(__uint_as_float(uw[j] & 0xFFFF0000u));
          }
        }
        if (d.flags & F_GELU_BWD) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[2 * j] *= gelu_grad_erf(__uint_as_float(mw[j] << 16));
            v[2 * j + 1] *= gelu_grad_erf(__uint_as_float(mw[j] & 0xFFFF0000u));
          }
        }
        RART_G2_ST(d.c + (size_t)row * d.ldc + col,
                   make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(512, 1) void k_gemm256_bf16(const RartGemm256Desc d) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * G2_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int n_tiles = d.N / G2_TN;
  // all column tiles of a row tile on one XCD (the A tile is re-read from its L2)
  const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
  const int m_tile = (slot / n_tiles) * 8 + xcd, n_tile = slot % n_tiles;
  if (m_tile * G2_TM >= d.M) return;
  const int m0 = m_tile * G2_TM, n0 = n_tile * G2_TN;
  const int lrow = lane >> 3, swz = ((8 * wave + lrow) >> 1) & 7, csrc = (lane & 7) ^ swz;
  const uint32_t wrow = (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 8u;
  const char* asrc[4];
  const char* bsrc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 64 * q + 8 * wave + lrow;
    asrc[q] = (m0 + r < d.M) ? reinterpret_cast<const char*>(d.a + (size_t)(m0 + r) * d.lda + csrc * 8) : nullptr;
    bsrc[q] = reinterpret_cast<const char*>(d.w + (size_t)(n0 + r) * d.K + csrc * 8);
  }
#define RART_G2_DL(SRC, DST)                                                                                    \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),                        \
                                   (__attribute__((address_space(3))) void*)(DST), 16, 0, 0);
#define RART_G2_ISSUE(KT, BUF)                                                                                  \
  {                                                                                                             \
    uint8_t* const st_ = lds + (BUF)*G2_STAGE;                                                                  \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                             \
      const char* s_ = asrc[q] ? asrc[q] + (size_t)(KT)*128 : reinterpret_cast<const char*>(g_gemm_zero16);     \
      RART_G2_DL(s_, st_ + (64 * q + wrow) * 128)                                                               \
    }                                                                                                           \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                             \
      RART_G2_DL(bsrc[q] + (size_t)(KT)*128, st_ + (G2_TM + 64 * q + wrow) * 128)                               \
    }                                                                                                           \
  }
  const int fr = lane & 31, h = lane >> 5;
  uint32_t xo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xo[ks] = (uint32_t)(fr * 128 + (((2 * ks + h) ^ ((fr >> 1) & 7)) << 4));
  // accumulators start at the bias of their column (lane & 31 is the column of a 32 x 32 MFMA tile), as in k_conv_igemm_bf16
  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float bv = d.bias ? d.bias[n0 + wn * 64 + j * 32 + fr] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
  }
  const int KT = d.K / 64;
  RART_G2_ISSUE(0, 0)
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) RART_G2_ISSUE(kt + 1, buf ^ 1)
    const uint8_t* Ab = lds + buf * G2_STAGE + wm * 128 * 128;
    const uint8_t* Bb = lds + buf * G2_STAGE + (G2_TM + wn * 64) * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[4], bfr[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * 128 + xo[ks]);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * 128 + xo[ks]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);          // the next tile has landed in LDS
    __syncthreads();
  }
#undef RART_G2_ISSUE
#undef RART_G2_DL
  g2_epilogue(d, lds, acc, m0, n0);
}

// ---- the same GEMM on the PING-PONG schedule (round 6; csrc/gemm_pair_pp.hip has the derivation and the measurements that led to it):
//      the tile loads cost k_gemm256_bf16 a third of its time although nothing waits for them (1 244-1 342 TFLOP/s without them, 880-900
//      with) -- all eight waves issue their eight LDS-DMA loads together, ~2 000 cycles of texture-address time in front of 2 048 cycles of
//      matrix work per SIMD.  Here the two wave groups (rows 0-127 / 128-255 of the tile; waves w and w + 4 share a SIMD) run one barrier
//      interval apart through  M0 | C0 | M1 | C1  per 64-deep K step: while one group issues 16 MFMAs (s_setprio 1) the other reads its
//      fragments and issues its share of the next stages' loads; the loads are buffer_load ... lds with ONE constant per-lane byte offset
//      and a scalar K offset (no vector instruction on the issue path, rows past M zero-filled by the range check, rart_lds_dma.h); every
//      8 KB region (a group's half of the A rows, half of a group's W rows) is refilled in the slot after its last reader's barrier and
//      waited for with a counted vmcnt one barrier before its first reader.  Same products in the same order, same epilogue:
//      BIT-IDENTICAL to k_gemm256_bf16 (tests/test_vit_gpu.py).
__device__ __forceinline__ void g2_slot_end() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__global__ __launch_bounds__(512, 1) void k_gemm256_pp(const RartGemm256Desc d) {
  constexpr int PLANE = G2_TM * 128;                      // the A rows of a stage, then the W rows: 32 KB each
  constexpr int NB = 4;                                   // loads of a wave per memory phase: 2 W pieces + 2 A pieces
  constexpr int VM_ALPHA = 3 * NB, VM_BETA = 2 + NB, VM_GAMMA = 2 + NB;
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * G2_STAGE];
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
  const int g = wm, w4 = wn, og = 1 - g;
  const int n_tiles = d.N / G2_TN;
  const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
  const int m_tile = (slot / n_tiles) * 8 + xcd, n_tile = slot % n_tiles;
  const int TMV = d.tile_rows;
  if (m_tile * TMV >= d.M) return;
  const int m0 = m_tile * TMV, n0 = n_tile * G2_TN;
  const bool skip_last = wm == 1 && TMV < G2_TM;         // (wave-uniform) rows 224 .. 255 of a 224-row tile are the next tile's
  // ---- loader: a piece = 8 rows x 128 B (lane -> row lane >> 3, slot lane & 7 <- the row's chunk (lane & 7) ^ ((row >> 1) & 7)).  A wave
  //      loads A rows of the OTHER group only: pieces 2 w4, 2 w4 + 1 of its two 64-row regions (q = 0: half `og`, refilled in M0; q = 1:
  //      half `g`, refilled in M1), and W pieces 4 w4 .. 4 w4 + 3 of its OWN group's 128 W rows (the first two in M1, the last two in M0)
  uint32_t av[2][2], wv[4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = og * 128 + (q == 0 ? og : g) * 64 + 8 * (2 * w4 + p) + (lane >> 3);
      const int csrc = (lane & 7) ^ ((r >> 1) & 7);
      av[q][p] = (m0 + r < d.M && r < TMV) ? (uint32_t)(((m0 + r) * d.lda + csrc * 8) * 2) : RART_DMA_OOR;   // (rows past a 224-row tile: not fetched)
    }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = g * 128 + 8 * (4 * w4 + p) + (lane >> 3);
    const int csrc = (lane & 7) ^ ((r >> 1) & 7);
    wv[p] = (uint32_t)(((n0 + r) * d.K + csrc * 8) * 2);
  }
  const rart_srd_t srd_a = rart_dma_srd(d.a), srd_w = rart_dma_srd(d.w);
  const int KT = d.K / 64;
#define RART_G2P_ISSUE_A(Q, ST)                                                                                  \
  {                                                                                                              \
    const int s_ = (ST);                                                                                         \
    const uint32_t dst_ = lds_base + (s_ & 1) * G2_STAGE + (og * 128 + ((Q) == 0 ? og : g) * 64 + 16 * w4) * 128; \
    rart_dma_load16(av[Q][0], srd_a, (uint32_t)s_ * 128u, dst_);                                                 \
    rart_dma_load16(av[Q][1], srd_a, (uint32_t)s_ * 128u, dst_ + 1024);                                          \
  }
#define RART_G2P_ISSUE_W(ST, P0, P1)                                                                             \
  {                                                                                                              \
    const int s_ = (ST);                                                                                         \
    const uint32_t dst_ = lds_base + (s_ & 1) * G2_STAGE + PLANE + (g * 128 + 32 * w4) * 128;                    \
    _Pragma("unroll") for (int p = (P0); p < (P1); ++p) rart_dma_load16(wv[p], srd_w, (uint32_t)s_ * 128u, dst_ + p * 1024); \
  }
  const int fr = lane & 31, h = lane >> 5;
  uint32_t xo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xo[ks] = (uint32_t)(fr * 128 + (((2 * ks + h) ^ ((fr >> 1) & 7)) << 4));
  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float bv = d.bias ? d.bias[n0 + wn * 64 + j * 32 + fr] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
  }
  // ---- prologue: stage 0 whole, stage 1 except what the first M0 phases issue
  RART_G2P_ISSUE_W(0, 0, 4)
  RART_G2P_ISSUE_A(0, 0)
  RART_G2P_ISSUE_A(1, 0)
  if (KT > 1) {
    RART_G2P_ISSUE_W(1, 0, 2)
    RART_G2P_ISSUE_A(1, 1)
    if (g) RART_G2P_ISSUE_A(0, 1)
  }
  rart_dma_wait<0>();
  __syncthreads();
  if (g) g2_slot_end();                     // the stagger: group 1 runs one slot behind group 0
  bf16x8 af[2][4], bfr[2][4];
#define RART_G2P_READ_A(HALF)                                                                                    \
  _Pragma("unroll") for (int ii = 0; ii < 2; ++ii) if (!((HALF) == 1 && ii == 1 && skip_last))                  \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                          \
          af[ii][ks] = *reinterpret_cast<const bf16x8*>(Ab + ((HALF)*2 + ii) * 32 * 128 + xo[ks]);
#define RART_G2P_MFMA(HALF)                                                                                      \
  __builtin_amdgcn_s_setprio(1);                                                                                 \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)             \
      if (!((HALF) == 1 && ii == 1 && skip_last)) _Pragma("unroll") for (int j = 0; j < 2; ++j)                 \
          acc[(HALF)*2 + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ii][ks], bfr[j][ks], acc[(HALF)*2 + ii][j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);
  for (int kt = 0; kt < KT; ++kt) {
    const uint8_t* const Ab = lds + (kt & 1) * G2_STAGE + wm * 128 * 128;
    const uint8_t* const Bb = lds + (kt & 1) * G2_STAGE + PLANE + wn * 64 * 128;
    const int tail = kt + 2 >= KT;          // some refill of this K step is skipped: the counts below do not hold, drain instead
    // ---- M0
    if (kt + 1 < KT) RART_G2P_ISSUE_W(kt + 1, 2, 4)
    if (kt + 1 + g < KT) RART_G2P_ISSUE_A(0, kt + 1 + g)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bfr[j][ks] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * 128 + xo[ks]);
    RART_G2P_READ_A(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    g2_slot_end();
    // ---- C0
    RART_G2P_MFMA(0)
    g2_slot_end();
    // ---- M1
    if (kt + 2 < KT) {
      RART_G2P_ISSUE_W(kt + 2, 0, 2)
      RART_G2P_ISSUE_A(1, kt + 2)
    }
    RART_G2P_READ_A(1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (tail) rart_dma_wait<0>();
    else if (g) rart_dma_wait<VM_GAMMA>();
    else rart_dma_wait<VM_ALPHA>();
    g2_slot_end();
    // ---- C1
    RART_G2P_MFMA(1)
    if (!g) {
      if (tail) rart_dma_wait<0>(); else rart_dma_wait<VM_BETA>();
    }
    g2_slot_end();
  }
  if (!g) g2_slot_end();                    // group 0's closing barrier pairs with group 1's last one
#undef RART_G2P_MFMA
#undef RART_G2P_READ_A
#undef RART_G2P_ISSUE_W
#undef RART_G2P_ISSUE_A
  rart_dma_wait<0>();
  __syncthreads();
  g2_epilogue(d, lds, acc, m0, n0, TMV);
}
}  // namespace

// tuning knob (tests / profiling): smallest K that takes the BK = 64 pipeline; a huge value disables it
static long long g_bk64_min_k = 256;    // A/B on MI355X (scratch/ab_bk64.py): ResNet-50 fwd+bwd 13.2 (never) / 12.0 (1024) / 11.9 ms (256)
extern "C" int rart_igemm_set_bk64_min_k(long long k) {
  g_bk64_min_k = k;
  return RART_OK;
}

// tuning knob (tests / profiling): 0 keeps every problem on the 128 x 128 kernel, 1 = the 256 x 256 x 64 kernel on round 2's two-stage loop,
// 2 (default) = the same kernel on the ping-pong schedule of round 6 (bit-identical outputs)
static int g_gemm256_enabled = 2;
static const int g_gemm256_rows224 = getenv("RART_GEMM256_ROWS224") ? atoi(getenv("RART_GEMM256_ROWS224")) : 1;    // lab switch
extern "C" int rart_igemm_set_gemm256(int enable) {
  g_gemm256_enabled = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
  return RART_OK;
}

// plain row-major products with at least two rounds of 256 x 256 tiles take k_gemm256_bf16
static bool gemm256_takes(const rart_conv_desc* h, long long* m_out) {
  if (!g_gemm256_enabled || h->n_taps != 1 || h->tap_dy[0] != 0 || h->tap_dx[0] != 0 || h->tap_src_off[0] != 0) return false;
  if (h->sy != 1 || h->sx != 1 || h->grid_w != 1 || h->src_w != 1 || h->dst_w != 1 || h->dst_sy != 1 || h->dst_oy != 0 || h->dst_ox != 0)
    return false;
  if (h->src_h != h->grid_h || h->dst_h != h->grid_h || h->n_batched > 1 || h->sign_out) return false;
  if (h->wgt_row_stride != 0 && h->wgt_row_stride != h->k_per_tap) return false;
  if (h->flags & ~(F_GELU | F_GELU_BWD | F_GELU_KEEP)) return false;
  if ((h->flags & (F_GELU_BWD | F_GELU_KEEP)) ? !h->mask : (h->mask != nullptr)) return false;
  if ((h->flags & F_GELU_KEEP) && (h->flags & (F_GELU | F_GELU_BWD))) return false;
  if (h->k_per_tap % 64 != 0 || h->n_cols % G2_TN != 0 || h->src_pix_stride % 8 != 0 || h->dst_pix_stride % 8 != 0) return false;
  const long long M = (long long)h->batch * h->grid_h;
  if (M * h->src_pix_stride >= (1ll << 31) || M * h->dst_pix_stride >= (1ll << 31)) return false;
  if (((M + G2_TM - 1) / G2_TM) * (h->n_cols / G2_TN) < 512) return false;
  *m_out = M;
  return true;
}

// would a plain rows x k (leading dimension src_ld) by n_cols product (output leading dimension dst_ld) take the 256 x 256 GEMM?
extern "C" int rart_gemm256_supported(long long rows, int k, int n_cols, int src_ld, int dst_ld) {
  if (!g_gemm256_enabled || rows <= 0 || k <= 0 || k % 64 != 0 || n_cols <= 0 || n_cols % G2_TN != 0 || src_ld % 8 != 0 || dst_ld % 8 != 0) return 0;
  if (rows * src_ld >= (1ll << 31) || rows * dst_ld >= (1ll << 31)) return 0;
  return ((rows + G2_TM - 1) / G2_TM) * (n_cols / G2_TN) >= 512 ? 1 : 0;
}

extern "C" int rart_conv_igemm_bf16(const rart_conv_desc* h, rart_stream_t stream) {
  RART_CHECK_ARG(h != nullptr, "rart_conv_igemm_bf16: null descriptor");
  RART_CHECK_ARG(h->src && h->wgt && h->dst, "rart_conv_igemm_bf16: null tensor pointer");
  RART_CHECK_ARG(h->batch > 0 && h->grid_h > 0 && h->grid_w > 0, "rart_conv_igemm_bf16: empty row grid");
  RART_CHECK_ARG(h->n_taps >= 1 && h->n_taps <= 32, "rart_conv_igemm_bf16: n_taps must be 1..32");
  RART_CHECK_ARG(h->k_per_tap > 0 && h->k_per_tap % 32 == 0, "rart_conv_igemm_bf16: k_per_tap must be a multiple of 32");
  RART_CHECK_ARG(h->n_cols > 0 && h->n_cols % 8 == 0, "rart_conv_igemm_bf16: n_cols must be a multiple of 8");
  RART_CHECK_ARG(h->src_pix_stride % 8 == 0 || h->src_pix_stride == 4,
                 "rart_conv_igemm_bf16: source pixels must keep 16-byte alignment of the K chunks");
  RART_CHECK_ARG(h->dst_pix_stride % 8 == 0, "rart_conv_igemm_bf16: dst_pix_stride must be a multiple of 8");
  {
    long long gm = 0;
    if (gemm256_takes(h, &gm)) {
      RartGemm256Desc g;
      g.a = (const uint16_t*)h->src; g.w = (const uint16_t*)h->wgt; g.bias = h->bias; g.res = (const uint16_t*)h->res;
      g.mask = (const uint16_t*)h->mask; g.c = (uint16_t*)h->dst;
      g.M = (int)gm; g.N = h->n_cols; g.K = h->k_per_tap; g.lda = h->src_pix_stride; g.ldc = h->dst_pix_stride; g.flags = h->flags;
      g.tile_rows = G2_TM;
      int m_tiles = (int)((gm + G2_TM - 1) / G2_TM), m8 = (m_tiles + 7) / 8 * 8;
      // the ping-pong form addresses its operands with 32-bit byte offsets (planes below 2 GiB: gemm256_takes checked 2^31 ELEMENTS)
      const bool pp = g_gemm256_enabled == 2 && gm * h->src_pix_stride < (1ll << 30) && (long long)g.N * g.K < (1ll << 30);
      if (pp && g_gemm256_rows224) {
        // a launch runs in passes of one tile per CU and XCD (workgroup i -> XCD i % 8, 32 CUs each; csrc/gemm_pair.hip, round 6): tiles that
        // step 224 rows where that does not add a pass -- ViT-B/16's 50432 x 768 outputs: 75 tiles of 256 rows or 87 of 224 per XCD, three
        // passes either way, 7/8 of the matrix work per tile
        const int n_t = g.N / G2_TN, mt224 = (int)((gm + 223) / 224);
        auto passes = [&](int mt) { return ((mt + 7) / 8 * n_t + 31) / 32; };
        if (passes(mt224) * 0.92 < passes(m_tiles) - 0.01) {
          g.tile_rows = 224;
          m_tiles = mt224;
          m8 = (m_tiles + 7) / 8 * 8;
        }
      }
      if (pp) hipLaunchKernelGGL(k_gemm256_pp, dim3((uint32_t)(m8 * (g.N / G2_TN))), dim3(512), 0, (hipStream_t)stream, g);
      else hipLaunchKernelGGL(k_gemm256_bf16, dim3((uint32_t)(m8 * (g.N / G2_TN))), dim3(512), 0, (hipStream_t)stream, g);
      RART_CHECK_LAUNCH("rart_conv_igemm_bf16 (256 x 256 GEMM)");
      return RART_OK;
    }
    if (h->flags & F_GELU_KEEP) {
      rart_set_error("rart_conv_igemm_bf16: flag 64 (GELU with the pre-activation kept) is served by the 256 x 256 GEMM only: ask "
                     "rart_gemm256_supported first");
      return RART_ERR_UNSUPPORTED;
    }
  }
  RartConvDescDev d;
  d.src = (const uint16_t*)h->src; d.wgt = (const uint16_t*)h->wgt; d.bias = h->bias;
  d.res = (const uint16_t*)h->res; d.mask = (const uint16_t*)h->mask; d.dst = h->dst;
  d.batch = h->batch; d.grid_h = h->grid_h; d.grid_w = h->grid_w;
  d.src_h = h->src_h; d.src_w = h->src_w; d.src_pix_stride = h->src_pix_stride;
  d.k_per_tap = h->k_per_tap; d.n_taps = h->n_taps; d.sy = h->sy; d.sx = h->sx;
  for (int i = 0; i < 32; ++i) { d.tap_dy[i] = h->tap_dy[i]; d.tap_dx[i] = h->tap_dx[i]; d.tap_src_off[i] = h->tap_src_off[i]; }
  d.n_cols = h->n_cols; d.dst_h = h->dst_h; d.dst_w = h->dst_w; d.dst_sy = h->dst_sy; d.dst_sx = h->dst_sx;
  d.dst_oy = h->dst_oy; d.dst_ox = h->dst_ox; d.dst_pix_stride = h->dst_pix_stride; d.flags = h->flags;
  const int nz = h->n_batched > 1 ? h->n_batched : 1;
  d.z_inner = h->z_inner > 0 ? h->z_inner : 1;
  d.wgt_row_stride = h->wgt_row_stride;
  d.src_zo = h->src_z_outer; d.src_zi = h->src_z_inner; d.wgt_zo = h->wgt_z_outer; d.wgt_zi = h->wgt_z_inner;
  d.dst_zo = h->dst_z_outer; d.dst_zi = h->dst_z_inner;
  d.sign_out = (uint8_t*)h->sign_out;
  d.stats_out = h->bn_stats_out;
  const bool pair = (h->flags & F_PAIR) != 0;
  RART_CHECK_ARG(!d.stats_out || (!pair && !d.bias && !d.res && !d.mask && !d.sign_out && h->n_batched <= 1 && d.flags == 0),
                 "rart_conv_igemm_bf16: bn_stats_out serves a plain bf16 convolution (no bias / residual / mask / flags, unbatched)");
  d.dst_pair_off = h->dst_pair_off; d.res_pair_off = h->res_pair_off;
  RART_CHECK_ARG(!pair || (nz == 1 && !(d.flags & (F_GELU | F_GELU_BWD)) && (!d.mask || (d.flags & F_MASK_BITS)) &&
                           ((d.flags & F_OUT_F32) ? (!d.res && !d.mask) : (d.dst_pair_off > 0 && d.dst_pair_off % 8 == 0)) &&
                           (!d.res || (d.res_pair_off > 0 && d.res_pair_off % 8 == 0))),
                 "rart_conv_igemm_bf16: split-bf16 (flag 32) needs an unbatched problem, 1-bit masks and 16-byte aligned lo planes");
  RART_CHECK_ARG(!(d.flags & F_MASK_RES) || (!pair && d.res && d.mask && (d.flags & F_MASK_BITS) && !(d.flags & F_OUT_F32)),
                 "rart_conv_igemm_bf16: flag 128 (mask on the residual) needs a residual, a 1-bit mask (flag 16) and bf16 output");
  RART_CHECK_ARG(!(d.sign_out || (d.flags & F_MASK_BITS)) || (nz == 1 && !(d.flags & (F_OUT_F32 | F_GELU_BWD))),
                 "rart_conv_igemm_bf16: sign_out / bit masks need a single bf16-output problem");
  {
    auto magic = [](uint32_t dv, uint32_t& mg, uint32_t& sh) {   // exact for dividends < 2^31
      uint32_t l = 0;
      while ((1ull << l) < dv) ++l;
      sh = 31 + l;
      mg = (uint32_t)(((1ull << sh) + dv - 1) / dv);
    };
    magic((uint32_t)d.grid_w, d.gw_magic, d.gw_shift);
    magic((uint32_t)d.grid_h, d.gh_magic, d.gh_shift);
    magic((uint32_t)(d.k_per_tap / ((d.k_per_tap % 64 == 0 && (long long)d.k_per_tap * d.n_taps >= g_bk64_min_k) ? 64 : 32)),
          d.tpt_magic, d.tpt_shift);
  }
  RART_CHECK_ARG(nz <= 65535, "rart_conv_igemm_bf16: n_batched must be <= 65535");
  RART_CHECK_ARG(d.wgt_row_stride == 0 || d.wgt_row_stride % 8 == 0, "rart_conv_igemm_bf16: wgt_row_stride must keep 16-byte alignment");
  const long long M = (long long)d.batch * d.grid_h * d.grid_w;
  // the kernel addresses with 32-bit element offsets from uniform bases (saves ~20 VGPRs of 64-bit math)
  const long long src_elems = (long long)d.batch * d.src_h * d.src_w * d.src_pix_stride;
  const long long dst_elems = (long long)d.batch * d.dst_h * d.dst_w * d.dst_pix_stride;
  RART_CHECK_ARG(M < (1ll << 31), "rart_conv_igemm_bf16: row grid must stay below 2^31 rows");
  RART_CHECK_ARG(src_elems < (1ll << 30) && dst_elems < (1ll << 31),
                 "rart_conv_igemm_bf16: a source plane must stay below 2 GiB (32-bit byte offsets of the buffer loads) and the destination below "
                 "2^31 elements (split the batch)");
  {
    const long long w_elems = (long long)d.n_cols * (d.wgt_row_stride > 0 ? d.wgt_row_stride : (long long)d.k_per_tap * d.n_taps);
    RART_CHECK_ARG(w_elems < (1ll << 30), "rart_conv_igemm_bf16: the weight table must stay below 2 GiB");
  }
  const int m_tiles = (int)((M + BM - 1) / BM);
  const bool wide = d.n_cols > 64;
  const int bn = wide ? 128 : 64;
  const int n_tiles = (d.n_cols + bn - 1) / bn;
  const int m_tiles8 = m_tiles >= 16 ? (m_tiles + 7) / 8 * 8 : m_tiles;  // the XCD remap enumerates row tiles in groups of 8
  const long long blocks = (long long)m_tiles8 * n_tiles;
  RART_CHECK_ARG(blocks < (1ll << 31), "rart_conv_igemm_bf16: grid too large");
  // K-deep problems take the BK = 64 pipeline (half the barriers per FLOP); shallow ones (K <= 128: the
  // memory-latency-bound 1x1 layers with 2-4 K steps) keep BK = 32 with loads two steps ahead
  const bool deep = (d.k_per_tap % 64 == 0) && ((long long)d.k_per_tap * d.n_taps >= g_bk64_min_k);
  const dim3 grid((uint32_t)blocks, nz);
  hipStream_t st = (hipStream_t)stream;
  if (pair) {
    if (wide && deep) hipLaunchKernelGGL((k_conv_igemm_bf16<128, 64, true>), grid, dim3(kThreads), 0, st, d);
    else if (wide) hipLaunchKernelGGL((k_conv_igemm_bf16<128, 32, true>), grid, dim3(kThreads), 0, st, d);
    else if (deep) hipLaunchKernelGGL((k_conv_igemm_bf16<64, 64, true>), grid, dim3(kThreads), 0, st, d);
    else hipLaunchKernelGGL((k_conv_igemm_bf16<64, 32, true>), grid, dim3(kThreads), 0, st, d);
  } else if (wide && deep) hipLaunchKernelGGL((k_conv_igemm_bf16<128, 64, false>), grid, dim3(kThreads), 0, st, d);
  else if (wide) hipLaunchKernelGGL((k_conv_igemm_bf16<128, 32, false>), grid, dim3(kThreads), 0, st, d);
  else if (deep) hipLaunchKernelGGL((k_conv_igemm_bf16<64, 64, false>), grid, dim3(kThreads), 0, st, d);
  else hipLaunchKernelGGL((k_conv_igemm_bf16<64, 32, false>), grid, dim3(kThreads), 0, st, d);
  RART_CHECK_LAUNCH("rart_conv_igemm_bf16");
  return RART_OK;
}
