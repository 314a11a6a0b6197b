// Split-bf16 ("fp32x") GEMM, PING-PONG schedule (round 6): the K loop of k_gemm_pair (csrc/gemm_pair.hip) rebuilt so that the memory
// phase of one half of the workgroup runs under the matrix phase of the other half.
//
// Why (profiles/r05_pair_knockouts.txt, scratch/r5/time_pair_ts.py): in k_gemm_pair all eight waves issue the next stage's LDS-DMA loads
// together (~2 000 cycles: 64 KB per K step through one texture-address unit), read their fragments together, multiply together
// (3 072 cycles of matrix work per SIMD) and meet at `s_waitcnt vmcnt(0)` + barrier: the phases ADD to ~5 700 cycles per K step.  The
// reference's arithmetic (fp32 logits and gradients, RobustART/noise/utils/adv/attack.py:20-23, Attacks/autoattack/autopgd_base.py:271-289)
// is reproduced by exactly the same products in exactly the same order as k_gemm_pair -- the outputs are BIT-IDENTICAL -- only the
// schedule differs:
//
//   * The eight waves form two groups (waves 0-3 = rows 0-127 of the tile, waves 4-7 = rows 128-255; wave w and w + 4 share a SIMD).
//     A K step (32 deep, four operand planes) of a wave is four phases  M0 | C0 | M1 | C1 :
//       M0  ds_read the W fragments of the step and the A fragments of the first half of the wave's row blocks, issue 2 LDS-DMA loads
//       C0  the 24 (TN = 256) MFMAs of those row blocks, s_setprio 1
//       M1  ds_read the A fragments of the second half, issue TN/64 + 2 LDS-DMA loads
//       C1  their MFMAs
//     separated by raw s_barrier; group 1 executes ONE extra barrier before its first phase, so in every barrier interval ("slot") one
//     group multiplies while the other reads / loads: the matrix pipe of a SIMD always has exactly one wave feeding it.
//   * No `vmcnt(0)` in the steady state.  The LDS-DMA loads are inline asm (hipcc neither counts them nor drains them at barriers), the
//     waits are counted: with two stage buffers every 8 KB region (a group's half of the A rows, hi + lo; a W plane) is refilled in the
//     slot after its last reader's barrier and waited for one barrier before its first reader, 4-6 slots (3 000-4 600 cycles) later:
//        slot 4kt   (g0 M0): A rows of (g1, half 1), stage kt+1        slot 4kt+1 (g1 M0): A rows of (g0, half 0), stage kt+2
//        slot 4kt+2 (g0 M1): W_hi + A rows of (g1, half 0), stage kt+2  slot 4kt+3 (g1 M1): W_lo + A rows of (g0, half 1), stage kt+2
//     Region free / needed slots and the vmcnt immediates are derived in DESIGN.md section 4.4 (round 6); gfx950 reports vector-memory
//     completion in issue order, so `vmcnt(N)` with N = the loads issued after the one needed is exact.  Every ds_read is retired
//     (lgkmcnt(0)) before the barrier that releases its region for refill.
//   * Epilogue, tile mapping, swizzle, operand conventions: k_gemm_pair's (rart_gemm_pair_dev.h).
#include "rart_common.h"
#include <stdlib.h>

#include "rart_gemm_pair_dev.h"
#include "rart_lds_dma.h"

namespace {

typedef rart_srd_t pp_srd_t;
__device__ __forceinline__ void pp_bload(uint32_t voff, pp_srd_t srd, uint32_t soff, uint32_t lds_addr) { rart_dma_load16(voff, srd, soff, lds_addr); }
__device__ __forceinline__ pp_srd_t pp_make_srd(const void* base) { return rart_dma_srd(base); }
constexpr uint32_t PP_OOR = RART_DMA_OOR;
template <int N>
__device__ __forceinline__ void pp_vmcnt() { rart_dma_wait<N>(); }
__device__ __forceinline__ void pp_slot_end() {      // phase boundary: nothing crosses it in either direction
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// Lab builds only (scratch/r6/build_pp_variants.sh): -DRART_PP_STAMPS sums s_memtime intervals per phase and group into g_pp_stamps (read by
// rart_debug_pp_stamps, an export that exists in that build alone); -DRART_PP_KO_* knock one component out of the K loop.
#ifdef RART_PP_STAMPS
__device__ unsigned long long g_pp_stamps[2][16];
__device__ unsigned long long g_pp_trace[4096][4];       // per workgroup (blockIdx.x < 4096): entry, K loop end, exit (s_memtime), HW_ID | XCC_ID << 32
#define PP_T(V) const unsigned long long V = __builtin_amdgcn_s_memtime();
#define PP_ACC(I, A, B) st_acc[I] += (B) - (A);
#else
#define PP_T(V)
#define PP_ACC(I, A, B)
#endif
#ifdef RART_PP_KO_NOLOAD
#define PP_LOOP_LOADS false
#else
#define PP_LOOP_LOADS true
#endif
#ifdef RART_PP_KO_NOWAIT
#define PP_VM(N) ((void)0)
#else
#define PP_VM(N) pp_vmcnt<N>()
#endif

// OPT bit 0: balanced refills (half of a wave's W pieces ride in M0: 16 pieces per slot instead of 8 / 8 / 24 / 24)
// OPT bit 1: the workgroup WALKS tiles (grid = one workgroup per CU, tile = blockIdx.x + k gridDim.x -- the order the dispatcher would have
//            used).  The per-workgroup trace of ViT-B/16's launches (scratch/r6/vit_pp_trace.py: s_memtime at entry / loop end / exit per CU)
//            put 8-10 k cycles between one workgroup's exit and the next one's first instruction and 9 k of prologue beside a 92 k-cycle
//            K loop (K = 768).  Walking removes the first; the second shrinks because the next tile's stage 0 is requested BEFORE the
//            epilogue (which stages through the bytes after stage 0, so the tile buffer is STAGE + staging long) and lands while the
//            epilogue's loads and stores run.  The epilogue itself is unchanged (gp_epilogue, in order): same bits.
template <int TN, bool CONV, int OPT>
__global__ __launch_bounds__(512, 1) void k_gemm_pair_pp(const GemmPairDev d) {
  constexpr bool BAL = OPT & 1, WALK = OPT & 2;
  constexpr int TM = 256, NW = 8, WN = TN / 64, WM = NW / WN, RW = TM / WM, MI = RW / 32, MH = MI / 2;
  static_assert(TN == 256 || TN == 128, "column tiles of 256 or 128");
  constexpr int PLANE_A = TM * 64, PLANE_B = TN * 64, STAGE = 2 * PLANE_A + 2 * PLANE_B;
  constexpr int WQ = TN / 64;                     // W pieces (one plane) per wave and K step
  constexpr int WH = WQ / 2, NB = WH + 2;         // balanced form: W pieces / loads of a wave per memory phase
  constexpr int N0 = 2, N1 = WQ + 2;              // loads a wave issues in M0 / M1
  constexpr int VM_ALPHA = BAL ? 3 * NB : 2 * N1 + N0, VM_BETA = BAL ? 2 + NB : N0 + N1, VM_GAMMA = BAL ? 2 + NB : 2 + N0 + N1,
                VM_DELTA = 2 * N0 + N1;
  constexpr int GP_STAGING = NW * 32 * GP_LDE * 4;
  static_assert(GP_STAGING + TM * 4 <= 2 * STAGE, "epilogue staging + row table must fit the tile buffers");
  constexpr int EP_OFF = WALK ? STAGE : 0;        // where the epilogue stages: WALK keeps stage 0 free for the next tile's first loads
  constexpr int LDS_BYTES = (WALK && STAGE + GP_STAGING + TM * 4 > 2 * STAGE) ? STAGE + GP_STAGING + TM * 4 : 2 * STAGE;
  __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
#ifdef RART_PP_STAMPS
  unsigned long long t_tile = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
  const int g = wave >> 2, w4 = wave & 3, og = 1 - g;
  const uint16_t *a_hi = d.a_hi, *a_lo = d.a_lo, *w_hi = d.w_hi, *w_lo = d.w_lo;
  long long c_off = 0;
  if (!CONV && gridDim.y > 1) {
    const int z = blockIdx.y, zo = z / d.z_inner, zi = z - zo * d.z_inner;
    const long long ao = zo * d.a_zo + zi * d.a_zi, wo = zo * d.w_zo + zi * d.w_zi;
    a_hi += ao; a_lo += ao; w_hi += wo; w_lo += wo;
    c_off = zo * d.c_zo + zi * d.c_zi;
  }
  const int TMV = d.tile_rows;                     // 256, or 224: a tile steps 224 rows and its last 32-row block is neither multiplied nor stored
  const int n_tiles = (d.N + TN - 1) / TN, m_tiles = (d.M - d.m_begin + TMV - 1) / TMV;
  // tile of (virtual) block index BID: XCD-aware enumeration, rows in groups of 8; `ok_` false for the holes past the last row tile
#define RART_PP_TILE_OF(BID, M0_, N0_, OK_)                                                                      \
  {                                                                                                              \
    const int bid_ = (BID);                                                                                      \
    int mt_, nt_;                                                                                                \
    if (m_tiles >= 16) {                                                                                         \
      const int xcd_ = bid_ & 7, slot_ = bid_ >> 3;                                                              \
      mt_ = (slot_ / n_tiles) * 8 + xcd_;                                                                        \
      nt_ = slot_ % n_tiles;                                                                                     \
    } else {                                                                                                     \
      mt_ = bid_ / n_tiles;                                                                                      \
      nt_ = bid_ - mt_ * n_tiles;                                                                                \
    }                                                                                                            \
    OK_ = mt_ < m_tiles;                                                                                         \
    M0_ = d.m_begin + mt_ * TMV; N0_ = nt_ * TN;                                                                             \
  }
  const int n_blocks = (m_tiles >= 16 ? (m_tiles + 7) / 8 * 8 : m_tiles) * n_tiles;
  int bid = blockIdx.x, m0 = 0, n0 = 0;
  {
    bool ok = false;
    while (bid < n_blocks) {
      RART_PP_TILE_OF(bid, m0, n0, ok)
      if (ok || !WALK) break;
      bid += gridDim.x;
    }
    if (!ok) return;
  }

  // ---- loader.  A wave only ever loads A rows of the OTHER group: piece w4 (16 rows) of its two 64-row regions -- q = 0: the region it
  //      refills in M0 (half `og` of group og), q = 1: the one it refills in M1 (half `g` of group og) -- and WQ pieces of ONE W plane
  //      (group 0: hi, group 1: lo).  Lane -> row (lane >> 2), LDS chunk (lane & 3) <- the row's chunk (lane & 3) ^ ((row >> 2) & 3).
  //      Addresses: a lane's byte offset inside the plane is ONE constant (avoff / wvoff; PP_OOR for rows past M / past the table: the
  //      buffer range check zero-fills them, the zero page of k_gemm_pair is not needed); the K step (and the tap) move the SCALAR offset.
  //      CONV with taps: the resource base is shifted down by the most negative tap offset so that every scalar offset is >= 0; a pixel
  //      outside the image for tap t has bit t of `nok` set and its offset ORed with PP_OOR (two vector instructions per piece and step).
  int pr[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int hh = q == 0 ? og : g;
    pr[q] = (WM == 2) ? og * 128 + hh * 64 + 16 * w4 : (2 * og + (w4 >> 1)) * 64 + hh * 32 + (w4 & 1) * 16;
  }
  uint32_t avoff[2], nok[2];
  int tapreg = 0, tap_min = 0;
  const bool one_tap = !CONV || d.n_taps == 1;
  if (CONV) {
    for (int t = 0; t < d.n_taps; ++t) tap_min = min(tap_min, (d.tap_dy[t] * d.src_w + d.tap_dx[t]) * d.lda * 2);
    if (lane < d.n_taps) tapreg = (d.tap_dy[lane] * d.src_w + d.tap_dx[lane]) * d.lda * 2 - tap_min;
  }
  const uint32_t tap0 = CONV ? (uint32_t)__builtin_amdgcn_readfirstlane((d.tap_dy[0] * d.src_w + d.tap_dx[0]) * d.lda * 2 - tap_min) : 0u;
  uint32_t wvoff[WQ];
#define RART_PP_ADDR(M0_, N0_)                                                                                   \
  {                                                                                                              \
    nok[0] = nok[1] = 0u;                                                                                        \
  _Pragma("unroll")                                                                                              \
    for (int q = 0; q < 2; ++q) {                                                                                \
      const int r = pr[q] + (lane >> 2);                                                                         \
      const int csrc = (lane & 3) ^ ((r >> 2) & 3);                                                              \
      const int m = (M0_) + r;                                                                                   \
      const bool ok = m < d.M && r < TMV;      /* rows past a 224-row tile are the next tile's: zero-filled, not fetched */ \
      if (CONV) {                                                                                                \
        const uint32_t mm = ok ? (uint32_t)m : 0u;                                                               \
        const uint32_t t = gp_fastdiv(mm, d.gw_magic, d.gw_shift);                                               \
        const int ox = (int)(mm - t * (uint32_t)d.grid_w);                                                       \
        const int n = (int)gp_fastdiv(t, d.gh_magic, d.gh_shift);                                                \
        const int oy = (int)(t - (uint32_t)n * (uint32_t)d.grid_h);                                              \
        const int by = oy * d.sy, bx = ox * d.sx;                                                                \
        avoff[q] = (uint32_t)((n * d.src_h * d.src_w + by * d.src_w + bx) * d.lda * 2 + csrc * 16);              \
        for (int t2 = 0; t2 < d.n_taps; ++t2) {                                                                  \
          const int iy = by + d.tap_dy[t2], ix = bx + d.tap_dx[t2];                                              \
          if (!(ok && (unsigned)iy < (unsigned)d.src_h && (unsigned)ix < (unsigned)d.src_w)) nok[q] |= 1u << t2; \
        }                                                                                                        \
        if (one_tap && (nok[q] & 1u)) avoff[q] = PP_OOR;                                                         \
      } else {                                                                                                   \
        long long srow = m;                                                                                      \
        if (d.map_rows) {                                                                                        \
          const int img = m / d.rpi;                                                                             \
          srow = (long long)img * d.src_rpi + (m - img * d.rpi);                                                 \
        }                                                                                                        \
        avoff[q] = ok ? (uint32_t)(((srow + d.src_off) * d.lda + csrc * 8) * 2) : PP_OOR;                        \
      }                                                                                                          \
    }                                                                                                            \
  _Pragma("unroll")                                                                                              \
    for (int q = 0; q < WQ; ++q) {                                                                               \
      const int r = 16 * (w4 + 4 * q) + (lane >> 2);                                                             \
      const int csrc = (lane & 3) ^ ((r >> 2) & 3);                                                              \
      const int n = (N0_) + r;                                                                                   \
      wvoff[q] = n < d.w_rows ? (uint32_t)((n * d.ldw + csrc * 8) * 2) : PP_OOR;                                 \
    }                                                                                                            \
  }
  RART_PP_ADDR(m0, n0)
  const pp_srd_t srd_ah = pp_make_srd(reinterpret_cast<const char*>(a_hi) + tap_min), srd_al = pp_make_srd(reinterpret_cast<const char*>(a_lo) + tap_min);
  const pp_srd_t srd_w = pp_make_srd(g ? w_lo : w_hi);
  const int w_step = (d.flags & GP_W_INTERLEAVED) ? 128 : 64;
  const uint32_t w_dst = lds_base + 2 * PLANE_A + g * PLANE_B + w4 * 1024;
  const int KT = d.K / GP_BK;

  // stage ST's A piece q (hi + lo): 2 loads
#define RART_PP_ISSUE_A(Q, ST)                                                                                   \
  {                                                                                                              \
    const int s_ = (ST);                                                                                         \
    const uint32_t dst_ = lds_base + (s_ & 1) * STAGE + pr[Q] * 64;                                              \
    if (one_tap) {                                                                                               \
      const uint32_t so_ = tap0 + (uint32_t)s_ * 64u;                                                            \
      pp_bload(avoff[Q], srd_ah, so_, dst_);                                                                     \
      pp_bload(avoff[Q], srd_al, so_, dst_ + PLANE_A);                                                           \
    } else {                                                                                                     \
      const int tap_ = s_ >> d.tpt_shift;                                                                        \
      const uint32_t so_ = (uint32_t)(__builtin_amdgcn_readlane(tapreg, tap_) + (s_ - (tap_ << d.tpt_shift)) * 64); \
      const uint32_t vo_ = avoff[Q] | ((uint32_t)__builtin_amdgcn_sbfe(nok[Q], tap_, 1) & PP_OOR);               \
      pp_bload(vo_, srd_ah, so_, dst_);                                                                          \
      pp_bload(vo_, srd_al, so_, dst_ + PLANE_A);                                                                \
    }                                                                                                            \
  }
  // stage ST's W pieces Q0..Q1-1 of this wave's plane
#define RART_PP_ISSUE_W(ST, Q0, Q1)                                                                              \
  {                                                                                                              \
    const int s_ = (ST);                                                                                         \
    const uint32_t so_ = (uint32_t)(s_ * w_step);                                                                \
    _Pragma("unroll") for (int q = (Q0); q < (Q1); ++q) pp_bload(wvoff[q], srd_w, so_, w_dst + (s_ & 1) * STAGE + q * 4096); \
  }

  const int fr = lane & 31, h = lane >> 5;
  uint32_t xo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) xo[ks] = (uint32_t)(fr * 64 + (((2 * ks + h) ^ ((fr >> 2) & 3)) << 4));
  f32x16 acc[MI][2];
  // ---- prologue, first part: stage 0 whole (WALK: for the following tiles this is issued before the previous tile's epilogue)
  RART_PP_ISSUE_W(0, 0, WQ)
  RART_PP_ISSUE_A(0, 0)
  RART_PP_ISSUE_A(1, 0)
  bf16x8 ah[MH][2], al[MH][2], bh[2][2], bl[2][2];
  const bool skip_last = wm * RW + MI * 32 > TMV;          // (wave-uniform) this wave's last row block lies past a 224-row tile
#define RART_PP_READ_A(HALF)                                                                                     \
  _Pragma("unroll") for (int ii = 0; ii < MH; ++ii) if (!((HALF) == 1 && ii == MH - 1 && skip_last))           \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                        \
    ah[ii][ks] = *reinterpret_cast<const bf16x8*>(Ah + ((HALF)*MH + ii) * 32 * 64 + xo[ks]);                     \
    al[ii][ks] = *reinterpret_cast<const bf16x8*>(Ah + PLANE_A + ((HALF)*MH + ii) * 32 * 64 + xo[ks]);           \
  }
#ifdef RART_PP_KO_NOMFMA
#define RART_PP_MFMA(HALF)                                                                                       \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int ii = 0; ii < MH; ++ii)           \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
          asm volatile("" ::"v"(al[ii][ks]), "v"(ah[ii][ks]), "v"(bh[j][ks]), "v"(bl[j][ks]));
#else
#define RART_PP_MFMA(HALF)                                                                                       \
  __builtin_amdgcn_s_setprio(1);                                                                                 \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int ii = 0; ii < MH; ++ii)           \
      if (!((HALF) == 1 && ii == MH - 1 && skip_last)) _Pragma("unroll") for (int j = 0; j < 2; ++j) {          \
    acc[(HALF)*MH + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ii][ks], bh[j][ks], acc[(HALF)*MH + ii][j], 0, 0, 0); \
    acc[(HALF)*MH + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ii][ks], bl[j][ks], acc[(HALF)*MH + ii][j], 0, 0, 0); \
    acc[(HALF)*MH + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ii][ks], bh[j][ks], acc[(HALF)*MH + ii][j], 0, 0, 0); \
  }                                                                                                              \
  __builtin_amdgcn_s_setprio(0);
#endif
#ifdef RART_PP_STAMPS
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (;;) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int bc = n0 + wn * 64 + j * 32 + fr;
    const float bv = (d.bias && bc < d.N) ? d.bias[bc] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
  }
  // ---- prologue, second part: stage 1 except what the first M0 phases issue
  if (KT > 1) {
    RART_PP_ISSUE_W(1, 0, BAL ? WH : WQ)      // balanced form: the second half of stage 1's W pieces is issued by the first M0
    RART_PP_ISSUE_A(1, 1)
    if (g) RART_PP_ISSUE_A(0, 1)
  }
  pp_vmcnt<0>();
  __syncthreads();
  if (g) pp_slot_end();                     // the stagger: group 1 runs one slot behind group 0
  PP_T(t_loop0)

  for (int kt = 0; kt < KT; ++kt) {
    const uint8_t* const Ah = lds + (kt & 1) * STAGE + (wm * RW) * 64;
    const uint8_t* const Bh = lds + (kt & 1) * STAGE + 2 * PLANE_A + (wn * 64) * 64;
    const int tail = kt + 2 >= KT;          // some refill of this K step is skipped: the counts below do not hold, drain instead
    // ---- M0 (the refill first: its flight time is what the counted waits budget)
    PP_T(t0)
    if (BAL && PP_LOOP_LOADS && kt + 1 < KT) RART_PP_ISSUE_W(kt + 1, WH, WQ)
    if (PP_LOOP_LOADS && kt + 1 + g < KT) RART_PP_ISSUE_A(0, kt + 1 + g)
#ifndef RART_PP_KO_NOREAD
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bh[j][ks] = *reinterpret_cast<const bf16x8*>(Bh + j * 32 * 64 + xo[ks]);
        bl[j][ks] = *reinterpret_cast<const bf16x8*>(Bh + PLANE_B + j * 32 * 64 + xo[ks]);
      }
    RART_PP_READ_A(0)
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_T(t1)
    if (g && !BAL) {
      if (tail) PP_VM(0); else PP_VM(VM_DELTA);
    }
    PP_T(t2)
    pp_slot_end();
    // ---- C0
    PP_T(t3)
    RART_PP_MFMA(0)
    PP_T(t4)
    pp_slot_end();
    // ---- M1
    PP_T(t5)
    if (PP_LOOP_LOADS && kt + 2 < KT) {
      RART_PP_ISSUE_W(kt + 2, 0, BAL ? WH : WQ)
      RART_PP_ISSUE_A(1, kt + 2)
    }
#ifndef RART_PP_KO_NOREAD
    RART_PP_READ_A(1)
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_T(t6)
    if (tail) PP_VM(0);
    else if (g) PP_VM(VM_GAMMA);
    else PP_VM(VM_ALPHA);
    PP_T(t7)
    pp_slot_end();
    // ---- C1
    PP_T(t8)
    RART_PP_MFMA(1)
    PP_T(t9)
    if (!g) {
      if (tail) PP_VM(0); else PP_VM(VM_BETA);
    }
    PP_T(t10)
    pp_slot_end();
    PP_T(t11)
    PP_ACC(0, t0, t1) PP_ACC(1, t1, t2) PP_ACC(2, t2, t3) PP_ACC(3, t3, t4) PP_ACC(4, t4, t5) PP_ACC(5, t5, t6)
    PP_ACC(6, t6, t7) PP_ACC(7, t7, t8) PP_ACC(8, t8, t9) PP_ACC(9, t9, t10) PP_ACC(10, t10, t11) PP_ACC(11, t0, t11)
  }
  PP_T(t_loop1)
  if (!g) pp_slot_end();                    // group 0's closing barrier pairs with group 1's last one
  pp_vmcnt<0>();
  __syncthreads();
  const int m0e = m0, n0e = n0;
#ifdef RART_PP_STAMPS
  const int bide = bid;
#endif
  bool more = false;
  if (WALK) {
    bid += gridDim.x;
    while (bid < n_blocks) {
      RART_PP_TILE_OF(bid, m0, n0, more)
      if (more) break;
      bid += gridDim.x;
    }
    if (more) {                             // the next tile's stage 0 travels while this tile's epilogue runs (it stages past stage 0)
      RART_PP_ADDR(m0, n0)
      RART_PP_ISSUE_W(0, 0, WQ)
      RART_PP_ISSUE_A(0, 0)
      RART_PP_ISSUE_A(1, 0)
    }
  }
  gp_epilogue<TM, TN, CONV, MI>(d, lds + EP_OFF, acc, m0e, n0e, c_off, TMV);
#ifdef RART_PP_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  PP_T(t_exit)
  if (lane == 0 && (wave & 3) == 0) {
    for (int i = 0; i < 12; ++i) { atomicAdd(&g_pp_stamps[g][i], st_acc[i]); st_acc[i] = 0; }
    atomicAdd(&g_pp_stamps[g][12], (unsigned long long)KT);
    atomicAdd(&g_pp_stamps[g][13], t_loop0 - t_tile);       // prologue
    atomicAdd(&g_pp_stamps[g][14], t_exit - t_loop1);       // closing barrier + epilogue
    atomicAdd(&g_pp_stamps[g][15], 1ull);                   // workgroups
  }
  if (tid == 0 && bide < 4096 && blockIdx.y == 0) {
    g_pp_trace[bide][0] = t_tile; g_pp_trace[bide][1] = t_loop1; g_pp_trace[bide][2] = t_exit;
    g_pp_trace[bide][3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
  }
#endif
#ifdef RART_PP_STAMPS
  t_tile = t_exit;
#endif
  if (!more) break;
  __syncthreads();                          // every wave's staging reads are done: stage 1 may be refilled
  }
#undef RART_PP_MFMA
#undef RART_PP_READ_A
#undef RART_PP_ISSUE_W
#undef RART_PP_ISSUE_A
#undef RART_PP_ADDR
#undef RART_PP_TILE_OF
}

// ---- PERSISTENT form (round 6; opt-in: rart_gemm_pair_set_schedule(2)): 256 x 128 tiles, one workgroup per CU walks a contiguous run of
//      tiles (consecutive column tiles of one row tile: the A rows stay in its L1 / L2), the ping-pong K loop above runs THROUGH the tile
//      boundaries (the refills of the last two K steps fetch the next tile's first stages, the counted waits never drain), and the epilogue of
//      tile t is DEFERRED: its accumulators move to a second register set and are written out in 12 micro-steps that ride in the phases of
//      tile t + 1's K loop -- the skip-pair loads and the hi / lo stores of the wide 1x1 layers (50176 x 256 -> 1024 and its like: 0.5 MB
//      per 256 x 256 of output against 8 K steps of matrix work) are 57 % of those launches and run with every CU's matrix pipes idle and
//      every CU storing at once.
//      A micro-step = a third of a 16-row half pass: (0) request the skip / mask operands of rows 0-7, write 16 accumulator registers per
//      column block to the wave's staging; (1) rows 0-7: read back transposed, point-wise tail (gp_finish_segment, the same code as
//      gp_epilogue: identical bits), stores, request rows 8-15; (2) rows 8-15.  The loads and stores of the micro-steps are ordinary
//      compiler-counted memory operations beside the inline-asm LDS-DMA: gfx950 completes vector memory operations in issue order, so a
//      counted `vmcnt(N)` of either kind can only wait for MORE than it needs when operations of the other kind sit behind its target --
//      never less (N <= the operations issued after the target).
//      One-tap problems without a GELU epilogue only (1x1 convolutions of any stride, plain products without row re-basing / batching).
//      MEASURED (scratch/r6/time_ps.py, profiles/r06_persistent.txt): bit-identical on every shape and on the whole engine, and SLOWER than the
//      256 x 256 ping-pong tiles it replaces -- 193 vs 176 us on 50176 x 256 -> 1024 (micro-steps in all four phases; 226 with them in the
//      memory phases only; 216 with the operands requested two micro-steps ahead: 255 VGPRs), 207 vs 166 us on 50176 x 1024 -> 512, the
//      gradient evaluation 20.4 vs 19.9 ms: a 128-column tile stages 1.5 x the bytes per MFMA (its K loop is bound by the texture-address
//      unit, ~1 800 cycles per K step for 1 536 of matrix work), and a micro-step's VALU work beside the partner's MFMAs costs what the
//      overlap gains.  A 256-column tile cannot hold two accumulator sets (2 x 128 registers).  Kept behind the switch, off by default.
template <bool CONV>
__global__ __launch_bounds__(512, 1) void k_gemm_pair_ps(const GemmPairDev d) {
  constexpr int TM = 256, TN = 128, NW = 8, WN = 2, RW = 64, MI = 2;
  constexpr int PLANE_A = TM * 64, PLANE_B = TN * 64, STAGE = 2 * PLANE_A + 2 * PLANE_B;
  constexpr int WQ = 2, WH = 1, NB = WH + 2;
  constexpr int VM_ALPHA = 3 * NB, VM_BETA = 2 + NB, VM_GAMMA = 2 + NB;
  constexpr int PS_STG = 16 * GP_LDE * 4;                    // a wave's staging region: 16 rows x 68 floats
  constexpr int OFF_STG = 2 * STAGE, OFF_TAB = OFF_STG + NW * PS_STG;
  constexpr int NMS = 12;                                    // micro-steps of a tile's epilogue: 4 half passes x 3
  __shared__ __attribute__((aligned(16))) uint8_t lds[OFF_TAB + 3 * TM * 4];
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
  const int g = wave >> 2, w4 = wave & 3, og = 1 - g;
  const int n_tiles = (d.N + TN - 1) / TN, m_tiles = (d.M + TM - 1) / TM, T = n_tiles * m_tiles;
  const int G = gridDim.x, base_n = T / G, rem_n = T % G;
  const int first = blockIdx.x * base_n + min((int)blockIdx.x, rem_n), count = base_n + ((int)blockIdx.x < rem_n ? 1 : 0);
  if (count == 0) return;
  const int KT = d.K / GP_BK;
  const int flags = d.flags;
  const bool out_f32 = flags & GP_OUT_F32;

  // ---- loader (k_gemm_pair_pp's, TN = 128, one tap): piece rows, per-tile lane offsets
  int pr[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int hh = q == 0 ? og : g;
    pr[q] = (2 * og + (w4 >> 1)) * 64 + hh * 32 + (w4 & 1) * 16;
  }
  const uint32_t tap0 = CONV ? (uint32_t)__builtin_amdgcn_readfirstlane((d.tap_dy[0] * d.src_w + d.tap_dx[0]) * d.lda * 2) : 0u;
  // byte offsets of this lane's two A pieces and two W pieces for tile t_
#define RART_PS_ADDR(t_, AV, WV)                                                                                 \
  {                                                                                                              \
    const int mt_ = (t_) / n_tiles, nt_ = (t_) - mt_ * n_tiles;                                                  \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                              \
      const int r = pr[q] + (lane >> 2);                                                                         \
      const int csrc = (lane & 3) ^ ((r >> 2) & 3);                                                              \
      const int m = mt_ * TM + r;                                                                                \
      const bool ok = m < d.M;                                                                                   \
      if (CONV) {                                                                                                \
        const uint32_t mm = ok ? (uint32_t)m : 0u;                                                               \
        const uint32_t tq = gp_fastdiv(mm, d.gw_magic, d.gw_shift);                                              \
        const int ox = (int)(mm - tq * (uint32_t)d.grid_w);                                                      \
        const int n = (int)gp_fastdiv(tq, d.gh_magic, d.gh_shift);                                               \
        const int oy = (int)(tq - (uint32_t)n * (uint32_t)d.grid_h);                                             \
        const int by = oy * d.sy, bx = ox * d.sx;                                                                \
        const int iy = by + d.tap_dy[0], ix = bx + d.tap_dx[0];                                                  \
        AV[q] = (ok && (unsigned)iy < (unsigned)d.src_h && (unsigned)ix < (unsigned)d.src_w)                     \
                    ? (uint32_t)((n * d.src_h * d.src_w + by * d.src_w + bx) * d.lda * 2 + csrc * 16) : PP_OOR;  \
      } else {                                                                                                   \
        AV[q] = ok ? (uint32_t)((((long long)m + d.src_off) * d.lda + csrc * 8) * 2) : PP_OOR;                   \
      }                                                                                                          \
    }                                                                                                            \
    _Pragma("unroll") for (int q = 0; q < WQ; ++q) {                                                             \
      const int r = 16 * (w4 + 4 * q) + (lane >> 2);                                                             \
      const int csrc = (lane & 3) ^ ((r >> 2) & 3);                                                              \
      const int n = nt_ * TN + r;                                                                                \
      WV[q] = n < d.w_rows ? (uint32_t)((n * d.ldw + csrc * 8) * 2) : PP_OOR;                                    \
    }                                                                                                            \
  }
  uint32_t av[2], wv[WQ], avn[2] = {PP_OOR, PP_OOR}, wvn[WQ] = {PP_OOR, PP_OOR};
  RART_PS_ADDR(first, av, wv)
  const pp_srd_t srd_ah = pp_make_srd(d.a_hi), srd_al = pp_make_srd(d.a_lo);
  const pp_srd_t srd_w = pp_make_srd(g ? d.w_lo : d.w_hi);
  const int w_step = (d.flags & GP_W_INTERLEAVED) ? 128 : 64;
  const uint32_t w_dst = lds_base + 2 * PLANE_A + g * PLANE_B + w4 * 1024;
  // stage S_ of the CURRENT tile's numbering (S_ >= KT: stage S_ - KT of the next tile, if there is one) -> buffer (par + S_) & 1
#define RART_PS_ISSUE_A(Q, S_)                                                                                   \
  {                                                                                                              \
    const int s_ = (S_);                                                                                         \
    const uint32_t dst_ = lds_base + ((par + s_) & 1) * STAGE + pr[Q] * 64;                                      \
    if (s_ < KT) {                                                                                               \
      const uint32_t so_ = tap0 + (uint32_t)s_ * 64u;                                                            \
      pp_bload(av[Q], srd_ah, so_, dst_);                                                                        \
      pp_bload(av[Q], srd_al, so_, dst_ + PLANE_A);                                                              \
    } else if (has_next) {                                                                                       \
      const uint32_t so_ = tap0 + (uint32_t)(s_ - KT) * 64u;                                                     \
      pp_bload(avn[Q], srd_ah, so_, dst_);                                                                       \
      pp_bload(avn[Q], srd_al, so_, dst_ + PLANE_A);                                                             \
    }                                                                                                            \
  }
#define RART_PS_ISSUE_W(S_, Q0, Q1)                                                                              \
  {                                                                                                              \
    const int s_ = (S_);                                                                                         \
    const uint32_t dst_ = w_dst + ((par + s_) & 1) * STAGE;                                                      \
    if (s_ < KT) {                                                                                               \
      _Pragma("unroll") for (int q = (Q0); q < (Q1); ++q) pp_bload(wv[q], srd_w, (uint32_t)(s_ * w_step), dst_ + q * 4096); \
    } else if (has_next) {                                                                                       \
      _Pragma("unroll") for (int q = (Q0); q < (Q1); ++q) pp_bload(wvn[q], srd_w, (uint32_t)((s_ - KT) * w_step), dst_ + q * 4096); \
    }                                                                                                            \
  }
  const int fr = lane & 31, h = lane >> 5;
  uint32_t xo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) xo[ks] = (uint32_t)(fr * 64 + (((2 * ks + h) ^ ((fr >> 2) & 3)) << 4));
  f32x16 acc[MI][2], accp[MI][2];
  int par = 0;
  bool has_next = count > 1;
  if (has_next) RART_PS_ADDR(first + 1, avn, wvn)

  // ---- deferred epilogue state: the previous tile (n0p, its row table tabp), the next micro-step esp (NMS = nothing pending)
  int esp = NMS, n0p = 0, tabp = 0;
  float* const sE = reinterpret_cast<float*>(lds + OFF_STG + wave * PS_STG);
  const int cw = lane & 7, rw = lane >> 3;
  uint4 erh = make_uint4(0, 0, 0, 0), erl = erh;
  uint32_t emb = 0xFFu;
  long long ee = -1;
  // request the operands of rows Q_ * 8 + rw of half pass HP_ of the pending tile
#define RART_PS_EP_LOAD(HP_, Q_)                                                                                 \
  {                                                                                                              \
    const int rt_ = wm * RW + ((HP_) >> 1) * 32 + ((HP_)&1) * 16 + (Q_)*8 + rw;                                  \
    const uint32_t off_ = reinterpret_cast<const uint32_t*>(lds + OFF_TAB)[tabp * TM + rt_];                     \
    const int col_ = n0p + wn * 64 + cw * 8;                                                                     \
    ee = (off_ != 0xFFFFFFFFu && col_ < d.N) ? (long long)off_ + col_ : -1;                                      \
    erh = erl = make_uint4(0, 0, 0, 0);                                                                          \
    emb = 0xFFu;                                                                                                 \
    if (ee >= 0) {                                                                                               \
      if (d.res_hi) {                                                                                            \
        erh = *reinterpret_cast<const uint4*>(d.res_hi + ee);                                                    \
        erl = *reinterpret_cast<const uint4*>(d.res_lo + ee);                                                    \
      }                                                                                                          \
      if (CONV && d.mask_bits) emb = d.mask_bits[ee >> 3];                                                       \
    }                                                                                                            \
  }
#define RART_PS_EP_PUT(I_, H_)                                                                                   \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int rr = 0; rr < 8; ++rr) {               \
    const int r_ = (H_)*8 + rr;      /* a constant after unrolling: the register index stays static */            \
    sE[((r_ & 3) + 8 * ((r_ >> 2) & 1) + 4 * h) * GP_LDE + j * 32 + fr] = accp[I_][j][r_];                       \
  }
  // one micro-step of the pending epilogue (nothing when none is pending)
#define RART_PS_EP_STEP()                                                                                        \
  if (esp < NMS) {                                                                                               \
    const int hp_ = esp / 3, ph_ = esp - 3 * hp_;                                                                \
    if (ph_ == 0) {                                                                                              \
      RART_PS_EP_LOAD(hp_, 0)                                                                                    \
      if (hp_ == 0) { RART_PS_EP_PUT(0, 0) } else if (hp_ == 1) { RART_PS_EP_PUT(0, 1) }                         \
      else if (hp_ == 2) { RART_PS_EP_PUT(1, 0) } else { RART_PS_EP_PUT(1, 1) }                                  \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                     \
      __builtin_amdgcn_wave_barrier();                                                                           \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                     \
    } else {                                                                                                     \
      const int r_ = (ph_ - 1) * 8 + rw;                                                                         \
      const float4 v0 = *reinterpret_cast<const float4*>(sE + r_ * GP_LDE + cw * 8);                             \
      const float4 v1 = *reinterpret_cast<const float4*>(sE + r_ * GP_LDE + cw * 8 + 4);                         \
      if (ee >= 0) {                                                                                             \
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};                                           \
        gp_finish_segment<CONV, false>(d, flags, out_f32, ee, v, erh, erl, erh, erl, emb);                       \
      }                                                                                                          \
      if (ph_ == 1) {                                                                                            \
        RART_PS_EP_LOAD(hp_, 1)                                                                                  \
      } else {                                                                                                   \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                   \
        __builtin_amdgcn_wave_barrier();      /* the reads of this half pass are done before the next one's writes */ \
      }                                                                                                          \
    }                                                                                                            \
    ++esp;                                                                                                       \
  }

  // ---- prologue: stage 0 whole, stage 1 except what the first M0 phases issue (the balanced form of k_gemm_pair_pp)
  RART_PS_ISSUE_W(0, 0, WQ)
  RART_PS_ISSUE_A(0, 0)
  RART_PS_ISSUE_A(1, 0)
  if (KT > 1 || has_next) {
    RART_PS_ISSUE_W(1, 0, WH)
    RART_PS_ISSUE_A(1, 1)
    if (g) RART_PS_ISSUE_A(0, 1)
  }
  pp_vmcnt<0>();
  __syncthreads();
  if (g) pp_slot_end();

  bf16x8 ah[2], al[2], bh[2][2], bl[2][2];
  for (int s = 0; s < count; ++s) {
    const int t = first + s, mt = t / n_tiles, m0 = mt * TM, n0 = (t - mt * n_tiles) * TN;
    // the row table of this tile (destination element offset of tile row r, ~0 past M); three tables: the one overwritten here belonged to
    // tile s - 3, whose deferred epilogue ended a whole tile ago
    if (tid < TM) {
      const uint32_t m = (uint32_t)(m0 + tid);
      uint32_t off = 0xFFFFFFFFu;
      if (m < (uint32_t)d.M) {
        if (CONV) {
          const uint32_t tq = gp_fastdiv(m, d.gw_magic, d.gw_shift);
          const int ox = (int)(m - tq * (uint32_t)d.grid_w);
          const int n = (int)gp_fastdiv(tq, d.gh_magic, d.gh_shift);
          const int oy = (int)(tq - (uint32_t)n * (uint32_t)d.grid_h);
          off = (uint32_t)(((n * d.dst_h + (oy * d.dst_sy + d.dst_oy)) * d.dst_w + (ox * d.dst_sx + d.dst_ox)) * d.ldc);
        } else {
          off = (uint32_t)((m + (uint32_t)d.dst_off) * (uint32_t)d.ldc);
        }
      }
      reinterpret_cast<uint32_t*>(lds + OFF_TAB)[(s % 3) * TM + tid] = off;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int bc = n0 + wn * 64 + j * 32 + fr;
      const float bv = (d.bias && bc < d.N) ? d.bias[bc] : 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
    }
    for (int kt = 0; kt < KT; ++kt) {
      const uint8_t* const Ah = lds + ((par + kt) & 1) * STAGE + (wm * RW) * 64;
      const uint8_t* const Bh = lds + ((par + kt) & 1) * STAGE + 2 * PLANE_A + (wn * 64) * 64;
      const int tail = !has_next && kt + 2 >= KT;
      // ---- M0
      RART_PS_ISSUE_W(kt + 1, WH, WQ)
      RART_PS_ISSUE_A(0, kt + 1 + g)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bh[j][ks] = *reinterpret_cast<const bf16x8*>(Bh + j * 32 * 64 + xo[ks]);
          bl[j][ks] = *reinterpret_cast<const bf16x8*>(Bh + PLANE_B + j * 32 * 64 + xo[ks]);
        }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        ah[ks] = *reinterpret_cast<const bf16x8*>(Ah + xo[ks]);
        al[ks] = *reinterpret_cast<const bf16x8*>(Ah + PLANE_A + xo[ks]);
      }
      RART_PS_EP_STEP()
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pp_slot_end();
      // ---- C0 (a micro-step after the MFMAs too: all eight waves advance their epilogue every slot)
#define RART_PS_MFMA(I_)                                                                                         \
  __builtin_amdgcn_s_setprio(1);                                                                                 \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < 2; ++j) {              \
    acc[I_][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh[j][ks], acc[I_][j], 0, 0, 0);               \
    acc[I_][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl[j][ks], acc[I_][j], 0, 0, 0);               \
    acc[I_][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh[j][ks], acc[I_][j], 0, 0, 0);               \
  }                                                                                                              \
  __builtin_amdgcn_s_setprio(0);
      RART_PS_MFMA(0)
      RART_PS_EP_STEP()
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pp_slot_end();
      // ---- M1
      RART_PS_ISSUE_W(kt + 2, 0, WH)
      RART_PS_ISSUE_A(1, kt + 2)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        ah[ks] = *reinterpret_cast<const bf16x8*>(Ah + 32 * 64 + xo[ks]);
        al[ks] = *reinterpret_cast<const bf16x8*>(Ah + PLANE_A + 32 * 64 + xo[ks]);
      }
      RART_PS_EP_STEP()
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (tail) pp_vmcnt<0>();
      else if (g) pp_vmcnt<VM_GAMMA>();
      else pp_vmcnt<VM_ALPHA>();
      pp_slot_end();
      // ---- C1
      RART_PS_MFMA(1)
      RART_PS_EP_STEP()
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!g) {
        if (tail) pp_vmcnt<0>(); else pp_vmcnt<VM_BETA>();
      }
      pp_slot_end();
    }
#undef RART_PS_MFMA
    // ---- tile boundary: what is left of the previous tile's epilogue, then this tile's accumulators become the pending ones
    while (esp < NMS) { RART_PS_EP_STEP() }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) accp[i][j] = acc[i][j];
    esp = 0; n0p = n0; tabp = s % 3;
    par = (par + KT) & 1;
#pragma unroll
    for (int q = 0; q < 2; ++q) av[q] = avn[q];
#pragma unroll
    for (int q = 0; q < WQ; ++q) wv[q] = wvn[q];
    has_next = s + 2 < count;
    if (has_next) RART_PS_ADDR(first + s + 2, avn, wvn)
  }
  if (!g) pp_slot_end();
  pp_vmcnt<0>();
  __syncthreads();
  while (esp < NMS) { RART_PS_EP_STEP() }
#undef RART_PS_EP_STEP
#undef RART_PS_EP_PUT
#undef RART_PS_EP_LOAD
#undef RART_PS_ISSUE_W
#undef RART_PS_ISSUE_A
#undef RART_PS_ADDR
}

#ifndef RART_PP_DEFAULT_OPT
#define RART_PP_DEFAULT_OPT 1
#endif
template <int OPT>
bool pp_launch_opt(const GemmPairDev& d, int tn, bool conv, dim3 grid, hipStream_t st, unsigned walk_wgs = 0) {
  if (walk_wgs && !(OPT & 2)) return pp_launch_opt<OPT | 2>(d, tn, conv, dim3(walk_wgs < grid.x ? walk_wgs : grid.x, 1), st);
  if (tn == 256) {
    if (conv) hipLaunchKernelGGL((k_gemm_pair_pp<256, true, OPT>), grid, dim3(512), 0, st, d);
    else hipLaunchKernelGGL((k_gemm_pair_pp<256, false, OPT>), grid, dim3(512), 0, st, d);
  } else if (tn == 128) {
    if (conv) hipLaunchKernelGGL((k_gemm_pair_pp<128, true, OPT>), grid, dim3(512), 0, st, d);
    else hipLaunchKernelGGL((k_gemm_pair_pp<128, false, OPT>), grid, dim3(512), 0, st, d);
  } else {
    return false;
  }
  return true;
}
}  // namespace

// Launch by column tile; `dev_desc` is gemm_pair.hip's filled GemmPairDev (same header, same layout).  Returns false when (tn) has no instance.
// walk_wgs > 0 (grid_y == 1 only): that many workgroups walk the grid_x tiles (OPT bit 1).
__attribute__((visibility("hidden"))) bool rart_gemm_pair_pp_launch(const void* dev_desc, int tn, bool conv, unsigned grid_x, unsigned grid_y,
                                                                    hipStream_t st, unsigned walk_wgs) {
  const GemmPairDev& d = *static_cast<const GemmPairDev*>(dev_desc);
  const dim3 grid(grid_x, grid_y);
#ifdef RART_PP_LAB              // lab build: both option values, chosen per call by RART_PP_OPT
  const char* e = getenv("RART_PP_OPT");
  if ((e ? atoi(e) : RART_PP_DEFAULT_OPT) & 1) return pp_launch_opt<1>(d, tn, conv, grid, st, walk_wgs);
  return pp_launch_opt<0>(d, tn, conv, grid, st, walk_wgs);
#else
  return pp_launch_opt<RART_PP_DEFAULT_OPT>(d, tn, conv, grid, st, walk_wgs);
#endif
}

// The persistent form: grid = `wgs` workgroups (one per CU), each walking T / wgs tiles of 256 x 128.
__attribute__((visibility("hidden"))) void rart_gemm_pair_ps_launch(const void* dev_desc, bool conv, unsigned wgs, hipStream_t st) {
  const GemmPairDev& d = *static_cast<const GemmPairDev*>(dev_desc);
  if (conv) hipLaunchKernelGGL((k_gemm_pair_ps<true>), dim3(wgs), dim3(512), 0, st, d);
  else hipLaunchKernelGGL((k_gemm_pair_ps<false>), dim3(wgs), dim3(512), 0, st, d);
}

#ifdef RART_PP_STAMPS
extern "C" int rart_debug_pp_trace(unsigned long long* out) {      // lab build only: out[4096][4] <- the per-workgroup trace of the last launch
  if (hipDeviceSynchronize() != hipSuccess) return RART_ERR_HIP;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_trace), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? RART_OK : RART_ERR_HIP;
}
// lab build only: out[2][16] <- the stamp sums (cycles; [g][12] = K steps summed over workgroups), then cleared
extern "C" int rart_debug_pp_stamps(unsigned long long* out) {
  if (hipDeviceSynchronize() != hipSuccess) return RART_ERR_HIP;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return RART_ERR_HIP;
  unsigned long long z[32] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_pp_stamps), z, sizeof(z)) != hipSuccess) return RART_ERR_HIP;
  return RART_OK;
}
#endif
