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
template <int TN, bool CONV, int OPT>
__global__ __launch_bounds__(512, 1) void k_gemm_pair_pp(const GemmPairDev d) {
  constexpr bool BAL = OPT & 1;
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
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * STAGE];
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  PP_T(t_entry)
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
  const int n_tiles = (d.N + TN - 1) / TN, m_tiles = (d.M + TM - 1) / TM;
  int m_tile, n_tile;
  if (m_tiles >= 16) {
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    m_tile = (slot / n_tiles) * 8 + xcd;
    n_tile = slot % n_tiles;
    if (m_tile >= m_tiles) return;
  } else {
    m_tile = blockIdx.x / n_tiles;
    n_tile = blockIdx.x - m_tile * n_tiles;
  }
  const int m0 = m_tile * TM, n0 = n_tile * TN;

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
    const int r = pr[q] + (lane >> 2);
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
      if (one_tap && (nok[q] & 1u)) avoff[q] = PP_OOR;
    } else {
      long long srow = m;
      if (d.map_rows) {
        const int img = m / d.rpi;
        srow = (long long)img * d.src_rpi + (m - img * d.rpi);
      }
      avoff[q] = ok ? (uint32_t)(((srow + d.src_off) * d.lda + csrc * 8) * 2) : PP_OOR;
    }
  }
  uint32_t wvoff[WQ];
#pragma unroll
  for (int q = 0; q < WQ; ++q) {
    const int r = 16 * (w4 + 4 * q) + (lane >> 2);
    const int csrc = (lane & 3) ^ ((r >> 2) & 3);
    const int n = n0 + r;
    wvoff[q] = n < d.w_rows ? (uint32_t)((n * d.ldw + csrc * 8) * 2) : PP_OOR;
  }
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
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int bc = n0 + wn * 64 + j * 32 + fr;
    const float bv = (d.bias && bc < d.N) ? d.bias[bc] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
  }

  // ---- prologue: stage 0 whole, stage 1 except what the first M0 phases issue
  RART_PP_ISSUE_W(0, 0, WQ)
  RART_PP_ISSUE_A(0, 0)
  RART_PP_ISSUE_A(1, 0)
  if (KT > 1) {
    RART_PP_ISSUE_W(1, 0, BAL ? WH : WQ)      // balanced form: the second half of stage 1's W pieces is issued by the first M0
    RART_PP_ISSUE_A(1, 1)
    if (g) RART_PP_ISSUE_A(0, 1)
  }
  pp_vmcnt<0>();
  __syncthreads();
  if (g) pp_slot_end();                     // the stagger: group 1 runs one slot behind group 0

  bf16x8 ah[MH][2], al[MH][2], bh[2][2], bl[2][2];
#define RART_PP_READ_A(HALF)                                                                                     \
  _Pragma("unroll") for (int ii = 0; ii < MH; ++ii) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {         \
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
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                           \
    acc[(HALF)*MH + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ii][ks], bh[j][ks], acc[(HALF)*MH + ii][j], 0, 0, 0); \
    acc[(HALF)*MH + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ii][ks], bl[j][ks], acc[(HALF)*MH + ii][j], 0, 0, 0); \
    acc[(HALF)*MH + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ii][ks], bh[j][ks], acc[(HALF)*MH + ii][j], 0, 0, 0); \
  }                                                                                                              \
  __builtin_amdgcn_s_setprio(0);
#endif
#ifdef RART_PP_STAMPS
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
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
#undef RART_PP_MFMA
#undef RART_PP_READ_A
#undef RART_PP_ISSUE_W
#undef RART_PP_ISSUE_A
  pp_vmcnt<0>();
  __syncthreads();
  gp_epilogue<TM, TN, CONV, MI>(d, lds, acc, m0, n0, c_off);
#ifdef RART_PP_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  PP_T(t_exit)
  if (lane == 0 && (wave & 3) == 0) {
    for (int i = 0; i < 12; ++i) atomicAdd(&g_pp_stamps[g][i], st_acc[i]);
    atomicAdd(&g_pp_stamps[g][12], (unsigned long long)KT);
    atomicAdd(&g_pp_stamps[g][13], t_loop0 - t_entry);      // prologue
    atomicAdd(&g_pp_stamps[g][14], t_exit - t_loop1);       // closing barrier + epilogue
    atomicAdd(&g_pp_stamps[g][15], 1ull);                   // workgroups
  }
#endif
}

#ifndef RART_PP_DEFAULT_OPT
#define RART_PP_DEFAULT_OPT 1
#endif
template <int OPT>
bool pp_launch_opt(const GemmPairDev& d, int tn, bool conv, dim3 grid, hipStream_t st) {
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
__attribute__((visibility("hidden"))) bool rart_gemm_pair_pp_launch(const void* dev_desc, int tn, bool conv, unsigned grid_x, unsigned grid_y,
                                                                    hipStream_t st) {
  const GemmPairDev& d = *static_cast<const GemmPairDev*>(dev_desc);
  const dim3 grid(grid_x, grid_y);
#ifdef RART_PP_LAB              // lab build: both option values, chosen per call by RART_PP_OPT
  const char* e = getenv("RART_PP_OPT");
  if ((e ? atoi(e) : RART_PP_DEFAULT_OPT) & 1) return pp_launch_opt<1>(d, tn, conv, grid, st);
  return pp_launch_opt<0>(d, tn, conv, grid, st);
#else
  return pp_launch_opt<RART_PP_DEFAULT_OPT>(d, tn, conv, grid, st);
#endif
}

#ifdef RART_PP_STAMPS
// lab build only: out[2][16] <- the stamp sums (cycles; [g][12] = K steps summed over workgroups), then cleared
extern "C" int rart_debug_pp_stamps(unsigned long long* out) {
  if (hipDeviceSynchronize() != hipSuccess) return RART_ERR_HIP;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return RART_ERR_HIP;
  unsigned long long z[32] = {0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_pp_stamps), z, sizeof(z)) != hipSuccess) return RART_ERR_HIP;
  return RART_OK;
}
#endif
