// Shared device/host helpers for the robustart HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/robustart_hip.h"
#include <mutex>
// The lazily built process-lifetime host tables (gaussian weights, disks, fixed-point fragments, the folded filter matrix) are shared by every
// thread that calls the library (ctypes releases the GIL: a loader thread and the main thread can be inside it together): one lock around
// their construction.
inline std::mutex& rart_host_table_mutex() {
  static std::mutex mu;
  return mu;
}

#define RART_VERSION RART_ABI_VERSION

// ---- error plumbing ---------------------------------------------------------------
void rart_set_error(const char* fmt, ...);
// Raise a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) to at least `bytes` on the CURRENT device.
// The attribute is per device, so the cache behind it is keyed by (kernel, device): a process driving several GPUs sets it on
// each of them.  Returns false (and sets the error string) when the runtime refuses.
bool rart_raise_dynamic_lds(const void* kernel, size_t bytes, const char* what);

#define RART_CHECK_ARG(cond, ...)                \
  do {                                           \
    if (!(cond)) {                               \
      rart_set_error(__VA_ARGS__);               \
      return RART_ERR_INVALID;                   \
    }                                            \
  } while (0)

#define RART_CHECK_LAUNCH(what)                                                   \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) {                                                       \
      rart_set_error("%s: %s", what, hipGetErrorString(e_));                      \
      return RART_ERR_HIP;                                                        \
    }                                                                             \
  } while (0)

static inline size_t rart_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Memory-bound launch geometry: 256-thread workgroups, grid capped at 256 CUs x 8 and
// grid-strided (cdna_hip_programming.md guideline 11).
static inline int rart_grid_for(size_t work_items, int block = 256, int max_blocks = 256 * 8) {
  size_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > (size_t)max_blocks) g = max_blocks;
  return (int)g;
}

// ---- per-severity tables (imagenet_c/corruptions.py) ------------
struct RartSeverity {
  static constexpr double gaussian_noise[5] = {.08, .12, 0.18, 0.26, 0.38};
  static constexpr double shot_noise[5] = {60, 25, 12, 5, 3};
  static constexpr double impulse_noise[5] = {.03, .06, .09, 0.17, 0.27};
  static constexpr double speckle_noise[5] = {.15, .2, 0.35, 0.45, 0.6};
  static constexpr double contrast[5] = {0.4, .3, .2, .1, .05};
  static constexpr double brightness[5] = {.1, .2, .3, .4, .5};
};

#ifdef __HIPCC__
// ---- Threefry-2x32 counter-based generator --------------------------------------------
// (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11.)  Chosen over Philox
// because gfx950 has no full-rate 32x32 multiplier: Threefry is add/rotate/xor only, all
// full-rate VALU ops.  13 rounds = the smallest Crush-resistant variant in the paper's
// Table 2 (20 is Random123's conservative default); both are pinned by the Random123
// known-answer vectors in tests/test_abi_cpu.py.  The kernels are VALU-bound on integer ops,
// so rounds are throughput.
// Counter layout used by every kernel in this library:
//   key = (seed_lo, seed_hi)
//   ctr = (block_index | stream_id << 28, global_sample_index)
// so a draw depends only on (seed, sample, element), never on launch geometry.
#ifndef RART_THREEFRY_ROUNDS
#define RART_THREEFRY_ROUNDS 13
#endif

__device__ __forceinline__ uint32_t rart_rotl(uint32_t x, int r) {
  return __builtin_rotateleft32(x, r);
}

template <int ROUNDS = RART_THREEFRY_ROUNDS>
__device__ __forceinline__ uint2 threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1) {
  constexpr int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  uint32_t ks[3] = {k0, k1, 0x1BD11BDAu ^ k0 ^ k1};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    x0 += x1;
    x1 = rart_rotl(x1, R[r & 7]);
    x1 ^= x0;
    if ((r & 3) == 3) {
      const int s = (r >> 2) + 1;
      x0 += ks[s % 3];
      x1 += ks[(s + 1) % 3] + (uint32_t)s;
    }
  }
  return make_uint2(x0, x1);
}

__device__ __forceinline__ uint32_t rart_ctr0(uint32_t block_index, int stream_id) {
  return (block_index & 0x0FFFFFFFu) | ((uint32_t)stream_id << 28);
}

// One 32-bit word -> two N(0,1) draws (Box-Muller, 16-bit radius uniform + 16-bit angle).
// u1 = (hi16 + 0.5)/65536 in (0,1) -> r = sqrt(-2 ln u1) <= 4.85; angle = (lo16 + 0.5)/65536 turns.
// v_sin_f32 / v_cos_f32 take their argument in revolutions, so no range reduction is needed.
__device__ __forceinline__ float2 rart_boxmuller16(uint32_t w) {
  const float u1 = ((float)(w >> 16) + 0.5f) * (1.0f / 65536.0f);
  const float u2 = ((float)(w & 0xFFFFu) + 0.5f) * (1.0f / 65536.0f);
  // -2 ln(u1) = -2 ln2 * log2(u1)
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
  return make_float2(r * __builtin_amdgcn_cosf(u2), r * __builtin_amdgcn_sinf(u2));
}

// Four normals for "quad" q of a sample: elements 4q .. 4q+3 of the flattened sample.
__device__ __forceinline__ float4 rart_normal4(uint32_t k0, uint32_t k1, uint32_t quad, int stream_id,
                                               uint32_t sample) {
  const uint2 w = threefry2x32(k0, k1, rart_ctr0(quad, stream_id), sample);
  const float2 a = rart_boxmuller16(w.x), b = rart_boxmuller16(w.y);
  return make_float4(a.x, a.y, b.x, b.y);
}

__device__ __forceinline__ float rart_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double rart_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float rart_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
// Workgroups are dealt to the 8 XCDs round-robin by block id; tiles of one image share halo rows / columns, so a kernel whose
// neighbouring tiles are consecutive block ids remaps: the returned LOGICAL id is contiguous per XCD (XCD k owns the k-th eighth of
// the ids, bijective for any grid size), and its L2 serves the overlap.
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t rart_xcd_block(uint32_t bid, uint32_t nb) {
  const uint32_t xcd = bid & 7u, slot = bid >> 3, q = nb >> 3, r = nb & 7u;
  return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + slot;
}
#endif

#endif  // __HIPCC__

// ---- internal launchers (one translation unit per kernel family) ---------------------
struct RartCorruptArgs {
  const uint8_t* in;
  uint8_t* out;
  int n, h, w;
  int severity;  // 1..5
  uint64_t seed, sample_offset;
  const void* const* injected;  // host array of device pointers, or nullptr
  int n_injected;
  void* workspace;
  size_t workspace_bytes;
  hipStream_t stream;
};

int rart_launch_pointwise(int corruption_id, const RartCorruptArgs& a);
size_t rart_ws_pointwise(int corruption_id, int severity, int n, int h, int w);
int rart_launch_resample(int corruption_id, const RartCorruptArgs& a);
size_t rart_ws_resample(int corruption_id, int severity, int n, int h, int w);
int rart_launch_jpeg(const RartCorruptArgs& a);
size_t rart_ws_jpeg(int severity, int n, int h, int w);
int rart_launch_stencil(int corruption_id, const RartCorruptArgs& a);
size_t rart_ws_stencil(int corruption_id, int severity, int n, int h, int w);
int rart_launch_composite(int corruption_id, const RartCorruptArgs& a);
size_t rart_ws_composite(int corruption_id, int severity, int n, int h, int w);

#ifdef __HIPCC__
// 16-byte non-temporal accesses (round 6): for streams of hundreds of MB that a kernel reads or writes once.  RART_NT_LAB builds switch the
// call sites marked "lab" in the fused bf16 kernels; the call sites of the pair GEMM / BatchNorm / implicit-GEMM epilogues use them always.
typedef __attribute__((ext_vector_type(4))) uint32_t rart_u32x4;
__device__ __forceinline__ void rart_nt_store16(void* p, const uint4& v) {
  rart_u32x4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<rart_u32x4*>(p));
}
__device__ __forceinline__ uint4 rart_nt_load16(const void* p) {
  const rart_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const rart_u32x4*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
#ifdef RART_NT_LAB
#define RART_LAB_STORE16(P, V) rart_nt_store16((P), (V))
#else
#define RART_LAB_STORE16(P, V) (*reinterpret_cast<uint4*>(P) = (V))
#endif
#endif
