// Variants of k_normal_noise_mfma<0> (gaussian_noise, B = 256) timed side by side on rotating buffers (> the Infinity Cache),
// each checked against V0 = the shipped kernel's arithmetic.  Standalone:  hipcc --offload-arch=gfx950 -O3 -I include
//   -I robustart_amd/csrc -mllvm -amdgpu-mfma-vgpr-form scratch/exp/noise_variants.hip -o scratch/exp/_bin/noise_variants
#include "rart_common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr int kBlock = 256;

struct HadamardTable { uint32_t w[64][4]; };
constexpr HadamardTable make_hadamard() {
  HadamardTable t{};
  for (int lane = 0; lane < 64; ++lane)
    for (int q = 0; q < 4; ++q) {
      uint32_t word = 0;
      for (int b = 0; b < 4; ++b) {
        const int k = 16 * (lane >> 5) + q * 4 + b, j = lane & 31;
        int x = k & j, par = 0;
        while (x) { par ^= x & 1; x >>= 1; }
        word |= (uint32_t)(par ? 0xFFu : 0x01u) << (8 * b);
      }
      t.w[lane][q] = word;
    }
  return t;
}
__device__ const HadamardTable g_hadamard = make_hadamard();
constexpr float kCltSigma = 418.0334915f;
constexpr float kCltA = 0.9953120f / kCltSigma;
constexpr float kCltB = 0.0015627f / (kCltSigma * kCltSigma * kCltSigma);

__device__ __forceinline__ i32x4 had(int lane) {
  i32x4 b = {(int)g_hadamard.w[lane][0], (int)g_hadamard.w[lane][1], (int)g_hadamard.w[lane][2], (int)g_hadamard.w[lane][3]};
  return b;
}
__device__ __forceinline__ void clt_sums16(uint32_t k0, uint32_t k1, uint32_t chunk, uint32_t sample, int lane, const i32x4 b, float* s) {
  const uint2 w0 = threefry2x32(k0, k1, rart_ctr0(chunk * 128u + lane * 2u, 14), sample);
  const uint2 w1 = threefry2x32(k0, k1, rart_ctr0(chunk * 128u + lane * 2u + 1u, 14), sample);
  const i32x4 a = {(int)(w0.x | 0x01010101u), (int)(w0.y | 0x01010101u), (int)(w1.x | 0x01010101u), (int)(w1.y | 0x01010101u)};
  i32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 16; ++e) s[e] = (float)acc[e];
}
__device__ __forceinline__ uint32_t pack4_floor_sat(float a, float b, float c, float d) {
  uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(a), 0, 0u);
  w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(b), 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(c), 2, w);
  return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(d), 3, w);
}
__device__ __forceinline__ uint32_t pack4_sat(float a, float b, float c, float d) {      // conversion's own rounding
  uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(a, 0, 0u);
  w = __builtin_amdgcn_cvt_pk_u8_f32(b, 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(c, 2, w);
  return __builtin_amdgcn_cvt_pk_u8_f32(d, 3, w);
}

// MODE 0: floor + cvt (shipped).  MODE 1: MODE register round-toward-zero for the whole wave, no floor.
// MODE 2: fold -0.5 into the addend (x - 0.5 by one packed add), nearest-even conversion, no floor.
template <int MODE>
__device__ __forceinline__ u32x4 process(const u32x4 cur, const float* s, f32x2 ga2, f32x2 gb2) {
  const uint32_t wi[4] = {cur[0], cur[1], cur[2], cur[3]};
  u32x4 wo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x2 y[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      f32x2 x = {(float)((wi[j] >> (16 * hh)) & 0xFFu), (float)((wi[j] >> (16 * hh + 8)) & 0xFFu)};
      if (MODE == 2) x = x - (f32x2){0.5f, 0.5f};
      const f32x2 sv = {s[j * 4 + 2 * hh], s[j * 4 + 2 * hh + 1]};
      const f32x2 t = __builtin_elementwise_fma(gb2, sv * sv, ga2);
      y[hh] = __builtin_elementwise_fma(t, sv, x);
    }
    wo[j] = MODE == 0 ? pack4_floor_sat(y[0].x, y[0].y, y[1].x, y[1].y) : pack4_sat(y[0].x, y[0].y, y[1].x, y[1].y);
  }
  return wo;
}

// ---- V0: the shipped kernel ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void v0(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint32_t cps, uint32_t total,
                                             float c, uint32_t k0, uint32_t k1, uint32_t sbase) {
  const int lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (g >= total) return;
  const u32x4 cur = in[(size_t)g * 64 + lane];
  const i32x4 hb = had(lane);
  const float gsc = 255.0f * c;
  const f32x2 ga2 = {gsc * kCltA, gsc * kCltA}, gb2 = {gsc * kCltB, gsc * kCltB};
  const uint32_t sample = g / cps, chunk = g - sample * cps;
  float s[16];
  clt_sums16(k0, k1, chunk, sbase + sample, lane, hb, s);
  out[(size_t)g * 64 + lane] = process<0>(cur, s, ga2, gb2);
}

// ---- V1: wave-uniform chunk index in SGPRs (readfirstlane), multiply-shift division, MODE = 1 or 2 --------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void v1(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint32_t cps, uint32_t total,
                                             float c, uint32_t k0, uint32_t k1, uint32_t sbase, uint32_t magic, uint32_t shift) {
  if (MODE == 1) __builtin_amdgcn_s_setreg(1 | (0 << 6) | (1 << 11), 3);      // MODE.fp_round (single precision) = toward zero
  const int lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (g >= total) return;
  const u32x4 cur = in[(size_t)g * 64 + lane];
  const i32x4 hb = had(lane);
  const float gsc = 255.0f * c;
  const f32x2 ga2 = {gsc * kCltA, gsc * kCltA}, gb2 = {gsc * kCltB, gsc * kCltB};
  const uint32_t sample = (uint32_t)(((uint64_t)g * magic) >> shift), chunk = g - sample * cps;
  float s[16];
  clt_sums16(k0, k1, chunk, sbase + sample, lane, hb, s);
  out[(size_t)g * 64 + lane] = process<MODE>(cur, s, ga2, gb2);
}

// ---- V2: NC chunks per wave, every load issued up front (straight-line code) ----------------------------------------------------
template <int MODE, int NC>
__global__ __launch_bounds__(kBlock) void v2(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint32_t cps, uint32_t total,
                                             float c, uint32_t k0, uint32_t k1, uint32_t sbase, uint32_t magic, uint32_t shift) {
  if (MODE == 1) __builtin_amdgcn_s_setreg(1 | (0 << 6) | (1 << 11), 3);
  const int lane = threadIdx.x & 63;
  const uint32_t w = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t nw = gridDim.x * (kBlock / 64);
  u32x4 cur[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    const uint32_t g = w + q * nw;
    cur[q] = (u32x4){0u, 0u, 0u, 0u};
    if (g < total) cur[q] = in[(size_t)g * 64 + lane];
  }
  const i32x4 hb = had(lane);
  const float gsc = 255.0f * c;
  const f32x2 ga2 = {gsc * kCltA, gsc * kCltA}, gb2 = {gsc * kCltB, gsc * kCltB};
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    const uint32_t g = w + q * nw;
    if (g >= total) break;
    const uint32_t sample = (uint32_t)(((uint64_t)g * magic) >> shift), chunk = g - sample * cps;
    float s[16];
    clt_sums16(k0, k1, chunk, sbase + sample, lane, hb, s);
    out[(size_t)g * 64 + lane] = process<MODE>(cur[q], s, ga2, gb2);
  }
}

// ---- V3: persistent waves, loads two chunks ahead.  Every VMEM instruction and every wait is inline asm, so the compiler inserts
//      no waits of its own.  Order per iteration: compute(i), store(i), load(i+2), then s_waitcnt vmcnt(1): outstanding are
//      load(i+1), store(i), load(i+2); loads return in order, so "at most one outstanding" implies load(i+1) has landed whatever
//      the store does. ------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void v3(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint32_t cps, uint32_t total,
                                             float c, uint32_t k0, uint32_t k1, uint32_t sbase, uint32_t magic, uint32_t shift) {
  if (MODE == 1) __builtin_amdgcn_s_setreg(1 | (0 << 6) | (1 << 11), 3);
  const int lane = threadIdx.x & 63;
  const uint32_t w = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t nw = gridDim.x * (kBlock / 64);
  if (w >= total) return;
  i32x4 hb = had(lane);
  asm volatile("" : "+v"(hb));                  // the table load (tracked by the compiler) is waited for HERE, not inside the loop
  const float gsc = 255.0f * c;
  const f32x2 ga2 = {gsc * kCltA, gsc * kCltA}, gb2 = {gsc * kCltB, gsc * kCltB};
  const u32x4* pin = in + lane;
  u32x4* pout = out + lane;
  u32x4 a, b;                                  // a = chunk i, b = chunk i + 1
#define LD(REG, G) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(REG) : "v"(pin + (size_t)(G)*64) : "memory")
#define ST(G, VAL) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(pout + (size_t)(G)*64), "v"(VAL) : "memory")
  LD(a, w);
  const uint32_t g1 = w + nw < total ? w + nw : w;          // clamp: a redundant in-range load instead of a branch
  LD(b, g1);
  asm volatile("s_waitcnt vmcnt(1)" : "+v"(a), "+v"(b));
  for (uint32_t g = w; g < total; g += 2 * nw) {
    {
      const uint32_t sample = (uint32_t)(((uint64_t)g * magic) >> shift), chunk = g - sample * cps;
      float s[16];
      clt_sums16(k0, k1, chunk, sbase + sample, lane, hb, s);
      const u32x4 r = process<MODE>(a, s, ga2, gb2);
      ST(g, r);
      const uint32_t g2 = g + 2 * nw < total ? g + 2 * nw : w;
      LD(a, g2);
      asm volatile("s_waitcnt vmcnt(1)" : "+v"(a), "+v"(b));     // b (loaded before this store) has landed
    }
    const uint32_t gb = g + nw;
    if (gb >= total) break;
    {
      const uint32_t sample = (uint32_t)(((uint64_t)gb * magic) >> shift), chunk = gb - sample * cps;
      float s[16];
      clt_sums16(k0, k1, chunk, sbase + sample, lane, hb, s);
      const u32x4 r = process<MODE>(b, s, ga2, gb2);
      ST(gb, r);
      const uint32_t g3 = gb + 2 * nw < total ? gb + 2 * nw : w;
      LD(b, g3);
      asm volatile("s_waitcnt vmcnt(1)" : "+v"(a), "+v"(b));     // a has landed
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b));
#undef LD
#undef ST
}

static void magic_for(uint32_t dv, uint32_t& mg, uint32_t& sh) {
  uint32_t l = 0;
  while ((1ull << l) < dv) ++l;
  sh = 31 + l;
  mg = (uint32_t)(((1ull << sh) + dv - 1) / dv);
}

int main() {
  const int B = 256, NP = 9;
  const size_t bytes = (size_t)B * 224 * 224 * 3;
  const uint32_t cps = 147, total = cps * B;
  uint32_t mg, sh;
  magic_for(cps, mg, sh);
  std::vector<uint8_t> h(bytes);
  srand(7);
  for (size_t i = 0; i < bytes; ++i) h[i] = (uint8_t)(rand() >> 7);
  std::vector<u32x4*> src(NP), dst(NP);
  for (int i = 0; i < NP; ++i) {
    hipMalloc(&src[i], bytes); hipMalloc(&dst[i], bytes);
    hipMemcpy(src[i], h.data(), bytes, hipMemcpyHostToDevice);
  }
  std::vector<uint8_t> ref(bytes), got(bytes);
  const float c = 0.18f;
  const uint32_t k0 = 0, k1 = 0, sb = 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, bool is_ref) {
    for (int i = 0; i < NP; ++i) launch(src[i], dst[i]);
    hipDeviceSynchronize();
    if (hipGetLastError() != hipSuccess) { printf("%-44s LAUNCH ERROR\n", name); return; }
    float tot = 0.f, best = 1e9f;
    const int L = 45;
    for (int i = 0; i < L; ++i) {
      hipEventRecord(e0); launch(src[i % NP], dst[i % NP]); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms; if (ms < best) best = ms;
    }
    hipMemcpy(is_ref ? ref.data() : got.data(), dst[0], bytes, hipMemcpyDeviceToHost);
    size_t diff = 0; int maxd = 0;
    if (!is_ref) for (size_t i = 0; i < bytes; ++i) { const int d = abs((int)got[i] - (int)ref[i]); if (d) { ++diff; if (d > maxd) maxd = d; } }
    const double us = tot / L * 1e3;
    printf("%-44s avg %.2f us  best %.2f us  %.2f TB/s  frac %.3f  | vs V0: %zu of %zu differ (max %d)\n", name, us, best * 1e3,
           2.0 * bytes / us / 1e6, 2.0 * bytes / (us * 1e-6) / 8e12, diff, bytes, maxd);
  };
  const dim3 g1((total + 3) / 4);
  run("V0 shipped", [&](u32x4* s, u32x4* d) { hipLaunchKernelGGL(v0, g1, dim3(kBlock), 0, 0, s, d, cps, total, c, k0, k1, sb); }, true);
  run("V1 scalar g + RTZ mode, no floor", [&](u32x4* s, u32x4* d) { hipLaunchKernelGGL(v1<1>, g1, dim3(kBlock), 0, 0, s, d, cps, total, c, k0, k1, sb, mg, sh); }, false);
  run("V1 scalar g + (x - 0.5), no floor", [&](u32x4* s, u32x4* d) { hipLaunchKernelGGL(v1<2>, g1, dim3(kBlock), 0, 0, s, d, cps, total, c, k0, k1, sb, mg, sh); }, false);
  run("V1 scalar g only (floor kept)", [&](u32x4* s, u32x4* d) { hipLaunchKernelGGL(v1<0>, g1, dim3(kBlock), 0, 0, s, d, cps, total, c, k0, k1, sb, mg, sh); }, false);
  for (int mode = 1; mode <= 2; ++mode) {
    char nm[96];
    snprintf(nm, sizeof nm, "V2 2 chunks/wave up front, mode %d", mode);
    const dim3 g2((total / 2 + 3) / 4);
    if (mode == 1) run(nm, [&](u32x4* s, u32x4* d) { hipLaunchKernelGGL((v2<1, 2>), g2, dim3(kBlock), 0, 0, s, d, cps, total, c, k0, k1, sb, mg, sh); }, false);
    else run(nm, [&](u32x4* s, u32x4* d) { hipLaunchKernelGGL((v2<2, 2>), g2, dim3(kBlock), 0, 0, s, d, cps, total, c, k0, k1, sb, mg, sh); }, false);
  }
  for (int wpc : {2048, 1536, 1280, 1024, 768}) {                 // persistent grid: workgroups (4 waves each)
    char nm[96];
    snprintf(nm, sizeof nm, "V3 persistent, %d WGs, prefetch 2, RTZ", wpc);
    run(nm, [&](u32x4* s, u32x4* d) { hipLaunchKernelGGL(v3<1>, dim3(wpc), dim3(kBlock), 0, 0, s, d, cps, total, c, k0, k1, sb, mg, sh); }, false);
    snprintf(nm, sizeof nm, "V3 persistent, %d WGs, prefetch 2, -0.5", wpc);
    run(nm, [&](u32x4* s, u32x4* d) { hipLaunchKernelGGL(v3<2>, dim3(wpc), dim3(kBlock), 0, 0, s, d, cps, total, c, k0, k1, sb, mg, sh); }, false);
  }
  return 0;
}
