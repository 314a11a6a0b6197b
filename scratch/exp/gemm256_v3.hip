// 256 x 256 x 64 bf16 GEMM prototype for the transformer layers (C[M][N] = A[M][K] . W[N][K]^T), gfx950:
// 8 waves (2 x 4), wave tile 128 x 64 (0.75 KiB of LDS fragment reads per MFMA instead of the 1 KiB of a 64 x 64 wave tile),
// tiles by global_load_lds_dwordx4 into unpadded, XOR-swizzled 128-byte rows, two LDS stages.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/exp/gemm256.hip -o scratch/exp/_bin/gemm256
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};
constexpr int TM = 256, TN = 256, TK = 64, STAGE = (TM + TN) * 128;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2; typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  f2 f = {lo, hi}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
}
#define DL(SRC, DST) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), 16, 0, 0);

__global__ __launch_bounds__(256, 1) void k_gemm256(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, uint16_t* __restrict__ C,
                                                    int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int n_tiles = N / TN;
  // all column tiles of a row tile on one XCD (the A tile is re-read from its L2)
  const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
  const int m_tile = (slot / n_tiles) * 8 + xcd, n_tile = slot % n_tiles;
  if (m_tile * TM >= M) return;
  const int m0 = m_tile * TM, n0 = n_tile * TN;
  const int lrow = lane >> 3, swz = ((8 * wave + lrow) >> 1) & 7, csrc = (lane & 7) ^ swz;     // (32 q is a multiple of 16: the swizzle does not depend on q)
  const uint32_t wrow = (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 8u;
  const char* asrc[8]; const char* bsrc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = 32 * q + 8 * wave + lrow;
    asrc[q] = (m0 + r < M) ? reinterpret_cast<const char*>(A + (size_t)(m0 + r) * K + csrc * 8) : nullptr;
    bsrc[q] = reinterpret_cast<const char*>(W + (size_t)(n0 + r) * K + csrc * 8);
  }
#define ISSUE(KT, BUF) { uint8_t* st_ = lds + (BUF)*STAGE; \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) { const char* s_ = asrc[q] ? asrc[q] + (size_t)(KT) * 128 : reinterpret_cast<const char*>(g_zero16); DL(s_, st_ + (32 * q + wrow) * 128) } \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) { DL(bsrc[q] + (size_t)(KT) * 128, st_ + (TM + 32 * q + wrow) * 128) } }
  const int fr = lane & 31, h = lane >> 5;
  uint32_t xo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) xo[ks] = (uint32_t)(fr * 128 + (((2 * ks + h) ^ ((fr >> 1) & 7)) << 4));
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int KT = K / TK;
  ISSUE(0, 0)
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) ISSUE(kt + 1, buf ^ 1)
    const uint8_t* Ab = lds + buf * STAGE + wm * 128 * 128;
    const uint8_t* Bb = lds + buf * STAGE + (TM + wn * 128) * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * 128 + xo[ks]);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * 128 + xo[ks]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
  }
  // epilogue: per wave, 32 rows x 64 columns at a time through LDS -> 128-byte row segments
  float* sE = reinterpret_cast<float*>(lds) + wave * 32 * 132;
  const int cw = lane & 15, rw = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sE[((r & 3) + 8 * (r >> 2) + 4 * h) * 132 + j * 32 + fr] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = q * 4 + rw, row = m0 + wm * 128 + i * 32 + r;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(sE + r * 132 + cw * 8), v1 = *reinterpret_cast<const f32x4*>(sE + r * 132 + cw * 8 + 4);
      if (row < M)
        *reinterpret_cast<uint4*>(C + (size_t)row * N + n0 + wn * 128 + cw * 8) =
            make_uint4(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3]));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
  }
}
static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
void run(int M, int N, int K) {
  std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K), hC((size_t)M * N);
  srand(1);
  for (auto& v : hA) v = f2bf((rand() % 17 - 8) / 8.0f);
  for (auto& v : hW) v = f2bf((rand() % 17 - 8) / 16.0f);
  uint16_t *A, *W, *C; hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, hC.size() * 2);
  hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  const int m_tiles = (M + TM - 1) / TM, m8 = (m_tiles + 7) / 8 * 8, blocks = m8 * (N / TN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k_gemm256, dim3(blocks), dim3(256), 0, 0, A, W, C, M, N, K); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int t = 0; t < 2000; ++t) {
    const int m = rand() % M, n = rand() % N; double s = 0;
    for (int k = 0; k < K; ++k) s += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hW[(size_t)n * K + k]);
    const double e = fabs(s - bf2f(hC[(size_t)m * N + n])) / (fabs(s) + 1.0); if (e > maxerr) maxerr = e;
  }
  printf("M %d N %d K %d: %.1f us  %.0f TFLOP/s  (%d blocks)  max rel err %.2e\n", M, N, K, best * 1e3, 2.0 * M * N * K / best / 1e9, blocks, maxerr);
  hipFree(A); hipFree(W); hipFree(C);
}
int main() { run(50432, 3072, 768); run(50432, 768, 3072); run(50432, 2304, 768); run(50432, 768, 768); run(8192, 8192, 8192); return 0; }
