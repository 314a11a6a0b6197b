// throughput of global_load_lds_dwordx4 vs global_load_dwordx4 (+ ds_write_b128) from an L2-resident buffer, 512 threads per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#define DL(SRC, DST) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), (__attribute__((address_space(3))) void*)(DST), 16, 0, 0);
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(const char* __restrict__ g, int iters, int* out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[131072];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t wbase = (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 8192u;
  const char* p = g + (size_t)(blockIdx.x & 15) * 65536 + wave * 8192 + lane * 16;
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) DL(p + q * 1024, lds + wbase + q * 1024)
      __builtin_amdgcn_s_waitcnt(0);
    } else {
      u32x4 r[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) r[q] = *reinterpret_cast<const u32x4*>(p + q * 1024);
#pragma unroll
      for (int q = 0; q < 8; ++q) { if (MODE == 1) *reinterpret_cast<u32x4*>(lds + wbase + q * 1024 + lane * 16) = r[q]; else acc += r[q]; }
    }
    __syncthreads();
  }
  if (MODE != 2) acc = *reinterpret_cast<u32x4*>(lds + threadIdx.x * 16);
  if (acc[0] == 0x12345678u) out[0] = 1;
}
template <int MODE> void run(const char* name, const char* g, int* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, g, 10, out);
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, g, iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * iters * 65536;
  printf("%-34s %.1f us  %.2f TB/s chip  %.1f B/clk/CU at 2.1 GHz\n", name, ms * 1e3, bytes / ms / 1e9, bytes / 256 / (ms * 1e-3) / 2.1e9);
}
int main() {
  char* g; int* out; hipMalloc(&g, 1 << 20); hipMalloc(&out, 4); hipMemset(g, 1, 1 << 20);
  run<0>("global_load_lds_dwordx4", g, out);
  run<1>("global_load_dwordx4 + ds_write_b128", g, out);
  run<2>("global_load_dwordx4 only", g, out);
  return 0;
}
