// calibration microbenchmarks: MFMA issue ceiling (with/without barriers), HBM write-only / read-only / copy rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC, bool BARRIER>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
  bf16x8 a, b;
  for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(float)(threadIdx.x + q); b[q] = (__bf16)(float)(threadIdx.x * 3 + q); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    if (BARRIER) __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[threadIdx.x] = s;
}
__global__ void k_write(uint4* dst, size_t n16, uint32_t v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) dst[i] = make_uint4(v, v, v, v);
}
__global__ void k_read(const uint4* src, size_t n16, uint32_t* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i < n16; i += stride) { uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345) sink[0] = acc;
}
__global__ void k_copy(const uint4* src, uint4* dst, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) dst[i] = src[i];
}
// one-shot (non grid-stride) write: 1 uint4 per thread
__global__ void k_write1(uint4* dst, uint32_t v) { dst[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = make_uint4(v, v, v, v); }

int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* out; CK(hipMalloc(&out, 4096));
  auto timeit = [&](auto fn, int reps) { fn(); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < reps; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps; };
  const int iters = 4000;
  for (int bpc : {1, 2, 3, 4}) {
    const int blocks = 256 * bpc;
    const double flop = (double)blocks * 4 * iters * 16 * 32768.0;
    float t1 = timeit([&] { hipLaunchKernelGGL((k_mfma<4, false>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    float t2 = timeit([&] { hipLaunchKernelGGL((k_mfma<4, true>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    float t3 = timeit([&] { hipLaunchKernelGGL((k_mfma<2, false>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    float t4 = timeit([&] { hipLaunchKernelGGL((k_mfma<1, false>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5);
    printf("mfma blocks/CU %d: 4acc %.0f TF  4acc+barrier %.0f TF  2acc %.0f TF  1acc %.0f TF\n", bpc, flop / t1 / 1e9, flop / t2 / 1e9, flop / t3 / 1e9, flop / t4 / 1e9);
  }
  for (size_t mb : {64, 256, 1024, 4096}) {
    const size_t bytes = mb << 20, n16 = bytes / 16;
    uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
    uint32_t* sink = (uint32_t*)out;
    for (int grid : {2048, 8192}) {
      float tw = timeit([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n16, 7u); }, 10);
      float tr = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n16, sink); }, 10);
      float tc = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n16); }, 10);
      printf("%5zu MiB grid %5d: write %.2f TB/s  read %.2f TB/s  copy(r+w) %.2f TB/s\n", mb, grid, bytes / tw / 1e9, bytes / tr / 1e9, 2.0 * bytes / tc / 1e9);
    }
    float tw1 = timeit([&] { hipLaunchKernelGGL(k_write1, dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, b, 7u); }, 10);
    float tm = timeit([&] { hipMemsetAsync(b, 0, bytes, 0); }, 10);
    printf("%5zu MiB one-shot write %.2f TB/s   hipMemset %.2f TB/s\n", mb, bytes / tw1 / 1e9, bytes / tm / 1e9);
    hipFree(a); hipFree(b);
  }
  return 0;
}
