// Round 6 probe: what a PLAIN HIP copy kernel reaches on the bytes one gaussian_noise launch at B = 256 moves (38.5 MB in,
// 38.5 MB out), in the noise kernel's own geometry (one wave per 1 KiB chunk, 16 bytes per lane) and in the usual alternatives, over rotating
// buffer pairs (> the 256 MiB Infinity Cache).  hipcc --offload-arch=gfx950 -O3 scratch/r6/copy_probe.hip -o scratch/r6/copy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void k_copy_chunk(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {   // one 16-byte vector per lane
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) out[i] = in[i];
}
template <int U>
__global__ __launch_bounds__(256) void k_copy_unroll(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {  // U vectors per lane, loads first
  const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  uint4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + (size_t)u * 256 < n16) v[u] = in[base + (size_t)u * 256];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + (size_t)u * 256 < n16) out[base + (size_t)u * 256] = v[u];
}
__global__ __launch_bounds__(256) void k_copy_stride(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {  // persistent grid-stride
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
int main() {
  const size_t bytes = 256ull * 224 * 224 * 3, n16 = bytes / 16;
  const int NB = 9;
  std::vector<uint8_t*> a(NB), b(NB);
  for (int i = 0; i < NB; ++i) { hipMalloc(&a[i], bytes); hipMalloc(&b[i], bytes); hipMemset(a[i], i + 1, bytes); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time_it = [&](const char* name, auto launch) {
    std::vector<float> ts;
    for (int rep = 0; rep < 5; ++rep) {
      for (int i = 0; i < NB; ++i) launch(i);            // warm
      hipEventRecord(e0);
      for (int k = 0; k < 36; ++k) launch(k % NB);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms / 36 * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    const double us = ts[2];
    printf("%-44s %7.2f us per launch  %6.3f of 8 TB/s on %.1f MB moved\n", name, us, 2.0 * bytes / (us * 1e-6) / 8e12, 2.0 * bytes / 1e6);
  };
  time_it("one 16-byte vector per lane (noise geometry)", [&](int i) { hipLaunchKernelGGL(k_copy_chunk, dim3((n16 + 255) / 256), dim3(256), 0, 0, (const uint4*)a[i], (uint4*)b[i], n16); });
  time_it("two vectors per lane, loads first", [&](int i) { hipLaunchKernelGGL(k_copy_unroll<2>, dim3((n16 + 511) / 512), dim3(256), 0, 0, (const uint4*)a[i], (uint4*)b[i], n16); });
  time_it("four vectors per lane, loads first", [&](int i) { hipLaunchKernelGGL(k_copy_unroll<4>, dim3((n16 + 1023) / 1024), dim3(256), 0, 0, (const uint4*)a[i], (uint4*)b[i], n16); });
  time_it("grid-stride, 2048 workgroups", [&](int i) { hipLaunchKernelGGL(k_copy_stride, dim3(2048), dim3(256), 0, 0, (const uint4*)a[i], (uint4*)b[i], n16); });
  time_it("grid-stride, 1024 workgroups", [&](int i) { hipLaunchKernelGGL(k_copy_stride, dim3(1024), dim3(256), 0, 0, (const uint4*)a[i], (uint4*)b[i], n16); });
  time_it("hipMemcpyAsync device to device", [&](int i) { hipMemcpyAsync(b[i], a[i], bytes, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
