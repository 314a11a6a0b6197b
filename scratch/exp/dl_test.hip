// semantics check of global_load_lds_dwordx4 (gfx950): LDS address = wave-uniform base + lane * 16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4* __restrict__ g, uint4* __restrict__ o) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // lane l of wave w fetches element (w * 64 + (l ^ 5)) : a per-lane permuted source
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + wave * 64 + (lane ^ 5)),
                                   (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  o[threadIdx.x] = *reinterpret_cast<uint4*>(lds + wave * 1024 + lane * 16);
}
int main() {
  std::vector<uint4> h(256), r(256);
  for (int i = 0; i < 256; ++i) h[i] = make_uint4(i, i * 2, i * 3, i * 4);
  uint4 *g, *o; hipMalloc(&g, 4096); hipMalloc(&o, 4096);
  hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, g, o);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) { const int w = i >> 6, l = i & 63, src = w * 64 + (l ^ 5); if (r[i].x != (unsigned)src || r[i].w != (unsigned)src * 4) ++bad; }
  printf("global_load_lds_dwordx4: %d mismatches (r[1].x = %u, expected %u)\n", bad, r[1].x, 1 ^ 5);
  return bad;
}
