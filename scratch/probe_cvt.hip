#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* f, unsigned* o, int n) {
  int i = threadIdx.x;
  if (i < n) o[i] = __builtin_amdgcn_cvt_pk_u8_f32(f[i], 0, 0u);
}
int main() {
  float h[] = {-5.f, -0.6f, -0.4f, 0.f, 0.4f, 0.5f, 0.6f, 0.999f, 1.0f, 1.5f, 2.5f, 3.5f, 127.49f, 127.5f, 254.5f, 254.99f, 255.0f, 255.4f, 255.6f, 300.f, 1e9f};
  int n = sizeof(h) / sizeof(float);
  float* d; unsigned* o; unsigned ho[32];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 32 * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, n);
  hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%g -> %u\n", h[i], ho[i]);
  return 0;
}
