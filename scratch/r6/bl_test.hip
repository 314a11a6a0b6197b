// Round 6 probe: semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 -- LDS placement (M0 base + 16 * lane?), out-of-range lanes (zero
// written? soffset part of the range check?).  hipcc --offload-arch=gfx950 scratch/r6/bl_test.hip -o scratch/r6/bl_test && gpurun -- scratch/r6/bl_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint32_t* src, uint32_t* out, unsigned nbytes, unsigned soff, unsigned bad_lane_mask_lo, unsigned bad_off) {
  __shared__ uint32_t lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  const unsigned long long b = (unsigned long long)src;
  u4 srd = {(unsigned)b, (unsigned)(b >> 32) & 0xFFFFu, nbytes, 0x00020000u};
  const unsigned lane = threadIdx.x;
  unsigned voff = lane * 64;                                     // lane l reads 16 bytes at l * 64
  if (lane < 32 && ((bad_lane_mask_lo >> lane) & 1u)) voff = bad_off;
  const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds + 1024;   // destination: byte 1024 of the array
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(srd), "s"(soff), "s"(la) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
  const int N = 1 << 16;
  std::vector<uint32_t> h(N);
  for (int i = 0; i < N; ++i) h[i] = i;
  uint32_t *d, *o;
  hipMalloc(&d, N * 4); hipMalloc(&o, 2048 * 4);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<uint32_t> r(2048);
  struct { unsigned nbytes, soff, mask, bad; const char* what; } cases[] = {
      {N * 4u, 0, 0, 0, "all in range, soffset 0"},
      {N * 4u, 256, 0, 0, "all in range, soffset 256"},
      {N * 4u, 256, 0x5, 0xFFFFFFFFu, "lanes 0 and 2 at voffset 0xFFFFFFFF, soffset 256"},
      {N * 4u, 256, 0x5, 0x80000000u, "lanes 0 and 2 at voffset 0x80000000, soffset 256"},
      {2048u, 1024, 0, 0, "num_records 2048, soffset 1024: lanes >= 32 out of range by voffset; lanes 16..31 only by voffset + soffset"},
  };
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, c.nbytes, c.soff, c.mask, c.bad);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    printf("%s\n", c.what);
    bool before_ok = true, after_ok = true;
    for (int i = 0; i < 256; ++i) before_ok &= r[i] == 0xDEADBEEFu;
    for (int i = 512; i < 2048; ++i) after_ok &= r[i] == 0xDEADBEEFu;
    printf("  untouched before / after the 1 KiB window: %d / %d\n", before_ok, after_ok);
    for (int l = 0; l < 64; l += 1) {
      if (l < 4 || (l >= 14 && l < 18) || (l >= 30 && l < 34) || l == 63)
        printf("  lane %2d -> lds words %d..: %u %u %u %u   (source word of lane: %u)\n", l, 256 + 4 * l, r[256 + 4 * l], r[256 + 4 * l + 1],
               r[256 + 4 * l + 2], r[256 + 4 * l + 3], (l * 64 + c.soff) / 4);
    }
  }
  return 0;
}
