// Phase timing of k_conv3x3_halo (cycle stamps per wave): hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DRART_HALO_TS
//   scratch/exp/halo_ts.hip -o /tmp/halo_ts && /tmp/halo_ts
#define RART_HALO_TS 1
#include "../../robustart_amd/csrc/conv3x3_halo.hip"
#include <cstdio>
#include <vector>
#include <cstdarg>
void rart_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }

template <int C>
void run(int n, int h, int w) {
  const size_t elems = (size_t)n * h * w * C;
  uint16_t *src, *dst, *wgt; float* bias; unsigned long long* ts;
  hipMalloc(&src, elems * 2); hipMalloc(&dst, elems * 2); hipMalloc(&wgt, (size_t)128 * 9 * C * 2); hipMalloc(&bias, C * 4);
  hipMemset(src, 0x3c, elems * 2); hipMemset(wgt, 0x3c, (size_t)128 * 9 * C * 2); hipMemset(bias, 0, C * 4);
  const int rpb = halo_rows_per_block(C, w);
  const size_t nblk = ((size_t)n * h + rpb - 1) / rpb;
  hipMalloc(&ts, nblk * 4 * 5 * 8);
  RartHaloDesc d{};
  d.src = src; d.wgt = wgt; d.bias = bias; d.mask_bits = nullptr; d.sign_out = nullptr; d.dst = dst;
  d.rows_total = n * h; d.h = h; d.w = w; d.relu = 1; d.ts = ts; d.rows_per_block = rpb;
  for (int t = 0; t < 9; ++t) { d.tap_dy[t] = t / 3 - 1; d.tap_dx[t] = t % 3 - 1; }
  magic_for(w, d.w_magic, d.w_shift); magic_for(h, d.h_magic, d.h_shift); magic_for(w + 2, d.w2_magic, d.w2_shift);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_conv3x3_halo<C>, dim3(nblk), dim3(256), 0, 0, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("C=%d n=%d %dx%d: %.1f us (%zu blocks)\n", C, n, h, w, ms * 1e3, nblk);
  }
  std::vector<unsigned long long> hts(nblk * 4 * 5);
  hipMemcpy(hts.data(), ts, hts.size() * 8, hipMemcpyDeviceToHost);
  double ph[4] = {0, 0, 0, 0};
  for (size_t b = 0; b < nblk * 4; ++b)
    for (int k = 0; k < 4; ++k) ph[k] += (double)(hts[b * 5 + k + 1] - hts[b * 5 + k]);
  const char* nm[4] = {"weights + halo staging issue/store", "geometry + barrier", "K loop (9 taps)", "barrier + epilogue"};
  for (int k = 0; k < 4; ++k) printf("   %-36s %9.0f cycles / wave\n", nm[k], ph[k] / (nblk * 4));
}
int main() { run<64>(256, 56, 56); run<128>(256, 28, 28); return 0; }
