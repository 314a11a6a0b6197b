// Phase timing of k_bottleneck56 (cycle stamps per wave):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include scratch/exp/bneck_ts.hip -o gpurun_out/bneck_ts && gpurun_out/bneck_ts
#define RART_BNECK_TS 1
#include "../../robustart_amd/csrc/bottleneck_fused.hip"
#include <cstdio>
#include <vector>
#include <cstdarg>
void rart_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }

template <bool BWD>
void run(int n) {
  const size_t P = (size_t)n * 56 * 56, elems = P * 256;
  uint16_t *x, *out, *w1, *w2, *w3; float* bias; uint8_t *m1, *m2, *m3; unsigned long long* ts;
  hipMalloc(&x, elems * 2); hipMalloc(&out, elems * 2); hipMalloc(&w1, 128 * 256 * 2); hipMalloc(&w2, 128 * 576 * 2); hipMalloc(&w3, 256 * 64 * 2);
  hipMalloc(&bias, 256 * 4); hipMalloc(&m1, P * 8); hipMalloc(&m2, P * 8); hipMalloc(&m3, P * 32);
  hipMemset(x, 0x3c, elems * 2); hipMemset(w1, 0x3c, 128 * 256 * 2); hipMemset(w2, 0x3c, 128 * 576 * 2); hipMemset(w3, 0x3c, 256 * 64 * 2);
  hipMemset(bias, 0, 256 * 4); hipMemset(m1, 0x5a, P * 8); hipMemset(m2, 0xa5, P * 8); hipMemset(m3, 0x3c, P * 32);
  const size_t nblk = (size_t)n * BF_TPI;
  hipMalloc(&ts, nblk * 4 * 7 * 8);
  RartBneckDesc d{};
  d.x = x; d.w4 = nullptr; d.w1 = w1; d.w2 = w2; d.w3 = w3; d.b1 = d.b2 = d.b3 = BWD ? nullptr : bias; d.m1 = m1; d.m2 = m2; d.m3 = m3; d.out = out;
  d.tiles = (uint32_t)nblk; d.ts = ts;
  for (int t = 0; t < 9; ++t) d.tap_off[t] = ((t / 3 - 1) * BF_SW + (t % 3 - 1)) * 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_bottleneck56<BWD, false>), dim3(nblk), dim3(256), 0, 0, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s n=%d: %.1f us (%zu blocks)\n", BWD ? "bwd" : "fwd", n, ms * 1e3, nblk);
  }
  std::vector<unsigned long long> hts(nblk * 4 * 7);
  hipMemcpy(hts.data(), ts, hts.size() * 8, hipMemcpyDeviceToHost);
  double ph[6] = {0, 0, 0, 0, 0, 0};
  for (size_t b = 0; b < nblk * 4; ++b)
    for (int k = 0; k < 6; ++k) ph[k] += (double)(hts[b * 7 + k + 1] - hts[b * 7 + k]);
  const char* nm[6] = {"prologue (W1 -> LDS) + barrier", "stage A (x . W1 -> T1)", "barrier", "W3 issue, sign pass, stage B (taps)", "W3 -> LDS + barrier", "stage C (+ residual, stores)"};
  for (int k = 0; k < 6; ++k) printf("   %-40s %9.0f cycles / wave\n", nm[k], ph[k] / (nblk * 4));
  hipFree(x); hipFree(out); hipFree(ts);
}
int main() { run<false>(256); run<true>(256); return 0; }
