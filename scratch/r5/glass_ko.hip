// Round 5: where a step of glass_blur's copy chain goes.  The table-driven kernel of corrupt_stencil.hip with parts knocked out (KO bits:
// 1 = no LDS reads/writes, 2 = no barrier, 4 = reads only, 8 = dword instead of byte accesses (wrong result, timing only)), 256 workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
struct GlassSched { int delta, iters, N, S, Toff, T, nthr; };
template <int KO>
__global__ __launch_bounds__(768) void k(uint8_t* __restrict__ img_all, const uint8_t* __restrict__ tab_all, GlassSched g) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int HW = 224;
  const int nthr = g.nthr, tid = threadIdx.x;
  uint8_t* gi = img_all + (size_t)blockIdx.x * HW * HW * 3;
  const uint4* g4 = reinterpret_cast<const uint4*>(gi);
  uint4* l4 = reinterpret_cast<uint4*>(lds);
  for (int i = tid; i < HW * HW * 3 / 16; i += nthr) l4[i] = g4[i];
  const int T16 = (g.T + 15) & ~15;
  const uint4* tab = reinterpret_cast<const uint4*>(tab_all + (size_t)blockIdx.x * T16 * nthr) + tid;
  const int it = tid >> 8, a = tid & 255;
  const bool row_ok = a < g.N;
  const int b0 = -it * g.Toff - a * g.S;
  const int pbase = ((HW - g.delta - a) * HW + (HW - g.delta)) * 3;
  const int nbase = -(g.delta * HW + g.delta) * 3;
  auto wanted = [&](int t0) { return row_ok && b0 + t0 + 15 >= 0 && b0 + t0 < g.N; };
  uint4 nxt = make_uint4(0, 0, 0, 0);
  if (wanted(0)) nxt = tab[0];
  __syncthreads();
  uint32_t sink = 0;
  for (int t0 = 0; t0 < T16; t0 += 16) {
    const uint4 cur = nxt;
    if (t0 + 16 < T16 && wanted(t0 + 16)) nxt = tab[(size_t)((t0 >> 4) + 1) * nthr];
    const uint32_t cw[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int b = b0 + t0 + j;
      if (row_ok && b >= 0 && b < g.N) {
        const uint32_t o = (cw[j >> 2] >> (8 * (j & 3))) & 255u;
        uint8_t* p = lds + pbase - 3 * b;
        const uint8_t* q = p + nbase + (int)(o >> 4) * (HW * 3) + (int)(o & 15u) * 3;
        if (KO & 1) {
          sink += (uint32_t)(size_t)q;
        } else if (KO & 8) {
          const uint32_t v = *(const uint32_t*)((size_t)q & ~(size_t)3);
          *(uint32_t*)((size_t)p & ~(size_t)3) = v;
        } else if (KO & 4) {
          sink += q[0] + q[1] + q[2];
        } else {
          const uint8_t q0 = q[0], q1 = q[1], q2 = q[2];
          p[0] = q0; p[1] = q1; p[2] = q2;
        }
      }
      if (!(KO & 2)) __syncthreads();
    }
  }
  if (sink == 0x12345678u) lds[tid] = 1;
  __syncthreads();
  uint4* o4 = reinterpret_cast<uint4*>(gi);
  for (int i = tid; i < HW * HW * 3 / 16; i += nthr) o4[i] = l4[i];
}
template <int KO>
float run(uint8_t* img, uint8_t* tab, GlassSched g, int n) {
  hipFuncSetAttribute((const void*)k<KO>, hipFuncAttributeMaxDynamicSharedMemorySize, 224 * 224 * 3);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KO>, dim3(n), dim3(g.nthr), 224 * 224 * 3, 0, img, tab, g);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<KO>, dim3(n), dim3(g.nthr), 224 * 224 * 3, 0, img, tab, g);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 3 * 1e3f;
}
int main() {
  const int n = 256;
  const int cfg[5][2] = {{1, 2}, {2, 1}, {2, 3}, {3, 2}, {4, 2}};
  uint8_t *img, *tab;
  hipMalloc(&img, (size_t)n * 224 * 224 * 3);
  hipMemset(img, 7, (size_t)n * 224 * 224 * 3);
  for (int s = 0; s < 5; ++s) {
    GlassSched g; g.delta = cfg[s][0]; g.iters = cfg[s][1]; g.N = 224 - 2 * g.delta; g.S = g.delta + 1; g.Toff = g.delta * g.S + g.delta + 1;
    g.T = (g.N - 1) * g.S + g.N + (g.iters - 1) * g.Toff; g.nthr = 256 * g.iters;
    const size_t tb = (size_t)((g.T + 15) & ~15) * g.nthr;
    std::vector<uint8_t> h(tb * n);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)((rand() % (2 * g.delta)) | ((rand() % (2 * g.delta)) << 4));
    hipMalloc(&tab, tb * n); hipMemcpy(tab, h.data(), tb * n, hipMemcpyHostToDevice);
    printf("sev %d (d %d, iters %d, %4d steps, %2d waves): full %7.1f us | no LDS ops %7.1f | no barrier %7.1f | reads only %7.1f | dword %7.1f | no LDS, no barrier %7.1f\n",
           s + 1, g.delta, g.iters, g.T, g.nthr / 64, run<0>(img, tab, g, n), run<1>(img, tab, g, n), run<2>(img, tab, g, n), run<4>(img, tab, g, n), run<8>(img, tab, g, n), run<3>(img, tab, g, n));
    hipFree(tab);
  }
  return 0;
}
