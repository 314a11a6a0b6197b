// spatter, water branch (severities 1-3) for gfx950 -- RobustART/noise/utils/imagenet_c/corruptions.py:305-328:
//   liquid (thresholded gaussian-blurred normal field) -> uint8 -> 255 - cv2.Canny(50, 150) -> cv2.distanceTransform(L2, 5)
//   -> threshold TRUNC 20 -> blur 3x3 -> uint8 -> equalizeHist -> filter2D(3x3 emboss) -> blur 3x3 -> m = liquid * dist,
//   m /= max(m), m *= c4 -> clip(x + m * turquoise, 0, 1).
// OpenCV is absent from the reference tree and from this image: the stages restate OpenCV 4.5's imgproc sources (SURVEY.md
// Appendix B; oracle/corruptions_np.py cv_canny_u8 ... cv_blur3_u8 are the checker) -- PARITY UNPINNED against cv2 itself.
// Every stage after the uint8 cast is integer / 16.16 fixed-point arithmetic, so these kernels match the oracle bit for bit.
//
// One workgroup per image (the stages are whole-image dependent: hysteresis, the two raster passes of the chamfer distance,
// the histogram, the global maximum); planes live in LDS (50 KB each) or, for the 32-bit distance plane, in the caller's
// workspace (208 KB per image, L2 resident).  The raster recurrences tmp[j] = min(cand[j], tmp[j-1] + a) of the distance
// transform are evaluated as a*j + prefix_min(cand[k] - a*k): a block-wide scan per row instead of a serial walk.
#include "rart_common.h"

#pragma clang fp contract(off)

namespace {
constexpr int HW = 224;
constexpr int NPIX = HW * HW;
constexpr int kT = 256;
constexpr int kTW = 1024;                                 // the whole-image kernels (one workgroup per image, 16 waves: round 5; 4 waves until then)
constexpr int DA = 65536, DB = 91750, DC = 143976;       // cvRound({1, 1.4, 2.1969} * 2^16)
constexpr int DINIT = 2147483647 >> 2;
constexpr int TW = HW + 4;                                // padded distance-plane row

__device__ __forceinline__ int reflect101(int i) { return i < 0 ? -i : (i >= HW ? 2 * HW - 2 - i : i); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// liquid fp64 (already blurred) -> thresholded uint8 plane: liquid[liquid < c3] = 0; (liquid * 255).astype(np.uint8)
__global__ __launch_bounds__(kT) void k_spatter_liquid_u8(const double* __restrict__ liquid, uint8_t* __restrict__ l8,
                                                          double thresh, size_t total) {
  for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < total; i += (size_t)gridDim.x * kT) {
    double v = liquid[i];
    if (v < thresh) v = 0.0;
    l8[i] = (uint8_t)(long long)(v * 255.0);
  }
}

// ---- Canny (aperture 3, L1 magnitude, thresholds low / high) -> src8 = 255 - edges -----------------------------------
__device__ __forceinline__ void sobel(const uint8_t* L, int y, int x, int& dx, int& dy) {
  const int ym = clampi(y - 1, 0, HW - 1), yp = clampi(y + 1, 0, HW - 1);       // BORDER_REPLICATE
  const int xm = clampi(x - 1, 0, HW - 1), xp = clampi(x + 1, 0, HW - 1);
  const int a = L[ym * HW + xm], b = L[ym * HW + x], c = L[ym * HW + xp];
  const int d = L[y * HW + xm], f = L[y * HW + xp];
  const int g = L[yp * HW + xm], h = L[yp * HW + x], i = L[yp * HW + xp];
  dx = (c + 2 * f + i) - (a + 2 * d + g);
  dy = (g + 2 * h + i) - (a + 2 * b + c);
}
__device__ __forceinline__ int mag_at(const uint8_t* L, int y, int x) {
  if ((unsigned)y >= (unsigned)HW || (unsigned)x >= (unsigned)HW) return 0;     // zero border of the magnitude buffer
  int dx, dy;
  sobel(L, y, x, dx, dy);
  return abs(dx) + abs(dy);
}

__global__ __launch_bounds__(kTW) void k_spatter_canny(const uint8_t* __restrict__ l8_all, uint8_t* __restrict__ src8_all,
                                                      int low, int high) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* L = lds;                 // [NPIX]
  uint8_t* map = lds + NPIX;        // [NPIX]: 0 candidate, 1 not an edge, 2 edge
  __shared__ int s_changed;
  const uint8_t* l8 = l8_all + (size_t)blockIdx.x * NPIX;
  for (int i = threadIdx.x; i < NPIX / 16; i += kTW) reinterpret_cast<uint4*>(L)[i] = reinterpret_cast<const uint4*>(l8)[i];
  __syncthreads();
  for (int p = threadIdx.x; p < NPIX; p += kTW) {
    const int y = p / HW, x = p - y * HW;
    int dx, dy;
    sobel(L, y, x, dx, dy);
    const int m = abs(dx) + abs(dy);
    uint8_t code = 1;
    if (m > low) {
      const long long ax = abs(dx), ay = (long long)abs(dy) << 15;
      const long long tg22x = ax * 13573;
      bool is_max;
      if (ay < tg22x) {
        is_max = m > mag_at(L, y, x - 1) && m >= mag_at(L, y, x + 1);
      } else {
        const long long tg67x = tg22x + (ax << 16);
        if (ay > tg67x) {
          is_max = m > mag_at(L, y - 1, x) && m >= mag_at(L, y + 1, x);
        } else {
          const int s = ((dx < 0) != (dy < 0)) ? -1 : 1;
          is_max = m > mag_at(L, y - 1, x - s) && m > mag_at(L, y + 1, x + s);
        }
      }
      if (is_max) code = m > high ? 2 : 0;
    }
    map[p] = code;
  }
  __syncthreads();
  // hysteresis: a candidate with an edge among its 8 neighbours becomes an edge; repeat to the fixed point
  for (;;) {
    if (threadIdx.x == 0) s_changed = 0;
    __syncthreads();
    int changed = 0;
    for (int p = threadIdx.x; p < NPIX; p += kTW) {
      if (map[p] != 0) continue;
      const int y = p / HW, x = p - y * HW;
      bool hit = false;
      for (int yy = max(y - 1, 0); yy <= min(y + 1, HW - 1) && !hit; ++yy)
        for (int xx = max(x - 1, 0); xx <= min(x + 1, HW - 1); ++xx)
          if (map[yy * HW + xx] == 2) { hit = true; break; }
      if (hit) { map[p] = 2; changed = 1; }
    }
    if (changed) s_changed = 1;
    __syncthreads();
    const int again = s_changed;
    __syncthreads();
    if (!again) break;
  }
  uint8_t* src8 = src8_all + (size_t)blockIdx.x * NPIX;
  for (int p = threadIdx.x; p < NPIX; p += kTW) src8[p] = map[p] == 2 ? 0 : 255;      // 255 - Canny
}

// ---- distanceTransform(DIST_L2, 5), 16.16 fixed point, truncated at 20 -------------------------------------------------
// block-wide inclusive prefix minimum over the 224 active threads (thread j holds element j).  Round 5: 32-bit candidates (|cand| < 2^30 +
// 224 * 2^16) scanned inside a wave with DPP row shifts and the two row broadcasts of gfx9 (VALU latency; the 64-bit __shfl_up version
// was six dependent pairs of ds_bpermute per step), one barrier for the carry across the four waves (the partials are double-buffered by
// step parity, the barrier that ends a step separates a slot's readers from its next writers).
__device__ __forceinline__ int wave_prefix_min(int v) {
  constexpr int BIG = 2147483647;
  v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x118, 0xf, 0xf, false));   // row_shr:8
  v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
  v = min(v, __builtin_amdgcn_update_dpp(BIG, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ int block_prefix_min(int v, int (*sh)[kT / 64], int parity) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_prefix_min(v);
  if (lane == 63) sh[parity][wave] = v;
  __syncthreads();
  int carry = 2147483647;
  for (int k = 0; k < wave; ++k) carry = min(carry, sh[parity][k]);
  return min(v, carry);
}

// Round 5: the source plane is staged in LDS once, and the pass-0 distances pass 1 folds in are fetched eight rows ahead (each was a
// dependent global / L2 read inside one of the 448 sequential row steps: 393 us per 256 images, ~2 100 cycles per step).
__global__ __launch_bounds__(kT) void k_spatter_dist(const uint8_t* __restrict__ src8_all, int* __restrict__ tplane_all) {
  __shared__ int sh[2][kT / 64];
  __shared__ int ring[3][TW];       // the two previous rows (+ the row being written), padded by 2 on each side
  __shared__ __attribute__((aligned(16))) uint8_t ssrc[NPIX];
  const uint8_t* src = src8_all + (size_t)blockIdx.x * NPIX;
  int* T = tplane_all + (size_t)blockIdx.x * NPIX;        // unpadded result plane
  for (int i = threadIdx.x; i < NPIX / 16; i += kT) reinterpret_cast<uint4*>(ssrc)[i] = reinterpret_cast<const uint4*>(src)[i];
  const int j = threadIdx.x;
  const bool act = j < HW;
  constexpr int PF = 8;
  for (int pass = 0; pass < 2; ++pass) {
    for (int k = threadIdx.x; k < 3 * TW; k += kT) (&ring[0][0])[k] = DINIT;
    __syncthreads();                                       // (also: ssrc staged; pass 0's stores to T visible to pass 1's loads)
    const int jj = pass == 0 ? j : HW - 1 - j;             // scan direction
    int tq[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) tq[k] = (pass == 1 && act) ? T[(HW - 1 - k) * HW + jj] : 0;
    for (int s0 = 0; s0 < HW; s0 += PF) {
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        const int step = s0 + k;
        const int i = pass == 0 ? step : HW - 1 - step;
        const int* r1 = ring[(step + 2) % 3];                // previous row in scan order
        const int* r2 = ring[(step + 1) % 3];                // the one before
        int cand = 2147483647;
        if (act) {
          const int J = jj + 2;
          // pass 0: rows i-1 / i-2, horizontal neighbour j-1; pass 1: rows i+1 / i+2, neighbour j+1 (mirror image)
          int c0 = min(min(r2[J - 1] + DC, r2[J + 1] + DC), min(r1[J - 2] + DC, r1[J + 2] + DC));
          c0 = min(c0, min(min(r1[J - 1] + DB, r1[J + 1] + DB), r1[J] + DA));
          if (pass == 0) c0 = ssrc[i * HW + jj] == 0 ? 0 : c0;
          else c0 = min(c0, tq[k]);
          cand = c0 - DA * j;                                // j = index along the scan direction
        }
        if (pass == 1 && act && step + PF < HW) tq[k] = T[(HW - 1 - (step + PF)) * HW + jj];
        const int pm = block_prefix_min(cand, sh, step & 1);
        int* rw = ring[step % 3];
        if (act) {
          const int v = pm + DA * j;
          rw[jj + 2] = v;
          T[i * HW + jj] = pass == 0 ? v : min(v, 20 * 65536);   // min(t0, DIST_MAX) then THRESH_TRUNC at 20
        }
        __syncthreads();
      }
    }
  }
}

// ---- blur -> equalizeHist -> filter2D -> blur -> m = liquid * dist, max, blend -------------------------------------------
__global__ __launch_bounds__(kTW) void k_spatter_finish(const uint8_t* __restrict__ in_all, uint8_t* __restrict__ out_all,
                                                       const uint8_t* __restrict__ l8_all, const int* __restrict__ tplane_all,
                                                       float c4) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* A = lds;                 // ping
  uint8_t* B = lds + NPIX;          // pong
  __shared__ uint32_t hist[256];
  __shared__ uint8_t lut[256];
  __shared__ float s_red[kTW / 64];
  const int* T = tplane_all + (size_t)blockIdx.x * NPIX;
  const uint8_t* l8 = l8_all + (size_t)blockIdx.x * NPIX;
  if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
  __syncthreads();
  // blur(dist, (3,3)).astype(np.uint8): exact 9-term sum of 16.16 values, float32(sum * (1/9)) truncated
  for (int p = threadIdx.x; p < NPIX; p += kTW) {
    const int y = p / HW, x = p - y * HW;
    long long s9 = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) s9 += T[reflect101(y + dy) * HW + reflect101(x + dx)];
    const float v = (float)(((double)s9 / 65536.0) * (1.0 / 9.0));
    const uint8_t q = (uint8_t)(int)v;
    A[p] = q;
    atomicAdd(&hist[q], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {           // equalizeHist LUT (256 sequential steps)
    int i0 = 0;
    while (i0 < 255 && hist[i0] == 0u) ++i0;
    if (hist[i0] == (uint32_t)NPIX) {
      for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)i0;
    } else {
      const float scale = 255.0f / (float)(NPIX - (int)hist[i0]);
      for (int i = 0; i <= i0; ++i) lut[i] = 0;
      uint32_t acc = 0;
      for (int i = i0 + 1; i < 256; ++i) {
        acc += hist[i];
        const int r = (int)rintf((float)acc * scale);
        lut[i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
      }
    }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < NPIX; p += kTW) A[p] = lut[A[p]];
  __syncthreads();
  // filter2D(dist, CV_8U, [[-2,-1,0],[-1,1,1],[0,1,2]]): correlation, BORDER_REFLECT_101, saturated
  for (int p = threadIdx.x; p < NPIX; p += kTW) {
    const int y = p / HW, x = p - y * HW;
    const int ym = reflect101(y - 1), yp = reflect101(y + 1), xm = reflect101(x - 1), xp = reflect101(x + 1);
    const int acc = -2 * A[ym * HW + xm] - A[ym * HW + x] - A[y * HW + xm] + A[y * HW + x] + A[y * HW + xp] + A[yp * HW + x] +
                    2 * A[yp * HW + xp];
    B[p] = (uint8_t)clampi(acc, 0, 255);
  }
  __syncthreads();
  // blur(dist, (3,3)) on uint8: round(sum / 9); then m = liquid_u8 * dist (exact in fp32) and its maximum
  float mx = 0.f;
  for (int p = threadIdx.x; p < NPIX; p += kTW) {
    const int y = p / HW, x = p - y * HW;
    int s9 = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) s9 += B[reflect101(y + dy) * HW + reflect101(x + dx)];
    const uint8_t d = (uint8_t)((2 * s9 + 9) / 18);
    A[p] = d;
    mx = fmaxf(mx, (float)l8[p] * (float)d);
  }
  mx = rart_wave_max(mx);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = s_red[0];
  for (int k = 1; k < kTW / 64; ++k) mx = fmaxf(mx, s_red[k]);
  const float col[3] = {(float)(175 / 255.), (float)(238 / 255.), (float)(238 / 255.)};
  const uint8_t* in = in_all + (size_t)blockIdx.x * NPIX * 3;
  uint8_t* out = out_all + (size_t)blockIdx.x * NPIX * 3;
  for (int p = threadIdx.x; p < NPIX; p += kTW) {
    float m = (float)l8[p] * (float)A[p];
    m = m / mx;
    m = m * c4;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = (float)in[p * 3 + c] / 255.0f;
      const float t = m * col[c];
      float v = x + t;
      v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
      out[p * 3 + c] = (uint8_t)(uint32_t)(v * 255.0f);
    }
  }
}
}  // namespace
#pragma clang fp contract(fast)

// liquid: fp64 [n][224][224], gaussian-blurred, NOT yet thresholded.  scratch_u8: >= 2 * n * 224 * 224 bytes, scratch_i32:
// >= n * 224 * 224 ints.  in / out: uint8 [n][224][224][3] (may alias).
int rart_launch_spatter_water(const uint8_t* in, uint8_t* out, const double* liquid, int n, double thresh, double c4,
                              uint8_t* scratch_u8, int* scratch_i32, hipStream_t st) {
  if (!rart_raise_dynamic_lds((const void*)k_spatter_canny, 2 * NPIX, "spatter") ||
      !rart_raise_dynamic_lds((const void*)k_spatter_finish, 2 * NPIX, "spatter"))
    return RART_ERR_HIP;
  uint8_t* l8 = scratch_u8;
  uint8_t* src8 = scratch_u8 + (size_t)n * NPIX;
  const size_t total = (size_t)n * NPIX;
  hipLaunchKernelGGL(k_spatter_liquid_u8, dim3(rart_grid_for(total, kT, 256 * 16)), dim3(kT), 0, st, liquid, l8, thresh, total);
  hipLaunchKernelGGL(k_spatter_canny, dim3(n), dim3(kTW), 2 * NPIX, st, (const uint8_t*)l8, src8, 50, 150);
  hipLaunchKernelGGL(k_spatter_dist, dim3(n), dim3(kT), 0, st, (const uint8_t*)src8, scratch_i32);
  hipLaunchKernelGGL(k_spatter_finish, dim3(n), dim3(kTW), 2 * NPIX, st, in, out, (const uint8_t*)l8, (const int*)scratch_i32,
                     (float)c4);
  return RART_OK;
}
