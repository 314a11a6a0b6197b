// Stencil corruptions for gfx950: gaussian_blur, glass_blur, defocus_blur, motion_blur.
// Reference: RobustART/noise/utils/imagenet_c/corruptions.py:26-51,162-216.
// Third-party semantics restated: skimage.filters.gaussian ==
// scipy.ndimage.gaussian_filter (separable, edge-replicate, truncate 4, fp64, symmetric-pair
// summation order of scipy's correlate1d); cv2.filter2D (correlation, BORDER_REFLECT_101, fp64);
// cv2.GaussianBlur on the aliased disk (fp32); ImageMagick MotionBlurImage (one-sided gaussian
// taps along the angle, edge virtual pixels, 8-bit requantisation).
#include "rart_common.h"
#include <map>
#include <utility>
#include <type_traits>
#include <math.h>
#include <string.h>
#include <vector>

#pragma clang fp contract(off)

namespace {
constexpr int kBlock = 256;
constexpr int MAXR = 24;  // gaussian_blur sigma 6 -> radius int(4*6+0.5)

struct GaussW {
  int radius;
  double w[2 * MAXR + 1];
};

// scipy _gaussian_kernel1d
GaussW make_gauss(double sigma, double truncate) {
  GaussW g;
  g.radius = (int)(truncate * sigma + 0.5);
  double sum = 0.0;
  for (int i = -g.radius; i <= g.radius; ++i) {
    const double v = exp(-0.5 / (sigma * sigma) * (double)(i * i));
    g.w[i + g.radius] = v;
    sum += v;
  }
  for (int i = 0; i <= 2 * g.radius; ++i) g.w[i] /= sum;
  return g;
}

// Pass 1 (axis 0 = image rows / H): u8 -> fp64 intermediate.  Pass 2 (axis 1 = W): fp64 -> u8.
// scipy correlate1d symmetric path: tmp = in[l]*w[r]; for jj=-r..-1: tmp += (in[l+jj] + in[l-jj]) * w[jj+r].
// MODE_IN 0: source is uint8 (value/255.0); 1: source is fp64.
// FINISH 0: store fp64; 1: np.uint8(v*255) (glass_blur first blur, no clip); 2: np.uint8(clip(v,0,1)*255).
template <int AXIS, int MODE_IN, int FINISH>
__global__ __launch_bounds__(kBlock) void k_gauss_pass(const void* __restrict__ src, void* __restrict__ dst, int n,
                                                       int h, int w, GaussW g) {
  const size_t total = (size_t)n * h * w * 3;
  const int len = AXIS == 0 ? h : w;
  const size_t stride = AXIS == 0 ? (size_t)w * 3 : 3;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const size_t pix = i / 3;
    const int xo = (int)(pix % w);
    const int yo = (int)((pix / w) % h);
    const int l = AXIS == 0 ? yo : xo;
    const size_t base = i - (size_t)l * stride;  // element at position 0 along the axis
    auto at = [&](int p) -> double {
      p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);  // mode='nearest'
      const size_t idx = base + (size_t)p * stride;
      if (MODE_IN == 0) return (double)((const uint8_t*)src)[idx] / 255.0;
      return ((const double*)src)[idx];
    };
    double tmp = at(l) * g.w[g.radius];
    for (int jj = -g.radius; jj < 0; ++jj) {
      const double pair = at(l + jj) + at(l - jj);
      tmp += pair * g.w[jj + g.radius];
    }
    if (FINISH == 0) {
      ((double*)dst)[i] = tmp;
    } else if (FINISH == 1) {
      ((uint8_t*)dst)[i] = (uint8_t)(uint32_t)(tmp * 255.0);
    } else {
      const double c = tmp < 0.0 ? 0.0 : (tmp > 1.0 ? 1.0 : tmp);
      ((uint8_t*)dst)[i] = (uint8_t)(uint32_t)(c * 255.0);
    }
  }
}

// Both passes in ONE kernel for radius <= 16 (gaussian_blur severities 1-4, both blurs of glass_blur): a workgroup owns a
// 16 x 32 pixel tile, stages the (16 + 2r) x (32 + 2r) x 3 clamped neighbourhood as the fp64 values u8 / 255.0 in LDS,
// writes the axis-0 pass (fp64, scipy's symmetric-pair order) to a second LDS array and runs the axis-1 pass from it.
// Identical arithmetic to the two-pass kernels above, bit for bit, without the 8-byte-per-element fp64 intermediate in HBM
// (pass 1 wrote 308 MB and pass 2 read it back: 515 + 357 us per 256-image batch, profiles/r02_corruption_kernels_before.csv).
constexpr int GF_TH = 16, GF_TW = 32, GF_RMAX = 16;
template <int FINISH>
__global__ __launch_bounds__(kBlock) void k_gauss_fused(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int h,
                                                        int w, GaussW g) {
  extern __shared__ __attribute__((aligned(16))) uint8_t gf_lds[];
  const int r = g.radius;
  const int RW = (GF_TW + 2 * r) * 3, RH = GF_TH + 2 * r;
  double* lut = reinterpret_cast<double*>(gf_lds);          // [256]
  double* sIn = lut + 256;                                    // [RH][RW]
  double* sP1 = sIn + RH * RW;                                // [GF_TH][RW]
  double* sWt = sP1 + GF_TH * RW;                             // [2r + 1]
  const int tid = threadIdx.x;
  lut[tid] = (double)tid / 255.0;
  if (tid <= 2 * r) sWt[tid] = g.w[tid];
  __syncthreads();
  const int x0 = blockIdx.x * GF_TW, y0 = blockIdx.y * GF_TH;
  const uint8_t* img = src + (size_t)blockIdx.z * h * w * 3;
  for (int i = tid; i < RH * RW; i += kBlock) {
    const int ry = i / RW, rxe = i - ry * RW;
    const int px = rxe / 3, c = rxe - px * 3;
    int yy = y0 - r + ry, xx = x0 - r + px;
    yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);              // mode='nearest'
    xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
    sIn[i] = lut[img[((size_t)yy * w + xx) * 3 + c]];
  }
  __syncthreads();
  for (int i = tid; i < GF_TH * RW; i += kBlock) {
    const int ty = i / RW, rxe = i - ty * RW;
    const double* col = sIn + (ty + r) * RW + rxe;
    double tmp = col[0] * sWt[r];
    for (int jj = -r; jj < 0; ++jj) {
      const double pair = col[jj * RW] + col[-jj * RW];
      tmp += pair * sWt[jj + r];
    }
    sP1[i] = tmp;
  }
  __syncthreads();
  uint8_t* out = dst + (size_t)blockIdx.z * h * w * 3;
  for (int i = tid; i < GF_TH * GF_TW * 3; i += kBlock) {
    const int ty = i / (GF_TW * 3), xe = i - ty * (GF_TW * 3);
    const int yy = y0 + ty, xx = x0 + xe / 3;
    if (yy >= h || xx >= w) continue;
    const double* row = sP1 + ty * RW + xe + 3 * r;
    double tmp = row[0] * sWt[r];
    for (int jj = -r; jj < 0; ++jj) {
      const double pair = row[3 * jj] + row[-3 * jj];
      tmp += pair * sWt[jj + r];
    }
    uint8_t o;
    if (FINISH == 1) {
      o = (uint8_t)(uint32_t)(tmp * 255.0);
    } else {
      const double c = tmp < 0.0 ? 0.0 : (tmp > 1.0 ? 1.0 : tmp);
      o = (uint8_t)(uint32_t)(c * 255.0);
    }
    out[((size_t)yy * w) * 3 + (size_t)x0 * 3 + xe] = o;
  }
}
size_t gauss_fused_lds(int r) {
  return (256 + (size_t)(GF_TH + 2 * r) * (GF_TW + 2 * r) * 3 + (size_t)GF_TH * (GF_TW + 2 * r) * 3 + 2 * r + 1) * sizeof(double);
}

// ---- glass_blur local shuffle -------------------------------------------------------------
// corruptions.py:176-182: for h in 224-d..d+1 (desc), w likewise: `x[h, w], x[h', w'] = x[h', w'], x[h, w]` with
// (h', w') = (h+dy, w+dx).  On a numpy array both right-hand sides are VIEWS, so the statement does not swap: after
// x[h, w] has received the neighbour's pixel, the second assignment copies that same pixel back onto the neighbour.  What the
// reference executes -- and what the golden vectors of the unmodified function show (tests/golden/make_golden_shim.py; rounds
// 1-2 implemented the swap the code appears to say) -- is the COPY CHAIN x[h, w] <- x[h+dy, w+dx] in scan order.
// Sequential per image: an operation reads a pixel within d of the one it writes, so two operations conflict (one reads what
// the other writes) only if both coordinates differ by <= d.  Rows are therefore pipelined with a skew of S = d+1 columns: at
// time t row index a (h = 224-d-a) handles column index b = t - a*S.  Operations issued in the same time step are >= S columns
// apart (no conflict) and every earlier-in-scan conflicting operation has a strictly smaller time ((a'-a)*S + (b'-b) >= S - d).
// One workgroup per image, image resident in LDS.
// Round 5 tried to take work out of the chain (scratch/r5/glass_offsets_prepass.patch, all variants bit-identical, us per 256 images at
// severity 3 for the whole corruption; this kernel: 1 197): the (dx, dy) draws by a parallel pre-pass, the chain only copying, two waves
// and a barrier per step 1 322; one wave per image, no barrier, four row slots per lane 1 637; the same with the four slots' reads
// issued before their writes 1 972.  A time step is bound by its dependent chain (offset -> LDS read -> LDS write) and by instruction
// issue per row slot, not by the Threefry call or the barrier: more lanes per step is what helps (256 threads, one row each: 1 154), and that is where
// the 222 rows run out.
constexpr int kGlassThreads = 256;

__global__ __launch_bounds__(kGlassThreads) void k_glass_shuffle(uint8_t* __restrict__ img_all, int delta, int iters,
                                                                 const int8_t* __restrict__ inj, uint32_t k0,
                                                                 uint32_t k1, uint32_t sample_base) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];  // 224*224*3
  constexpr int HW = 224;
  uint8_t* g = img_all + (size_t)blockIdx.x * HW * HW * 3;
  const uint4* g4 = reinterpret_cast<const uint4*>(g);
  uint4* l4 = reinterpret_cast<uint4*>(lds);
  for (int i = threadIdx.x; i < HW * HW * 3 / 16; i += kGlassThreads) l4[i] = g4[i];
  __syncthreads();
  const int N = HW - 2 * delta;  // loop trip count per axis
  const int S = delta + 1;
  const int8_t* dr = inj ? inj + (size_t)blockIdx.x * iters * N * N * 2 : nullptr;
  const uint32_t sample = sample_base + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    const int T = (N - 1) * S + N;
    for (int t = 0; t < T; ++t) {
      for (int a = threadIdx.x; a < N; a += kGlassThreads) {
        const int b = t - a * S;
        if (b >= 0 && b < N) {
          int dx, dy;
          const size_t di = ((size_t)it * N + a) * N + b;
          if (dr) {
            dx = dr[di * 2];
            dy = dr[di * 2 + 1];
          } else {
            const uint2 wv = threefry2x32(k0, k1, rart_ctr0((uint32_t)di, 4), sample);
            dx = (int)__umulhi(wv.x, (uint32_t)(2 * delta)) - delta;  // randint(-d, d): upper bound exclusive
            dy = (int)__umulhi(wv.y, (uint32_t)(2 * delta)) - delta;
          }
          const int hh = HW - delta - a, ww = HW - delta - b;
          uint8_t* p = lds + ((size_t)hh * HW + ww) * 3;
          const uint8_t* q = lds + ((size_t)(hh + dy) * HW + (ww + dx)) * 3;
          const uint8_t q0 = q[0], q1 = q[1], q2 = q[2];
          p[0] = q0; p[1] = q1; p[2] = q2;
        }
      }
      __syncthreads();
    }
  }
  uint4* o4 = reinterpret_cast<uint4*>(g);
  for (int i = threadIdx.x; i < HW * HW * 3 / 16; i += kGlassThreads) o4[i] = l4[i];
}

// Round 5, second half: the ITERATIONS of the chain overlap, and the draws leave the chain.  Operation (it, a, b) conflicts with
// (it', a', b') only when both index differences are <= d, whatever the iterations, so with time(it, a, b) = it * Toff + a * S + b,
// Toff = d * S + d + 1, every conflicting pair that the reference executes in the order X, Y has time(Y) - time(X) >= 1 (same iteration: as
// above; later iteration: Toff - d * S - d): a valid schedule with 256 lanes PER ITERATION in flight and T + (iters - 1) * Toff barriers
// instead of iters * T (severity 3: 895 instead of 2 631).  A step of the kernel above is one wave per SIMD executing ~85 dependent
// instructions (Threefry, address, LDS read, LDS write, loop control: ~700 cycles); here a parallel pre-pass (k_glass_offsets) writes the
// (dx, dy) of every (step, lane) as one byte, [step / 16][lane][16], so that a lane fetches sixteen steps with one 16-byte load a block ahead
// and a step is: extract, address, three byte reads, three byte writes, barrier (scratch/r5/glass_ko.hip: 167 us per 256 images at severity 3,
// of which 45 us are the LDS accesses and 80 us the barriers; the old kernel: 815 us).  Same operations in an order the dependencies allow:
// bit-identical (test_glass_shuffle_overlapped_iterations_equal_the_serial_kernel); RART_GLASS_SERIAL=1 keeps the kernel above.
struct GlassSched {
  int delta, iters, N, S, Toff, T, nthr;      // T = (N - 1) S + N + (iters - 1) Toff steps; nthr = 256 iters lanes
};
__global__ __launch_bounds__(256) void k_glass_offsets(uint4* __restrict__ tab_all, GlassSched g, const int8_t* __restrict__ inj,
                                                       uint32_t k0, uint32_t k1, uint32_t sample_base) {
  const int T16 = (g.T + 15) & ~15;
  // thread = (k-th 16-step block that overlaps the lane's window of N steps, lane), lane fastest: one 16-byte store per thread, and every
  // thread of a wave has work (dealing ALL blocks of a lane left a wave 53 % active: its 64 windows are 3 S steps apart)
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int nblk = (g.N + 15) / 16 + 1;                    // a window of N steps touches at most this many aligned blocks
  if (e >= nblk * g.nthr) return;
  const int k = e / g.nthr, lane = e - k * g.nthr;
  const int it = lane >> 8, a = lane & 255;
  const int t_first = it * g.Toff + a * g.S;               // the step of the lane's column 0
  const int blk = (t_first >> 4) + k;
  const int bb = blk * 16 - t_first;                       // column of the block's first step
  if (a >= g.N || bb >= g.N || blk >= (T16 >> 4)) return;  // never read
  const int8_t* dr = inj ? inj + (size_t)blockIdx.y * g.iters * g.N * g.N * 2 : nullptr;
  uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int b = bb + j;
    if (b >= 0 && b < g.N) {
      const size_t di = ((size_t)it * g.N + a) * g.N + b;
      int dx, dy;
      if (dr) {
        dx = dr[di * 2];
        dy = dr[di * 2 + 1];
      } else {
        const uint2 wv = threefry2x32(k0, k1, rart_ctr0((uint32_t)di, 4), sample_base + blockIdx.y);
        dx = (int)__umulhi(wv.x, (uint32_t)(2 * g.delta)) - g.delta;  // randint(-d, d): upper bound exclusive
        dy = (int)__umulhi(wv.y, (uint32_t)(2 * g.delta)) - g.delta;
      }
      w[j >> 2] |= (uint32_t)((dx + g.delta) | ((dy + g.delta) << 4)) << (8 * (j & 3));
    }
  }
  tab_all[(size_t)blockIdx.y * (T16 >> 4) * g.nthr + (size_t)blk * g.nthr + lane] = make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ __launch_bounds__(768) void k_glass_shuffle_overlap(uint8_t* __restrict__ img_all, const uint8_t* __restrict__ tab_all,
                                                               GlassSched g) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];  // 224*224*3
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
  const int b0 = -it * g.Toff - a * g.S;                   // this lane's column at step 0
  // the lane's pixel at column b: lds + ((HW - d - a) * HW + (HW - d - b)) * 3; the neighbour adds (dy * HW + dx) * 3, dy = (o >> 4) - d, dx = (o & 15) - d
  const int pbase = ((HW - g.delta - a) * HW + (HW - g.delta)) * 3;
  const int nbase = -(g.delta * HW + g.delta) * 3;
  auto wanted = [&](int t0) { return row_ok && b0 + t0 + 15 >= 0 && b0 + t0 < g.N; };
  uint4 nxt = make_uint4(0, 0, 0, 0);
  if (wanted(0)) nxt = tab[0];
  __syncthreads();
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
        const uint8_t q0 = q[0], q1 = q[1], q2 = q[2];
        p[0] = q0; p[1] = q1; p[2] = q2;
      }
      __syncthreads();
    }
  }
  uint4* o4 = reinterpret_cast<uint4*>(gi);
  for (int i = tid; i < HW * HW * 3 / 16; i += nthr) o4[i] = l4[i];
}

// ---- defocus_blur ---------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  i %= period;
  if (i < 0) i += period;
  return i >= n ? period - i : i;
}

// cv2.filter2D(plane fp64, -1, kernel): row-major tap order, fp64 multiply then add (not contracted).
// A workgroup owns a 16 x 64 pixel tile: the reflected (16 + 2r) x (64 + 2r) x 3 neighbourhood is staged ONCE in LDS as the
// fp64 values u8 / 255.0 (the first version re-fetched every tap from global memory through a byte LUT: 3.9 ms per
// 256-image batch), the disk kernel sits in LDS too (broadcast reads), and a thread accumulates 4 horizontally adjacent
// pixels x 3 channels in registers, tap by tap in the reference order -- bit-identical results.  The kernel is bound by the
// fp64 vector rate: 441 taps x 3 channels x (mul + add) per pixel = 34 G instructions per batch, 0.87 ms at the MI355X
// fp64 peak; the 30 %-of-HBM target of the byte-sized corruptions does not apply to a 21 x 21 dense filter.
constexpr int F2_TH = 16, F2_TW = 64, F2_PX = 4;
template <int KSZ>
__global__ __launch_bounds__(kBlock) void k_filter2d(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int n,
                                                     int h, int w, const double* __restrict__ kern) {
  constexpr int ksz = KSZ;
  extern __shared__ __attribute__((aligned(16))) uint8_t f2_lds[];
  const int r = ksz / 2;
  const int RW = (F2_TW + 2 * r) * 3, RH = F2_TH + 2 * r;
  double* lut = reinterpret_cast<double*>(f2_lds);     // [256]
  double* sK = lut + 256;                                // [ksz * ksz]
  double* sIn = sK + ksz * ksz;                          // [RH][RW]
  const int tid = threadIdx.x;
  lut[tid] = (double)tid / 255.0;
  for (int i = tid; i < ksz * ksz; i += kBlock) sK[i] = kern[i];
  __syncthreads();
  const int x0 = blockIdx.x * F2_TW, y0 = blockIdx.y * F2_TH;
  const uint8_t* img = in + (size_t)blockIdx.z * h * w * 3;
  for (int i = tid; i < RH * RW; i += kBlock) {
    const int ry = i / RW, rxe = i - ry * RW;
    const int px = rxe / 3, c = rxe - px * 3;
    const int yy = reflect101(y0 - r + ry, h), xx = reflect101(x0 - r + px, w);
    sIn[i] = lut[img[((size_t)yy * w + xx) * 3 + c]];
  }
  __syncthreads();
  // thread -> (row ty, 4 pixels starting at tx4): 16 rows x 16 groups
  const int ty = tid >> 4, tx4 = (tid & 15) * F2_PX;
  double acc[F2_PX * 3];
#pragma unroll
  for (int k = 0; k < F2_PX * 3; ++k) acc[k] = 0.0;
  for (int a = 0; a < ksz; ++a) {
    // the (KSZ + 3) x 3 values a row of taps touches are loaded once into registers; the b loop is fully unrolled so
    // the sliding window costs no LDS re-reads (the runtime-trip-count version re-read 12 doubles per tap: LDS bound)
    const double* row = sIn + (ty + a) * RW + tx4 * 3;
    const double* kr = sK + a * ksz;
    double win[(KSZ + F2_PX - 1) * 3];
#pragma unroll
    for (int k = 0; k < (KSZ + F2_PX - 1) * 3; ++k) win[k] = row[k];
#pragma unroll
    for (int b = 0; b < KSZ; ++b) {
      const double kv = kr[b];
#pragma unroll
      for (int k = 0; k < F2_PX * 3; ++k) {
        const double t = win[b * 3 + k] * kv;
        acc[k] += t;
      }
    }
  }
  const int yy = y0 + ty;
  if (yy < h) {
    uint8_t* o = out + ((size_t)blockIdx.z * h + yy) * w * 3;
#pragma unroll
    for (int k = 0; k < F2_PX * 3; ++k) {
      const int xx = x0 + tx4 + k / 3;
      if (xx < w) {
        const double cl = acc[k] < 0.0 ? 0.0 : (acc[k] > 1.0 ? 1.0 : acc[k]);
        o[(size_t)xx * 3 + k % 3] = (uint8_t)(uint32_t)(cl * 255.0);
      }
    }
  }
}
size_t filter2d_lds(int ksz) {
  const int r = ksz / 2;
  return (256 + (size_t)ksz * ksz + (size_t)(F2_TH + 2 * r) * (F2_TW + 2 * r) * 3) * sizeof(double);
}

// ---- defocus_blur on the matrix cores (round 5) --------------------------------------------------------------------
// k_filter2d above is bound by the fp64 vector rate (289 taps x 3 channels x (mul + add) per pixel: 2.2 ms per 256-image batch).  The filter's
// REAL value needs no fp64: the inputs are 8-bit integers and the 17 x 17 weights are fp32 values, so with W = round(w 2^F) (32-bit
// fixed point, exact for severities 2-4 whose weights span 31 bits) T = sum W p is an exact integer and the reference's result is
// floor(T / 2^F) -- unless T / 2^F lies within `band` (the fixed-point error bound + 1e-9 for the reference's own fp64 rounding) of
// an integer, where the ORDER of the reference's fp64 additions decides (flat regions: every tap of a saturated neighbourhood lands on
// p (1 +- 1e-16)).  So:
//   * fast path, every pixel: T on v_mfma_i32_16x16x64_i8.  W is split into four signed base-256 digits, the pixels go in as q = p - 128
//     (a constant 128 sum W is added back).  One MFMA = 16 output columns (M) x 16 output rows (N) x two kernel rows of a 32-pixel window
//     (K = 2 x 32): the A operand is the banded (Toeplitz) weight matrix of a row pair and a digit -- 9 pairs x 4 digits = 36 fragments,
//     resident in 144 VGPRs for the whole workgroup -- the B operand one ds_read_b128 per lane from the channel-planar tile in LDS,
//     reused by the four digit MFMAs.  Only the CONSISTENCY of the (lane >> 4, byte) -> k mapping between A and B is relied upon.
//   * flagged 16 x 16 tiles (any output within `band` of an integer, T != 0): recomputed by the whole workgroup in fp64, tap by tap in
//     the reference's order from the same LDS tile -- bit-identical to k_filter2d.  Random images flag nothing; flat regions pay the old rate.
// Output identical to k_filter2d's on every input (tests/test_corruptions_gpu.py::test_defocus_fast_path_equals_the_ordered_fp64_kernel).
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef int i32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
// KSZ x KSZ kernel, MOUT outputs per MFMA row block (MOUT + KSZ - 1 = 32 = the window two kernel rows share one K = 64 step over):
// 17 x 17 -> 16 outputs (14 column tiles, 9 row pairs, 16-byte aligned window reads); 21 x 21 (severity 5) -> 12 outputs (19 column tiles,
// 11 row pairs, the window starts every 12 bytes: 4-byte aligned reads, rows 4 of 16 of the result block unused).
template <int KSZ, int MOUT>
struct FiCfg {
  static constexpr int R = KSZ / 2, TH = 32, RH = TH + KSZ, STEPS = (KSZ + 1) / 2, XT = (224 + MOUT - 1) / MOUT;
  static constexpr int RS = MOUT == 16 ? 240 : 252;                  // bytes per staged row: x = -R .. ; 4 x odd for the 4-byte reads
  static constexpr int PLANES = 3 * RH * RS, PLANES_PAD = (PLANES + 15) / 16 * 16;
  static constexpr int OUT = TH * 224 * 3, TILES = 3 * 2 * XT;
  static constexpr size_t LDS = PLANES_PAD + OUT + 256 * sizeof(double) + (TILES + 4) * sizeof(int);
  static_assert(MOUT + KSZ - 1 == 32, "two kernel rows of a 32-pixel window per K = 64 step");
  static_assert((XT - 1) * MOUT + 31 + 1 <= RS, "the last column tile's window stays inside the staged row");
};
constexpr int FI_TH = 32;
struct FilterI8Meta {
  int F;
  long long corr, band;
};

template <int KSZ, int MOUT>
__global__ __launch_bounds__(kBlock, 2) void k_filter2d_i8(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           const uint4* __restrict__ frags, const double* __restrict__ kern,
                                                           FilterI8Meta meta) {
  using C = FiCfg<KSZ, MOUT>;
  extern __shared__ __attribute__((aligned(16))) uint8_t fi_lds[];
  uint8_t* const sP = fi_lds;
  uint8_t* const sO = fi_lds + C::PLANES_PAD;
  double* const lut = reinterpret_cast<double*>(fi_lds + C::PLANES_PAD + C::OUT);
  int* const sList = reinterpret_cast<int*>(lut + 256);           // [0] = count, [1 ..] = flagged tile-channels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int y0 = blockIdx.x * C::TH;
  const size_t img = blockIdx.y;
  // the weight fragments of this lane (same for every tile): requested first, consumed after the staging
  i32x4 A[C::STEPS][4];
#pragma unroll
  for (int j = 0; j < C::STEPS; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) A[j][k] = __builtin_bit_cast(i32x4, frags[(j * 4 + k) * 64 + lane]);
  lut[tid] = (double)tid / 255.0;
  if (tid == 0) sList[0] = 0;
  // ---- stage rows y0 - R .. y0 + TH + R (reflect-101) as three byte planes, x = -R at byte 0 of a row
  const uint8_t* base = in + img * (size_t)(224 * 224 * 3);
  for (int i = tid; i < C::RH * 42; i += kBlock) {
    const int ry = i / 42, cx = i - ry * 42;
    const int yy = reflect101(y0 - C::R + ry, 224);
    const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)yy * 672 + cx * 16);
    const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
    int e = cx * 16, px = e / 3, c = e - px * 3;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      sP[(c * C::RH + ry) * C::RS + px + C::R] = (uint8_t)(((wv[b >> 2] >> (8 * (b & 3))) & 0xFFu) ^ 0x80u);
      if (++c == 3) { c = 0; ++px; }
    }
  }
  __syncthreads();
  constexpr int HALO = C::RS - 224;                                  // R bytes left of x = 0, the rest right of x = 223
  for (int i = tid; i < 3 * C::RH * HALO; i += kBlock) {             // x = -k <- p[k]; x = 223 + k <- p[223 - k]
    const int row = i / HALO, k = i - row * HALO;
    uint8_t* r = sP + row * C::RS;
    if (k < C::R) r[k] = r[C::R + (C::R - k)];
    else r[C::R + 224 + (k - C::R)] = r[C::R + 222 - (k - C::R)];
  }
  __syncthreads();
  // ---- fast path: tile-channel tc = (channel, row block, column tile); lane = (n = output row of the block, g = K group)
  const int n = lane & 15, g = lane >> 4;
  const long long one = 1ll << meta.F;
  for (int tc = wave; tc < C::TILES; tc += 4) {
    const int c = tc % 3, rest = tc / 3, blk = rest / C::XT, xt = rest - blk * C::XT;
    const uint8_t* pb = sP + (c * C::RH + blk * 16 + n + (g >> 1)) * C::RS + xt * MOUT + 16 * (g & 1);
    i32x4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = (i32x4){0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < C::STEPS; ++j) {
      i32x4 B;
      if (MOUT == 16) B = *reinterpret_cast<const i32x4*>(pb + 2 * j * C::RS);
      else B = *reinterpret_cast<const i32x4_a4*>(pb + 2 * j * C::RS);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[j][k], B, acc[k], 0, 0, 0);
    }
    bool flag = false;
    const int xo = xt * MOUT + 4 * g;
    uint8_t* o = sO + ((blk * 16 + n) * 224 + xo) * 3 + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (4 * g + r < MOUT && xo + r < 224) {                          // (MOUT = 12: the fourth row group of the block is unused)
        const long long T = (long long)acc[0][r] + (long long)acc[1][r] * 256ll + (long long)acc[2][r] * 65536ll +
                            (long long)acc[3][r] * 16777216ll + meta.corr;
        const long long fl = T >> meta.F, fr = T - (fl << meta.F);
        flag |= T != 0 && (fr < meta.band || fr > one - meta.band);
        o[3 * r] = (uint8_t)(T <= 0 ? 0 : (fl > 255 ? 255 : fl));
      }
    }
    if (__ballot(flag) != 0ull && lane == 0) sList[1 + atomicAdd(&sList[0], 1)] = tc;
  }
  __syncthreads();
  // ---- flagged tiles: the reference's fp64 sum, row-major tap order, multiply then add (k_filter2d's arithmetic), one output per thread
  const int nflag = sList[0];
  for (int i = 0; i < nflag; ++i) {
    const int tc = sList[1 + i];
    const int c = tc % 3, rest = tc / 3, blk = rest / C::XT, xt = rest - blk * C::XT;
    const int ty = tid >> 4, tx = tid & 15;
    if (tx < MOUT && xt * MOUT + tx < 224) {
      const uint8_t* pb = sP + (c * C::RH + blk * 16 + ty) * C::RS + xt * MOUT + tx;
      double acc = 0.0;
      for (int a = 0; a < KSZ; ++a) {
        const uint8_t* row = pb + a * C::RS;
        const double* kr = kern + a * KSZ;
#pragma unroll
        for (int b = 0; b < KSZ; ++b) {
          const double t = lut[row[b] ^ 0x80u] * kr[b];
          acc += t;
        }
      }
      const double cl = acc < 0.0 ? 0.0 : (acc > 1.0 ? 1.0 : acc);
      sO[((blk * 16 + ty) * 224 + xt * MOUT + tx) * 3 + c] = (uint8_t)(uint32_t)(cl * 255.0);
    }
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(out + img * (size_t)(224 * 224 * 3) + (size_t)y0 * 672);
  const uint4* so4 = reinterpret_cast<const uint4*>(sO);
  for (int i = tid; i < C::OUT / 16; i += kBlock) dst[i] = so4[i];
}

// Host side of the fast path: the fixed-point weights, their four signed base-256 digits as MFMA A-operand fragments
// (fragment (pair j, digit k), lane (m = lane & 15, g = lane >> 4), byte i: kernel row 2 j + (g >> 1), window column 16 (g & 1) + i ->
// tap b = column - m), the constant 128 sum W and the ambiguity band.
struct FilterI8Host {
  std::vector<uint8_t> frags;        // [steps][4][64][16]
  FilterI8Meta meta;
  int steps = 0, mout = 0;
  bool ok = false;
};
FilterI8Host make_filter_i8(const std::vector<double>& kern, int ksz) {
  FilterI8Host h;
  if (ksz != 17 && ksz != 21) return h;
  const int mout = 33 - ksz, steps = (ksz + 1) / 2;
  double wmax = 0.0;
  for (double v : kern) {
    if (v < 0.0) return h;                       // the digit split below assumes non-negative weights (every disk is)
    wmax = v > wmax ? v : wmax;
  }
  if (!(wmax > 0.0)) return h;
  int ex;
  frexp(wmax, &ex);                              // wmax = f 2^ex, f in [0.5, 1)
  const int F = 30 - ex;                         // W <= 2^30
  if (F < 8 || F > 46) return h;
  std::vector<long long> W(kern.size());
  long long sumW = 0;
  double dq = 0.0;
  for (size_t i = 0; i < kern.size(); ++i) {
    W[i] = llrint(ldexp(kern[i], F));
    sumW += W[i];
    dq += fabs(ldexp((double)W[i], -F) - kern[i]);
  }
  h.frags.assign((size_t)steps * 4 * 64 * 16, 0);
  for (int j = 0; j < steps; ++j)
    for (int lane = 0; lane < 64; ++lane) {
      const int m = lane & 15, g = lane >> 4, a = 2 * j + (g >> 1);
      for (int i = 0; i < 16; ++i) {
        const int b = 16 * (g & 1) + i - m;
        long long w = (m < mout && a < ksz && b >= 0 && b < ksz) ? W[(size_t)a * ksz + b] : 0;
        for (int k = 0; k < 4; ++k) {
          const long long lo = ((w + 128) & 255) - 128;     // signed digit in [-128, 127]
          h.frags[(((size_t)(j * 4 + k) * 64 + lane) * 16) + i] = (uint8_t)(int8_t)lo;
          w = (w - lo) / 256;
        }
        if (w != 0) return h;                                // does not fit four digits
      }
    }
  h.meta.F = F;
  h.meta.corr = 128 * sumW;
  h.meta.band = (long long)ceil(ldexp(255.0 * dq + 1e-9, F)) + 1;
  h.steps = steps;
  h.mout = mout;
  h.ok = true;
  return h;
}

// ---- gaussian_blur / glass_blur's two blurs on the matrix cores (round 5) ------------------------------------------------
// The same idea as k_filter2d_i8, for the SEPARABLE filter: W_a = round(w_a 2^31) in four signed base-256 digits, q = p - 128.
//   pass 1 (along H): U'[y][x] = sum_a W_a q[y + a - r][x] -- an exact integer, |U'| < 2^38.1 -- as one 16 x 16 x 64 i8 MFMA per digit:
//          M = 16 output rows (A = the banded weight matrix over a 64-row window), N = 16 columns, B from the TRANSPOSED tile [c][x][row];
//   U' is split into five signed digits (carry-propagated in 32-bit from the four accumulators) and written to five byte planes [row][x + r];
//   pass 2 (along W): T = sum_b W_b U'[y][x + b - r] + 128 (sum W)^2: digit pairs (dw, du) with dw + du >= 3 (14 MFMAs; the six dropped pairs
//          are below 2^-24.8 of an output step and part of the band), M = 16 output columns (the SAME A fragments), N = 16 rows.
//   out = floor(T / 2^62) unless T / 2^62 is within `band` of an integer: those 16 x 16 tiles are recomputed by the workgroup in fp64 in
//   scipy's order (pass 1 into a scratch tile, pass 2 from it) -- k_gauss_fused's arithmetic, bit for bit.
// 16 + 2 r <= 64 covers every radius of gaussian_blur (4 .. 24) and glass_blur (3 .. 6).  18 MFMAs per 256 outputs and channel.
constexpr int GI_TH = 16, GI_WR = 64, GI_RSU = 272, GI_THREADS = 448, GI_XT = 14;
constexpr int GI_PT = 3 * 224 * GI_WR;                 // 43 008 B: q transposed, [channel][x][window row]
constexpr int GI_U = 5 * GI_TH * GI_RSU;               // 21 760 B: the five digit planes of U' of ONE channel (then the fallback's fp64 scratch)
constexpr int GI_OUT = GI_TH * 224 * 3;                // 10 752 B
constexpr size_t GI_LDS = GI_PT + GI_U + GI_OUT + 256 * sizeof(double) + (3 * GI_XT + 4) * sizeof(int);
static_assert(GI_U >= GI_TH * GI_WR * (int)sizeof(double), "the fallback's pass-1 scratch lives in the digit planes");
struct GaussI8Meta {
  long long corr, band;      // 128 (sum W)^2 >> 24; ambiguity band, both in units of 2^-38 of an output step
};

template <int FINISH>
__global__ __launch_bounds__(GI_THREADS, 2) void k_gauss_i8(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                            const uint4* __restrict__ frags, GaussW gw, GaussI8Meta meta) {
  extern __shared__ __attribute__((aligned(16))) uint8_t gi_lds[];
  uint8_t* const sPT = gi_lds;
  int8_t* const sU = reinterpret_cast<int8_t*>(gi_lds + GI_PT);
  uint8_t* const sO = gi_lds + GI_PT + GI_U;
  double* const lut = reinterpret_cast<double*>(gi_lds + GI_PT + GI_U + GI_OUT);
  int* const sList = reinterpret_cast<int*>(lut + 256);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = gw.radius, nrow = GI_TH + 2 * r;
  const int y0 = blockIdx.x * GI_TH;
  const size_t img = blockIdx.y;
  i32x4 A[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) A[k] = __builtin_bit_cast(i32x4, frags[k * 64 + lane]);
  if (tid < 256) lut[tid] = (double)tid / 255.0;
  if (tid == 0) sList[0] = 0;
  // ---- stage window rows y0 - r .. y0 + 15 + r (mode 'nearest') transposed: sPT[(c * 224 + x) * 64 + window row]
  const uint8_t* base = in + img * (size_t)(224 * 224 * 3);
  for (int i = tid; i < nrow * 42; i += GI_THREADS) {
    const int k = i / 42, cx = i - k * 42;
    int yy = y0 - r + k;
    yy = yy < 0 ? 0 : (yy > 223 ? 223 : yy);
    const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)yy * 672 + cx * 16);
    const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
    int e = cx * 16, px = e / 3, c = e - px * 3;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      sPT[(c * 224 + px) * GI_WR + k] = (uint8_t)(((wv[b >> 2] >> (8 * (b & 3))) & 0xFFu) ^ 0x80u);
      if (++c == 3) { c = 0; ++px; }
    }
  }
  __syncthreads();
  const int n = lane & 15, g = lane >> 4;
  for (int c = 0; c < 3; ++c) {
    // ---- pass 1: column tile xt of channel c -> five digit planes
    for (int xt = wave; xt < GI_XT; xt += GI_THREADS / 64) {
      const i32x4 B = *reinterpret_cast<const i32x4*>(sPT + (c * 224 + xt * 16 + n) * GI_WR + 16 * g);
      i32x4 acc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[k], B, (i32x4){0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {                 // lane: column xt*16 + n, rows 4g + q
        int8_t* u = sU + (4 * g + q) * GI_RSU + r + xt * 16 + n;
        int t = acc[0][q];
        int d = (int)(int8_t)t;
        u[0] = (int8_t)d;
        t = acc[1][q] + ((t - d) >> 8);
        d = (int)(int8_t)t;
        u[GI_TH * GI_RSU] = (int8_t)d;
        t = acc[2][q] + ((t - d) >> 8);
        d = (int)(int8_t)t;
        u[2 * GI_TH * GI_RSU] = (int8_t)d;
        t = acc[3][q] + ((t - d) >> 8);
        d = (int)(int8_t)t;
        u[3 * GI_TH * GI_RSU] = (int8_t)d;
        u[4 * GI_TH * GI_RSU] = (int8_t)((t - d) >> 8);
      }
    }
    __syncthreads();
    for (int i = tid; i < 5 * GI_TH * (GI_RSU - 224); i += GI_THREADS) {       // 'nearest' along W: both halos <- the edge columns
      const int row = i / (GI_RSU - 224), k = i - row * (GI_RSU - 224);
      int8_t* u = sU + row * GI_RSU;
      if (k < r) u[k] = u[r];
      else u[224 + k] = u[r + 223];
    }
    __syncthreads();
    // ---- pass 2: lane = (n = row, g = K group); result lane: row n, columns xt*16 + 4g + q
    for (int xt = wave; xt < GI_XT; xt += GI_THREADS / 64) {
      i32x4 Bu[5];
#pragma unroll
      for (int du = 0; du < 5; ++du) Bu[du] = *reinterpret_cast<const i32x4*>(sU + (du * GI_TH + n) * GI_RSU + xt * 16 + 16 * g);
      i32x4 acc[5];
#pragma unroll
      for (int s = 0; s < 5; ++s) acc[s] = (i32x4){0, 0, 0, 0};
#pragma unroll
      for (int dw = 0; dw < 4; ++dw)
#pragma unroll
        for (int du = 0; du < 5; ++du)
          if (dw + du >= 3) acc[dw + du - 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[dw], Bu[du], acc[dw + du - 3], 0, 0, 0);
      bool flag = false;
      uint8_t* o = sO + (n * 224 + xt * 16 + 4 * g) * 3 + c;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long long T = (long long)acc[0][q] + (long long)acc[1][q] * 256ll + (long long)acc[2][q] * 65536ll +
                            (long long)acc[3][q] * 16777216ll + (long long)acc[4][q] * 4294967296ll + meta.corr;
        const long long fl = T >> 38, fr = T - (fl << 38);
        flag |= fr < meta.band || fr > (1ll << 38) - meta.band;
        o[3 * q] = (uint8_t)(fl < 0 ? 0 : (fl > 255 ? 255 : fl));
      }
      if (__ballot(flag) != 0ull && lane == 0) sList[1 + atomicAdd(&sList[0], 1)] = c * GI_XT + xt;
    }
    __syncthreads();
  }
  // ---- flagged tiles: scipy's two passes in fp64 (k_gauss_fused's arithmetic: centre tap, then symmetric pairs added before weighting)
  const int nflag = sList[0];
  double* const P1 = reinterpret_cast<double*>(sU);              // [16][64]
  const int ncol = 16 + 2 * r;
  for (int i = 0; i < nflag; ++i) {
    const int tc = sList[1 + i], c = tc / GI_XT, xt = tc - c * GI_XT;
    for (int j = tid; j < GI_TH * ncol; j += GI_THREADS) {
      const int ty = j / ncol, jx = j - ty * ncol;
      int xx = xt * 16 - r + jx;
      xx = xx < 0 ? 0 : (xx > 223 ? 223 : xx);
      const uint8_t* col = sPT + (c * 224 + xx) * GI_WR + ty + r;
      double tmp = lut[col[0] ^ 0x80u] * gw.w[r];
      for (int jj = -r; jj < 0; ++jj) {
        const double pair = lut[col[jj] ^ 0x80u] + lut[col[-jj] ^ 0x80u];
        tmp += pair * gw.w[jj + r];
      }
      P1[ty * GI_WR + jx] = tmp;
    }
    __syncthreads();
    if (tid < 256) {
      const int ty = tid >> 4, tx = tid & 15;
      const double* row = P1 + ty * GI_WR + tx + r;
      double tmp = row[0] * gw.w[r];
      for (int jj = -r; jj < 0; ++jj) {
        const double pair = row[jj] + row[-jj];
        tmp += pair * gw.w[jj + r];
      }
      uint8_t o;
      if (FINISH == 1) {
        o = (uint8_t)(uint32_t)(tmp * 255.0);
      } else {
        const double cl = tmp < 0.0 ? 0.0 : (tmp > 1.0 ? 1.0 : tmp);
        o = (uint8_t)(uint32_t)(cl * 255.0);
      }
      sO[(ty * 224 + xt * 16 + tx) * 3 + c] = o;
    }
    __syncthreads();
  }
  uint4* dst = reinterpret_cast<uint4*>(out + img * (size_t)(224 * 224 * 3) + (size_t)y0 * 672);
  const uint4* so4 = reinterpret_cast<const uint4*>(sO);
  for (int i = tid; i < GI_OUT / 16; i += GI_THREADS) dst[i] = so4[i];
}

struct GaussI8Host {
  double sigma = -1.0;
  std::vector<uint8_t> frags;        // [4 digits][64 lanes][16]
  GaussI8Meta meta;
  bool ok = false;
};
GaussI8Host make_gauss_i8(const GaussW& gw, double sigma) {
  GaussI8Host h;
  h.sigma = sigma;
  const int r = gw.radius, nt = 2 * r + 1;
  if (16 + 2 * r > GI_WR) return h;
  long long W[2 * MAXR + 1];
  __int128 sumW = 0;
  double errw = 0.0;
  for (int i = 0; i < nt; ++i) {
    W[i] = llrint(ldexp(gw.w[i], 31));
    sumW += W[i];
    errw += fabs(ldexp((double)W[i], -31) - gw.w[i]);
  }
  h.frags.assign(4 * 64 * 16, 0);
  for (int lane = 0; lane < 64; ++lane) {
    const int m = lane & 15, g = lane >> 4;
    for (int i = 0; i < 16; ++i) {
      const int t = 16 * g + i - m;
      long long w = (t >= 0 && t < nt) ? W[t] : 0;
      for (int k = 0; k < 4; ++k) {
        const long long lo = ((w + 128) & 255) - 128;
        h.frags[((size_t)k * 64 + lane) * 16 + i] = (uint8_t)(int8_t)lo;
        w = (w - lo) / 256;
      }
      if (w != 0) return h;            // a weight >= 0.996 (sigma < 0.3): not a blur this library issues
    }
  }
  const __int128 corr = (__int128)128 * sumW * sumW;
  h.meta.corr = (long long)(corr >> 24);
  const double bq = 255.0 * (2.0 * errw + errw * errw) + 1e-9;
  const long long dropped = ((long long)nt * 16384ll * (1ll + 2ll * 256ll + 3ll * 65536ll)) >> 24;
  h.meta.band = (long long)ceil(ldexp(bq, 38)) + dropped + 8;
  h.ok = true;
  return h;
}
// Process-lifetime table per (sigma, radius), NEVER evicted (std::map nodes do not move): the reference handed out stays valid while another
// thread inserts, and the fragments behind an in-flight hipMemcpyAsync are never overwritten (the 16-slot array reused slot 15).
const GaussI8Host& gauss_i8_for(const GaussW& gw, double sigma) {
  static std::map<std::pair<double, int>, GaussI8Host> cache;
  std::lock_guard<std::mutex> lk(rart_host_table_mutex());
  const auto key = std::make_pair(sigma, gw.radius);
  auto it = cache.find(key);
  if (it == cache.end()) it = cache.emplace(key, make_gauss_i8(gw, sigma)).first;
  return it->second;
}

// ---- motion_blur (ImageMagick) -----------------------------------------------------------------
struct MotionTab {
  int offx[41], offy[41];
  int ext[4];                 // min / max of offx, min / max of offy over the taps (both ranges contain 0: tap 0 is the pixel itself)
};

// one block per image: offsets from the angle (injected or drawn from host-mirrored stream 8)
__global__ void k_motion_offsets(MotionTab* __restrict__ tab, const double* __restrict__ angles, int width,
                                 double lo, double hi, uint32_t k0, uint32_t k1, uint32_t sample_base) {
  const int img = blockIdx.x;
  double ang;
  if (angles) {
    ang = angles[img];
  } else {
    const uint2 wv = threefry2x32(k0, k1, rart_ctr0(0, 8), sample_base + img);
    const double u = ((double)(wv.x >> 5) * 67108864.0 + (double)(wv.y >> 6)) / 9007199254740992.0;
    ang = lo + (hi - lo) * u;
  }
  const int i = threadIdx.x;
  if (i < width) {
    const double a = ang * (M_PI / 180.0);
    const double px = (double)width * sin(a), py = (double)width * cos(a);
    const double hyp = hypot(px, py);
    tab[img].offx[i] = (int)ceil((double)i * py / hyp - 0.5);
    tab[img].offy[i] = (int)ceil((double)i * px / hyp - 0.5);
  }
  if (blockDim.x == 64) {     // (both launches use one wave per image) the extents of the offsets, for the LDS-tiled kernel's halo
    int ox = 0, oy = 0;
    if (i < width) {
      const double a = ang * (M_PI / 180.0);
      const double px = (double)width * sin(a), py = (double)width * cos(a);
      const double hyp = hypot(px, py);
      ox = (int)ceil((double)i * py / hyp - 0.5);
      oy = (int)ceil((double)i * px / hyp - 0.5);
    }
    int mnx = ox, mxx = ox, mny = oy, mxy = oy;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      mnx = min(mnx, __shfl_xor(mnx, d, 64));
      mxx = max(mxx, __shfl_xor(mxx, d, 64));
      mny = min(mny, __shfl_xor(mny, d, 64));
      mxy = max(mxy, __shfl_xor(mxy, d, 64));
    }
    if (i == 0) {
      tab[img].ext[0] = mnx; tab[img].ext[1] = mxx; tab[img].ext[2] = mny; tab[img].ext[3] = mxy;
    }
  }
}

struct MotionK {
  int width;
  double k[41];
};

// CH = 3 (RGB image) or 1 (snow layer); out = floor(sum_i k[i] * in(x+offx[i], y+offy[i]) + 0.5), edge clamp
template <int CH>
__global__ __launch_bounds__(kBlock) void k_motion_blur(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                        int n, int h, int w, const MotionTab* __restrict__ tab,
                                                        MotionK mk) {
  __shared__ MotionTab st;
  const size_t per_img = (size_t)h * w;
  // grid: (chunks, n)
  const int img = blockIdx.y;
  if (threadIdx.x < 41) {
    st.offx[threadIdx.x] = tab[img].offx[threadIdx.x];
    st.offy[threadIdx.x] = tab[img].offy[threadIdx.x];
  }
  __syncthreads();
  const uint8_t* src = in + (size_t)img * per_img * CH;
  uint8_t* dst = out + (size_t)img * per_img * CH;
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < per_img; p += (size_t)gridDim.x * kBlock) {
    const int xo = (int)(p % w), yo = (int)(p / w);
    double acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0.0;
    for (int t = 0; t < mk.width; ++t) {
      int yy = yo + st.offy[t], xx = xo + st.offx[t];
      yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
      xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
      const uint8_t* s = src + ((size_t)yy * w + xx) * CH;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const double term = mk.k[t] * (double)s[c];
        acc[c] += term;
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      double v = floor(acc[c] + 0.5);
      v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
      dst[p * CH + c] = (uint8_t)(uint32_t)v;
    }
  }
}

// Round 5: the same sum from an LDS tile.  k_motion_blur reads every tap of every channel as a byte from global memory (31 taps x 3 at
// severity 3: 1.2 G byte loads per 256 images, 441 us); here a workgroup stages its 32 x 32 output tile + the halo the image's offsets reach
// (edge clamp applied while staging, RGB packed into one dword per pixel) and a tap is one ds_read_b32 per pixel.  Same terms, same order,
// same fp64 operations: bit-identical (test_motion_blur_tile_kernel_equals_the_direct_kernel).  RART_MOTION_DIRECT=1 keeps the old kernel.
constexpr int MB_T = 32, MB_HALO = 40, MB_P = MB_T + MB_HALO;
template <int CH>
__global__ __launch_bounds__(kBlock) void k_motion_blur_tile(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int h, int w,
                                                             const MotionTab* __restrict__ tab, MotionK mk) {
  typedef typename std::conditional<CH == 3, uint32_t, uint8_t>::type PX;
  __shared__ PX tile[MB_P * MB_P];
  __shared__ int s_off[41];
  const int img = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * MB_T, y0 = blockIdx.y * MB_T;
  const MotionTab* mt = tab + img;
  const int mnx = mt->ext[0], mxx = mt->ext[1], mny = mt->ext[2], mxy = mt->ext[3];
  const int pitch = MB_T + mxx - mnx, rows = MB_T + mxy - mny;
  if (tid < mk.width) s_off[tid] = mt->offy[tid] * pitch + mt->offx[tid];
  const uint8_t* src = in + (size_t)img * h * w * CH;
  for (int r = wave; r < rows; r += kBlock / 64) {
    int gy = y0 + mny + r;
    gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
    for (int c = lane; c < pitch; c += 64) {
      int gx = x0 + mnx + c;
      gx = gx < 0 ? 0 : (gx > w - 1 ? w - 1 : gx);
      const uint8_t* sp = src + ((size_t)gy * w + gx) * CH;
      if (CH == 3) tile[r * pitch + c] = (PX)((uint32_t)sp[0] | ((uint32_t)sp[1] << 8) | ((uint32_t)sp[2] << 16));
      else tile[r * pitch + c] = (PX)sp[0];
    }
  }
  __syncthreads();
  constexpr int NP = MB_T * MB_T / kBlock;          // 4 pixels per thread: rows ly, ly + 8, ...
  const int lx = tid & (MB_T - 1), ly = tid >> 5;
  const int base = (ly - mny) * pitch + (lx - mnx);
  double acc[NP][CH];
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[j][c] = 0.0;
  for (int t = 0; t < mk.width; ++t) {
    const int o = base + s_off[t];
    const double kt = mk.k[t];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const uint32_t px = (uint32_t)tile[o + j * (kBlock / MB_T) * pitch];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const double term = kt * (double)((px >> (8 * c)) & 255u);
        acc[j][c] += term;
      }
    }
  }
  uint8_t* dst = out + (size_t)img * h * w * CH;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int yo = y0 + ly + j * (kBlock / MB_T), xo = x0 + lx;
    if (yo < h && xo < w) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        double v = floor(acc[j][c] + 0.5);
        v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
        dst[((size_t)yo * w + xo) * CH + c] = (uint8_t)(uint32_t)v;
      }
    }
  }
}

template <int CH>
void launch_motion_blur(const uint8_t* in, uint8_t* out, int n, int h, int w, const MotionTab* tab, const MotionK& mk, hipStream_t s) {
  if (mk.width - 1 <= MB_HALO && n <= 65535 && getenv("RART_MOTION_DIRECT") == nullptr) {
    hipLaunchKernelGGL(k_motion_blur_tile<CH>, dim3((w + MB_T - 1) / MB_T, (h + MB_T - 1) / MB_T, n), dim3(kBlock), 0, s, in, out, h, w, tab, mk);
    return;
  }
  const uint32_t gx = (uint32_t)(((size_t)h * w + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_motion_blur<CH>, dim3(gx, n), dim3(kBlock), 0, s, in, out, n, h, w, tab, mk);
}

MotionK make_motion_kernel(double radius, double sigma) {
  MotionK mk;
  mk.width = (int)(2.0 * ceil(radius) + 1.0);
  double norm = 0.0;
  for (int i = 0; i < mk.width; ++i) {
    mk.k[i] = exp(-((double)i * (double)i) / (2.0 * sigma * sigma)) / (sqrt(2.0 * M_PI) * sigma);
    norm += mk.k[i];
  }
  for (int i = 0; i < mk.width; ++i) mk.k[i] /= norm;
  return mk;
}

// ---- host: the anti-aliased disk of defocus_blur (corruptions.py:26-38) ------------------------
// float32 arithmetic mirrors numpy: aliased /= sum (float32), then cv2.GaussianBlur ksize 3 or 5 as two
// float32 passes with BORDER_REFLECT_101.
int host_reflect101(int i, int n) {
  const int period = 2 * (n - 1);
  i %= period;
  if (i < 0) i += period;
  return i >= n ? period - i : i;
}

std::vector<double> make_disk(int radius, double alias, int* ksz_out) {
  int half, ks;
  if (radius <= 8) { half = 8; ks = 3; } else { half = radius; ks = 5; }
  const int n = 2 * half + 1;
  std::vector<float> d((size_t)n * n);
  float sum = 0.f;
  // np.sum(float32 array) uses pairwise summation; every term is 0 or 1 so any order is exact here
  for (int y = -half; y <= half; ++y)
    for (int x = -half; x <= half; ++x) {
      const float v = (x * x + y * y) <= radius * radius ? 1.f : 0.f;
      d[(size_t)(y + half) * n + (x + half)] = v;
      sum += v;
    }
  for (auto& v : d) v = v / sum;
  // cv2.getGaussianKernel(ks, alias, CV_32F)
  std::vector<float> k(ks);
  {
    std::vector<double> kd(ks);
    double s = 0.0;
    for (int i = 0; i < ks; ++i) {
      const double x = (double)i - (ks - 1) * 0.5;
      kd[i] = exp(-(x * x) / (2.0 * alias * alias));
      s += kd[i];
    }
    for (int i = 0; i < ks; ++i) k[i] = (float)(kd[i] / s);
  }
  const int r = ks / 2;
  std::vector<float> tmp((size_t)n * n, 0.f), out((size_t)n * n, 0.f);
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) {
      volatile float acc = 0.f;
      for (int t = 0; t < ks; ++t) {
        volatile float prod = d[(size_t)y * n + host_reflect101(x + t - r, n)] * k[t];
        acc = acc + prod;
      }
      tmp[(size_t)y * n + x] = acc;
    }
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) {
      volatile float acc = 0.f;
      for (int t = 0; t < ks; ++t) {
        volatile float prod = tmp[(size_t)host_reflect101(y + t - r, n) * n + x] * k[t];
        acc = acc + prod;
      }
      out[(size_t)y * n + x] = acc;
    }
  *ksz_out = n;
  return std::vector<double>(out.begin(), out.end());
}

const double kGaussBlurSigma[5] = {1, 2, 3, 4, 6};
const double kGlass[5][3] = {{0.7, 1, 2}, {0.9, 2, 1}, {1, 2, 3}, {1.1, 3, 2}, {1.5, 4, 2}};
const double kDefocus[5][2] = {{3, 0.1}, {4, 0.5}, {6, 0.5}, {8, 0.5}, {10, 0.5}};
const double kMotion[5][2] = {{10, 3}, {15, 5}, {15, 8}, {15, 12}, {20, 15}};

// ---- the small Gaussians (radius 3, 4, 6, 8: both blurs of glass_blur at every severity, gaussian_blur severities 1-2) -- round 5 ---------------------
// k_gauss_i8 costs 166-186 us per 256 images whatever the radius (the digit split and the 18 MFMAs do not shrink with it); k_gauss_fused re-reads
// every tap of every output from LDS (300 us at radius 4).  Here the ordered fp64 sums run from REGISTER windows: pass 1 (along H) is one
// thread per column element sliding down 12 + 2 R rows (one coalesced byte load per row, the 2 R + 1 window in registers, compile-time ring), pass 2
// (along W) one thread per run of 4 pixels and channel by channel a 4 + 2 R window read once from the LDS row -- 3 reads per output instead of
// 2 R + 1.  The same operations in the same order as k_gauss_pass / k_gauss_fused (scipy's symmetric correlate1d): bit-identical to them and hence to
// k_gauss_i8 (test_gaussian_fast_path_equals_the_ordered_fp64_kernels covers all three).  RART_GAUSS_SMALL_OFF=1 disables.
constexpr int GS_TH = 12, GS_THREADS = 704, GS_TILES = (224 + GS_TH - 1) / GS_TH;
template <int R, int FINISH>
__global__ __launch_bounds__(GS_THREADS) void k_gauss_small(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, GaussW g) {
  constexpr int HW = 224, RWD = (HW + 2 * R) * 3;            // an LDS row: R halo pixels on each side
  __shared__ double lut[256];
  __shared__ double p1[GS_TH * RWD];
  const int tid = threadIdx.x;
  if (tid < 256) lut[tid] = (double)tid / 255.0;
  __syncthreads();
  const int y0 = blockIdx.x * GS_TH;
  const uint8_t* img = src + (size_t)blockIdx.y * HW * HW * 3;
  if (tid < HW * 3) {
    // pass 1, axis 0: element tid of a row; win[k] = in(y + k - R), mode='nearest'
    const int px = tid / 3;
    double win[2 * R + 1];
#pragma unroll
    for (int k = 0; k < 2 * R; ++k) {
      int yy = y0 - R + k;
      yy = yy < 0 ? 0 : (yy > HW - 1 ? HW - 1 : yy);
      win[k + 1] = lut[img[(size_t)yy * (HW * 3) + tid]];
    }
#pragma unroll
    for (int ty = 0; ty < GS_TH; ++ty) {
#pragma unroll
      for (int k = 0; k < 2 * R; ++k) win[k] = win[k + 1];
      int yy = y0 + ty + R;
      yy = yy < 0 ? 0 : (yy > HW - 1 ? HW - 1 : yy);
      win[2 * R] = lut[img[(size_t)yy * (HW * 3) + tid]];
      double tmp = win[R] * g.w[R];
#pragma unroll
      for (int jj = -R; jj < 0; ++jj) {
        const double pair = win[R + jj] + win[R - jj];
        tmp += pair * g.w[jj + R];
      }
      double* row = p1 + ty * RWD;
      row[R * 3 + tid] = tmp;
      if (px == 0) {
#pragma unroll
        for (int k = 0; k < R; ++k) row[k * 3 + tid] = tmp;                       // columns -R .. -1 = column 0 (tid = its channel)
      }
      if (px == HW - 1) {
#pragma unroll
        for (int k = 1; k <= R; ++k) row[(R + k) * 3 + tid] = tmp;                // columns 224 .. 223 + R = column 223
      }
    }
  }
  __syncthreads();
  if (tid < GS_TH * (HW / 4)) {
    // pass 2, axis 1: row ty, pixels 4 run .. 4 run + 3, channel by channel
    const int ty = tid / (HW / 4), run = tid - ty * (HW / 4);
    const int yy = y0 + ty;
    if (yy < HW) {
      const double* row = p1 + ty * RWD + (4 * run) * 3;                           // element of pixel (4 run - R)
      uint32_t ob[3] = {0u, 0u, 0u};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double win[4 + 2 * R];
#pragma unroll
        for (int k = 0; k < 4 + 2 * R; ++k) win[k] = row[k * 3 + c];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          double tmp = win[o + R] * g.w[R];
#pragma unroll
          for (int jj = -R; jj < 0; ++jj) {
            const double pair = win[o + R + jj] + win[o + R - jj];
            tmp += pair * g.w[jj + R];
          }
          uint32_t b;
          if (FINISH == 1) {
            b = (uint32_t)(uint8_t)(uint32_t)(tmp * 255.0);
          } else {
            const double cl = tmp < 0.0 ? 0.0 : (tmp > 1.0 ? 1.0 : tmp);
            b = (uint32_t)(uint8_t)(uint32_t)(cl * 255.0);
          }
          const int byte = o * 3 + c;                                              // byte of the 12-byte run
          ob[byte >> 2] |= b << (8 * (byte & 3));
        }
      }
      uint32_t* o32 = reinterpret_cast<uint32_t*>(dst + (size_t)blockIdx.y * HW * HW * 3 + ((size_t)yy * HW + 4 * run) * 3);
      o32[0] = ob[0]; o32[1] = ob[1]; o32[2] = ob[2];
    }
  }
}

template <int FINISH>
bool launch_gauss_small(const uint8_t* in, uint8_t* out, int n, const GaussW& g, hipStream_t s) {
  const dim3 grid(GS_TILES, (unsigned)n), blk(GS_THREADS);
  switch (g.radius) {
    case 3: hipLaunchKernelGGL((k_gauss_small<3, FINISH>), grid, blk, 0, s, in, out, g); return true;
    case 4: hipLaunchKernelGGL((k_gauss_small<4, FINISH>), grid, blk, 0, s, in, out, g); return true;
    case 6: hipLaunchKernelGGL((k_gauss_small<6, FINISH>), grid, blk, 0, s, in, out, g); return true;
    case 8: hipLaunchKernelGGL((k_gauss_small<8, FINISH>), grid, blk, 0, s, in, out, g); return true;
  }
  return false;
}

template <int FINISH>
void gauss_u8_to_u8(const uint8_t* in, uint8_t* out, double* tmp, int n, int h, int w, const GaussW& g,
                    hipStream_t s, double sigma = 0.0, void* frag_ws = nullptr) {
  if (FINISH != 0 && h == 224 && w == 224 && n <= 65535 && ((uintptr_t)out & 3) == 0 && getenv("RART_GAUSS_FP64") == nullptr &&
      getenv("RART_GAUSS_SMALL_OFF") == nullptr && launch_gauss_small<FINISH>(in, out, n, g, s))
    return;
  if (frag_ws && h == 224 && w == 224 && n <= 65535 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
      getenv("RART_GAUSS_FP64") == nullptr) {
    // the matrix-core path (exact fixed point + ordered-fp64 recompute of ambiguous tiles); anything else: the fp64 kernels below
    const GaussI8Host& f = gauss_i8_for(g, sigma);
    if (f.ok && rart_raise_dynamic_lds((const void*)k_gauss_i8<FINISH>, GI_LDS, "gaussian filter (matrix-core path)") &&
        hipMemcpyAsync(frag_ws, f.frags.data(), f.frags.size(), hipMemcpyHostToDevice, s) == hipSuccess) {
      hipLaunchKernelGGL((k_gauss_i8<FINISH>), dim3(224 / GI_TH, n), dim3(GI_THREADS), GI_LDS, s, in, out, (const uint4*)frag_ws, g, f.meta);
      return;
    }
  }
  if (g.radius <= GF_RMAX && n <= 65535) {
    const size_t lds = gauss_fused_lds(g.radius);
    if (rart_raise_dynamic_lds((const void*)k_gauss_fused<FINISH>, lds, "gaussian filter")) {
      hipLaunchKernelGGL((k_gauss_fused<FINISH>), dim3((w + GF_TW - 1) / GF_TW, (h + GF_TH - 1) / GF_TH, n), dim3(kBlock), lds, s,
                         in, out, h, w, g);
      return;
    }
  }
  const int grid = rart_grid_for((size_t)n * h * w * 3, kBlock, 256 * 16);
  hipLaunchKernelGGL((k_gauss_pass<0, 0, 0>), dim3(grid), dim3(kBlock), 0, s, (const void*)in, (void*)tmp, n, h, w, g);
  hipLaunchKernelGGL((k_gauss_pass<1, 1, FINISH>), dim3(grid), dim3(kBlock), 0, s, (const void*)tmp, (void*)out, n,
                     h, w, g);
}
}  // namespace
#pragma clang fp contract(fast)

// Shared with corrupt_composite.hip (snow uses the same ImageMagick motion blur on its 1-channel layer)
int rart_motion_blur_gray(const uint8_t* in, uint8_t* out, int n, int h, int w, double radius, double sigma,
                          const double* angles_dev, double lo, double hi, uint64_t seed, uint64_t sample_offset,
                          void* tab_ws, hipStream_t s) {
  const MotionK mk = make_motion_kernel(radius, sigma);
  MotionTab* tab = (MotionTab*)tab_ws;
  hipLaunchKernelGGL(k_motion_offsets, dim3(n), dim3(64), 0, s, tab, angles_dev, mk.width, lo, hi, (uint32_t)seed,
                     (uint32_t)(seed >> 32), (uint32_t)sample_offset);
  launch_motion_blur<1>(in, out, n, h, w, tab, mk, s);
  return RART_OK;
}
size_t rart_motion_tab_bytes(int n) { return rart_align_up((size_t)n * sizeof(MotionTab), 256); }

extern "C" int rart_stencil_fixed_point_info(int corruption_id, int severity, rart_fixed_point_info* info, unsigned char* frags,
                                             size_t frags_bytes) {
  RART_CHECK_ARG(info != nullptr, "rart_stencil_fixed_point_info: null info");
  RART_CHECK_ARG(severity >= 1 && severity <= 5, "rart_stencil_fixed_point_info: severity %d outside 1..5", severity);
  memset(info, 0, sizeof(*info));
  const int s = severity - 1;
  const std::vector<uint8_t>* fr = nullptr;
  FilterI8Host f2;
  GaussI8Host f1;
  if (corruption_id == RART_DEFOCUS_BLUR) {
    int ksz = 0;
    const std::vector<double> disk = make_disk((int)kDefocus[s][0], kDefocus[s][1], &ksz);
    f2 = make_filter_i8(disk, ksz);
    if (!f2.ok) return RART_ERR_UNSUPPORTED;
    info->kind = 1; info->ksize = ksz; info->n_steps = f2.steps; info->frac_bits = f2.meta.F; info->out_frac_bits = f2.meta.F;
    info->corr = f2.meta.corr; info->band = f2.meta.band;
    for (double v : disk) {
      const double e = fabs(ldexp((double)llrint(ldexp(v, f2.meta.F)), -f2.meta.F) - v);
      info->max_abs_weight_error = e > info->max_abs_weight_error ? e : info->max_abs_weight_error;
      info->sum_abs_weight_error += e;
    }
    fr = &f2.frags;
  } else if (corruption_id == RART_GAUSSIAN_BLUR || corruption_id == RART_GLASS_BLUR) {
    const double sigma = corruption_id == RART_GAUSSIAN_BLUR ? kGaussBlurSigma[s] : kGlass[s][0];
    const GaussW g = make_gauss(sigma, 4.0);
    f1 = make_gauss_i8(g, sigma);
    if (!f1.ok) return RART_ERR_UNSUPPORTED;
    info->kind = 2; info->ksize = 2 * g.radius + 1; info->n_steps = 1; info->frac_bits = 31; info->out_frac_bits = 38;
    info->corr = f1.meta.corr; info->band = f1.meta.band;
    for (int i = 0; i <= 2 * g.radius; ++i) {
      const double e = fabs(ldexp((double)llrint(ldexp(g.w[i], 31)), -31) - g.w[i]);
      info->max_abs_weight_error = e > info->max_abs_weight_error ? e : info->max_abs_weight_error;
      info->sum_abs_weight_error += e;
    }
    fr = &f1.frags;
  } else {
    return RART_ERR_UNSUPPORTED;
  }
  if (frags) {
    RART_CHECK_ARG(frags_bytes >= fr->size(), "rart_stencil_fixed_point_info: fragment buffer of %zu bytes required", fr->size());
    memcpy(frags, fr->data(), fr->size());
  }
  return RART_OK;
}

size_t rart_ws_stencil(int id, int /*severity*/, int n, int h, int w) {
  const size_t tmp = rart_align_up((size_t)n * h * w * 3 * sizeof(double), 256);
  switch (id) {
    case RART_GAUSSIAN_BLUR: return tmp + 4096;                                              // + the weight fragments of k_gauss_i8
    case RART_GLASS_BLUR: return tmp + rart_align_up((size_t)n * h * w * 3, 256) + 4096;
    case RART_DEFOCUS_BLUR: return rart_align_up(21 * 21 * sizeof(double), 4096) + rart_align_up((size_t)11 * 4 * 64 * 16, 256);
    case RART_MOTION_BLUR: return rart_motion_tab_bytes(n);
  }
  return 0;
}

int rart_launch_stencil(int id, const RartCorruptArgs& a) {
  const int s = a.severity - 1;
  const void* inj0 = (a.injected && a.n_injected > 0) ? a.injected[0] : nullptr;
  switch (id) {
    case RART_GAUSSIAN_BLUR: {
      const GaussW g = make_gauss(kGaussBlurSigma[s], 4.0);
      void* fr = (uint8_t*)a.workspace + rart_align_up((size_t)a.n * a.h * a.w * 3 * sizeof(double), 256);
      gauss_u8_to_u8<2>(a.in, a.out, (double*)a.workspace, a.n, a.h, a.w, g, a.stream, kGaussBlurSigma[s], fr);
      break;
    }
    case RART_GLASS_BLUR: {
      RART_CHECK_ARG(a.h == 224 && a.w == 224, "glass_blur: reference hard-codes 224x224 (corruptions.py:177-178)");
      const GaussW g = make_gauss(kGlass[s][0], 4.0);
      double* tmp = (double*)a.workspace;
      uint8_t* mid = (uint8_t*)a.workspace + rart_align_up((size_t)a.n * a.h * a.w * 3 * sizeof(double), 256);
      void* fr = mid + rart_align_up((size_t)a.n * a.h * a.w * 3, 256);
      gauss_u8_to_u8<1>(a.in, mid, tmp, a.n, a.h, a.w, g, a.stream, kGlass[s][0], fr);
      if (!rart_raise_dynamic_lds((const void*)k_glass_shuffle, 224 * 224 * 3, "glass_blur") ||
          !rart_raise_dynamic_lds((const void*)k_glass_shuffle_overlap, 224 * 224 * 3, "glass_blur"))
        return RART_ERR_HIP;
      GlassSched gs;
      gs.delta = (int)kGlass[s][1]; gs.iters = (int)kGlass[s][2];
      gs.N = 224 - 2 * gs.delta; gs.S = gs.delta + 1; gs.Toff = gs.delta * gs.S + gs.delta + 1;
      gs.T = (gs.N - 1) * gs.S + gs.N + (gs.iters - 1) * gs.Toff; gs.nthr = 256 * gs.iters;
      const size_t tab_img = (size_t)((gs.T + 15) & ~15) * gs.nthr;           // bytes of the offset table per image: lives in `tmp` (free between the two blurs)
      if (gs.iters <= 3 && tab_img <= (size_t)a.h * a.w * 3 * sizeof(double) && a.n <= 65535 && getenv("RART_GLASS_SERIAL") == nullptr) {
        hipLaunchKernelGGL(k_glass_offsets, dim3((unsigned)((((gs.N + 15) / 16 + 1) * gs.nthr + 255) / 256), a.n), dim3(256), 0, a.stream, (uint4*)tmp, gs,
                           (const int8_t*)inj0, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)a.sample_offset);
        hipLaunchKernelGGL(k_glass_shuffle_overlap, dim3(a.n), dim3(gs.nthr), 224 * 224 * 3, a.stream, mid, (const uint8_t*)tmp, gs);
      } else {
        hipLaunchKernelGGL(k_glass_shuffle, dim3(a.n), dim3(kGlassThreads), 224 * 224 * 3, a.stream, mid,
                           (int)kGlass[s][1], (int)kGlass[s][2], (const int8_t*)inj0, (uint32_t)a.seed,
                           (uint32_t)(a.seed >> 32), (uint32_t)a.sample_offset);
      }
      gauss_u8_to_u8<2>(mid, a.out, tmp, a.n, a.h, a.w, g, a.stream, kGlass[s][0], fr);
      break;
    }
    case RART_DEFOCUS_BLUR: {
      // disk tables are tiny and fixed per severity: built once, kept for the process lifetime so the
      // async upload below never reads freed host memory
      static std::vector<double> disks[5];
      static int ksz[5];
      {
        std::lock_guard<std::mutex> lk(rart_host_table_mutex());
        if (disks[s].empty()) disks[s] = make_disk((int)kDefocus[s][0], kDefocus[s][1], &ksz[s]);
      }
      if (hipMemcpyAsync(a.workspace, disks[s].data(), disks[s].size() * sizeof(double), hipMemcpyHostToDevice,
                         a.stream) != hipSuccess) {
        rart_set_error("defocus_blur: kernel upload failed");
        return RART_ERR_HIP;
      }
      RART_CHECK_ARG(a.n <= 65535, "defocus_blur: at most 65535 images per call");
      {
        // the matrix-core path (17 x 17 disks = severities 1-4, 224 x 224 images, 16-byte aligned batches); everything else: k_filter2d
        static FilterI8Host fast[5];
        static bool fast_made[5] = {false, false, false, false, false};
        {
          std::lock_guard<std::mutex> lk(rart_host_table_mutex());
          if (!fast_made[s]) { fast[s] = make_filter_i8(disks[s], ksz[s]); fast_made[s] = true; }
        }
        const bool use_fast = fast[s].ok && a.h == 224 && a.w == 224 && ((uintptr_t)a.in & 15) == 0 && ((uintptr_t)a.out & 15) == 0 &&
                              getenv("RART_DEFOCUS_FP64") == nullptr;
        const void* kfn = ksz[s] == 17 ? (const void*)k_filter2d_i8<17, 16> : (const void*)k_filter2d_i8<21, 12>;
        const size_t flds = ksz[s] == 17 ? FiCfg<17, 16>::LDS : FiCfg<21, 12>::LDS;
        if (use_fast && rart_raise_dynamic_lds(kfn, flds, "defocus_blur (matrix-core path)")) {
          uint8_t* ftab = (uint8_t*)a.workspace + rart_align_up(21 * 21 * sizeof(double), 4096);
          if (hipMemcpyAsync(ftab, fast[s].frags.data(), fast[s].frags.size(), hipMemcpyHostToDevice, a.stream) != hipSuccess) {
            rart_set_error("defocus_blur: fragment table upload failed");
            return RART_ERR_HIP;
          }
          if (ksz[s] == 17)
            hipLaunchKernelGGL((k_filter2d_i8<17, 16>), dim3(224 / FI_TH, a.n), dim3(kBlock), flds, a.stream, a.in, a.out, (const uint4*)ftab,
                               (const double*)a.workspace, fast[s].meta);
          else
            hipLaunchKernelGGL((k_filter2d_i8<21, 12>), dim3(224 / FI_TH, a.n), dim3(kBlock), flds, a.stream, a.in, a.out, (const uint4*)ftab,
                               (const double*)a.workspace, fast[s].meta);
          break;
        }
      }
      const size_t lds = filter2d_lds(ksz[s]);
      if (!rart_raise_dynamic_lds((const void*)k_filter2d<17>, filter2d_lds(21), "defocus_blur") ||
          !rart_raise_dynamic_lds((const void*)k_filter2d<21>, filter2d_lds(21), "defocus_blur"))
        return RART_ERR_HIP;
      const dim3 f2grid((a.w + F2_TW - 1) / F2_TW, (a.h + F2_TH - 1) / F2_TH, a.n);
      if (ksz[s] == 17)
        hipLaunchKernelGGL(k_filter2d<17>, f2grid, dim3(kBlock), lds, a.stream, a.in, a.out, a.n, a.h, a.w, (const double*)a.workspace);
      else if (ksz[s] == 21)
        hipLaunchKernelGGL(k_filter2d<21>, f2grid, dim3(kBlock), lds, a.stream, a.in, a.out, a.n, a.h, a.w, (const double*)a.workspace);
      else {
        rart_set_error("defocus_blur: unexpected disk size %d", ksz[s]);
        return RART_ERR_INVALID;
      }
      break;
    }
    case RART_MOTION_BLUR: {
      const MotionK mk = make_motion_kernel(kMotion[s][0], kMotion[s][1]);
      MotionTab* tab = (MotionTab*)a.workspace;
      hipLaunchKernelGGL(k_motion_offsets, dim3(a.n), dim3(64), 0, a.stream, tab, (const double*)inj0, mk.width,
                         -45.0, 45.0, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)a.sample_offset);
      launch_motion_blur<3>(a.in, a.out, a.n, a.h, a.w, tab, mk, a.stream);
      break;
    }
    default:
      rart_set_error("rart_launch_stencil: bad id %d", id);
      return RART_ERR_INVALID;
  }
  RART_CHECK_LAUNCH("stencil corruption launch");
  return RART_OK;
}
