// Composite corruptions for gfx950: fog (diamond-square plasma), snow, elastic_transform, spatter (mud).
// Reference: RobustART/noise/utils/imagenet_c/corruptions.py:55-114,235-241,265-342,395-424.
// Third-party semantics restated: scipy.ndimage.zoom / gaussian_filter /
// map_coordinates, cv2.getAffineTransform / warpAffine (1/32-pixel fixed point) / cvtColor, and the
// ImageMagick motion blur shared with corrupt_stencil.hip.
#include "rart_common.h"
#include <math.h>
#include <vector>

int rart_motion_blur_gray(const uint8_t* in, uint8_t* out, int n, int h, int w, double radius, double sigma,
                          const double* angles_dev, double lo, double hi, uint64_t seed, uint64_t sample_offset,
                          void* tab_ws, hipStream_t s);
size_t rart_motion_tab_bytes(int n);
int rart_launch_spatter_water(const uint8_t* in, uint8_t* out, const double* liquid, int n, double thresh, double c4,
                              uint8_t* scratch_u8, int* scratch_i32, hipStream_t st);

#pragma clang fp contract(off)

namespace {
constexpr int kBlock = 256;
constexpr int HW = 224;

__device__ __forceinline__ double u53(uint2 w) {
  return ((double)(w.x >> 5) * 67108864.0 + (double)(w.y >> 6)) / 9007199254740992.0;
}

// =====================================================================================
// fog: plasma_fractal (corruptions.py:55-101), one workgroup per image, map in HBM/L2 (fp64 256x256)
// =====================================================================================
constexpr int MAPN = 256;
constexpr int PLASMA_DRAWS = 65535;
constexpr int PLASMA_T = 1024;            // 16 waves per image (round 5; 4 until then: 274 us per 256 images, the last two levels are 80 % of the draws)
// (Round 5 also ran the last two levels as grid-wide launches, one thread per element, + a min / max reduction kernel: 34 + 47 + 84 + 20 = 185 us
//  against 196 for this kernel -- the levels are bound by the strided fp64 map traffic through L2 (134 MB of maps per 256 images, every line of it
//  read and half of it written by the last level), not by the one workgroup's latency.  Not kept: eight launches for 6 %.)

__global__ __launch_bounds__(PLASMA_T) void k_plasma(double* __restrict__ maps, double* __restrict__ minmax,
                                                   const double* __restrict__ inj, double wibbledecay, uint32_t k0,
                                                   uint32_t k1, uint32_t sample_base) {
  __shared__ double red[PLASMA_T];
  double* M = maps + (size_t)blockIdx.x * MAPN * MAPN;
  const double* dr = inj ? inj + (size_t)blockIdx.x * PLASMA_DRAWS : nullptr;
  const uint32_t sample = sample_base + blockIdx.x;
  if (threadIdx.x == 0) M[0] = 0.0;
  __syncthreads();
  double wibble = 100.0;
  uint32_t base = 0;
  auto draw = [&](uint32_t idx) -> double {
    if (dr) return dr[idx];
    const uint2 wv = threefry2x32(k0, k1, rart_ctr0(idx, 5), sample);
    return -wibble + (wibble - (-wibble)) * u53(wv);  // np.random.uniform(-wibble, wibble)
  };
  for (int step = MAPN; step >= 2; step >>= 1) {
    const int m = MAPN / step, half = step / 2;
    // fillsquares
    for (int e = threadIdx.x; e < m * m; e += PLASMA_T) {
      const int i = e / m, j = e % m, i1 = (i + 1) % m, j1 = (j + 1) % m;
      const double a = M[(i * step) * MAPN + j * step] + M[(i1 * step) * MAPN + j * step];
      const double b = M[(i * step) * MAPN + j1 * step] + M[(i1 * step) * MAPN + j1 * step];
      const double sq = a + b;
      const double nz = wibble * draw(base + e);
      M[(half + i * step) * MAPN + half + j * step] = sq / 4.0 + nz;
    }
    __syncthreads();
    // filldiamonds (two independent sets)
    for (int e = threadIdx.x; e < m * m; e += PLASMA_T) {
      const int i = e / m, j = e % m, i1 = (i + 1) % m, j1 = (j + 1) % m, im = (i + m - 1) % m, jm = (j + m - 1) % m;
      const double dr_ij = M[(half + i * step) * MAPN + half + j * step];
      const double ul_ij = M[(i * step) * MAPN + j * step];
      const double ldr = dr_ij + M[(half + im * step) * MAPN + half + j * step];
      const double lul = ul_ij + M[(i * step) * MAPN + j1 * step];
      const double lt = ldr + lul;
      const double tdr = dr_ij + M[(half + i * step) * MAPN + half + jm * step];
      const double tul = ul_ij + M[(i1 * step) * MAPN + j * step];
      const double tt = tdr + tul;
      const double n2 = wibble * draw(base + m * m + e);
      const double n3 = wibble * draw(base + 2 * m * m + e);
      M[(i * step) * MAPN + half + j * step] = lt / 4.0 + n2;
      M[(half + i * step) * MAPN + j * step] = tt / 4.0 + n3;
    }
    __syncthreads();
    base += 3 * m * m;
    wibble /= wibbledecay;
  }
  // min, then max of (M - min)
  double mn = INFINITY;
  for (int e = threadIdx.x; e < MAPN * MAPN; e += PLASMA_T) mn = fmin(mn, M[e]);
  red[threadIdx.x] = mn;
  __syncthreads();
  for (int s = PLASMA_T / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmin(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  mn = red[0];
  __syncthreads();
  double mx = -INFINITY;
  for (int e = threadIdx.x; e < MAPN * MAPN; e += PLASMA_T) mx = fmax(mx, M[e] - mn);
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = PLASMA_T / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    minmax[blockIdx.x * 2] = mn;
    minmax[blockIdx.x * 2 + 1] = red[0];
  }
}

__global__ __launch_bounds__(kBlock) void k_image_max(const uint8_t* __restrict__ in, uint32_t* __restrict__ mx,
                                                      uint32_t elems) {
  const uint8_t* src = in + (size_t)blockIdx.y * elems;
  uint32_t m = 0;
  if ((elems & 15u) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    // 16 bytes per load; the byte-wise maximum of dwords by v_pk-free bit tricks is not needed: max of the four bytes of each dword
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < elems / 16; e += gridDim.x * kBlock) {
      const uint4 v = s4[e];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) m = max(max(m, w[k] & 255u), max(max((w[k] >> 8) & 255u, (w[k] >> 16) & 255u), w[k] >> 24));
    }
  } else {
    for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < elems; e += gridDim.x * kBlock) m = max(m, (uint32_t)src[e]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(&mx[blockIdx.y], m);
}

__global__ __launch_bounds__(kBlock) void k_fog_blend(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                      const double* __restrict__ maps,
                                                      const double* __restrict__ minmax,
                                                      const uint32_t* __restrict__ imax, double c0) {
  __shared__ double lut[256];                    // b / 255.0: the same division, once per workgroup instead of three times per pixel (round 5)
  lut[threadIdx.x] = (double)threadIdx.x / 255.0;
  __syncthreads();
  const int img = blockIdx.y;
  const double mn = minmax[img * 2], mxm = minmax[img * 2 + 1];
  const double max_val = (double)imax[img] / 255.0;
  const double* M = maps + (size_t)img * MAPN * MAPN;
  const size_t base = (size_t)img * HW * HW * 3;
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < HW * HW; p += gridDim.x * kBlock) {
    const int y = p / HW, x = p % HW;
    const double sh = M[y * MAPN + x] - mn;
    const double pl = sh / mxm;
    const double add = c0 * pl;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double v = lut[in[base + p * 3 + c]] + add;
      const double num = v * max_val;
      double r = num / (max_val + c0);
      r = r < 0.0 ? 0.0 : (r > 1.0 ? 1.0 : r);
      out[base + p * 3 + c] = (uint8_t)(uint32_t)(r * 255.0);
    }
  }
}

// =====================================================================================
// generic single-channel field helpers
// =====================================================================================

// field[n][HW][HW] = loc + scale * N(0,1) (native) -- injected fields are used directly
__global__ __launch_bounds__(kBlock) void k_normal_field(double* __restrict__ f, double loc, double scale, uint32_t k0,
                                                         uint32_t k1, uint32_t sample_base, int stream_id) {
  const uint32_t sample = blockIdx.y;
  for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < HW * HW / 4; q += gridDim.x * kBlock) {
    const float4 z = rart_normal4(k0, k1, q, stream_id, sample_base + sample);
    double* d = f + (size_t)sample * HW * HW + q * 4;
    d[0] = loc + scale * (double)z.x;
    d[1] = loc + scale * (double)z.y;
    d[2] = loc + scale * (double)z.z;
    d[3] = loc + scale * (double)z.w;
  }
}

// field = U(-1, 1) (native elastic displacement seeds)
__global__ __launch_bounds__(kBlock) void k_uniform_field(double* __restrict__ f, uint32_t k0, uint32_t k1,
                                                          uint32_t sample_base, int stream_id) {
  const uint32_t sample = blockIdx.y;
  for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < HW * HW; e += gridDim.x * kBlock) {
    const uint2 wv = threefry2x32(k0, k1, rart_ctr0(e, stream_id), sample_base + sample);
    f[(size_t)sample * HW * HW + e] = -1.0 + 2.0 * u53(wv);
  }
}

// scipy gaussian_filter1d along AXIS of [n][h][w] fields; REFLECT: mode='reflect' (half-sample symmetric)
// else 'nearest'.  TIN/TOUT in {double, float}; scipy computes each line in double and casts on store.
template <int AXIS, bool REFLECT, typename TIN, typename TOUT>
__global__ __launch_bounds__(kBlock) void k_field_gauss(const TIN* __restrict__ src, TOUT* __restrict__ dst, int n,
                                                        int h, int w, const double* __restrict__ wts, int radius,
                                                        double post_scale) {
  const size_t total = (size_t)n * h * w;
  const int len = AXIS == 0 ? h : w;
  const size_t stride = AXIS == 0 ? (size_t)w : 1;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int xo = (int)(i % w), yo = (int)((i / w) % h);
    const int l = AXIS == 0 ? yo : xo;
    const size_t base = i - (size_t)l * stride;
    auto at = [&](int p) -> double {
      if (REFLECT) {
        const int period = 2 * len;
        p %= period;
        if (p < 0) p += period;
        if (p >= len) p = period - 1 - p;
      } else {
        p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
      }
      return (double)src[base + (size_t)p * stride];
    };
    double tmp = at(l) * wts[radius];
    for (int jj = -radius; jj < 0; ++jj) {
      const double pair = at(l + jj) + at(l - jj);
      tmp += pair * wts[jj + radius];
    }
    if (post_scale != 1.0) tmp = tmp * post_scale;
    dst[i] = (TOUT)tmp;
  }
}

// The same filter with the lines resident in LDS (round 4).  k_field_gauss walks global memory with two integer modulos per tap; at
// elastic_transform's severity 1 (sigma = 170.8 px: radius 512, 1 025 taps per output and pass) that was ~10 ms per batch and 4 % of a
// ViT-B/16 ImageNet-C sweep.  Here a workgroup owns FL_N lines of one image: every line is staged ONCE, already extended by `radius`
// reflected / clamped positions on either side (the modulo runs len + 2 radius times per line, not per tap), the weights sit beside it,
// and the tap loop is two ds_read_b64 + a broadcast weight read + the reference's add and multiply-add -- in scipy's order, term by term
// (symmetric pairs first added, then weighted; fp contract off): bit-identical output.
//   AXIS 1 (along w): ext[line][pos], a thread owns position tid of each of the FL_N rows (adjacent lanes, adjacent addresses);
//   AXIS 0 (along h): ext[pos][line], lane -> (line = tid & 7, pos = tid >> 3 + 32 k): 64 contiguous bytes per position.
constexpr int FL_N = 8;
// GEN (round 5, elastic_transform's native fields): the staged values are drawn here -- U(-1, 1) of the counter generator at the element's
// own index, exactly k_uniform_field's expression -- instead of being written to HBM by a launch of their own and read back (2 x 34 us + a
// 103 MB round trip per field at B = 256); the reflected extension re-draws the element it mirrors.
struct FieldGen {
  uint32_t k0, k1, sample_base;
  int stream_id;
};
template <int AXIS, bool REFLECT, typename TIN, typename TOUT, bool GEN = false>
__global__ __launch_bounds__(kBlock) void k_field_gauss_lds(const TIN* __restrict__ src, TOUT* __restrict__ dst, int h, int w,
                                                            const double* __restrict__ wts, int radius, double post_scale,
                                                            FieldGen gen = FieldGen{0, 0, 0, 0}) {
  extern __shared__ __attribute__((aligned(16))) double fl_s[];
  const int len = AXIS == 0 ? h : w, other = AXIS == 0 ? w : h;
  const int ext = len + 2 * radius;
  const int extp = AXIS == 1 ? (ext | 1) : ext;                     // AXIS 1: odd line stride
  double* const wl = fl_s;                                          // [radius + 1]
  double* const e = fl_s + ((radius + 2) & ~1);                     // the extended lines
  const int groups = (other + FL_N - 1) / FL_N;
  const int img = blockIdx.x / groups, l0 = (blockIdx.x - img * groups) * FL_N;
  const size_t ibase = (size_t)img * h * w;
  for (int i = threadIdx.x; i <= radius; i += kBlock) wl[i] = wts[i];
  for (int i = threadIdx.x; i < FL_N * ext; i += kBlock) {
    const int line = AXIS == 0 ? i % FL_N : i / ext, q = AXIS == 0 ? i / FL_N : i - line * ext;
    int p = q - radius;
    if (REFLECT) {
      const int period = 2 * len;
      p %= period;
      if (p < 0) p += period;
      if (p >= len) p = period - 1 - p;
    } else {
      p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
    }
    double v = 0.0;
    if (l0 + line < other) {
      const size_t el = AXIS == 0 ? (size_t)p * w + (l0 + line) : (size_t)(l0 + line) * w + p;
      if (GEN) v = -1.0 + 2.0 * u53(threefry2x32(gen.k0, gen.k1, rart_ctr0((uint32_t)el, gen.stream_id), gen.sample_base + (uint32_t)img));
      else v = (double)src[ibase + el];
    }
    e[AXIS == 0 ? q * FL_N + line : line * extp + q] = v;
  }
  __syncthreads();
  const int per = AXIS == 0 ? (len * FL_N + kBlock - 1) / kBlock : FL_N;
  for (int k = 0; k < per; ++k) {
    int line, l;
    if (AXIS == 0) {
      const int o = k * kBlock + threadIdx.x;
      line = o % FL_N;
      l = o / FL_N;
    } else {
      line = k;
      l = threadIdx.x;
    }
    if (l >= len || l0 + line >= other) continue;
    const double* c = AXIS == 0 ? e + (size_t)(l + radius) * FL_N + line : e + (size_t)line * extp + l + radius;
    const int st = AXIS == 0 ? FL_N : 1;
    double tmp = c[0] * wl[radius];
    for (int jj = -radius; jj < 0; ++jj) {
      const double pair = c[jj * st] + c[-jj * st];
      tmp += pair * wl[jj + radius];
    }
    if (post_scale != 1.0) tmp = tmp * post_scale;
    dst[ibase + (AXIS == 0 ? (size_t)l * w + (l0 + line) : (size_t)(l0 + line) * w + l)] = (TOUT)tmp;
  }
}

// ---- the field filter as a dense matrix product on the fp64 matrix cores (round 5) ---------------------------------------------------
// elastic_transform's severity 1 filters its two 224 x 224 displacement fields with sigma = 170.8 px: radius 512, 1 025 taps per output and
// pass over a signal of 224 samples that `mode='reflect'` wraps four and a half times -- 10.1 ms per 256-image batch, 29 % of a sweep over all
// 19 corruptions x 5 severities (scratch/r5/sweep_all_severities.py).  Along one axis the filter IS a 224 x 224 matrix: out = M in with
// M[l][p] = sum of the weights w[j] whose tap l + j reflects onto p.  Both passes become fp64 GEMMs against that one matrix,
// Y = M X (axis 0) and Z = Y M^T (axis 1), 224 instead of 1 025 x 1.5 operations per output, on v_mfma_f64_16x16x4_f64 (the data operand from
// an LDS slab laid out so that its reads are conflict free, the matrix operand in fragment order straight from L2, 56 fragments = 112 VGPRs
// per 16-row block).  NOT the reference's summation order: the fp64 results differ from scipy's in the last bits (~1e-16 relative), which
// survives the cast of the displacement to fp32 for ~1e-9 of its elements; the corrupted images stay inside elastic_transform's stated
// tolerance (<= 1 LSB on <= 1e-4 of the pixels; measured: identical on the test batches,
// test_elastic_dense_field_filter_matches_the_ordered_kernels).  Used when the kernel is longer than half the signal (4 radius + 2 > 224:
// severity 1, and severity 2 -- sigma 19.5 px, 119 taps -- whose matrix is banded: the zero K steps are skipped); RART_ELASTIC_ORDERED=1
// keeps the ordered kernels.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int FD_THREADS = 448, FD_SLAB = 32, FD_GROUPS = 18; // 7 waves; a slab = 32 columns (pass 0) / 32 rows (pass 1); 14 x 18 = 252 persistent workgroups
constexpr int FD_LDX = 48;                                   // pass 0: row stride of the [224][32] slab in doubles (96 dwords: K groups 32 banks apart)
constexpr int FD_LDY = 226;                                  // pass 1: row stride of the [32][224] slab in doubles (452 dwords = 4 mod 64)
constexpr int FD_PF = HW * FD_SLAB / FD_THREADS;             // 16 slab elements per thread

// AXIS 0: Y[f][l][c] = sum_p M[l][p] X[f][p][c]            (TOUT double, post_scale 1)
// AXIS 1: Z[f][r][c] = post_scale * sum_p Y[f][r][p] M[c][p]   (TOUT float)
// grid (7 slabs x 2 halves, min(n, FD_GROUPS)), persistent: wave w of half h owns the 16 output rows (AXIS 0) / columns (AXIS 1) of block
// 7 h + w and keeps that block's 56 matrix fragments in registers for EVERY field the workgroup walks (f = group, group + groups, ...); the data
// slab goes through LDS once per field, the next field's slab already in flight into registers while this one is multiplied.  (First version: a
// workgroup per two fields, slab staged and fragments re-read in the open: 201 / 188 us per 256 fields = 29 TFLOP/s.)
template <int AXIS, typename TOUT>
__global__ __launch_bounds__(FD_THREADS) void k_field_dense(const double* __restrict__ src, TOUT* __restrict__ dst,
                                                            const double* __restrict__ mfrag, double post_scale, int n, int radius) {
  extern __shared__ __attribute__((aligned(16))) double fd_s[];     // AXIS 0: [224][FD_LDX]; AXIS 1: [32][FD_LDY]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slab = (int)blockIdx.x >> 1, blk = ((int)blockIdx.x & 1) * 7 + wave;
  const int j = lane & 15, kk = lane >> 4;
  // element e = tid + FD_THREADS * q of a slab: where it comes from (offset inside the field) and where it goes in LDS
  int goff[FD_PF], loff[FD_PF];
#pragma unroll
  for (int q = 0; q < FD_PF; ++q) {
    const int e = tid + FD_THREADS * q;
    if (AXIS == 0) {
      goff[q] = (e >> 5) * HW + slab * FD_SLAB + (e & 31);
      loff[q] = (e >> 5) * FD_LDX + (e & 31);
    } else {
      const int rr = e / HW;
      goff[q] = slab * FD_SLAB * HW + e;
      loff[q] = rr * FD_LDY + (e - rr * HW);
    }
  }
  double pf[FD_PF];
  int f = blockIdx.y;
  if (f < n) {
#pragma unroll
    for (int q = 0; q < FD_PF; ++q) pf[q] = src[(size_t)f * HW * HW + goff[q]];
  }
  // rows 16 blk .. 16 blk + 15 of M are zero outside columns [16 blk - radius, 16 blk + 15 + radius] (taps past an edge fold back INTO that range):
  // K steps outside it multiply by exact zeros and are skipped (severity 2: 34 of 56 steps per block)
  const int ublk = __builtin_amdgcn_readfirstlane(blk);
  const int s_lo = max(0, 16 * ublk - radius) >> 2, s_hi = (min(HW - 1, 16 * ublk + 15 + radius) >> 2) + 1;
  double m[56];
#pragma unroll
  for (int s = 0; s < 56; ++s) m[s] = (s >= s_lo && s < s_hi) ? mfrag[((size_t)blk * 56 + s) * 64 + lane] : 0.0;      // M[blk * 16 + j][4 s + kk]
  for (; f < n; f += gridDim.y) {
    const size_t fbase = (size_t)f * HW * HW;
#pragma unroll
    for (int q = 0; q < FD_PF; ++q) fd_s[loff[q]] = pf[q];
    __syncthreads();
    if (f + (int)gridDim.y < n) {
#pragma unroll
      for (int q = 0; q < FD_PF; ++q) pf[q] = src[(size_t)(f + gridDim.y) * HW * HW + goff[q]];
    }
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      const double* lp = AXIS == 0 ? fd_s + kk * FD_LDX + t * 16 + j : fd_s + (t * 16 + j) * FD_LDY + kk;
#pragma unroll
      for (int s = 0; s < 56; ++s) {
        // v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]; D[row = (lane >> 4) + 4 reg][col = lane & 15]
        if (s >= s_lo && s < s_hi) {
          if (AXIS == 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m[s], lp[4 * s * FD_LDX], acc, 0, 0, 0);
          else           acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lp[4 * s], m[s], acc, 0, 0, 0);
        }
        if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (AXIS == 0) {
          dst[fbase + (size_t)(blk * 16 + kk + 4 * q) * HW + slab * FD_SLAB + t * 16 + j] = (TOUT)acc[q];
        } else {
          double v = acc[q];
          if (post_scale != 1.0) v = v * post_scale;
          dst[fbase + (size_t)(slab * FD_SLAB + t * 16 + kk + 4 * q) * HW + blk * 16 + j] = (TOUT)v;
        }
      }
    }
    __syncthreads();
  }
}

// scipy gaussian_filter1d along AXIS of n fields [h][w]: the LDS-resident kernel whenever its lines fit, else the global-memory walk
template <int AXIS, bool REFLECT, typename TIN, typename TOUT>
static int launch_field_gauss(const TIN* src, TOUT* dst, int n, int h, int w, const double* wts, int radius, double post_scale,
                              hipStream_t st) {
  const int len = AXIS == 0 ? h : w, other = AXIS == 0 ? w : h;
  const int ext = len + 2 * radius, extp = AXIS == 1 ? (ext | 1) : ext;
  const size_t lds = (size_t)(((radius + 2) & ~1) + (size_t)FL_N * extp) * sizeof(double);
  if (len <= kBlock && lds <= 150 * 1024 &&
      rart_raise_dynamic_lds((const void*)k_field_gauss_lds<AXIS, REFLECT, TIN, TOUT>, lds, "gaussian field filter")) {
    hipLaunchKernelGGL((k_field_gauss_lds<AXIS, REFLECT, TIN, TOUT>), dim3((unsigned)(n * ((other + FL_N - 1) / FL_N))), dim3(kBlock), lds, st,
                       src, dst, h, w, wts, radius, post_scale);
    return 0;
  }
  const int g = rart_grid_for((size_t)n * h * w, kBlock, 256 * 16);
  hipLaunchKernelGGL((k_field_gauss<AXIS, REFLECT, TIN, TOUT>), dim3(g), dim3(kBlock), 0, st, src, dst, n, h, w, wts, radius, post_scale);
  return 0;
}

// =====================================================================================
// snow (corruptions.py:265-290)
// =====================================================================================
struct SnowZoom {
  int ch, top, out_n, trim;
};

// clipped_zoom of the fp64 layer (order-1, grid_mode False) -> threshold -> uint8 'L' image
__global__ __launch_bounds__(kBlock) void k_snow_layer_u8(const double* __restrict__ layer, uint8_t* __restrict__ out,
                                                          SnowZoom z, double thresh) {
  const double* L = layer + (size_t)blockIdx.y * HW * HW;
  uint8_t* o = out + (size_t)blockIdx.y * HW * HW;
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < HW * HW; p += gridDim.x * kBlock) {
    const int yo = p / HW, xo = p % HW;
    const double sy = z.out_n > 1 ? (double)((long long)(yo + z.trim) * (z.ch - 1)) / (double)(z.out_n - 1) : 0.0;
    const double sx = z.out_n > 1 ? (double)((long long)(xo + z.trim) * (z.ch - 1)) / (double)(z.out_n - 1) : 0.0;
    int y0 = (int)floor(sy), x0 = (int)floor(sx);
    y0 = y0 < 0 ? 0 : (y0 > z.ch - 1 ? z.ch - 1 : y0);
    x0 = x0 < 0 ? 0 : (x0 > z.ch - 1 ? z.ch - 1 : x0);
    const int y1 = y0 + 1 < z.ch ? y0 + 1 : z.ch - 1, x1 = x0 + 1 < z.ch ? x0 + 1 : z.ch - 1;
    const double ty = sy - (double)y0, tx = sx - (double)x0;
    const double* r0 = L + (size_t)(z.top + y0) * HW + z.top;
    const double* r1 = L + (size_t)(z.top + y1) * HW + z.top;
    const double omty = 1.0 - ty, omtx = 1.0 - tx;
    const double l0 = r0[x0] * omty, l1 = r1[x0] * ty;
    const double left = l0 + l1;
    const double g0 = r0[x1] * omty, g1 = r1[x1] * ty;
    const double right = g0 + g1;
    const double v0 = left * omtx, v1 = right * tx;
    double v = v0 + v1;
    if (v < thresh) v = 0.0;
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    o[p] = (uint8_t)(uint32_t)(v * 255.0);
  }
}

__global__ __launch_bounds__(kBlock) void k_snow_blend(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                       const uint8_t* __restrict__ layer_u8, float c6, float omc6) {
  __shared__ double lutd[256];                   // b / 255.0 and b / 255.0f: the same divisions, once per workgroup (round 5)
  __shared__ float lutf[256];
  lutd[threadIdx.x] = (double)threadIdx.x / 255.0;
  lutf[threadIdx.x] = (float)threadIdx.x / 255.0f;
  __syncthreads();
  const int img = blockIdx.y;
  const size_t base = (size_t)img * HW * HW * 3;
  const uint8_t* L = layer_u8 + (size_t)img * HW * HW;
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < HW * HW; p += gridDim.x * kBlock) {
    const int y = p / HW, x = p % HW;
    const float r = lutf[in[base + p * 3]], g = lutf[in[base + p * 3 + 1]], b = lutf[in[base + p * 3 + 2]];
    const float g0 = 0.299f * r, g1 = 0.587f * g, g2 = 0.114f * b;
    const float gray = (g0 + g1) + g2;
    const float gg = gray * 1.5f;
    const float lift = gg + 0.5f;
    const double s1 = lutd[L[p]];
    const double s2 = lutd[L[(HW - 1 - y) * HW + (HW - 1 - x)]];  // np.rot90(k=2)
    const float px[3] = {r, g, b};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float mxv = fmaxf(px[c], lift);
      const float a = c6 * px[c];
      const float bb = omc6 * mxv;
      const float xn = a + bb;
      const double t = (double)xn + s1;
      double v = t + s2;
      v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
      out[base + p * 3 + c] = (uint8_t)(uint32_t)(v * 255.0);
    }
  }
}

// =====================================================================================
// elastic_transform (corruptions.py:395-424)
// =====================================================================================
__device__ __forceinline__ int reflect101(int i, int n) {
  const int period = 2 * (n - 1);
  i %= period;
  if (i < 0) i += period;
  return i >= n ? period - i : i;
}

// per image: jitter -> cv2.getAffineTransform(pts1, pts2) -> inverse map coefficients (warpAffine inverts M)
__global__ void k_elastic_affine(double* __restrict__ inv, const float* __restrict__ jitter_inj, float c2, uint32_t k0,
                                 uint32_t k1, uint32_t sample_base) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= (int)gridDim.x * (int)blockDim.x) return;
  // pts1 for a 224x224 image: centre 112, square 74 (corruptions.py:407-411)
  const float p1[3][2] = {{186.f, 186.f}, {186.f, 38.f}, {38.f, 38.f}};
  float p2[3][2];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 2; ++j) {
      float jt;
      if (jitter_inj) {
        jt = jitter_inj[img * 6 + i * 2 + j];
      } else {
        const uint2 wv = threefry2x32(k0, k1, rart_ctr0(i * 2 + j, 6), sample_base + img);
        const double lo = -(double)c2;
        jt = (float)(lo + ((double)c2 - lo) * u53(wv));   // np.random.uniform(-c2, c2).astype(float32)
      }
      p2[i][j] = p1[i][j] + jt;
    }
  // solve M * [x y 1]^T = [x' y']^T for the three correspondences (Cramer, double)
  const double x0 = p1[0][0], y0 = p1[0][1], x1 = p1[1][0], y1 = p1[1][1], x2 = p1[2][0], y2 = p1[2][1];
  const double det = x0 * (y1 - y2) - y0 * (x1 - x2) + (x1 * y2 - x2 * y1);
  double M[2][3];
  for (int r = 0; r < 2; ++r) {
    const double u0 = p2[0][r], u1 = p2[1][r], u2 = p2[2][r];
    M[r][0] = (u0 * (y1 - y2) - y0 * (u1 - u2) + (u1 * y2 - u2 * y1)) / det;
    M[r][1] = (x0 * (u1 - u2) - u0 * (x1 - x2) + (x1 * u2 - x2 * u1)) / det;
    M[r][2] = (x0 * (y1 * u2 - y2 * u1) - y0 * (x1 * u2 - x2 * u1) + u0 * (x1 * y2 - x2 * y1)) / det;
  }
  double D = M[0][0] * M[1][1] - M[0][1] * M[1][0];
  D = D != 0.0 ? 1.0 / D : 0.0;
  const double A11 = M[1][1] * D, A22 = M[0][0] * D, A12 = -M[0][1] * D, A21 = -M[1][0] * D;
  const double b1 = -A11 * M[0][2] - A12 * M[1][2];
  const double b2 = -A21 * M[0][2] - A22 * M[1][2];
  double* o = inv + (size_t)img * 6;
  o[0] = A11; o[1] = A12; o[2] = b1; o[3] = A21; o[4] = A22; o[5] = b2;
}

// cv2.warpAffine INTER_LINEAR BORDER_REFLECT_101 on float32 (x/255): 1/32-pixel fixed-point coordinates
__global__ __launch_bounds__(kBlock) void k_warp_affine(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                        const double* __restrict__ inv) {
  const int img = blockIdx.y;
  const double* A = inv + (size_t)img * 6;
  const uint8_t* src = in + (size_t)img * HW * HW * 3;
  float* dst = out + (size_t)img * HW * HW * 3;
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < HW * HW; p += gridDim.x * kBlock) {
    const int y = p / HW, x = p % HW;
    const long long adelta = llrint(A[0] * (double)x * 1024.0), bdelta = llrint(A[3] * (double)x * 1024.0);
    const long long X0 = llrint((A[1] * (double)y + A[2]) * 1024.0) + 16;
    const long long Y0 = llrint((A[4] * (double)y + A[5]) * 1024.0) + 16;
    const long long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const int sx = (int)(X >> 5), sy = (int)(Y >> 5);
    const float fx = (float)(X & 31) / 32.0f, fy = (float)(Y & 31) / 32.0f;
    const int xa = reflect101(sx, HW), xb = reflect101(sx + 1, HW), ya = reflect101(sy, HW), yb = reflect101(sy + 1, HW);
    const float w00 = (1.0f - fy) * (1.0f - fx), w01 = (1.0f - fy) * fx, w10 = fy * (1.0f - fx), w11 = fy * fx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v00 = (float)src[(ya * HW + xa) * 3 + c] / 255.0f, v01 = (float)src[(ya * HW + xb) * 3 + c] / 255.0f;
      const float v10 = (float)src[(yb * HW + xa) * 3 + c] / 255.0f, v11 = (float)src[(yb * HW + xb) * 3 + c] / 255.0f;
      const float t0 = v00 * w00, t1 = v01 * w01, t2 = v10 * w10, t3 = v11 * w11;
      dst[p * 3 + c] = ((t0 + t1) + t2) + t3;
    }
  }
}

// scipy map_coordinates(order=1, mode='reflect') of the warped fp32 image at (y+dy, x+dx, c)
__device__ __forceinline__ double reflect_coord(double c, int n) {
  c = c + 0.5;
  const double p = 2.0 * n;
  c = c - p * floor(c / p);
  if (c > (double)n) c = p - c;
  return c - 0.5;
}

__global__ __launch_bounds__(kBlock) void k_elastic_gather(const float* __restrict__ img_all,
                                                           const float* __restrict__ dx_all,
                                                           const float* __restrict__ dy_all, uint8_t* __restrict__ out) {
  const int img = blockIdx.y;
  const float* I = img_all + (size_t)img * HW * HW * 3;
  const float* DX = dx_all + (size_t)img * HW * HW;
  const float* DY = dy_all + (size_t)img * HW * HW;
  uint8_t* o = out + (size_t)img * HW * HW * 3;
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < HW * HW; p += gridDim.x * kBlock) {
    const int y = p / HW, x = p % HW;
    const double cy = reflect_coord((double)y + (double)DY[p], HW);
    const double cx = reflect_coord((double)x + (double)DX[p], HW);
    const double fy0 = floor(cy), fx0 = floor(cx);
    const double ty = cy - fy0, tx = cx - fx0;
    int y0 = (int)fy0, x0 = (int)fx0;
    int y1 = y0 + 1, x1 = x0 + 1;
    y0 = y0 < 0 ? 0 : (y0 > HW - 1 ? HW - 1 : y0);
    y1 = y1 < 0 ? 0 : (y1 > HW - 1 ? HW - 1 : y1);
    x0 = x0 < 0 ? 0 : (x0 > HW - 1 ? HW - 1 : x0);
    x1 = x1 < 0 ? 0 : (x1 > HW - 1 ? HW - 1 : x1);
    const double wy0 = 1.0 - ty, wx0 = 1.0 - tx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double t = 0.0;
      t += ((double)I[(y0 * HW + x0) * 3 + c] * wy0) * wx0;
      t += ((double)I[(y0 * HW + x1) * 3 + c] * wy0) * tx;
      t += ((double)I[(y1 * HW + x0) * 3 + c] * ty) * wx0;
      t += ((double)I[(y1 * HW + x1) * 3 + c] * ty) * tx;
      float v = (float)t;
      v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
      o[p * 3 + c] = (uint8_t)(uint32_t)(v * 255.0f);
    }
  }
}

// =====================================================================================
// spatter, mud branch (corruptions.py:329-342)
// =====================================================================================
__global__ __launch_bounds__(kBlock) void k_spatter_mask(const double* __restrict__ liquid, float* __restrict__ m,
                                                         double thresh, size_t total) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    double v = liquid[i];
    if (v < thresh) v = 0.0;
    m[i] = v > thresh ? 1.0f : 0.0f;
  }
}

__global__ __launch_bounds__(kBlock) void k_spatter_mud(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                        const float* __restrict__ m_all) {
  const int img = blockIdx.y;
  const size_t base = (size_t)img * HW * HW * 3;
  const float* Mk = m_all + (size_t)img * HW * HW;
  const float col[3] = {(float)(63 / 255.), (float)(42 / 255.), (float)(20 / 255.)};
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < HW * HW; p += gridDim.x * kBlock) {
    float m = Mk[p];
    if (m < 0.8f) m = 0.f;
    const float om = 1.0f - m;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = (float)in[base + p * 3 + c] / 255.0f;
      const float cm = col[c] * m;
      const float xm = x * om;
      float v = xm + cm;
      v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
      out[base + p * 3 + c] = (uint8_t)(uint32_t)(v * 255.0f);
    }
  }
}

// ---- host tables --------------------------------------------------------------------------------
std::vector<double> gauss_weights(double sigma, double truncate, int* radius) {
  const int r = (int)(truncate * sigma + 0.5);
  std::vector<double> w(2 * r + 1);
  double sum = 0.0;
  for (int i = -r; i <= r; ++i) {
    w[i + r] = exp(-0.5 / (sigma * sigma) * (double)i * (double)i);
    sum += w[i + r];
  }
  for (auto& v : w) v /= sum;
  *radius = r;
  return w;
}

// the folded filter matrix of `mode='reflect'` in MFMA fragment order: frag[(block * 56 + s) * 64 + lane] = M[block * 16 + (lane & 15)][4 s + (lane >> 4)]
const std::vector<double>& cached_dense_matrix(int slot, const std::vector<double>& w, int radius) {
  static std::vector<double> tabs[16];
  std::lock_guard<std::mutex> lk(rart_host_table_mutex());
  if (tabs[slot].empty()) {
    std::vector<long double> M((size_t)HW * HW, 0.0L);
    for (int l = 0; l < HW; ++l)
      for (int jx = -radius; jx <= radius; ++jx) {
        int p = (l + jx) % (2 * HW);
        if (p < 0) p += 2 * HW;
        if (p >= HW) p = 2 * HW - 1 - p;
        M[(size_t)l * HW + p] += (long double)w[jx + radius];
      }
    tabs[slot].resize((size_t)(HW / 16) * 56 * 64);
    for (int blk = 0; blk < HW / 16; ++blk)
      for (int sidx = 0; sidx < 56; ++sidx)
        for (int lane = 0; lane < 64; ++lane)
          tabs[slot][((size_t)blk * 56 + sidx) * 64 + lane] = (double)M[(size_t)(blk * 16 + (lane & 15)) * HW + 4 * sidx + (lane >> 4)];
  }
  return tabs[slot];
}

// The same matrix RESIDENT on the current device: allocated and uploaded once per (device, slot), for the life of the process (401 KB per
// severity).  Round 5 copied it from pageable host memory with hipMemcpyAsync on every call -- a host stall per call, and not capturable in
// a hipGraph.  The first call per device must not run under stream capture (it allocates and copies synchronously).
const double* cached_dense_matrix_dev(int slot, const std::vector<double>& host) {
  static const double* dev_tabs[64][16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(rart_host_table_mutex());
  if (!dev_tabs[dev][slot]) {
    double* p = nullptr;
    if (hipMalloc((void**)&p, host.size() * sizeof(double)) != hipSuccess) return nullptr;
    if (hipMemcpy(p, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(p);
      return nullptr;
    }
    dev_tabs[dev][slot] = p;
  }
  return dev_tabs[dev][slot];
}

const double kFog[5][2] = {{1.5, 2}, {2., 2}, {2.5, 1.7}, {2.5, 1.5}, {3., 1.4}};
const double kSnow[5][7] = {{0.1, 0.3, 3, 0.5, 10, 4, 0.8}, {0.2, 0.3, 2, 0.5, 12, 4, 0.7}, {0.55, 0.3, 4, 0.9, 12, 8, 0.7},
                            {0.55, 0.3, 4.5, 0.85, 12, 8, 0.65}, {0.55, 0.3, 2.5, 0.85, 12, 12, 0.55}};
const double kElastic[5][3] = {{244 * 2, 244 * 0.7, 244 * 0.1}, {244 * 2, 244 * 0.08, 244 * 0.2},
                               {244 * 0.05, 244 * 0.01, 244 * 0.02}, {244 * 0.07, 244 * 0.01, 244 * 0.02},
                               {244 * 0.12, 244 * 0.01, 244 * 0.02}};
const double kSpatter[5][6] = {{0.65, 0.3, 4, 0.69, 0.6, 0}, {0.65, 0.3, 3, 0.68, 0.6, 0}, {0.65, 0.3, 2, 0.68, 0.5, 0},
                               {0.65, 0.3, 1, 0.65, 1.5, 1}, {0.67, 0.4, 1, 0.65, 1.5, 1}};

dim3 img_grid(int n) { return dim3((HW * HW + kBlock - 1) / kBlock, n); }
size_t A256(size_t v) { return rart_align_up(v, 256); }

// process-lifetime weight tables (async uploads read them after the call returns)
const std::vector<double>& cached_weights(int slot, double sigma, double truncate, int* radius) {
  static std::vector<double> tabs[16];
  static int radii[16];
  std::lock_guard<std::mutex> lk(rart_host_table_mutex());
  if (tabs[slot].empty()) tabs[slot] = gauss_weights(sigma, truncate, &radii[slot]);
  *radius = radii[slot];
  return tabs[slot];
}
}  // namespace
#pragma clang fp contract(fast)

size_t rart_ws_composite(int id, int /*severity*/, int n, int h, int w) {
  if (h != HW || w != HW) return 0;
  const size_t field = A256((size_t)n * HW * HW * sizeof(double));
  switch (id) {
    case RART_FOG: return A256((size_t)n * MAPN * MAPN * sizeof(double)) + A256((size_t)n * 2 * sizeof(double)) +
                          A256((size_t)n * sizeof(uint32_t));
    case RART_SNOW: return field + 2 * A256((size_t)n * HW * HW) + rart_motion_tab_bytes(n);
    case RART_ELASTIC_TRANSFORM:
      return A256((size_t)n * 6 * sizeof(double)) + A256((size_t)n * HW * HW * 3 * sizeof(float)) + 2 * field +
             2 * A256((size_t)n * HW * HW * sizeof(float)) + A256(1025 * sizeof(double)) + A256((size_t)(HW / 16) * 56 * 64 * sizeof(double));
    case RART_SPATTER: return 2 * field + 2 * A256((size_t)n * HW * HW * sizeof(float)) + 2 * A256(64 * sizeof(double));
  }
  return 0;
}

int rart_launch_composite(int id, const RartCorruptArgs& a) {
  RART_CHECK_ARG(a.h == HW && a.w == HW, "%s: reference hard-codes 224x224", rart_corruption_name(id));
  const int s = a.severity - 1;
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32), sb = (uint32_t)a.sample_offset;
  auto inj = [&](int i) -> const void* { return (a.injected && a.n_injected > i) ? a.injected[i] : nullptr; };
  uint8_t* ws = (uint8_t*)a.workspace;
  const size_t field = A256((size_t)a.n * HW * HW * sizeof(double));
  hipStream_t st = a.stream;
  switch (id) {
    case RART_FOG: {
      double* maps = (double*)ws;
      ws += A256((size_t)a.n * MAPN * MAPN * sizeof(double));
      double* minmax = (double*)ws;
      ws += A256((size_t)a.n * 2 * sizeof(double));
      uint32_t* imax = (uint32_t*)ws;
      if (hipMemsetAsync(imax, 0, (size_t)a.n * sizeof(uint32_t), st) != hipSuccess) {
        rart_set_error("fog: memset failed");
        return RART_ERR_HIP;
      }
      hipLaunchKernelGGL(k_plasma, dim3(a.n), dim3(PLASMA_T), 0, st, maps, minmax, (const double*)inj(0), kFog[s][1], k0,
                         k1, sb);
      hipLaunchKernelGGL(k_image_max, dim3(8, a.n), dim3(kBlock), 0, st, a.in, imax, (uint32_t)(HW * HW * 3));
      hipLaunchKernelGGL(k_fog_blend, img_grid(a.n), dim3(kBlock), 0, st, a.in, a.out, maps, minmax, imax, kFog[s][0]);
      break;
    }
    case RART_SNOW: {
      const double* c = kSnow[s];
      double* layer = (double*)ws;
      ws += field;
      uint8_t* l8a = ws;
      ws += A256((size_t)a.n * HW * HW);
      uint8_t* l8b = ws;
      ws += A256((size_t)a.n * HW * HW);
      void* tab = ws;
      const double* layer_src = (const double*)inj(0);
      if (!layer_src) {
        hipLaunchKernelGGL(k_normal_field, dim3(HW * HW / 4 / kBlock, a.n), dim3(kBlock), 0, st, layer, c[0], c[1], k0,
                           k1, sb, 7);
        layer_src = layer;
      }
      SnowZoom z;
      z.ch = (int)ceil((double)HW / c[2]);
      z.top = (HW - z.ch) / 2;
      z.out_n = (int)nearbyint((double)z.ch * c[2]);
      z.trim = (z.out_n - HW) / 2;
      hipLaunchKernelGGL(k_snow_layer_u8, img_grid(a.n), dim3(kBlock), 0, st, layer_src, l8a, z, c[3]);
      rart_motion_blur_gray(l8a, l8b, a.n, HW, HW, c[4], c[5], (const double*)inj(1), -135.0, -45.0, a.seed,
                            a.sample_offset, tab, st);
      const float c6 = (float)c[6], omc6 = (float)(1.0 - c[6]);
      hipLaunchKernelGGL(k_snow_blend, img_grid(a.n), dim3(kBlock), 0, st, a.in, a.out, l8b, c6, omc6);
      break;
    }
    case RART_ELASTIC_TRANSFORM: {
      const double* c = kElastic[s];
      double* inv = (double*)ws;
      ws += A256((size_t)a.n * 6 * sizeof(double));
      float* warped = (float*)ws;
      ws += A256((size_t)a.n * HW * HW * 3 * sizeof(float));
      double* f0 = (double*)ws;
      ws += field;
      double* f1 = (double*)ws;
      ws += field;
      float* dx = (float*)ws;
      ws += A256((size_t)a.n * HW * HW * sizeof(float));
      float* dy = (float*)ws;
      ws += A256((size_t)a.n * HW * HW * sizeof(float));
      double* wdev = (double*)ws;
      int radius;
      const std::vector<double>& wt = cached_weights(s, c[1], 3.0, &radius);
      RART_CHECK_ARG(radius <= 512, "elastic_transform: gaussian radius too large");
      if (hipMemcpyAsync(wdev, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) {
        rart_set_error("elastic_transform: weight upload failed");
        return RART_ERR_HIP;
      }
      hipLaunchKernelGGL(k_elastic_affine, dim3(a.n), dim3(1), 0, st, inv, (const float*)inj(0), (float)c[2], k0, k1, sb);
      hipLaunchKernelGGL(k_warp_affine, img_grid(a.n), dim3(kBlock), 0, st, a.in, warped, inv);
      // the kernel is longer than half the signal (severities 1 and 2): both passes as fp64 matrix products against the folded filter matrix
      bool dense = 4 * radius + 2 > HW && getenv("RART_ELASTIC_ORDERED") == nullptr;          // severities 1 (radius 512) and 2 (radius 59)
      const double* mdev = nullptr;
      const size_t lds0 = (size_t)HW * FD_LDX * sizeof(double), lds1 = (size_t)FD_SLAB * FD_LDY * sizeof(double);
      if (dense) {
        mdev = cached_dense_matrix_dev(s, cached_dense_matrix(s, wt, radius));          // resident on the device, uploaded once
        dense = mdev != nullptr &&
                rart_raise_dynamic_lds((const void*)k_field_dense<0, double>, lds0, "elastic field filter (dense)") &&
                rart_raise_dynamic_lds((const void*)k_field_dense<1, float>, lds1, "elastic field filter (dense)");
      }
      for (int which = 0; which < 2 && dense; ++which) {
        const double* fsrc = (const double*)inj(1 + which);
        if (!fsrc) {
          hipLaunchKernelGGL(k_uniform_field, img_grid(a.n), dim3(kBlock), 0, st, f0, k0, k1, sb, 11 + which);
          fsrc = f0;
        }
        const dim3 grid(2 * (HW / FD_SLAB), (unsigned)(a.n < FD_GROUPS ? a.n : FD_GROUPS));
        hipLaunchKernelGGL((k_field_dense<0, double>), grid, dim3(FD_THREADS), lds0, st, fsrc, f1, (const double*)mdev, 1.0, a.n, radius);
        hipLaunchKernelGGL((k_field_dense<1, float>), grid, dim3(FD_THREADS), lds1, st, (const double*)f1, which == 0 ? dx : dy,
                           (const double*)mdev, c[0], a.n, radius);
      }
      const bool dense_done = dense;
      for (int which = 0; which < 2 && !dense_done; ++which) {
        const double* fsrc = (const double*)inj(1 + which);
        const size_t gen_lds = (size_t)(((radius + 2) & ~1) + (size_t)FL_N * (HW + 2 * radius)) * sizeof(double);
        if (!fsrc && gen_lds <= 150 * 1024 && getenv("RART_ELASTIC_FIELD_KERNEL") == nullptr &&
            rart_raise_dynamic_lds((const void*)k_field_gauss_lds<0, true, double, double, true>, gen_lds, "gaussian field filter")) {
          // native fields: drawn inside the first pass's staging loop (no field tensor of their own)
          hipLaunchKernelGGL((k_field_gauss_lds<0, true, double, double, true>), dim3((unsigned)(a.n * ((HW + FL_N - 1) / FL_N))), dim3(kBlock),
                             gen_lds, st, (const double*)nullptr, f1, HW, HW, (const double*)wdev, radius, 1.0, FieldGen{k0, k1, sb, 11 + which});
          launch_field_gauss<1, true, double, float>((const double*)f1, which == 0 ? dx : dy, a.n, HW, HW, (const double*)wdev, radius, c[0], st);
          continue;
        }
        if (!fsrc) {
          hipLaunchKernelGGL(k_uniform_field, img_grid(a.n), dim3(kBlock), 0, st, f0, k0, k1, sb, 11 + which);
          fsrc = f0;
        }
        launch_field_gauss<0, true, double, double>(fsrc, f1, a.n, HW, HW, (const double*)wdev, radius, 1.0, st);
        launch_field_gauss<1, true, double, float>((const double*)f1, which == 0 ? dx : dy, a.n, HW, HW, (const double*)wdev, radius, c[0], st);
      }
      hipLaunchKernelGGL(k_elastic_gather, img_grid(a.n), dim3(kBlock), 0, st, (const float*)warped, (const float*)dx,
                         (const float*)dy, a.out);
      break;
    }
    case RART_SPATTER: {
      const double* c = kSpatter[s];
      double* layer = (double*)ws;
      ws += field;
      double* tmpd = (double*)ws;
      ws += field;
      float* m0 = (float*)ws;
      ws += A256((size_t)a.n * HW * HW * sizeof(float));
      float* m1 = (float*)ws;
      ws += A256((size_t)a.n * HW * HW * sizeof(float));
      double* w1dev = (double*)ws;
      ws += A256(64 * sizeof(double));
      double* w2dev = (double*)ws;
      int r1, r2;
      const bool water = c[5] == 0;
      const std::vector<double>& wt1 = cached_weights(water ? 12 + s : 8 + (s - 3) * 2, c[2], 4.0, &r1);
      const std::vector<double>& wt2 = water ? wt1 : cached_weights(9 + (s - 3) * 2, c[4], 4.0, &r2);
      if (water) r2 = r1;
      if (hipMemcpyAsync(w1dev, wt1.data(), wt1.size() * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
          hipMemcpyAsync(w2dev, wt2.data(), wt2.size() * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) {
        rart_set_error("spatter: weight upload failed");
        return RART_ERR_HIP;
      }
      const double* lsrc = (const double*)inj(0);
      if (!lsrc) {
        hipLaunchKernelGGL(k_normal_field, dim3(HW * HW / 4 / kBlock, a.n), dim3(kBlock), 0, st, layer, c[0], c[1], k0,
                           k1, sb, 13);
        lsrc = layer;
      }
      const int g = rart_grid_for((size_t)a.n * HW * HW, kBlock, 256 * 16);
      // gaussian(liquid_layer, sigma=c2): fp64, mode nearest, truncate 4
      launch_field_gauss<0, false, double, double>(lsrc, tmpd, a.n, HW, HW, (const double*)w1dev, r1, 1.0, st);
      launch_field_gauss<1, false, double, double>((const double*)tmpd, layer, a.n, HW, HW, (const double*)w1dev, r1, 1.0, st);
      if (water) {
        // severities 1-3: the Canny / distance-transform / equalizeHist pipeline (corrupt_spatter.hip); tmpd and m0 are free
        const int rc = rart_launch_spatter_water(a.in, a.out, (const double*)layer, a.n, c[3], c[4], (uint8_t*)m0, (int*)tmpd, st);
        if (rc != RART_OK) return rc;
        break;
      }
      hipLaunchKernelGGL(k_spatter_mask, dim3(g), dim3(kBlock), 0, st, (const double*)layer, m0, c[3],
                         (size_t)a.n * HW * HW);
      // gaussian(m.astype(float32), sigma=c4): float32 in/out, the axis-0 result is stored as float32
      launch_field_gauss<0, false, float, float>((const float*)m0, m1, a.n, HW, HW, (const double*)w2dev, r2, 1.0, st);
      launch_field_gauss<1, false, float, float>((const float*)m1, m0, a.n, HW, HW, (const double*)w2dev, r2, 1.0, st);
      hipLaunchKernelGGL(k_spatter_mud, img_grid(a.n), dim3(kBlock), 0, st, a.in, a.out, (const float*)m0);
      break;
    }
    default:
      rart_set_error("rart_launch_composite: bad id %d", id);
      return RART_ERR_INVALID;
  }
  RART_CHECK_LAUNCH("composite corruption launch");
  return RART_OK;
}
