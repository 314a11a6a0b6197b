// Pointwise ImageNet-C corruptions for gfx950: gaussian / speckle / shot / impulse noise,
// contrast, brightness, saturate, frost blend.
// Reference: RobustART/noise/utils/imagenet_c/corruptions.py:122-147,244-262,345-372.
//
// Two paths per noise corruption:
//  * native  : counter-based Threefry draws generated in-kernel, fp32 arithmetic in 0..255
//              scale, 16-byte coalesced loads/stores (HBM-roofline path);
//  * injected: the caller supplies what np.random returned; arithmetic follows the
//              reference's fp64 operation order exactly (bit-exact parity path).
#include "rart_common.h"

namespace {

constexpr int kBlock = 256;

// =====================================================================================
// native noise kernels (fp32, 16 B per lane)
// =====================================================================================

__device__ __forceinline__ uint32_t pack4_trunc(float a, float b, float c, float d) {
  // inputs already clamped to [0,255]; (unsigned) cast truncates toward zero like np.uint8()
  return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
}

__device__ __forceinline__ float clamp255(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 255.0f); }

// KIND 0: gaussian_noise  y = clip(x + 255*c*z)          (corruptions.py:122-126)
// KIND 1: speckle_noise   y = clip(x + x*c*z)            (corruptions.py:143-147)
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_normal_noise_native(
    const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t vec_per_sample, float c,
    uint32_t k0, uint32_t k1, uint32_t sample_base) {
  const uint32_t sample = blockIdx.y;
  const uint4* src = in + (size_t)sample * vec_per_sample;
  uint4* dst = out + (size_t)sample * vec_per_sample;
  const uint32_t gsample = sample_base + sample;
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < vec_per_sample; v += gridDim.x * kBlock) {
    const uint4 p = src[v];
    const uint32_t wi[4] = {p.x, p.y, p.z, p.w};
    uint32_t wo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 z = rart_normal4(k0, k1, v * 4u + j, 0, gsample);
      const float x0 = (float)(wi[j] & 0xFFu), x1 = (float)((wi[j] >> 8) & 0xFFu);
      const float x2 = (float)((wi[j] >> 16) & 0xFFu), x3 = (float)(wi[j] >> 24);
      float y0, y1, y2, y3;
      if (KIND == 0) {
        const float s = 255.0f * c;
        y0 = fmaf(s, z.x, x0); y1 = fmaf(s, z.y, x1); y2 = fmaf(s, z.z, x2); y3 = fmaf(s, z.w, x3);
      } else {
        y0 = fmaf(x0 * c, z.x, x0); y1 = fmaf(x1 * c, z.y, x1);
        y2 = fmaf(x2 * c, z.z, x2); y3 = fmaf(x3 * c, z.w, x3);
      }
      wo[j] = pack4_trunc(clamp255(y0), clamp255(y1), clamp255(y2), clamp255(y3));
    }
    dst[v] = make_uint4(wo[0], wo[1], wo[2], wo[3]);
  }
}

// scalar tail / odd-size path: one quad (4 elements) per thread, byte accesses
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_normal_noise_native_scalar(
    const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t elems_per_sample,
    uint32_t first_quad, float c, uint32_t k0, uint32_t k1, uint32_t sample_base) {
  const uint32_t sample = blockIdx.y;
  const uint8_t* src = in + (size_t)sample * elems_per_sample;
  uint8_t* dst = out + (size_t)sample * elems_per_sample;
  const uint32_t nquad = (elems_per_sample + 3) / 4;
  for (uint32_t q = first_quad + blockIdx.x * kBlock + threadIdx.x; q < nquad; q += gridDim.x * kBlock) {
    const float4 z = rart_normal4(k0, k1, q, 0, sample_base + sample);
    const float zz[4] = {z.x, z.y, z.z, z.w};
    for (int j = 0; j < 4; ++j) {
      const uint32_t e = q * 4 + j;
      if (e < elems_per_sample) {
        const float x = (float)src[e];
        const float y = KIND == 0 ? fmaf(255.0f * c, zz[j], x) : fmaf(x * c, zz[j], x);
        dst[e] = (uint8_t)(uint32_t)clamp255(y);
      }
    }
  }
}

// ---- matrix-core normal generator -----------------------------------------------------------------
// The Box-Muller kernels above are bound by integer VALU throughput (~4 cycles per wave64 instruction on
// gfx950): random BITS alone cost 12 instructions per 16-bit pair with Threefry-13.  This path spends ONE
// random byte per normal and moves the bits->gaussian transform onto the otherwise idle matrix cores:
//   S = A x H,  A = 32x32 iid uniform int8 (Threefry bytes), H = 32x32 Hadamard (+-1, orthogonal)
// one v_mfma_i32_32x32x32_i8 per wave yields 1024 sums of 32 uniform bytes: by orthogonality the 32 sums of
// a row are uncorrelated, each is a 32-term CLT sum (excess kurtosis -1.2/32), and a cubic Cornish-Fisher
// term z = y + (0.0375/24)(y^3 - 3y) removes that kurtosis (residual: 6th cumulant, < 1e-4 in density).
// Known limitation: the 32 outputs that share a matrix row are uncorrelated but not
// independent (their sum of squares equals that of the 32 input bytes).
// The field is a pure function of (seed, sample, 1 KiB chunk index): one wave = one aligned chunk.
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

struct HadamardTable {
  uint32_t w[64][4];
};
constexpr HadamardTable make_hadamard() {
  HadamardTable t{};
  for (int lane = 0; lane < 64; ++lane)
    for (int q = 0; q < 4; ++q) {
      uint32_t word = 0;
      for (int b = 0; b < 4; ++b) {
        const int k = 16 * (lane >> 5) + q * 4 + b, j = lane & 31;
        int x = k & j, par = 0;
        while (x) { par ^= x & 1; x >>= 1; }
        word |= (uint32_t)(par ? 0xFFu : 0x01u) << (8 * b);   // int8 -1 / +1
      }
      t.w[lane][q] = word;
    }
  return t;
}
__device__ const HadamardTable g_hadamard = make_hadamard();

// A bytes get their LSB forced to 1: uniform over the 128 odd values -127..127, i.e. exactly zero mean
// (plain int8 has mean -1/2, which the all-ones Hadamard column would turn into a -16 offset).
constexpr float kCltSigma = 418.0334915f;        // sqrt(32 * (128^2 - 1) / 3)
constexpr float kCltA = 0.9953120f / kCltSigma;  // z = s * (A + B s^2): Cornish-Fisher inverse, kappa4 = -1.20015/32
constexpr float kCltB = 0.0015627f / (kCltSigma * kCltSigma * kCltSigma);
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 16 raw CLT sums for this lane's 16 elements of chunk `chunk` (all 64 lanes of the wave must call this)
__device__ __forceinline__ i32x4 clt_hadamard_operand(int lane) {
  i32x4 b = {(int)g_hadamard.w[lane][0], (int)g_hadamard.w[lane][1], (int)g_hadamard.w[lane][2],
             (int)g_hadamard.w[lane][3]};
  return b;
}

__device__ __forceinline__ void clt_sums16(uint32_t k0, uint32_t k1, uint32_t chunk, uint32_t sample, int lane,
                                           const i32x4 b, float* s) {
  const uint2 w0 = threefry2x32(k0, k1, rart_ctr0(chunk * 128u + lane * 2u, 14), sample);
  const uint2 w1 = threefry2x32(k0, k1, rart_ctr0(chunk * 128u + lane * 2u + 1u, 14), sample);
  const i32x4 a = {(int)(w0.x | 0x01010101u), (int)(w0.y | 0x01010101u), (int)(w1.x | 0x01010101u),
                   (int)(w1.y | 0x01010101u)};
  // (the builtin, not inline asm: hipcc drains every outstanding load with s_waitcnt vmcnt(0) in front of an
  //  asm statement, which would serialise the prefetched next chunk behind this chunk's arithmetic)
  i32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 16; ++e) s[e] = (float)acc[e];
}

// v_cvt_pk_u8_f32 saturates to [0,255] and converts with the wave's CURRENT rounding mode; the kernel below runs with MODE.fp_round
// = toward zero, which makes it np.uint8's truncation of the clipped value (negative values saturate to 0 either way) without a
// v_floor per element.
__device__ __forceinline__ uint32_t pack4_sat(float a, float b, float c, float d) {
  uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(a, 0, 0u);
  w = __builtin_amdgcn_cvt_pk_u8_f32(b, 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(c, 2, w);
  return __builtin_amdgcn_cvt_pk_u8_f32(d, 3, w);
}

// One wave = one aligned 1 KiB chunk (chunk ids run across samples, so any batch packs exactly).
// Round 3 (scratch/exp/noise_variants.hip, all variants in one run, outputs compared byte for byte): 207 -> 164 VALU
// instructions per wave and 20.75 -> 18.2 us per launch at B = 256 (0.46 -> 0.53 of the 8 TB/s peak; a plain copy of the same
// bytes takes ~16 us on this part) from
//   * the chunk index held in SGPRs (readfirstlane of the wave id: hipcc otherwise runs the sample / chunk division, a ~25
//     instruction sequence, on the vector unit) with a host-computed multiply-shift reciprocal;
//   * MODE.fp_round = toward zero for the wave (s_setreg), so the 16 conversions need no v_floor; the packed FMAs round toward
//     zero as well, which left all 38.5 M output bytes of the test batch unchanged.
// Measured and not adopted: 2 chunks per wave with both loads issued up front (24.5 us: 2.3 rounds of waves -> grid tail);
// persistent waves with inline-asm loads two chunks ahead and s_waitcnt vmcnt(1) (21.7-23 us at 768-2048 workgroups): the
// kernel is not waiting on load latency -- 8 resident waves per SIMD already cover it.
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_normal_noise_mfma(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                              uint32_t chunks_per_sample, uint32_t total_chunks,
                                                              float c, uint32_t k0, uint32_t k1, uint32_t sample_base,
                                                              uint32_t cps_magic, uint32_t cps_shift) {
  __builtin_amdgcn_s_setreg(1 | (0 << 6) | (1 << 11), 3);        // hwreg(HW_REG_MODE, 0, 2): single-precision rounding = toward zero
  const int lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (g >= total_chunks) return;                                // wave-uniform
  const uint4 cur = in[(size_t)g * 64 + lane];
  const i32x4 hb = clt_hadamard_operand(lane);
  const float gsc = KIND == 0 ? 255.0f * c : c;
  const float ga = gsc * kCltA, gb = gsc * kCltB;
  const f32x2 ga2 = {ga, ga}, gb2 = {gb, gb};
  const uint32_t sample = (uint32_t)(((uint64_t)g * cps_magic) >> cps_shift), chunk = g - sample * chunks_per_sample;
  float s[16];
  clt_sums16(k0, k1, chunk, sample_base + sample, lane, hb, s);
  const uint32_t wi[4] = {cur.x, cur.y, cur.z, cur.w};
  uint32_t wo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // two elements per v_pk_* instruction (the packed fp32 pipe is the only 2-results-per-issue VALU path)
    f32x2 y[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const f32x2 x = {(float)((wi[j] >> (16 * hh)) & 0xFFu), (float)((wi[j] >> (16 * hh + 8)) & 0xFFu)};
      const f32x2 sv = {s[j * 4 + 2 * hh], s[j * 4 + 2 * hh + 1]};
      f32x2 t = __builtin_elementwise_fma(gb2, sv * sv, ga2);   // g * (A + B s^2)
      if (KIND == 1) t = t * x;                                  // speckle: x + x*c*z
      y[hh] = __builtin_elementwise_fma(t, sv, x);
    }
    wo[j] = pack4_sat(y[0].x, y[0].y, y[1].x, y[1].y);
  }
  out[(size_t)g * 64 + lane] = make_uint4(wo[0], wo[1], wo[2], wo[3]);
}

// Several severities (or, generally, several (scale, seed) draws) of ONE source batch in one launch: a wave reads its 1 KiB chunk once and
// writes `ns` outputs, each with its own field -- output i is bit-identical to k_normal_noise_mfma launched with (c[i], seed i) on the
// same input.  ImageNet-C generation corrupts every image at five severities (imagenet_c/__init__.py:13-35 is called once per (image,
// severity) by the generation scripts): five launches move 10 x the batch through HBM, this one 6 x, and a wave that lives five times as
// long amortises the launch ramp / tail that bounds the 16 us single-severity launch at B = 256.
struct NoiseMultiArgs {
  uint4* out[8];
  float c[8];
  uint32_t k0[8], k1[8];
  int ns;
};
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_normal_noise_mfma_multi(const uint4* __restrict__ in, const NoiseMultiArgs m,
                                                                    uint32_t chunks_per_sample, uint32_t total_chunks,
                                                                    uint32_t sample_base, uint32_t cps_magic, uint32_t cps_shift) {
  __builtin_amdgcn_s_setreg(1 | (0 << 6) | (1 << 11), 3);        // hwreg(HW_REG_MODE, 0, 2): single-precision rounding = toward zero
  const int lane = threadIdx.x & 63;
  const uint32_t g = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (g >= total_chunks) return;                                // wave-uniform
  const uint4 cur = in[(size_t)g * 64 + lane];
  const i32x4 hb = clt_hadamard_operand(lane);
  const uint32_t sample = (uint32_t)(((uint64_t)g * cps_magic) >> cps_shift), chunk = g - sample * chunks_per_sample;
  const uint32_t wi[4] = {cur.x, cur.y, cur.z, cur.w};
  for (int i = 0; i < m.ns; ++i) {
    const float gsc = KIND == 0 ? 255.0f * m.c[i] : m.c[i];
    const float ga = gsc * kCltA, gb = gsc * kCltB;
    const f32x2 ga2 = {ga, ga}, gb2 = {gb, gb};
    float s[16];
    clt_sums16(m.k0[i], m.k1[i], chunk, sample_base + sample, lane, hb, s);
    uint32_t wo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 y[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x2 x = {(float)((wi[j] >> (16 * hh)) & 0xFFu), (float)((wi[j] >> (16 * hh + 8)) & 0xFFu)};
        const f32x2 sv = {s[j * 4 + 2 * hh], s[j * 4 + 2 * hh + 1]};
        f32x2 t = __builtin_elementwise_fma(gb2, sv * sv, ga2);   // g * (A + B s^2)
        if (KIND == 1) t = t * x;                                  // speckle: x + x*c*z
        y[hh] = __builtin_elementwise_fma(t, sv, x);
      }
      wo[j] = pack4_sat(y[0].x, y[0].y, y[1].x, y[1].y);
    }
    m.out[i][(size_t)g * 64 + lane] = make_uint4(wo[0], wo[1], wo[2], wo[3]);
  }
}

// replay of the field for the parity tests: z[sample][element] exactly as the kernel above forms it
__global__ __launch_bounds__(kBlock) void k_clt_field(float* __restrict__ out, uint32_t vec_per_sample, uint32_t k0,
                                                      uint32_t k1, uint32_t sample_base) {
  const uint32_t sample = blockIdx.y;
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= vec_per_sample) return;
  float s[16];
  clt_sums16(k0, k1, v >> 6, sample_base + sample, threadIdx.x & 63, clt_hadamard_operand(threadIdx.x & 63), s);
  float* o = out + ((size_t)sample * vec_per_sample + v) * 16;
#pragma unroll
  for (int e = 0; e < 16; ++e) o[e] = s[e] * fmaf(kCltB, s[e] * s[e], kCltA);
}

// impulse_noise (corruptions.py:136-140 -> skimage random_noise 's&p'): each element flips with
// probability `amount` (24-bit threshold compare on the raw word), salt/pepper by one more bit.
// One Threefry call (2 words) serves 2 elements: word>>8 = flip uniform, bit 0 = salt.
__global__ __launch_bounds__(kBlock) void k_impulse_native(
    const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t vec_per_sample, uint32_t thresh24,
    uint32_t k0, uint32_t k1, uint32_t sample_base) {
  const uint32_t sample = blockIdx.y;
  const uint4* src = in + (size_t)sample * vec_per_sample;
  uint4* dst = out + (size_t)sample * vec_per_sample;
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < vec_per_sample; v += gridDim.x * kBlock) {
    const uint4 p = src[v];
    uint32_t wi[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // pair index within the sample: element e = v*16 + j*4 + h*2 (+0, +1)
        const uint2 w = threefry2x32(k0, k1, rart_ctr0(v * 8u + j * 2u + h, 0), sample_base + sample);
        const uint32_t ww[2] = {w.x, w.y};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int sh = (h * 2 + t) * 8;
          if ((ww[t] >> 8) < thresh24) {
            const uint32_t val = (ww[t] & 1u) ? 0xFFu : 0u;
            wi[j] = (wi[j] & ~(0xFFu << sh)) | (val << sh);
          }
        }
      }
    }
    dst[v] = make_uint4(wi[0], wi[1], wi[2], wi[3]);
  }
}

// shot_noise (corruptions.py:129-133): k ~ Poisson(x/255*c); y = uint8(clip(k/c, 0, 1)*255).
// lambda takes only 256 values (one per input byte) and every count k >= c saturates the output, so the whole sampler is a table:
// each workgroup builds, in fp64, the 32-bit cumulative thresholds floor(CDF_x(k) * 2^32) for k = 0 .. c-1 of all 256 levels
// (66 KB of LDS, rows padded to 65 words so that lanes on different levels hit different banks) and the 64 output bytes; a sample is
// one 32-bit uniform and a 6-step binary search -- exact inversion up to the 2^-32 quantisation of the CDF.  One Threefry call serves
// two elements.  (The first version walked the CDF / ran Hormann's PTRS rejection per element with byte loads: 594 us per 256 images.)
constexpr int kShotRow = 65;
template <bool VEC>
__global__ __launch_bounds__(kBlock) void k_shot_native(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                        uint32_t elems_per_sample, int c, uint32_t k0, uint32_t k1, uint32_t sample_base) {
  __shared__ uint32_t thr[256 * kShotRow];
  __shared__ uint8_t outv[64];
  {
    const int t = threadIdx.x;                          // kBlock == 256: one input level per thread
    const double lam = (double)t / 255.0 * (double)c;
    double p = exp(-lam), F = p;
    for (int k = 0; k < 64; ++k) {
      uint32_t v = 0xFFFFFFFFu;
      if (k < c) {
        const double q = F * 4294967296.0;
        v = q >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)q;
        p *= lam / (double)(k + 1);
        F += p;
      }
      thr[t * kShotRow + k] = v;
    }
    if (t < 64) outv[t] = (uint8_t)(fmin((double)t / (double)c, 1.0) * 255.0);
  }
  __syncthreads();
  const uint32_t sample = blockIdx.y;
  const uint8_t* src = in + (size_t)sample * elems_per_sample;
  uint8_t* dst = out + (size_t)sample * elems_per_sample;
  auto draw = [&](uint32_t xb, uint32_t u) -> uint32_t {
    if (xb == 0) return 0u;                              // lambda = 0
    const uint32_t* row = thr + xb * kShotRow;
    uint32_t pos = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1)
      if (row[pos + step - 1] <= u) pos += step;
    return outv[pos];                                    // pos = the count k (>= c: saturated)
  };
  if (VEC) {
    const uint32_t nvec = elems_per_sample / 16;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v < nvec; v += gridDim.x * kBlock) {
      const uint4 pk = s4[v];
      const uint32_t wi[4] = {pk.x, pk.y, pk.z, pk.w};
      // the 16 searches of a vector run in lock step (16 independent LDS reads per step: a dependent chain per element would leave
      // the wave waiting on one LDS round trip at a time)
      uint32_t u[16], base[16], pos[16];
#pragma unroll
      for (int q = 0; q < 8; ++q) {                       // element e = v*16 + 2q (+0, +1): pair index v*8 + q
        const uint2 w = threefry2x32(k0, k1, rart_ctr0(v * 8u + q, 0), sample_base + sample);
        u[2 * q] = w.x;
        u[2 * q + 1] = w.y;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        base[i] = ((wi[i >> 2] >> (8 * (i & 3))) & 0xFFu) * kShotRow;
        pos[i] = 0;
      }
#pragma unroll
      for (int step = 32; step >= 1; step >>= 1) {
        uint32_t t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = thr[base[i] + pos[i] + step - 1];
#pragma unroll
        for (int i = 0; i < 16; ++i) pos[i] += t[i] <= u[i] ? step : 0;
      }
      uint32_t wo[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 16; ++i) wo[i >> 2] |= (base[i] == 0 ? 0u : (uint32_t)outv[pos[i]]) << (8 * (i & 3));
      d4[v] = make_uint4(wo[0], wo[1], wo[2], wo[3]);
    }
  } else {
    for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < elems_per_sample; e += gridDim.x * kBlock) {
      const uint2 w = threefry2x32(k0, k1, rart_ctr0(e >> 1, 0), sample_base + sample);
      dst[e] = (uint8_t)draw(src[e], (e & 1u) ? w.y : w.x);
    }
  }
}

// per-image per-channel integer sums for contrast (corruptions.py:349 np.mean over H,W)
__global__ __launch_bounds__(kBlock) void k_channel_sums(const uint8_t* __restrict__ in,
                                                         unsigned long long* __restrict__ sums,
                                                         uint32_t pixels_per_sample) {
  const uint32_t sample = blockIdx.y;
  const uint8_t* src = in + (size_t)sample * pixels_per_sample * 3;
  uint32_t s0 = 0, s1 = 0, s2 = 0;
  const uint32_t nquad = pixels_per_sample / 4;  // 4 pixels = 12 bytes = 3 dwords
  const uint32_t* src32 = reinterpret_cast<const uint32_t*>(src);
  const bool aligned = ((pixels_per_sample * 3) % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 3) == 0);
  if (aligned) {
    for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < nquad; q += gridDim.x * kBlock) {
      const uint32_t a = src32[q * 3], b = src32[q * 3 + 1], c = src32[q * 3 + 2];
      // bytes: a = r0 g0 b0 r1 | b = g1 b1 r2 g2 | c = b2 r3 g3 b3
      s0 += (a & 0xFF) + (a >> 24) + ((b >> 16) & 0xFF) + ((c >> 8) & 0xFF);
      s1 += ((a >> 8) & 0xFF) + (b & 0xFF) + (b >> 24) + ((c >> 16) & 0xFF);
      s2 += ((a >> 16) & 0xFF) + ((b >> 8) & 0xFF) + (c & 0xFF) + (c >> 24);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      for (uint32_t p = nquad * 4; p < pixels_per_sample; ++p) {
        s0 += src[p * 3]; s1 += src[p * 3 + 1]; s2 += src[p * 3 + 2];
      }
    }
  } else {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < pixels_per_sample; p += gridDim.x * kBlock) {
      s0 += src[p * 3]; s1 += src[p * 3 + 1]; s2 += src[p * 3 + 2];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_xor(s0, off, 64);
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&sums[sample * 3 + 0], (unsigned long long)s0);
    atomicAdd(&sums[sample * 3 + 1], (unsigned long long)s1);
    atomicAdd(&sums[sample * 3 + 2], (unsigned long long)s2);
  }
}

}  // namespace

// =====================================================================================
// fp64 parity kernels: follow the reference's operation order, no FMA contraction
// =====================================================================================
#pragma clang fp contract(off)
namespace {

__device__ __forceinline__ uint8_t finish_unit(double v) {
  // np.clip(v, 0, 1) * 255 -> np.uint8 (truncation)
  v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
  return (uint8_t)(uint32_t)(v * 255.0);
}

// KIND 0 gaussian, 1 speckle; noise = what np.random.normal(scale=c) returned
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_normal_noise_injected(const uint8_t* __restrict__ in,
                                                                  uint8_t* __restrict__ out,
                                                                  const double* __restrict__ noise,
                                                                  size_t total) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const double x = (double)in[i] / 255.0;
    const double nz = noise[i];
    double v;
    if (KIND == 0) {
      v = x + nz;
    } else {
      const double xn = x * nz;
      v = x + xn;
    }
    out[i] = finish_unit(v);
  }
}

__global__ __launch_bounds__(kBlock) void k_shot_injected(uint8_t* __restrict__ out,
                                                          const int32_t* __restrict__ counts, double c,
                                                          size_t total) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    out[i] = finish_unit((double)counts[i] / c);
  }
}

__global__ __launch_bounds__(kBlock) void k_impulse_injected(const uint8_t* __restrict__ in,
                                                             uint8_t* __restrict__ out,
                                                             const uint8_t* __restrict__ code, size_t total) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const uint8_t cd = code[i];
    // x/255*255 truncated == x for every byte value (checked in tests), salt -> 255, pepper -> 0
    out[i] = cd == 1 ? 255 : (cd == 2 ? 0 : finish_unit((double)in[i] / 255.0));
  }
}

// contrast (corruptions.py:345-350): (x - m)*c + m with m the per-channel mean of x/255
__global__ __launch_bounds__(kBlock) void k_contrast_apply(const uint8_t* __restrict__ in,
                                                           uint8_t* __restrict__ out,
                                                           const unsigned long long* __restrict__ sums,
                                                           uint32_t pixels_per_sample, double c) {
  const uint32_t sample = blockIdx.y;
  const size_t base = (size_t)sample * pixels_per_sample * 3;
  const double npx = (double)pixels_per_sample;
  const double m0 = (double)sums[sample * 3 + 0] / 255.0 / npx;
  const double m1 = (double)sums[sample * 3 + 1] / 255.0 / npx;
  const double m2 = (double)sums[sample * 3 + 2] / 255.0 / npx;
  const uint32_t elems = pixels_per_sample * 3;
  for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < elems; e += gridDim.x * kBlock) {
    const uint32_t ch = e % 3;
    const double m = ch == 0 ? m0 : (ch == 1 ? m1 : m2);
    const double x = (double)in[base + e] / 255.0;
    const double d = x - m;
    const double dc = d * c;
    out[base + e] = finish_unit(dc + m);
  }
}

// contrast as ONE kernel, one workgroup per image with the image resident in LDS (150 528 B of the 160 KiB): HBM is touched
// once each way.  The result of a byte depends only on (channel, byte value, channel mean), so after the exact integer channel
// sums each workgroup evaluates the reference's fp64 expression 768 times into a lookup table (the same statements as
// k_contrast_apply: bit-identical by construction) and the 150 528 bytes go through the table -- no fp64 per element.
// Round 2's two-launch form (k_channel_sums with 64-bit atomics + a per-element fp64 kernel) took 178 us per 256-image batch.
constexpr int kCiThreads = 1024;
__global__ __launch_bounds__(kCiThreads) void k_contrast_image(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                               uint32_t pixels_per_sample, double c) {
  extern __shared__ __attribute__((aligned(16))) uint8_t ci_lds[];      // image bytes, then sums[3] (u32), then lut[768]
  const uint32_t elems = pixels_per_sample * 3;                         // multiple of 48 (host check): 16-byte vectors start at
  uint32_t* const ssum = reinterpret_cast<uint32_t*>(ci_lds + elems);   //   element 16 v, i.e. channel phase v % 3
  uint8_t* const lut = ci_lds + elems + 16;
  const uint4* src = reinterpret_cast<const uint4*>(in + (size_t)blockIdx.x * elems);
  uint4* img = reinterpret_cast<uint4*>(ci_lds);
  const uint32_t nvec = elems / 16;
  if (threadIdx.x < 3) ssum[threadIdx.x] = 0;
  // byte j of a vector belongs to class j % 3; per dword q the class of byte b is (4 q + b) % 3 = (q + b) % 3
  uint32_t t0 = 0, t1 = 0, t2 = 0;                                      // per-thread sums by (phase-corrected) channel
  for (uint32_t v = threadIdx.x; v < nvec; v += kCiThreads) {
    const uint4 p = src[v];
    img[v] = p;
    const uint32_t w[4] = {p.x, p.y, p.z, p.w};
    uint32_t a0 = 0, a1 = 0, a2 = 0;                                    // class sums of this vector
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // masks of the bytes of dword q whose class (q + b) % 3 is 0 / 1 / 2
      const uint32_t m0 = q == 0 ? 0xFF0000FFu : q == 1 ? 0x00FF0000u : q == 2 ? 0x0000FF00u : 0xFF0000FFu;
      const uint32_t m1 = q == 0 ? 0x0000FF00u : q == 1 ? 0xFF0000FFu : q == 2 ? 0x00FF0000u : 0x0000FF00u;
      const uint32_t m2 = q == 0 ? 0x00FF0000u : q == 1 ? 0x0000FF00u : q == 2 ? 0xFF0000FFu : 0x00FF0000u;
      a0 = __builtin_amdgcn_sad_u8(w[q] & m0, 0u, a0);
      a1 = __builtin_amdgcn_sad_u8(w[q] & m1, 0u, a1);
      a2 = __builtin_amdgcn_sad_u8(w[q] & m2, 0u, a2);
    }
    const uint32_t ph = v % 3;                                          // channel of the vector's first byte
    t0 += ph == 0 ? a0 : (ph == 1 ? a2 : a1);                           // channel 0 = class (0 - ph) mod 3
    t1 += ph == 0 ? a1 : (ph == 1 ? a0 : a2);
    t2 += ph == 0 ? a2 : (ph == 1 ? a1 : a0);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    t0 += __shfl_xor(t0, off, 64);
    t1 += __shfl_xor(t1, off, 64);
    t2 += __shfl_xor(t2, off, 64);
  }
  __syncthreads();                                                      // ssum zeroed, image in LDS
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&ssum[0], t0);                                            // integer sums: order does not matter
    atomicAdd(&ssum[1], t1);
    atomicAdd(&ssum[2], t2);
  }
  __syncthreads();
  if (threadIdx.x < 768) {
    const uint32_t ch = threadIdx.x >> 8, xb = threadIdx.x & 255;
    const double npx = (double)pixels_per_sample;
    const double m = (double)(unsigned long long)ssum[ch] / 255.0 / npx;
    const double x = (double)xb / 255.0;
    const double d = x - m;
    const double dc = d * c;
    lut[threadIdx.x] = finish_unit(dc + m);
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(out + (size_t)blockIdx.x * elems);
  for (uint32_t v = threadIdx.x; v < nvec; v += kCiThreads) {
    const uint4 p = img[v];
    const uint32_t w[4] = {p.x, p.y, p.z, p.w};
    uint32_t o[4];
    uint32_t chn = v % 3;                                               // channel of the next byte, advanced as we go
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t r = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        r |= (uint32_t)lut[chn * 256 + ((w[q] >> (8 * b)) & 0xFFu)] << (8 * b);
        chn = chn == 2 ? 0 : chn + 1;
      }
      o[q] = r;
    }
    dst[v] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// brightness / saturate (corruptions.py:353-372): skimage rgb2hsv -> edit V or S -> hsv2rgb, fp64.
// MODE 0: V = clip(V + p0, 0, 1);  MODE 1: S = clip(S*p0 + p1, 0, 1)
// One pixel: r, g, b already x / 255.0.  fmod(hh, 1.0) of the reference's `% 1.` is the identity here (|hh| <= 5/6 < 1, and
// fmod is exact), so only numpy's sign fix-up remains.
template <int MODE>
__device__ __forceinline__ void hsv_edit_pixel(double r, double g, double b, double p0, double p1, uint32_t& ro8, uint32_t& go8,
                                               uint32_t& bo8) {
  double v = fmax(fmax(r, g), b);
  const double mn = fmin(fmin(r, g), b);
  const double delta = v - mn;
  double s = 0.0, h = 0.0;
  if (delta != 0.0) {
    s = delta / v;
    double hh;
    if (b == v) {
      const double t = (r - g) / delta;
      hh = 4.0 + t;
    } else if (g == v) {
      const double t = (b - r) / delta;
      hh = 2.0 + t;
    } else {
      hh = (g - b) / delta;
    }
    hh = hh / 6.0;
    double m = hh;
    if (m != 0.0 && m < 0.0) m += 1.0;
    h = m;
  }
  if (MODE == 0) {
    v = v + p0;
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
  } else {
    const double sp = s * p0;
    s = sp + p1;
    s = s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s);
  }
  const double h6 = h * 6.0;
  const double hi = floor(h6);
  const double f = h6 - hi;
  const double one_s = 1.0 - s;
  const double p = v * one_s;
  const double fs = f * s;
  const double q = v * (1.0 - fs);
  const double omf = 1.0 - f;
  const double omfs = omf * s;
  const double t = v * (1.0 - omfs);
  const int sector = ((int)hi) % 6;
  double ro, go, bo;
  switch (sector) {
    case 0: ro = v; go = t; bo = p; break;
    case 1: ro = q; go = v; bo = p; break;
    case 2: ro = p; go = v; bo = t; break;
    case 3: ro = p; go = q; bo = v; break;
    case 4: ro = t; go = p; bo = v; break;
    default: ro = v; go = p; bo = q; break;
  }
  ro8 = finish_unit(ro);
  go8 = finish_unit(go);
  bo8 = finish_unit(bo);
}

// Four pixels (12 bytes = 3 dwords) per thread and step: coalesced dword traffic instead of six byte accesses per pixel, and
// x / 255.0 from a 256-entry fp64 table in LDS (the same division, done once per workgroup) instead of three fp64 divisions per
// pixel.  `vec` = the buffers are 4-byte aligned and pixels % 4 == 0 (224 x 224 is); otherwise one pixel per step.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_hsv_edit(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                     size_t pixels, double p0, double p1, int vec) {
  __shared__ double unit[256];
  unit[threadIdx.x] = (double)threadIdx.x / 255.0;
  __syncthreads();
  if (vec) {
    const uint32_t* in32 = reinterpret_cast<const uint32_t*>(in);
    uint32_t* out32 = reinterpret_cast<uint32_t*>(out);
    const size_t nq = pixels / 4;
    for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < nq; q += (size_t)gridDim.x * kBlock) {
      const uint32_t a = in32[q * 3], b = in32[q * 3 + 1], c = in32[q * 3 + 2];
      // bytes: a = r0 g0 b0 r1 | b = g1 b1 r2 g2 | c = b2 r3 g3 b3
      uint32_t r0, g0, b0, r1, g1, b1, r2, g2, b2, r3, g3, b3;
      hsv_edit_pixel<MODE>(unit[a & 0xFF], unit[(a >> 8) & 0xFF], unit[(a >> 16) & 0xFF], p0, p1, r0, g0, b0);
      hsv_edit_pixel<MODE>(unit[a >> 24], unit[b & 0xFF], unit[(b >> 8) & 0xFF], p0, p1, r1, g1, b1);
      hsv_edit_pixel<MODE>(unit[(b >> 16) & 0xFF], unit[b >> 24], unit[c & 0xFF], p0, p1, r2, g2, b2);
      hsv_edit_pixel<MODE>(unit[(c >> 8) & 0xFF], unit[(c >> 16) & 0xFF], unit[c >> 24], p0, p1, r3, g3, b3);
      out32[q * 3] = r0 | (g0 << 8) | (b0 << 16) | (r1 << 24);
      out32[q * 3 + 1] = g1 | (b1 << 8) | (r2 << 16) | (g2 << 24);
      out32[q * 3 + 2] = b2 | (r3 << 8) | (g3 << 16) | (b3 << 24);
    }
    return;
  }
  for (size_t px = (size_t)blockIdx.x * kBlock + threadIdx.x; px < pixels; px += (size_t)gridDim.x * kBlock) {
    uint32_t ro, go, bo;
    hsv_edit_pixel<MODE>(unit[in[px * 3]], unit[in[px * 3 + 1]], unit[in[px * 3 + 2]], p0, p1, ro, go, bo);
    out[px * 3] = (uint8_t)ro;
    out[px * 3 + 1] = (uint8_t)go;
    out[px * 3 + 2] = (uint8_t)bo;
  }
}

// frost (corruptions.py:262): clip(a*x + b*texture, 0, 255) -> uint8
__global__ __launch_bounds__(kBlock) void k_frost_blend(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                        const uint8_t* __restrict__ tex, double a, double b,
                                                        size_t total) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const double ax = a * (double)in[i];
    const double bt = b * (double)tex[i];
    double v = ax + bt;
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    out[i] = (uint8_t)(uint32_t)v;
  }
}


// frost with the photographs resident on the device (round 5): the crop of image i is drawn AND read here -- texture index
// floor(u_8 * min(5, k)) (corruptions.py:250: randint(5) over the six-entry list), crop origin floor(u_9 * (height - 224)),
// floor(u_10 * (width - 224)) (corruptions.py:259), u_s = the counter generator's uniform of stream s at element 0 of the sample --
// the draws robustart_amd/noise/imagenet_c.py made on the host (rng.host_uniform_many) before gathering the n crops with a torch
// index expression (211 us + nine small launches + an upload per 256 images; the blend itself is 50 us).  grid (37, n): a thread blends 16 bytes.
struct FrostTex {
  int k_draw, sh, sw;
  int th[8], tw[8];
};
__global__ __launch_bounds__(kBlock) void k_frost_textures(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           const uint8_t* __restrict__ stack, FrostTex t, double a, double b,
                                                           uint32_t k0, uint32_t k1, uint32_t sample_base) {
  const uint32_t sample = sample_base + blockIdx.y;
  auto u53 = [&](int stream) {
    const uint2 w = threefry2x32(k0, k1, rart_ctr0(0u, stream), sample);
    return ((double)(w.x >> 5) * 67108864.0 + (double)(w.y >> 6)) / 9007199254740992.0;
  };
  const int idx = (int)(u53(8) * (double)t.k_draw);
  const int xs = (int)(u53(9) * (double)(t.th[idx] - 224)), ys = (int)(u53(10) * (double)(t.tw[idx] - 224));
  const int q = blockIdx.x * kBlock + threadIdx.x;                   // 16-byte piece of the image: 42 per row
  if (q >= 224 * 42) return;
  const int row = q / 42, cb = (q - row * 42) * 16;
  const size_t o = (size_t)blockIdx.y * (224 * 672) + (size_t)q * 16;
  const uint8_t* tp = stack + (((size_t)idx * t.sh + (xs + row)) * t.sw + ys) * 3 + cb;
  const uint4 pin4 = *(const uint4*)(in + o);
  const uint32_t pin[4] = {pin4.x, pin4.y, pin4.z, pin4.w};
  uint32_t r[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double ax = a * (double)((pin[j >> 2] >> (8 * (j & 3))) & 255u);
    const double bt = b * (double)tp[j];
    double v = ax + bt;
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    r[j >> 2] |= (uint32_t)v << (8 * (j & 3));
  }
  *(uint4*)(out + o) = make_uint4(r[0], r[1], r[2], r[3]);
}

}  // namespace
#pragma clang fp contract(fast)

// =====================================================================================
// host launchers
// =====================================================================================

// One item per thread when the launch stays below ~16k workgroups (no grid-stride tail
// imbalance: 9408 vectors per 224x224 image = 36.75 x 256), grid-strided beyond that.
// 1 = matrix-core CLT generator for gaussian/speckle noise when a sample is a whole number of 1 KiB chunks
// (default), 0 = Threefry + Box-Muller everywhere.  The generator is part of the (seed -> field) definition.
static int g_normal_generator = 1;
// ---- calibration (bench.py's hbm_roofline_gaussian_noise block): a PLAIN copy of n bytes in k_normal_noise_mfma's own geometry -- one wave
//      per 1 KiB chunk, 16 bytes per lane (variant 0) -- and as a grid-stride loop over 1 024 workgroups (variant 1, the fastest plain copy
//      of 77 MB measured on this part: scratch/r6/copy_probe.hip).  What a kernel that does NOTHING but move the launch's bytes reaches at
//      this size is the ceiling the noise kernel is read against.
namespace {
__global__ __launch_bounds__(256) void k_copy_calib_chunk(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_copy_calib_stride(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
}  // namespace
extern "C" int rart_copy_calibration(const void* in, void* out, size_t bytes, int variant, rart_stream_t stream) {
  RART_CHECK_ARG(in && out && bytes % 16 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0,
                 "rart_copy_calibration: 16-byte aligned buffers of a multiple of 16 bytes");
  RART_CHECK_ARG(variant == 0 || variant == 1, "rart_copy_calibration: variant must be 0 (one vector per lane) or 1 (grid-stride)");
  const size_t n16 = bytes / 16;
  if (n16 == 0) return RART_OK;
  RART_CHECK_ARG((n16 + 255) / 256 < (1ull << 31), "rart_copy_calibration: buffer too large");
  if (variant == 0) hipLaunchKernelGGL(k_copy_calib_chunk, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, (uint4*)out, n16);
  else hipLaunchKernelGGL(k_copy_calib_stride, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const uint4*)in, (uint4*)out, n16);
  RART_CHECK_LAUNCH("rart_copy_calibration");
  return RART_OK;
}

extern "C" int rart_set_normal_generator(int kind) {
  if (kind != 0 && kind != 1) {
    rart_set_error("rart_set_normal_generator: kind must be 0 (Box-Muller) or 1 (matrix-core CLT)");
    return RART_ERR_INVALID;
  }
  g_normal_generator = kind;
  return RART_OK;
}
extern "C" int rart_get_normal_generator(void) { return g_normal_generator; }

static dim3 grid2d(uint32_t items_per_sample, int n) {
  uint32_t gx = (items_per_sample + kBlock - 1) / kBlock;
  uint32_t cap = (uint32_t)(16384 / (n < 1 ? 1 : n));
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return dim3(gx, (uint32_t)n, 1);
}

size_t rart_ws_pointwise(int corruption_id, int /*severity*/, int n, int /*h*/, int /*w*/) {
  if (corruption_id == RART_CONTRAST) return rart_align_up((size_t)n * 3 * sizeof(unsigned long long), 256);
  return 0;
}

int rart_launch_pointwise(int id, const RartCorruptArgs& a) {
  const int s = a.severity - 1;
  const size_t pixels = (size_t)a.h * a.w;
  const size_t eps = pixels * 3;  // elements per sample
  const size_t total = eps * a.n;
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  const uint32_t sbase = (uint32_t)a.sample_offset;
  const bool vec_ok = (eps % 16 == 0) && ((reinterpret_cast<uintptr_t>(a.in) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0);
  RART_CHECK_ARG(eps < (1u << 28) * 4ull, "image too large for the counter layout (h*w*3 must be < 2^30)");
  const void* inj0 = (a.injected && a.n_injected > 0) ? a.injected[0] : nullptr;

  switch (id) {
    case RART_GAUSSIAN_NOISE:
    case RART_SPECKLE_NOISE: {
      const double c = id == RART_GAUSSIAN_NOISE ? RartSeverity::gaussian_noise[s] : RartSeverity::speckle_noise[s];
      if (inj0) {
        const int g = rart_grid_for(total);
        if (id == RART_GAUSSIAN_NOISE)
          hipLaunchKernelGGL(k_normal_noise_injected<0>, dim3(g), dim3(kBlock), 0, a.stream, a.in, a.out,
                             (const double*)inj0, total);
        else
          hipLaunchKernelGGL(k_normal_noise_injected<1>, dim3(g), dim3(kBlock), 0, a.stream, a.in, a.out,
                             (const double*)inj0, total);
      } else if (vec_ok && g_normal_generator == 1 && eps % 1024 == 0) {
        const uint32_t cps = (uint32_t)(eps / 1024), total = cps * (uint32_t)a.n;
        RART_CHECK_ARG((unsigned long long)cps * (unsigned long long)a.n < (1ull << 31), "noise: too many 1 KiB chunks per call");
        uint32_t lg = 0;                                     // exact g / cps for g < 2^31: q = (g * magic) >> (31 + ceil(log2 cps))
        while ((1ull << lg) < cps) ++lg;
        const uint32_t cshift = 31 + lg, cmagic = (uint32_t)(((1ull << cshift) + cps - 1) / cps);
        const dim3 g((total + kBlock / 64 - 1) / (kBlock / 64));
        if (id == RART_GAUSSIAN_NOISE)
          hipLaunchKernelGGL(k_normal_noise_mfma<0>, g, dim3(kBlock), 0, a.stream, (const uint4*)a.in, (uint4*)a.out,
                             cps, total, (float)c, k0, k1, sbase, cmagic, cshift);
        else
          hipLaunchKernelGGL(k_normal_noise_mfma<1>, g, dim3(kBlock), 0, a.stream, (const uint4*)a.in, (uint4*)a.out,
                             cps, total, (float)c, k0, k1, sbase, cmagic, cshift);
      } else if (vec_ok) {
        const uint32_t vps = (uint32_t)(eps / 16);
        const dim3 g = grid2d(vps, a.n);
        if (id == RART_GAUSSIAN_NOISE)
          hipLaunchKernelGGL(k_normal_noise_native<0>, g, dim3(kBlock), 0, a.stream, (const uint4*)a.in,
                             (uint4*)a.out, vps, (float)c, k0, k1, sbase);
        else
          hipLaunchKernelGGL(k_normal_noise_native<1>, g, dim3(kBlock), 0, a.stream, (const uint4*)a.in,
                             (uint4*)a.out, vps, (float)c, k0, k1, sbase);
      } else {
        const dim3 g = grid2d((uint32_t)((eps + 3) / 4), a.n);
        if (id == RART_GAUSSIAN_NOISE)
          hipLaunchKernelGGL(k_normal_noise_native_scalar<0>, g, dim3(kBlock), 0, a.stream, a.in, a.out,
                             (uint32_t)eps, 0u, (float)c, k0, k1, sbase);
        else
          hipLaunchKernelGGL(k_normal_noise_native_scalar<1>, g, dim3(kBlock), 0, a.stream, a.in, a.out,
                             (uint32_t)eps, 0u, (float)c, k0, k1, sbase);
      }
      break;
    }
    case RART_SHOT_NOISE: {
      const double c = RartSeverity::shot_noise[s];
      if (inj0) {
        hipLaunchKernelGGL(k_shot_injected, dim3(rart_grid_for(total)), dim3(kBlock), 0, a.stream, a.out,
                           (const int32_t*)inj0, c, total);
      } else {
        // (every severity's c is a whole number <= 60; the table has 64 columns)
        RART_CHECK_ARG(c == (double)(int)c && c >= 1 && c <= 63, "shot_noise: the count scale must be a whole number in 1..63");
        // few, long-lived workgroups: each builds the 66 KB table once
        const uint32_t per = vec_ok ? (uint32_t)(eps / 16) : (uint32_t)eps;
        uint32_t gx = (per + kBlock - 1) / kBlock, cap = (uint32_t)(2048 / (a.n < 1 ? 1 : a.n));
        gx = gx > (cap < 1 ? 1 : cap) ? (cap < 1 ? 1 : cap) : gx;
        if (vec_ok) hipLaunchKernelGGL(k_shot_native<true>, dim3(gx, a.n), dim3(kBlock), 0, a.stream, a.in, a.out, (uint32_t)eps, (int)c, k0, k1, sbase);
        else hipLaunchKernelGGL(k_shot_native<false>, dim3(gx, a.n), dim3(kBlock), 0, a.stream, a.in, a.out, (uint32_t)eps, (int)c, k0, k1, sbase);
      }
      break;
    }
    case RART_IMPULSE_NOISE: {
      if (inj0) {
        hipLaunchKernelGGL(k_impulse_injected, dim3(rart_grid_for(total)), dim3(kBlock), 0, a.stream, a.in,
                           a.out, (const uint8_t*)inj0, total);
      } else {
        RART_CHECK_ARG(vec_ok, "impulse_noise native path needs h*w*3 %% 16 == 0 and 16-byte aligned buffers");
        const uint32_t thresh = (uint32_t)(RartSeverity::impulse_noise[s] * 16777216.0);
        const uint32_t vps = (uint32_t)(eps / 16);
        hipLaunchKernelGGL(k_impulse_native, grid2d(vps, a.n), dim3(kBlock), 0, a.stream, (const uint4*)a.in,
                           (uint4*)a.out, vps, thresh, k0, k1, sbase);
      }
      break;
    }
    case RART_CONTRAST: {
      const size_t lds = eps + 16 + 768;
      if (vec_ok && eps % 48 == 0 && lds <= 160 * 1024 && pixels < (1u << 24)) {
        // one kernel, image resident in LDS; 255 * pixels < 2^32 keeps the channel sums in 32 bits
        if (!rart_raise_dynamic_lds((const void*)k_contrast_image, lds, "contrast")) return RART_ERR_HIP;
        hipLaunchKernelGGL(k_contrast_image, dim3(a.n), dim3(kCiThreads), lds, a.stream, a.in, a.out, (uint32_t)pixels,
                           RartSeverity::contrast[s]);
        break;
      }
      const size_t need = rart_ws_pointwise(id, a.severity, a.n, a.h, a.w);
      if (!a.workspace || a.workspace_bytes < need) {
        rart_set_error("contrast: workspace of %zu bytes required", need);
        return RART_ERR_WORKSPACE;
      }
      unsigned long long* sums = (unsigned long long*)a.workspace;
      if (hipMemsetAsync(sums, 0, (size_t)a.n * 3 * sizeof(unsigned long long), a.stream) != hipSuccess) {
        rart_set_error("contrast: hipMemsetAsync failed");
        return RART_ERR_HIP;
      }
      hipLaunchKernelGGL(k_channel_sums, grid2d((uint32_t)(pixels / 4 + 1), a.n), dim3(kBlock), 0, a.stream,
                         a.in, sums, (uint32_t)pixels);
      hipLaunchKernelGGL(k_contrast_apply, grid2d((uint32_t)eps, a.n), dim3(kBlock), 0, a.stream, a.in, a.out,
                         sums, (uint32_t)pixels, RartSeverity::contrast[s]);
      break;
    }
    case RART_BRIGHTNESS: {
      const int hv = ((pixels * a.n) % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 3) == 0) ? 1 : 0;
      hipLaunchKernelGGL(k_hsv_edit<0>, dim3(rart_grid_for(pixels * a.n / (hv ? 4 : 1), kBlock, 256 * 16)), dim3(kBlock), 0, a.stream, a.in,
                         a.out, pixels * a.n, RartSeverity::brightness[s], 0.0, hv);
      break;
    }
    case RART_SATURATE: {
      static const double sat[5][2] = {{0.3, 0}, {0.1, 0}, {2, 0}, {5, 0.1}, {20, 0.2}};
      const int hv = ((pixels * a.n) % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 3) == 0) ? 1 : 0;
      hipLaunchKernelGGL(k_hsv_edit<1>, dim3(rart_grid_for(pixels * a.n / (hv ? 4 : 1), kBlock, 256 * 16)), dim3(kBlock), 0, a.stream, a.in,
                         a.out, pixels * a.n, sat[s][0], sat[s][1], hv);
      break;
    }
    case RART_FROST: {
      RART_CHECK_ARG(a.h == 224 && a.w == 224, "frost: reference hard-codes 224x224 (corruptions.py:259-260)");
      RART_CHECK_ARG(inj0 != nullptr,
                     "frost: injected[0] (uint8 texture crops [n][224][224][3]) is required: the reference's "
                     "frost photos are not part of its repository");
      static const double fr[5][2] = {{1, 0.4}, {0.8, 0.6}, {0.7, 0.7}, {0.65, 0.7}, {0.6, 0.75}};
      hipLaunchKernelGGL(k_frost_blend, dim3(rart_grid_for(total)), dim3(kBlock), 0, a.stream, a.in, a.out,
                         (const uint8_t*)inj0, fr[s][0], fr[s][1], total);
      break;
    }
    default:
      rart_set_error("rart_launch_pointwise: corruption id %d is not pointwise", id);
      return RART_ERR_INVALID;
  }
  RART_CHECK_LAUNCH("pointwise corruption launch");
  return RART_OK;
}

// The standard-normal field gaussian_noise / speckle_noise draw for `elems` elements per sample, replayed
// for the parity tests (same generator selection as the corruption kernels).
extern "C" int rart_rng_noise_field_f32(float* out, int n_samples, size_t elems, uint64_t seed, uint64_t sample_offset,
                                        rart_stream_t stream) {
  RART_CHECK_ARG(out && n_samples > 0 && elems > 0 && elems < (1ull << 30), "rart_rng_noise_field_f32: bad arguments");
  if (g_normal_generator == 1 && elems % 1024 == 0) {
    const uint32_t vps = (uint32_t)(elems / 16);
    hipLaunchKernelGGL(k_clt_field, dim3((vps + kBlock - 1) / kBlock, n_samples), dim3(kBlock), 0,
                       (hipStream_t)stream, out, vps, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)sample_offset);
    RART_CHECK_LAUNCH("rart_rng_noise_field_f32");
    return RART_OK;
  }
  return rart_rng_normal_f32(out, n_samples, elems, seed, sample_offset, 0, stream);
}


// gaussian_noise / speckle_noise at `ns` (severity, seed) pairs of ONE source batch in one launch (k_normal_noise_mfma_multi):
// outs[i] is bit-identical to rart_corrupt_u8(in, outs[i], ..., corruption_id, severities[i], seeds[i], sample_offset) with the native
// matrix-core generator.  Needs h*w*3 a multiple of 1024 and 16-byte aligned buffers (no other path: callers fall back to ns launches).
extern "C" int rart_noise_multi_u8(const uint8_t* in, uint8_t* const* outs, int ns, int n, int h, int w, int corruption_id,
                                   const int* severities, const uint64_t* seeds, uint64_t sample_offset, rart_stream_t stream) {
  RART_CHECK_ARG(in && outs && severities && seeds && ns >= 1 && ns <= 8 && n > 0 && h > 0 && w > 0, "rart_noise_multi_u8: bad arguments (1..8 outputs)");
  RART_CHECK_ARG(corruption_id == RART_GAUSSIAN_NOISE || corruption_id == RART_SPECKLE_NOISE, "rart_noise_multi_u8: gaussian_noise / speckle_noise only");
  const size_t eps = (size_t)h * w * 3;
  if (g_normal_generator != 1 || eps % 1024 != 0 || (reinterpret_cast<uintptr_t>(in) & 15)) {
    rart_set_error("rart_noise_multi_u8: needs the matrix-core generator, h*w*3 %% 1024 == 0 and 16-byte aligned buffers");
    return RART_ERR_UNSUPPORTED;
  }
  NoiseMultiArgs m;
  m.ns = ns;
  for (int i = 0; i < 8; ++i) { m.out[i] = nullptr; m.c[i] = 0.f; m.k0[i] = m.k1[i] = 0; }
  for (int i = 0; i < ns; ++i) {
    RART_CHECK_ARG(outs[i] && !(reinterpret_cast<uintptr_t>(outs[i]) & 15) && outs[i] != in, "rart_noise_multi_u8: outputs must be 16-byte aligned and distinct from the input");
    RART_CHECK_ARG(severities[i] >= 1 && severities[i] <= 5, "rart_noise_multi_u8: severity must be 1..5");
    m.out[i] = (uint4*)outs[i];
    m.c[i] = (float)(corruption_id == RART_GAUSSIAN_NOISE ? RartSeverity::gaussian_noise[severities[i] - 1] : RartSeverity::speckle_noise[severities[i] - 1]);
    m.k0[i] = (uint32_t)seeds[i];
    m.k1[i] = (uint32_t)(seeds[i] >> 32);
  }
  const uint32_t cps = (uint32_t)(eps / 1024), total = cps * (uint32_t)n;
  RART_CHECK_ARG((unsigned long long)cps * (unsigned long long)n < (1ull << 31), "rart_noise_multi_u8: too many 1 KiB chunks per call");
  uint32_t lg = 0;
  while ((1ull << lg) < cps) ++lg;
  const uint32_t cshift = 31 + lg, cmagic = (uint32_t)(((1ull << cshift) + cps - 1) / cps);
  const dim3 g((total + kBlock / 64 - 1) / (kBlock / 64));
  if (corruption_id == RART_GAUSSIAN_NOISE)
    hipLaunchKernelGGL(k_normal_noise_mfma_multi<0>, g, dim3(kBlock), 0, (hipStream_t)stream, (const uint4*)in, m, cps, total,
                       (uint32_t)sample_offset, cmagic, cshift);
  else
    hipLaunchKernelGGL(k_normal_noise_mfma_multi<1>, g, dim3(kBlock), 0, (hipStream_t)stream, (const uint4*)in, m, cps, total,
                       (uint32_t)sample_offset, cmagic, cshift);
  RART_CHECK_LAUNCH("rart_noise_multi_u8");
  return RART_OK;
}

// frost from device-resident photographs: stack = uint8 [k_tex][sh][sw][3] (every photograph padded to a common sh x sw), dims_host = host int[2 k_tex]
// (height, width of each); the per-image texture index and crop origin are the counter generator's (streams 8, 9, 10).  Bit-identical to
// rart_corrupt_u8(RART_FROST) with those crops injected.
extern "C" int rart_frost_textures_u8(const uint8_t* in, uint8_t* out, int n, int h, int w, int severity, const uint8_t* stack, int k_tex,
                                      int sh, int sw, const int* dims_host, uint64_t seed, uint64_t sample_offset, rart_stream_t stream) {
  RART_CHECK_ARG(in && out && stack && dims_host && n > 0 && n <= 65535, "rart_frost_textures_u8: bad arguments");
  RART_CHECK_ARG(h == 224 && w == 224, "frost: reference hard-codes 224x224 (corruptions.py:259-260)");
  RART_CHECK_ARG(severity >= 1 && severity <= 5, "rart_frost_textures_u8: severity must be 1..5");
  RART_CHECK_ARG(k_tex >= 1 && k_tex <= 8, "rart_frost_textures_u8: 1..8 photographs");
  RART_CHECK_ARG(!(reinterpret_cast<uintptr_t>(in) & 15) && !(reinterpret_cast<uintptr_t>(out) & 15), "rart_frost_textures_u8: 16-byte aligned batches");
  FrostTex t;
  t.k_draw = k_tex < 5 ? k_tex : 5;
  t.sh = sh;
  t.sw = sw;
  for (int i = 0; i < 8; ++i) {
    t.th[i] = i < k_tex ? dims_host[2 * i] : 224;
    t.tw[i] = i < k_tex ? dims_host[2 * i + 1] : 224;
    RART_CHECK_ARG(t.th[i] >= 224 && t.tw[i] >= 224 && (i >= k_tex || (t.th[i] <= sh && t.tw[i] <= sw)),
                   "rart_frost_textures_u8: every photograph must be at least 224 x 224 and fit the stack");
  }
  static const double fr[5][2] = {{1, 0.4}, {0.8, 0.6}, {0.7, 0.7}, {0.65, 0.7}, {0.6, 0.75}};
  hipLaunchKernelGGL(k_frost_textures, dim3((224 * 42 + kBlock - 1) / kBlock, n), dim3(kBlock), 0, (hipStream_t)stream, in, out, stack, t, fr[severity - 1][0],
                     fr[severity - 1][1], (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)sample_offset);
  RART_CHECK_LAUNCH("rart_frost_textures_u8");
  return RART_OK;
}
