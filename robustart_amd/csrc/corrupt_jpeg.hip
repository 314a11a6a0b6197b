// jpeg_compression for gfx950: the Pillow save(JPEG, quality) -> open round trip restated as an
// all-integer kernel (baseline, 4:2:0, libjpeg ISLOW FDCT/IDCT, fancy h2v2 upsampling), bit-exact.
// Reference: RobustART/noise/utils/imagenet_c/corruptions.py:375-382; arithmetic: SURVEY.md
// Appendix A.2 (the Huffman stage is lossless and skipped).
//
// One workgroup per image: Y (h*w bytes) and the two half-resolution chroma planes live in LDS
// (224x224: 75 KB), so the three phases -- colour convert + downsample, per-8x8-block codec,
// upsample + colour convert -- touch HBM exactly once each way (u8 in, u8 out).
#include "rart_common.h"

namespace {
constexpr int kThreads = 512;

constexpr int FIXI(double x) { return (int)(x * 65536.0 + 0.5); }

__constant__ int c_std_luma[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                   14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                   18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                   49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
__constant__ int c_std_chroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                     24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

#define F_0_298 2446
#define F_0_390 3196
#define F_0_541 4433
#define F_0_765 6270
#define F_0_899 7373
#define F_1_175 9633
#define F_1_501 12299
#define F_1_847 15137
#define F_1_961 16069
#define F_2_053 16819
#define F_2_562 20995
#define F_3_072 25172

__device__ __forceinline__ int ds(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jfdctint.c, one 1-D pass over d[0..7] (stride s)
template <bool PASS1>
__device__ __forceinline__ void fdct8(int* d, int s) {
  const int d0 = d[0], d1 = d[s], d2 = d[2 * s], d3 = d[3 * s], d4 = d[4 * s], d5 = d[5 * s], d6 = d[6 * s],
            d7 = d[7 * s];
  int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6, t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4,
      t4 = d3 - d4;
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  constexpr int sh = PASS1 ? 11 : 15;
  if (PASS1) {
    d[0] = (t10 + t11) << 2;
    d[4 * s] = (t10 - t11) << 2;
  } else {
    d[0] = ds(t10 + t11, 2);
    d[4 * s] = ds(t10 - t11, 2);
  }
  int z1 = (t12 + t13) * F_0_541;
  d[2 * s] = ds(z1 + t13 * F_0_765, sh);
  d[6 * s] = ds(z1 - t12 * F_1_847, sh);
  z1 = t4 + t7;
  int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const int z5 = (z3 + z4) * F_1_175;
  t4 *= F_0_298; t5 *= F_2_053; t6 *= F_3_072; t7 *= F_1_501;
  z1 *= -F_0_899; z2 *= -F_2_562;
  z3 = z3 * -F_1_961 + z5;
  z4 = z4 * -F_0_390 + z5;
  d[7 * s] = ds(t4 + z1 + z3, sh);
  d[5 * s] = ds(t5 + z2 + z4, sh);
  d[3 * s] = ds(t6 + z2 + z3, sh);
  d[s] = ds(t7 + z1 + z4, sh);
}

// jidctint.c, one 1-D pass, descale by SH
template <int SH>
__device__ __forceinline__ void idct8(int* x, int s) {
  const int x0 = x[0], x1 = x[s], x2 = x[2 * s], x3 = x[3 * s], x4 = x[4 * s], x5 = x[5 * s], x6 = x[6 * s],
            x7 = x[7 * s];
  int z1 = (x2 + x6) * F_0_541;
  const int t2 = z1 - x6 * F_1_847, t3 = z1 + x2 * F_0_765;
  const int t0 = (x0 + x4) << 13, t1 = (x0 - x4) << 13;
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  int a0 = x7, a1 = x5, a2 = x3, a3 = x1;
  z1 = a0 + a3;
  int z2 = a1 + a2, z3 = a0 + a2, z4 = a1 + a3;
  const int z5 = (z3 + z4) * F_1_175;
  a0 *= F_0_298; a1 *= F_2_053; a2 *= F_3_072; a3 *= F_1_501;
  z1 *= -F_0_899; z2 *= -F_2_562;
  z3 = z3 * -F_1_961 + z5;
  z4 = z4 * -F_0_390 + z5;
  a0 += z1 + z3; a1 += z2 + z4; a2 += z2 + z3; a3 += z1 + z4;
  x[0] = ds(t10 + a3, SH); x[7 * s] = ds(t10 - a3, SH);
  x[s] = ds(t11 + a2, SH); x[6 * s] = ds(t11 - a2, SH);
  x[2 * s] = ds(t12 + a1, SH); x[5 * s] = ds(t12 - a1, SH);
  x[3 * s] = ds(t13 + a0, SH); x[4 * s] = ds(t13 - a0, SH);
}

// FDCT -> quantise -> dequantise -> IDCT of the 8x8 block at (by, bx) of an LDS plane, in place
__device__ __forceinline__ void codec_block(uint8_t* plane, int stride, int by, int bx, const int* qt) {
  int b[64];
  uint8_t* p = plane + (size_t)by * 8 * stride + bx * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint2 v = *reinterpret_cast<const uint2*>(p + r * stride);  // rows are 8-byte aligned
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      b[r * 8 + c] = (int)((v.x >> (8 * c)) & 0xFF) - 128;
      b[r * 8 + 4 + c] = (int)((v.y >> (8 * c)) & 0xFF) - 128;
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) fdct8<true>(b + r * 8, 1);
#pragma unroll
  for (int c = 0; c < 8; ++c) fdct8<false>(b + c, 8);
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int t = qt[i];
    const int d = t << 3;
    const int v = b[i];
    const int av = v < 0 ? -v : v;
    int q = (av + (d >> 1)) / d;
    q = v < 0 ? -q : q;
    b[i] = q * t;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) idct8<11>(b + c, 8);
#pragma unroll
  for (int r = 0; r < 8; ++r) idct8<18>(b + r * 8, 1);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int v0 = b[r * 8 + c] + 128, v1 = b[r * 8 + 4 + c] + 128;
      v0 = v0 < 0 ? 0 : (v0 > 255 ? 255 : v0);
      v1 = v1 < 0 ? 0 : (v1 > 255 ? 255 : v1);
      lo |= (uint32_t)v0 << (8 * c);
      hi |= (uint32_t)v1 << (8 * c);
    }
    *reinterpret_cast<uint2*>(p + r * stride) = make_uint2(lo, hi);
  }
}

__global__ __launch_bounds__(kThreads) void k_jpeg(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                   int h, int w, int quality_scale) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int cw = w / 2, chh = h / 2;
  uint8_t* Y = smem;
  uint8_t* Cb = Y + (size_t)h * w;
  uint8_t* Cr = Cb + (size_t)chh * cw;
  int* qy = reinterpret_cast<int*>(Cr + (size_t)chh * cw);
  int* qc = qy + 64;
  const uint8_t* src = in + (size_t)blockIdx.x * h * w * 3;
  uint8_t* dst = out + (size_t)blockIdx.x * h * w * 3;
  if (threadIdx.x < 128) {
    const int i = threadIdx.x & 63;
    const int base = threadIdx.x < 64 ? c_std_luma[i] : c_std_chroma[i];
    int t = (base * quality_scale + 50) / 100;
    t = t < 1 ? 1 : (t > 255 ? 255 : t);
    (threadIdx.x < 64 ? qy : qc)[i] = t;
  }
  // phase A: RGB -> YCbCr (16-bit fixed point), h2v2 chroma downsample with alternating bias
  constexpr int H = 32768;
  for (int q = threadIdx.x; q < chh * cw; q += kThreads) {
    const int cy = q / cw, cx = q % cw;
    int sb = 0, sr = 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const uint8_t* p = src + ((size_t)(2 * cy + dy) * w + 2 * cx) * 3;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int r = p[dx * 3], g = p[dx * 3 + 1], b = p[dx * 3 + 2];
        Y[(2 * cy + dy) * w + 2 * cx + dx] =
            (uint8_t)((FIXI(0.29900) * r + FIXI(0.58700) * g + FIXI(0.11400) * b + H) >> 16);
        sb += (-FIXI(0.16874) * r - FIXI(0.33126) * g + FIXI(0.50000) * b + (128 << 16) + H - 1) >> 16;
        sr += (FIXI(0.50000) * r - FIXI(0.41869) * g - FIXI(0.08131) * b + (128 << 16) + H - 1) >> 16;
      }
    }
    const int bias = (cx & 1) ? 2 : 1;
    Cb[cy * cw + cx] = (uint8_t)((sb + bias) >> 2);
    Cr[cy * cw + cx] = (uint8_t)((sr + bias) >> 2);
  }
  __syncthreads();
  // phase B: per-block codec, in place in LDS
  const int nby = h / 8, nbx = w / 8, ncy = chh / 8, ncx = cw / 8;
  const int nblk = nby * nbx + 2 * ncy * ncx;
  for (int b = threadIdx.x; b < nblk; b += kThreads) {
    if (b < nby * nbx) {
      codec_block(Y, w, b / nbx, b % nbx, qy);
    } else {
      int c = b - nby * nbx;
      uint8_t* plane = Cb;
      if (c >= ncy * ncx) {
        c -= ncy * ncx;
        plane = Cr;
      }
      codec_block(plane, cw, c / ncx, c % ncx, qc);
    }
  }
  __syncthreads();
  // phase C: fancy h2v2 upsample (triangle filter) + YCbCr -> RGB, one chroma sample (2x2 pixels) per thread
  for (int q = threadIdx.x; q < chh * cw; q += kThreads) {
    const int cy = q / cw, cx = q % cw;
    const int ya = cy > 0 ? cy - 1 : 0, yb = cy < chh - 1 ? cy + 1 : chh - 1;
    const int xl = cx > 0 ? cx - 1 : 0, xr = cx < cw - 1 ? cx + 1 : cw - 1;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int yn = dy == 0 ? ya : yb;  // far row: above for the upper output row, below for the lower
      int cs_b[3], cs_r[3];
      const int xs[3] = {xl, cx, xr};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        cs_b[k] = 3 * (int)Cb[cy * cw + xs[k]] + (int)Cb[yn * cw + xs[k]];
        cs_r[k] = 3 * (int)Cr[cy * cw + xs[k]] + (int)Cr[yn * cw + xs[k]];
      }
      uint8_t* o = dst + ((size_t)(2 * cy + dy) * w + 2 * cx) * 3;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        int vb, vr;
        if (dx == 0) {
          vb = cx == 0 ? (4 * cs_b[1] + 8) >> 4 : (3 * cs_b[1] + cs_b[0] + 8) >> 4;
          vr = cx == 0 ? (4 * cs_r[1] + 8) >> 4 : (3 * cs_r[1] + cs_r[0] + 8) >> 4;
        } else {
          vb = cx == cw - 1 ? (4 * cs_b[1] + 7) >> 4 : (3 * cs_b[1] + cs_b[2] + 7) >> 4;
          vr = cx == cw - 1 ? (4 * cs_r[1] + 7) >> 4 : (3 * cs_r[1] + cs_r[2] + 7) >> 4;
        }
        vb -= 128;
        vr -= 128;
        const int yy = Y[(2 * cy + dy) * w + 2 * cx + dx];
        int r = yy + ((FIXI(1.40200) * vr + H) >> 16);
        int b = yy + ((FIXI(1.77200) * vb + H) >> 16);
        int g = yy + ((-FIXI(0.34414) * vb + H - FIXI(0.71414) * vr) >> 16);
        r = r < 0 ? 0 : (r > 255 ? 255 : r);
        g = g < 0 ? 0 : (g > 255 ? 255 : g);
        b = b < 0 ? 0 : (b > 255 ? 255 : b);
        o[dx * 3] = (uint8_t)r;
        o[dx * 3 + 1] = (uint8_t)g;
        o[dx * 3 + 2] = (uint8_t)b;
      }
    }
  }
}
}  // namespace

size_t rart_ws_jpeg(int, int, int, int) { return 0; }

int rart_launch_jpeg(const RartCorruptArgs& a) {
  static const int quality[5] = {25, 18, 15, 10, 7};
  RART_CHECK_ARG(a.h % 16 == 0 && a.w % 16 == 0,
                 "jpeg_compression: h and w must be multiples of 16 (MCU-aligned; 224 = 14*16)");
  const size_t lds = (size_t)a.h * a.w + 2 * (size_t)(a.h / 2) * (a.w / 2) + 128 * sizeof(int);
  RART_CHECK_ARG(lds <= 160 * 1024, "jpeg_compression: image too large for the one-image-per-CU LDS layout");
  const int q = quality[a.severity - 1];
  const int scale = q < 50 ? 5000 / q : 200 - q * 2;
  if (!rart_raise_dynamic_lds((const void*)k_jpeg, 160 * 1024, "jpeg_compression")) return RART_ERR_HIP;
  hipLaunchKernelGGL(k_jpeg, dim3(a.n), dim3(kThreads), lds, a.stream, a.in, a.out, a.h, a.w, scale);
  RART_CHECK_LAUNCH("jpeg_compression");
  return RART_OK;
}
