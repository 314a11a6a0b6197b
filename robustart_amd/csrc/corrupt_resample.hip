// Resampling corruptions for gfx950: pixelate (Pillow BOX resize, integer, bit-exact) and
// zoom_blur (scipy.ndimage.zoom order-1 accumulation).
// Reference: RobustART/noise/utils/imagenet_c/corruptions.py:104-114,219-232,385-391.
// Restatements: SURVEY.md Appendix A.1 (Pillow Resample.c fixed point) and A.3 (scipy zoom).
#include "rart_common.h"
#include <math.h>

#pragma clang fp contract(off)

namespace {
constexpr int kBlock = 256;
constexpr int KMAX = 4;  // BOX taps per output for every size pair pixelate uses (<= 3)

struct BoxEntry {
  int xmin, n;
  int k[KMAX];
};

// Pillow precompute_coeffs + normalize_coeffs_8bpc for the BOX filter (support 0.5), one thread per
// output index; double arithmetic in the same order as Resample.c.
__device__ __forceinline__ BoxEntry box_entry(int xx, int in_size, int out_size) {
  const double scale = (double)in_size / (double)out_size;
  const double fs = scale < 1.0 ? 1.0 : scale;
  const double support = 0.5 * fs;
  const double center = ((double)xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double w[KMAX] = {0, 0, 0, 0};
  double ww = 0.0;
  const int n = xmax > KMAX ? KMAX : xmax;
  for (int x = 0; x < n; ++x) {
    const double t = ((double)(x + xmin) - center + 0.5) / fs;
    const double v = (t > -0.5 && t <= 0.5) ? 1.0 : 0.0;
    w[x] = v;
    ww += v;
  }
  BoxEntry e;
  e.xmin = xmin;
  e.n = n;
  for (int x = 0; x < KMAX; ++x) {
    double v = w[x];
    if (ww != 0.0) v = v / ww;
    e.k[x] = (int)(0.5 + v * 4194304.0);  // 1 << 22; weights are >= 0
  }
  return e;
}

__global__ void k_box_table(BoxEntry* __restrict__ tab, int in_size, int out_size) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= out_size) return;
  tab[xx] = box_entry(xx, in_size, out_size);
}

// One separable pass.  AXIS 1: along width (in [n][h][win][3] -> out [n][h][wout][3]);
// AXIS 0: along height (in [n][hin][w][3] -> out [n][hout][w][3]).
template <int AXIS>
__global__ __launch_bounds__(kBlock) void k_box_pass(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                     const BoxEntry* __restrict__ tab, int n, int hin, int win,
                                                     int hout, int wout) {
  const size_t total = (size_t)n * hout * wout * 3;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % 3);
    const size_t p = i / 3;
    const int xo = (int)(p % wout);
    const size_t q = p / wout;
    const int yo = (int)(q % hout);
    const int img = (int)(q / hout);
    const BoxEntry e = tab[AXIS == 1 ? xo : yo];
    int acc = 1 << 21;
    for (int j = 0; j < e.n; ++j) {
      const size_t src = AXIS == 1 ? (((size_t)img * hin + yo) * win + (e.xmin + j)) * 3 + c
                                   : (((size_t)img * hin + (e.xmin + j)) * win + xo) * 3 + c;
      acc += (int)in[src] * e.k[j];
    }
    acc >>= 22;
    out[i] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
  }
}

// ---- pixelate as ONE kernel: one workgroup per 224 x 224 image, every intermediate in LDS --------------------------------------
// Pillow's resize is two separable passes with a uint8 intermediate (horizontal, then vertical), and pixelate resizes twice
// (224 -> s -> 224, BOX): four passes, which round 2 ran as four launches through HBM buffers (350 us per 256-image batch, 0.03 of
// the HBM roofline).  Here a workgroup owns an image: the input streams in three 75-row chunks through LDS (16-byte loads, the
// next chunk requested before the current one is reduced), b1 [224][s][3], b2 [s][s][3] and b3 [s][224][3] (over b1) never leave
// LDS, and the last pass writes whole dwords.  Same integer arithmetic per output as k_box_pass: bit-identical.
constexpr int kPxThreads = 1024;
constexpr int PX_HW = 224, PX_ROW = PX_HW * 3;                       // 672 bytes per image row
constexpr int PX_SMAX = 134;                                          // int(224 * 0.6)
constexpr int PX_TAB = 512 * (int)sizeof(BoxEntry);                   // tdown[256] | tup[256]
constexpr int PX_B1 = PX_HW * PX_SMAX * 3;                            // 90 048: b1, later b3
constexpr int PX_CHUNK_ROWS = 75, PX_CHUNK = PX_CHUNK_ROWS * PX_ROW;  // 50 400 B per input chunk
constexpr int PX_R2 = PX_SMAX * PX_SMAX * 3 > PX_CHUNK ? PX_SMAX * PX_SMAX * 3 : PX_CHUNK;    // b2 / chunk buffer: 53 868
constexpr int PX_LDS = PX_TAB + PX_B1 + ((PX_R2 + 15) / 16) * 16;
static_assert(PX_LDS <= 160 * 1024, "pixelate: LDS budget");
static_assert(PX_B1 % 16 == 0 && PX_CHUNK % 16 == 0 && PX_TAB % 16 == 0, "pixelate: 16-byte aligned LDS regions");

__global__ __launch_bounds__(kPxThreads) void k_pixelate_image(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int s) {
  extern __shared__ __attribute__((aligned(16))) uint8_t px_lds[];
  BoxEntry* const tdown = reinterpret_cast<BoxEntry*>(px_lds);
  BoxEntry* const tup = tdown + 256;
  uint8_t* const b1 = px_lds + PX_TAB;                                // [224][s][3]; later b3 [s][224][3]
  uint8_t* const r2 = b1 + PX_B1;                                     // input chunk during pass 1; then b2 [s][s][3]
  const int tid = threadIdx.x;
  const uint8_t* img = in + (size_t)blockIdx.x * PX_HW * PX_ROW;
  const int s3 = s * 3;
  if (tid < s) tdown[tid] = box_entry(tid, PX_HW, s);
  else if (tid >= 256 && tid < 256 + PX_HW) tup[tid - 256] = box_entry(tid - 256, s, PX_HW);

  // ---- pass 1 (horizontal, 224 -> s) over 3 chunks of input rows; chunk k + 1 is in flight while chunk k is reduced
  constexpr int NV = PX_CHUNK / 16;                                   // 3 150 vectors per full chunk: <= 4 per thread
  uint4 q0, q1, q2, q3;
  q0 = q1 = q2 = q3 = make_uint4(0, 0, 0, 0);
#define RART_PX_FETCH(CK)                                                                                       \
  {                                                                                                             \
    const int r0_ = (CK)*PX_CHUNK_ROWS, rows_ = (PX_HW - r0_) < PX_CHUNK_ROWS ? (PX_HW - r0_) : PX_CHUNK_ROWS;  \
    const int nv_ = rows_ * PX_ROW / 16;                                                                        \
    const uint4* g_ = reinterpret_cast<const uint4*>(img + (size_t)r0_ * PX_ROW);                               \
    if (tid < nv_) q0 = g_[tid];                                                                                \
    if (tid + 1024 < nv_) q1 = g_[tid + 1024];                                                                  \
    if (tid + 2048 < nv_) q2 = g_[tid + 2048];                                                                  \
    if (tid + 3072 < nv_) q3 = g_[tid + 3072];                                                                  \
  }
  RART_PX_FETCH(0)
  const int col = tid & 511, sub = tid >> 9;                          // two rows per sweep, one output column (x, c) per thread
  const bool col_ok = col < s3;
  BoxEntry ed;
  int cc = 0;
  __syncthreads();                                                    // tables visible
  if (col_ok) { ed = tdown[col / 3]; cc = col % 3; }
  for (int ck = 0; ck < 3; ++ck) {
    const int r0 = ck * PX_CHUNK_ROWS, rows = (PX_HW - r0) < PX_CHUNK_ROWS ? (PX_HW - r0) : PX_CHUNK_ROWS;
    uint4* c4 = reinterpret_cast<uint4*>(r2);
    c4[tid] = q0;                                                     // NV = 3150 < 4096: slots past the chunk are scratch space
    if (tid + 1024 < NV) c4[tid + 1024] = q1;
    if (tid + 2048 < NV) c4[tid + 2048] = q2;
    if (tid + 3072 < NV) c4[tid + 3072] = q3;
    __syncthreads();
    if (ck + 1 < 3) RART_PX_FETCH(ck + 1)
    if (col_ok) {
      // four outputs per sweep, all their LDS reads issued before the first store (byte stores into LDS would otherwise order
      // every read behind them: the chain of dependent ds_read latencies was most of round 3's first version)
      for (int y = sub; y < rows; y += 8) {
        int acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int yu = y + 2 * u < rows ? y + 2 * u : y;
          const uint8_t* row = r2 + yu * PX_ROW + ed.xmin * 3 + cc;
          acc[u] = 1 << 21;
#pragma unroll
          for (int j = 0; j < KMAX; ++j)                                // static indices: the entry stays in registers
            if (j < ed.n) acc[u] += (int)row[3 * j] * ed.k[j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int a = acc[u] >> 22;
          if (y + 2 * u < rows) b1[(r0 + y + 2 * u) * s3 + col] = (uint8_t)(a < 0 ? 0 : (a > 255 ? 255 : a));
        }
      }
    }
    __syncthreads();                                                  // chunk consumed before the next one overwrites it
  }
#undef RART_PX_FETCH
  // ---- pass 2 (vertical, 224 -> s): b2[yo][col] from b1
  uint8_t* const b2 = r2;
  if (col_ok) {
    for (int yo = sub; yo < s; yo += 8) {
      int acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const BoxEntry e = tdown[yo + 2 * u < s ? yo + 2 * u : yo];
        const uint8_t* colp = b1 + e.xmin * s3 + col;
        acc[u] = 1 << 21;
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
          if (j < e.n) acc[u] += (int)colp[j * s3] * e.k[j];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int a = acc[u] >> 22;
        if (yo + 2 * u < s) b2[(yo + 2 * u) * s3 + col] = (uint8_t)(a < 0 ? 0 : (a > 255 ? 255 : a));
      }
    }
  }
  __syncthreads();
  // ---- pass 3 (horizontal, s -> 224): b3[yo][x][c] over b1's memory
  uint8_t* const b3 = b1;
  for (int i0 = tid; i0 < s * PX_ROW; i0 += 4 * kPxThreads) {
    int acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * kPxThreads < s * PX_ROW ? i0 + u * kPxThreads : i0;
      const int yo = i / PX_ROW, cx = i - yo * PX_ROW, x = cx / 3, c = cx - x * 3;
      const BoxEntry e = tup[x];
      const uint8_t* row = b2 + yo * s3 + e.xmin * 3 + c;
      acc[u] = 1 << 21;
#pragma unroll
      for (int j = 0; j < KMAX; ++j)
        if (j < e.n) acc[u] += (int)row[3 * j] * e.k[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int a = acc[u] >> 22;
      if (i0 + u * kPxThreads < s * PX_ROW) b3[i0 + u * kPxThreads] = (uint8_t)(a < 0 ? 0 : (a > 255 ? 255 : a));
    }
  }
  __syncthreads();
  // ---- pass 4 (vertical, s -> 224): four output bytes (one dword) per step, coalesced stores
  uint32_t* o32 = reinterpret_cast<uint32_t*>(out + (size_t)blockIdx.x * PX_HW * PX_ROW);
  for (int d = tid; d < PX_HW * (PX_ROW / 4); d += kPxThreads) {
    const int y = d / (PX_ROW / 4), cd = d - y * (PX_ROW / 4);
    const BoxEntry e = tup[y];
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21, a3 = 1 << 21;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j >= e.n) break;
      const uint32_t w = *reinterpret_cast<const uint32_t*>(b3 + (e.xmin + j) * PX_ROW + cd * 4);
      const int k = e.k[j];
      a0 += (int)(w & 0xFFu) * k;
      a1 += (int)((w >> 8) & 0xFFu) * k;
      a2 += (int)((w >> 16) & 0xFFu) * k;
      a3 += (int)(w >> 24) * k;
    }
    a0 >>= 22; a1 >>= 22; a2 >>= 22; a3 >>= 22;
    a0 = a0 < 0 ? 0 : (a0 > 255 ? 255 : a0);
    a1 = a1 < 0 ? 0 : (a1 > 255 ? 255 : a1);
    a2 = a2 < 0 ? 0 : (a2 > 255 ? 255 : a2);
    a3 = a3 < 0 ? 0 : (a3 > 255 ? 255 : a3);
    // hipcc (ROCm 7.2) fuses "arithmetic shift + clamp to 0..255 + pack two bytes" into gfx950's v_ashr_pk_u8_i32 and then ORs
    // bytes 2 and 3 into its result as if the upper half were zero -- on the MI355X it is not (measured: bytes 2 and 3 of every
    // dword came out as value | garbage).  An empty asm between the clamp and the pack keeps the four values apart.
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    o32[d] = (uint32_t)a0 | ((uint32_t)a1 << 8) | ((uint32_t)a2 << 16) | ((uint32_t)a3 << 24);
  }
}

// ---- zoom_blur --------------------------------------------------------------------------
struct ZoomParams {
  int count;
  int ch[16], top[16], out_n[16], trim[16];
};

__global__ __launch_bounds__(kBlock) void k_zoom_blur(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                      int n, int h, int w, ZoomParams zp) {
  __shared__ float lut[256];  // (np.array(x) / 255.).astype(np.float32)
  lut[threadIdx.x] = (float)((double)threadIdx.x / 255.0);
  __syncthreads();
  const size_t pixels = (size_t)n * h * w;
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < pixels; p += (size_t)gridDim.x * kBlock) {
    const int xo = (int)(p % w);
    const size_t q = p / w;
    const int yo = (int)(q % h);
    const uint8_t* img = in + (q / h) * (size_t)h * w * 3;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int z = 0; z < zp.count; ++z) {
      const int ch = zp.ch[z], top = zp.top[z], on = zp.out_n[z], trim = zp.trim[z];
      // source coordinate inside the crop: (o + trim) * (ch - 1) / (out_n - 1)  (grid_mode=False)
      const double sy = on > 1 ? (double)((long long)(yo + trim) * (ch - 1)) / (double)(on - 1) : 0.0;
      const double sx = on > 1 ? (double)((long long)(xo + trim) * (ch - 1)) / (double)(on - 1) : 0.0;
      int y0 = (int)floor(sy), x0 = (int)floor(sx);
      y0 = y0 < 0 ? 0 : (y0 > ch - 1 ? ch - 1 : y0);
      x0 = x0 < 0 ? 0 : (x0 > ch - 1 ? ch - 1 : x0);
      const int y1 = y0 + 1 < ch ? y0 + 1 : ch - 1, x1 = x0 + 1 < ch ? x0 + 1 : ch - 1;
      const double ty = sy - (double)y0, tx = sx - (double)x0;
      const uint8_t* r0 = img + ((size_t)(top + y0) * w + top) * 3;
      const uint8_t* r1 = img + ((size_t)(top + y1) * w + top) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double a00 = (double)lut[r0[x0 * 3 + c]], a01 = (double)lut[r0[x1 * 3 + c]];
        const double a10 = (double)lut[r1[x0 * 3 + c]], a11 = (double)lut[r1[x1 * 3 + c]];
        const double omty = 1.0 - ty, omtx = 1.0 - tx;
        const double l0 = a00 * omty, l1 = a10 * ty;
        const double left = l0 + l1;
        const double g0 = a01 * omty, g1 = a11 * ty;
        const double right = g0 + g1;
        const double v0 = left * omtx, v1 = right * tx;
        const double v = v0 + v1;
        acc[c] += (float)v;  // out += clipped_zoom(x, z): fp32 accumulate of the fp32-cast zoom
      }
    }
    const float denom = (float)(zp.count + 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = (lut[in[p * 3 + c]] + acc[c]) / denom;
      v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
      out[p * 3 + c] = (uint8_t)(uint32_t)(v * 255.0f);  // float32 * 255 -> np.uint8
    }
  }
}

// ---- zoom_blur, table driven and row factored (round 5) ----------------------------------------------------------------------
// k_zoom_blur above evaluates the source coordinate of every (pixel, zoom factor) with two fp64 divisions, a floor and the clamps, and
// gathers its 12 source bytes one at a time through a float LUT: 780 us per 256-image batch at severity 3.  Two observations:
//   * the coordinate of an output row / column under zoom z does not depend on the pixel: k_zoom_table writes (i0, i1, t) once per
//     (z, row-or-column) with the SAME expressions;
//   * the reference's bilinear value is  (a00 (1-ty) + a10 ty) (1-tx) + (a01 (1-ty) + a11 ty) tx  with both inner sums depending only on
//     (output row, z, SOURCE column): L[x] = lut(src[y0][x]) (1-ty) + lut(src[y1][x]) ty is computed once per source column of the crop --
//     two coalesced row reads, no gathers -- into LDS, and an output pixel is L[x0] (1-tx) + L[x1] tx.  Same operations on the same
//     operands in the same order as the reference (`left` = L[x0], `right` = L[x1]), fp32 cast and fp32 accumulate per zoom: bit-identical,
//     with 6 instead of 12 LUT look-ups and 6 instead of 9 fp64 operations per (pixel, zoom, channel triple).
// A workgroup owns ZB_ROWS output rows of one image; each of its four waves owns 56 output columns and is self-sufficient: under a zoom
// >= 1 those columns read at most 57 consecutive source columns (x0 grows by <= 1 per output column), so lane l computes L for source
// column x0(first output column) + l into the wave's private LDS region and the blend reads it back after a wave barrier -- no
// workgroup barrier inside the (row, zoom) loop.  Measured on the way (us per 256 images, severity 3; all bit-identical): rounds 1-4's
// per-pixel kernel 620-780; per-pixel gathers with the tables 465; one L row per workgroup with two __syncthreads per (row, zoom) 574;
// this version 454; the same with a bank-replicated 64 KiB LUT 676 (occupancy); L in registers fetched by cross-lane reads with the
// zoom loop's loads free to move: 444 VGPRs or spills; this version with the strip's source rows staged in LDS, i.e. no global load
// inside the loop: 459 (not kept).  The last one settles what binds the kernel: instruction issue -- 440 us are ~94 issue slots per wave
// and (row, zoom), of which 24 are the arithmetic the reference prescribes (18 fp64 operations, 3 casts, 3 fp32 adds) and 12 the LUT
// reads; the rest is byte extraction, LUT / L address arithmetic and the wave fences.  The per-column tables live in registers (the zoom
// loop is unrolled over the 16 possible factors).
constexpr int ZB_ROWS = 8, ZB_WCOLS = 56;

__global__ void k_zoom_table(double* __restrict__ tt, uint32_t* __restrict__ ii, int h, ZoomParams zp) {
  const int z = blockIdx.x, o = threadIdx.x;
  if (z >= zp.count || o >= h) return;
  const int ch = zp.ch[z], on = zp.out_n[z], trim = zp.trim[z];
  const double sc = on > 1 ? (double)((long long)(o + trim) * (ch - 1)) / (double)(on - 1) : 0.0;
  int i0 = (int)floor(sc);
  i0 = i0 < 0 ? 0 : (i0 > ch - 1 ? ch - 1 : i0);
  const int i1 = i0 + 1 < ch ? i0 + 1 : ch - 1;
  tt[z * 224 + o] = sc - (double)i0;
  ii[z * 224 + o] = (uint32_t)i0 | ((uint32_t)i1 << 16);
}

__global__ __launch_bounds__(kBlock) void k_zoom_blur_rows(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           const double* __restrict__ g_tt, const uint32_t* __restrict__ g_ii,
                                                           ZoomParams zp) {
  __shared__ double lut[256];
  __shared__ double Lw[4][2][64 * 3];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool col = lane < ZB_WCOLS;
  const int xo = wave * ZB_WCOLS + (col ? lane : ZB_WCOLS - 1);
  lut[tid] = (double)(float)((double)tid / 255.0);
  double txr[16];
  uint32_t xir[16];
#pragma unroll
  for (int z = 0; z < 16; ++z) {
    txr[z] = 0.0;
    xir[z] = 0u;
    if (z < zp.count) {
      txr[z] = g_tt[z * 224 + xo];
      xir[z] = g_ii[z * 224 + xo];
    }
  }
  __syncthreads();
  const uint8_t* img = in + (size_t)blockIdx.y * (224 * 224 * 3);
  uint8_t* dst = out + (size_t)blockIdx.y * (224 * 224 * 3);
  const float denom = (float)(zp.count + 1);
  for (int row = 0; row < ZB_ROWS; ++row) {
    const int yo = blockIdx.x * ZB_ROWS + row;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int z = 0; z < 16; ++z) {
      if (z < zp.count) {
        const int top = zp.top[z], ch = zp.ch[z];
        const uint32_t yi = g_ii[z * 224 + yo];
        const double ty = g_tt[z * 224 + yo], omty = 1.0 - ty;
        const int y0 = (int)(yi & 0xFFFFu), y1 = (int)(yi >> 16);
        const int x0 = (int)(xir[z] & 0xFFFFu), x1 = (int)(xir[z] >> 16);
        const int xs0 = __builtin_amdgcn_readfirstlane(x0);           // first source column this wave reads
        double* Lb = Lw[wave][z & 1];
        const int xs = xs0 + lane;                                     // L phase: lane -> source column (<= 57 of them are read)
        if (xs < ch) {
          // the 3 bytes of source pixel (top + xs) of both rows as one 4-byte load each that never leaves the image row (672 bytes)
          const int bo = (top + xs) * 3, ba = bo > 668 ? 668 : bo, sh = (bo - ba) * 8;
          uint32_t u0, u1;
          __builtin_memcpy(&u0, img + (size_t)(top + y0) * 672 + ba, 4);
          __builtin_memcpy(&u1, img + (size_t)(top + y1) * 672 + ba, 4);
          u0 >>= sh;
          u1 >>= sh;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const double a0 = lut[(u0 >> (8 * c)) & 0xFFu], a1 = lut[(u1 >> (8 * c)) & 0xFFu];
            const double l0 = a0 * omty, l1 = a1 * ty;
            Lb[lane * 3 + c] = l0 + l1;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
          const double tx = txr[z], omtx = 1.0 - tx;
          const int i0 = (x0 - xs0) * 3, i1 = (x1 - xs0) * 3;          // both < 64 * 3: x1 <= x0(last column) + 1 <= xs0 + 56
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const double q0 = Lb[i0 + c] * omtx, q1 = Lb[i1 + c] * tx;
            const double v = q0 + q1;
            acc[c] += (float)v;
          }
        }
        // (the buffer written two zooms later is the one read here: the next zoom's wave barrier orders the two)
      }
    }
    if (col) {
      const size_t e = ((size_t)yo * 224 + xo) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = ((float)lut[img[e + c]] + acc[c]) / denom;
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
        dst[e + c] = (uint8_t)(uint32_t)(v * 255.0f);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                                   // the next row's first L phase reuses buffer 0
  }
}

}  // namespace
#pragma clang fp contract(fast)

static const double kPixelate[5] = {0.6, 0.5, 0.4, 0.3, 0.25};

static int zoom_factors(int severity, double* f) {
  // np.arange(start, stop, step): n = ceil((stop - start)/step), value_i = start + i*step
  static const double stop[5] = {1.11, 1.16, 1.21, 1.26, 1.31}, step[5] = {0.01, 0.01, 0.02, 0.02, 0.03};
  const int n = (int)ceil((stop[severity - 1] - 1.0) / step[severity - 1]);
  for (int i = 0; i < n; ++i) f[i] = 1.0 + i * step[severity - 1];
  return n;
}

size_t rart_ws_resample(int id, int severity, int n, int h, int w) {
  if (id == RART_ZOOM_BLUR) return rart_align_up(16 * 224 * (sizeof(double) + sizeof(uint32_t)), 256);     // k_zoom_table's (t, i0 | i1) tables
  if (id != RART_PIXELATE || severity < 1 || severity > 5) return 0;
  const int s = (int)(224 * kPixelate[severity - 1]);
  // tables (2 x 256 entries) + three intermediates: [h][s], [s][s], [s][w]
  return rart_align_up(2 * 256 * sizeof(BoxEntry), 256) + rart_align_up((size_t)n * h * s * 3, 256) +
         rart_align_up((size_t)n * s * s * 3, 256) + rart_align_up((size_t)n * s * w * 3, 256);
}

int rart_launch_resample(int id, const RartCorruptArgs& a) {
  if (id == RART_PIXELATE) {
    RART_CHECK_ARG(a.h == 224 && a.w == 224, "pixelate: reference hard-codes 224x224 (corruptions.py:388-389)");
    const int s = (int)(224 * kPixelate[a.severity - 1]);
    if (s <= PX_SMAX && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 3) == 0) {
      // one kernel, one workgroup per image, intermediates in LDS
      if (!rart_raise_dynamic_lds((const void*)k_pixelate_image, PX_LDS, "pixelate")) return RART_ERR_HIP;
      hipLaunchKernelGGL(k_pixelate_image, dim3(a.n), dim3(kPxThreads), PX_LDS, a.stream, a.in, a.out, s);
      RART_CHECK_LAUNCH("pixelate");
      return RART_OK;
    }
    uint8_t* ws = (uint8_t*)a.workspace;
    BoxEntry* t_down = (BoxEntry*)ws;
    BoxEntry* t_up = t_down + 256;
    ws += rart_align_up(2 * 256 * sizeof(BoxEntry), 256);
    uint8_t* b1 = ws;  // [n][224][s]
    ws += rart_align_up((size_t)a.n * a.h * s * 3, 256);
    uint8_t* b2 = ws;  // [n][s][s]
    ws += rart_align_up((size_t)a.n * s * s * 3, 256);
    uint8_t* b3 = ws;  // [n][s][224]
    hipLaunchKernelGGL(k_box_table, dim3(1), dim3(256), 0, a.stream, t_down, 224, s);
    hipLaunchKernelGGL(k_box_table, dim3(1), dim3(256), 0, a.stream, t_up, s, 224);
    // Pillow: horizontal pass first (uint8 intermediate), then vertical -- twice
    hipLaunchKernelGGL(k_box_pass<1>, dim3(rart_grid_for((size_t)a.n * 224 * s * 3)), dim3(kBlock), 0, a.stream,
                       a.in, b1, t_down, a.n, 224, 224, 224, s);
    hipLaunchKernelGGL(k_box_pass<0>, dim3(rart_grid_for((size_t)a.n * s * s * 3)), dim3(kBlock), 0, a.stream, b1,
                       b2, t_down, a.n, 224, s, s, s);
    hipLaunchKernelGGL(k_box_pass<1>, dim3(rart_grid_for((size_t)a.n * s * 224 * 3)), dim3(kBlock), 0, a.stream, b2,
                       b3, t_up, a.n, s, s, s, 224);
    hipLaunchKernelGGL(k_box_pass<0>, dim3(rart_grid_for((size_t)a.n * 224 * 224 * 3)), dim3(kBlock), 0, a.stream,
                       b3, a.out, t_up, a.n, s, 224, 224, 224);
    RART_CHECK_LAUNCH("pixelate");
    return RART_OK;
  }
  if (id == RART_ZOOM_BLUR) {
    RART_CHECK_ARG(a.h == a.w, "zoom_blur: square images only (clipped_zoom crops h x h, corruptions.py:105-110)");
    RART_CHECK_ARG(a.in != a.out, "zoom_blur cannot run in place through this entry (gather reads neighbours); "
                                  "the Python layer stages a copy");
    double f[16];
    ZoomParams zp;
    zp.count = zoom_factors(a.severity, f);
    for (int i = 0; i < zp.count; ++i) {
      const int ch = (int)ceil((double)a.h / f[i]);
      zp.ch[i] = ch;
      zp.top[i] = (a.h - ch) / 2;
      zp.out_n[i] = (int)nearbyint((double)ch * f[i]);  // Python round(): half to even
      zp.trim[i] = (zp.out_n[i] - a.h) / 2;
    }
    if (a.h == 224 && a.w == 224 && a.n <= 65535 && a.workspace && getenv("RART_ZOOM_DIRECT") == nullptr) {
      double* tt = (double*)a.workspace;
      uint32_t* ii = (uint32_t*)(tt + 16 * 224);
      hipLaunchKernelGGL(k_zoom_table, dim3(zp.count), dim3(256), 0, a.stream, tt, ii, a.h, zp);
      hipLaunchKernelGGL(k_zoom_blur_rows, dim3(224 / ZB_ROWS, a.n), dim3(kBlock), 0, a.stream, a.in, a.out, (const double*)tt,
                         (const uint32_t*)ii, zp);
      RART_CHECK_LAUNCH("zoom_blur");
      return RART_OK;
    }
    hipLaunchKernelGGL(k_zoom_blur, dim3(rart_grid_for((size_t)a.n * a.h * a.w, kBlock, 256 * 16)), dim3(kBlock), 0,
                       a.stream, a.in, a.out, a.n, a.h, a.w, zp);
    RART_CHECK_LAUNCH("zoom_blur");
    return RART_OK;
  }
  rart_set_error("rart_launch_resample: bad id %d", id);
  return RART_ERR_INVALID;
}
