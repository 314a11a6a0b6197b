// ImageNet-S resize operators, OpenCV family, on gfx950: cv2.resize(img, (w, h), interpolation=...) for 8-bit
// 3-channel images -- INTER_NEAREST / LINEAR / CUBIC / AREA / LANCZOS4 -- followed by a crop.
// Reference: RobustART/noise/utils/imagenet_s_gen.py:27-33 (opencv_resize_mode_dict), :138-146 (val transform:
// cv2.resize to (256, 256), centre crop 224).  OpenCV itself is an unvendored dependency (requirements: opencv-python,
// unpinned) and is absent from the build container: PARITY UNPINNED.  This restates the published algorithm of
// opencv/modules/imgproc/src/resize.cpp (4.x):
//   * coordinates: fx = (float)((dx + 0.5) * scale - 0.5) (float!), sx = floor, fx -= sx;
//   * LINEAR / CUBIC (A = -0.75) / LANCZOS4: coefficients in float, converted to int16 by saturate_cast<short>(c * 2048)
//     (round half to even); horizontal pass into int32 rows (edge taps replicate), vertical pass
//       linear:  (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
//       cubic / lanczos4:  (sum b_k S_k + 2^21) >> 22, saturated;
//   * LINEAR with an exact 2x2 decimation is executed as AREA (resize.cpp: "is_area_fast && iscale == 2");
//   * AREA, both scales >= 1: integer scales average in integers ((a+b+c+d+2)>>2 for 2x2, cvRound(sum * (1.f/area))
//     otherwise); fractional scales use the float DecimateAlpha tables, accumulated in OpenCV's order; AREA with an
//     up-scaling axis is the linear code with area-mode coordinates;
//   * NEAREST: sx = min(floor(dx * scale), w - 1).
// Coefficient tables are built on the host in the same float / double arithmetic as OpenCV and uploaded; the device
// does only integer (or ordered float, FMA contraction off) accumulation.
#include "rart_common.h"
#include <math.h>
#include <float.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <tuple>
#include <vector>

#pragma clang fp contract(off)

namespace {
constexpr int kBlock = 256;
enum { CV_NEAREST = 0, CV_LINEAR = 1, CV_CUBIC = 2, CV_AREA = 3, CV_LANCZOS4 = 4 };

inline int cv_round(double v) { return (int)lrint(v); }           // round half to even (default rounding mode)
inline short sat_short(float v) {
  int i = (int)lrintf(v);
  return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i));
}
void interpolate_cubic(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}
void interpolate_lanczos4(float x, float* c) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
  if (x < FLT_EPSILON) {
    for (int i = 0; i < 8; ++i) c[i] = 0;
    c[3] = 1;
    return;
  }
  float sum = 0;
  const double y0 = -(x + 3) * M_PI * 0.25, s0 = sin(y0), c0 = cos(y0);
  for (int i = 0; i < 8; ++i) {
    const double y = -(x + 3 - i) * M_PI * 0.25;
    c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
    sum += c[i];
  }
  sum = 1.f / sum;
  for (int i = 0; i < 8; ++i) c[i] *= sum;
}

// per output index: [first tap source index (may be outside: taps clamp), k0 .. k_{ksize-1}] as int32
struct AxisTable {
  int ksize;
  std::vector<int> data;
};
const AxisTable& linear_family_table(int ssize, int dsize, int interp, bool is_x) {
  // cached for the process lifetime: the asynchronous upload reads the host vector after this call returns
  static std::map<std::tuple<int, int, int, bool>, AxisTable> cache;
  std::lock_guard<std::mutex> lk(rart_host_table_mutex());      // (loader threads: see rart_common.h)
  const auto key = std::make_tuple(ssize, dsize, interp, is_x);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const double inv_scale = (double)dsize / ssize, scale = 1.0 / inv_scale;
  const bool area_mode = interp == CV_AREA;
  const int ksize = interp == CV_CUBIC ? 4 : (interp == CV_LANCZOS4 ? 8 : 2), ksize2 = ksize / 2;
  AxisTable t;
  t.ksize = ksize;
  t.data.assign((size_t)dsize * (ksize + 1), 0);
  for (int d = 0; d < dsize; ++d) {
    float f;
    int s;
    if (!area_mode) {
      f = (float)((d + 0.5) * scale - 0.5);
      s = (int)floorf(f);
      f -= s;
    } else {
      s = (int)floor(d * scale);
      f = (float)((d + 1) - (s + 1) * inv_scale);
      f = f <= 0 ? 0.f : f - floorf(f);
    }
    if (is_x && ksize == 2) {            // the x table pins border samples (resize.cpp: xmin / xmax handling)
      if (s < 0) { f = 0; s = 0; }
      if (s >= ssize - 1) { f = 0; s = ssize - 1; }
    }
    float cbuf[8];
    if (interp == CV_CUBIC) interpolate_cubic(f, cbuf);
    else if (interp == CV_LANCZOS4) interpolate_lanczos4(f, cbuf);
    else { cbuf[0] = 1.f - f; cbuf[1] = f; }
    int* row = &t.data[(size_t)d * (ksize + 1)];
    row[0] = s - (ksize2 - 1);
    for (int k = 0; k < ksize; ++k) row[1 + k] = sat_short(cbuf[k] * 2048.f);
  }
  return cache.emplace(key, std::move(t)).first->second;
}

// AREA (fractional decimation): per output index [count, (source index, alpha as float bits) x count]
struct AreaTable {
  int max_count;
  std::vector<int> data;
};
const AreaTable& area_table(int ssize, int dsize) {
  static std::map<std::pair<int, int>, AreaTable> cache;
  std::lock_guard<std::mutex> lk(rart_host_table_mutex());      // (loader threads: see rart_common.h)
  const auto key = std::make_pair(ssize, dsize);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const double scale = (double)ssize / dsize;
  std::vector<std::vector<std::pair<int, float>>> rows(dsize);
  int mx = 0;
  for (int dx = 0; dx < dsize; ++dx) {
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    const double cell = std::min(scale, ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = std::min(sx2, ssize - 1);
    sx1 = std::min(sx1, sx2);
    if (sx1 - fsx1 > 1e-3) rows[dx].push_back({sx1 - 1, (float)((sx1 - fsx1) / cell)});
    for (int sx = sx1; sx < sx2; ++sx) rows[dx].push_back({sx, (float)(1.0 / cell)});
    if (fsx2 - sx2 > 1e-3) rows[dx].push_back({sx2, (float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)});
    mx = std::max(mx, (int)rows[dx].size());
  }
  AreaTable t;
  t.max_count = mx;
  t.data.assign((size_t)dsize * (1 + 2 * mx), 0);
  for (int dx = 0; dx < dsize; ++dx) {
    int* r = &t.data[(size_t)dx * (1 + 2 * mx)];
    r[0] = (int)rows[dx].size();
    for (size_t k = 0; k < rows[dx].size(); ++k) {
      r[1 + 2 * k] = rows[dx][k].first;
      float a = rows[dx][k].second;
      memcpy(&r[2 + 2 * k], &a, 4);
    }
  }
  return cache.emplace(key, std::move(t)).first->second;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(kBlock) void k_cv_nearest(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int n, int h,
                                                       int w, double sx, double sy, int cy, int cx, int ch, int cw) {
  const size_t total = (size_t)n * ch * cw;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int xo = (int)(i % cw), yo = (int)((i / cw) % ch), img = (int)(i / ((size_t)cw * ch));
    int x = (int)floor((cx + xo) * sx), y = (int)floor((cy + yo) * sy);
    x = x < w - 1 ? x : w - 1;
    y = y < h - 1 ? y : h - 1;
    const uint8_t* s = in + (((size_t)img * h + y) * w + x) * 3;
    out[i * 3] = s[0];
    out[i * 3 + 1] = s[1];
    out[i * 3 + 2] = s[2];
  }
}
// horizontal pass: tmp[n][y_count][cw][3] int32 = sum_k S[clamp(first + k)] * a_k
__global__ __launch_bounds__(kBlock) void k_cv_h(const uint8_t* __restrict__ in, int* __restrict__ tmp,
                                                 const int* __restrict__ tab, int ksize, int n, int h, int w, int cx, int cw,
                                                 int y_first, int y_count) {
  const size_t total = (size_t)n * y_count * cw * 3;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % 3);
    size_t p = i / 3;
    const int xo = (int)(p % cw);
    p /= cw;
    const int yo = (int)(p % y_count), img = (int)(p / y_count);
    const int* row = tab + (size_t)(cx + xo) * (ksize + 1);
    const uint8_t* src = in + ((size_t)img * h + (y_first + yo)) * w * 3 + c;
    int acc = 0;
    for (int k = 0; k < ksize; ++k) acc += (int)src[(size_t)clampi(row[0] + k, 0, w - 1) * 3] * row[1 + k];
    tmp[i] = acc;
  }
}
__global__ __launch_bounds__(kBlock) void k_cv_v(const int* __restrict__ tmp, uint8_t* __restrict__ out,
                                                 const int* __restrict__ tab, int ksize, int n, int h, int y_first,
                                                 int y_count, int cy, int ch, int cw) {
  const size_t total = (size_t)n * ch * cw * 3;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const size_t col = i % ((size_t)cw * 3);
    size_t p = i / ((size_t)cw * 3);
    const int yo = (int)(p % ch), img = (int)(p / ch);
    const int* row = tab + (size_t)(cy + yo) * (ksize + 1);
    const int* base = tmp + (size_t)img * y_count * cw * 3 + col;
    int v;
    if (ksize == 2) {
      const int s0 = base[(size_t)(clampi(row[0], 0, h - 1) - y_first) * cw * 3];
      const int s1 = base[(size_t)(clampi(row[0] + 1, 0, h - 1) - y_first) * cw * 3];
      v = (((row[1] * (s0 >> 4)) >> 16) + ((row[2] * (s1 >> 4)) >> 16) + 2) >> 2;
    } else {
      int acc = 0;
      for (int k = 0; k < ksize; ++k) acc += base[(size_t)(clampi(row[0] + k, 0, h - 1) - y_first) * cw * 3] * row[1 + k];
      v = (acc + (1 << 21)) >> 22;
    }
    out[i] = (uint8_t)clampi(v, 0, 255);
  }
}
// AREA, fractional decimation: float accumulation in OpenCV's order (columns of a source row, then rows)
__global__ __launch_bounds__(kBlock) void k_cv_area(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                    const int* __restrict__ xtab, int xmax, const int* __restrict__ ytab,
                                                    int ymax, int n, int h, int w, int cy, int cx, int ch, int cw) {
  const size_t total = (size_t)n * ch * cw * 3;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % 3);
    size_t p = i / 3;
    const int xo = (int)(p % cw);
    p /= cw;
    const int yo = (int)(p % ch), img = (int)(p / ch);
    const int* xr = xtab + (size_t)(cx + xo) * (1 + 2 * xmax);
    const int* yr = ytab + (size_t)(cy + yo) * (1 + 2 * ymax);
    float sum = 0.f;
    for (int j = 0; j < yr[0]; ++j) {
      const uint8_t* src = in + ((size_t)img * h + yr[1 + 2 * j]) * w * 3 + c;
      float buf = 0.f;
      for (int k = 0; k < xr[0]; ++k) buf += (float)src[(size_t)xr[1 + 2 * k] * 3] * __int_as_float(xr[2 + 2 * k]);
      const float beta = __int_as_float(yr[2 + 2 * j]);
      sum = j == 0 ? beta * buf : sum + beta * buf;
    }
    out[i] = (uint8_t)clampi((int)rintf(sum), 0, 255);
  }
}
// AREA, integer scales
__global__ __launch_bounds__(kBlock) void k_cv_area_fast(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int n, int h,
                                                         int w, int isx, int isy, float inv_area, int cy, int cx, int ch,
                                                         int cw) {
  const size_t total = (size_t)n * ch * cw * 3;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % 3);
    size_t p = i / 3;
    const int xo = (int)(p % cw);
    p /= cw;
    const int yo = (int)(p % ch), img = (int)(p / ch);
    const uint8_t* src = in + (((size_t)img * h + (size_t)(cy + yo) * isy) * w + (size_t)(cx + xo) * isx) * 3 + c;
    int sum = 0;
    for (int y = 0; y < isy; ++y)
      for (int x = 0; x < isx; ++x) sum += src[((size_t)y * w + x) * 3];
    const int v = (isx == 2 && isy == 2) ? (sum + 2) >> 2 : (int)rintf((float)sum * inv_area);
    out[i] = (uint8_t)clampi(v, 0, 255);
  }
}
int grid_for(size_t items) { return rart_grid_for(items, kBlock, 256 * 16); }

struct CvPlan {
  int mode;          // 0 nearest, 1 two-pass fixed point, 2 area tables, 3 area fast
  int interp;        // effective interpolation of the two-pass path
  int y_first, y_count;
  const AxisTable *tx, *ty;
  const AreaTable *ax, *ay;
  size_t tabx_bytes, taby_bytes, tmp_bytes;
};
CvPlan make_plan(int n, int h, int w, int rh, int rw, int interp, int cy, int ch, int cw) {
  CvPlan p{};
  const double scale_x = 1.0 / ((double)rw / w), scale_y = 1.0 / ((double)rh / h);
  if (interp == CV_NEAREST) { p.mode = 0; return p; }
  const int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
  const bool area_fast = fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON;
  if (interp == CV_LINEAR && area_fast && isx == 2 && isy == 2) interp = CV_AREA;
  if (interp == CV_AREA && scale_x >= 1 && scale_y >= 1) {
    if (area_fast) { p.mode = 3; return p; }
    p.mode = 2;
    p.ax = &area_table(w, rw);
    p.ay = &area_table(h, rh);
    p.tabx_bytes = rart_align_up(p.ax->data.size() * 4, 256);
    p.taby_bytes = rart_align_up(p.ay->data.size() * 4, 256);
    return p;
  }
  p.mode = 1;
  p.interp = interp;
  p.tx = &linear_family_table(w, rw, interp, true);
  p.ty = &linear_family_table(h, rh, interp, false);
  int first = h, last = 0;
  for (int yy = cy; yy < cy + ch; ++yy) {
    const int* row = &p.ty->data[(size_t)yy * (p.ty->ksize + 1)];
    const int a = std::max(0, std::min(h - 1, row[0])), b = std::max(0, std::min(h - 1, row[0] + p.ty->ksize - 1));
    first = std::min(first, a);
    last = std::max(last, b + 1);
  }
  p.y_first = first;
  p.y_count = last - first;
  p.tabx_bytes = rart_align_up(p.tx->data.size() * 4, 256);
  p.taby_bytes = rart_align_up(p.ty->data.size() * 4, 256);
  p.tmp_bytes = rart_align_up((size_t)n * p.y_count * cw * 3 * 4, 256);
  return p;
}
}  // namespace

extern "C" {

size_t rart_cv_resize_workspace_bytes(int n, int h, int w, int resize_h, int resize_w, int interpolation, int crop_y, int crop_x,
                                      int crop_h, int crop_w) {
  if (n <= 0 || h <= 0 || w <= 0 || resize_h <= 0 || resize_w <= 0 || interpolation < 0 || interpolation > 4) return 0;
  if (crop_y < 0 || crop_x < 0 || crop_h <= 0 || crop_w <= 0 || crop_y + crop_h > resize_h || crop_x + crop_w > resize_w)
    return 0;
  const CvPlan p = make_plan(n, h, w, resize_h, resize_w, interpolation, crop_y, crop_h, crop_w);
  return p.tabx_bytes + p.taby_bytes + p.tmp_bytes;
}

int rart_cv_resize_u8(const uint8_t* in, uint8_t* out, int n, int h, int w, int resize_h, int resize_w, int interpolation,
                      int crop_y, int crop_x, int crop_h, int crop_w, void* workspace, size_t workspace_bytes,
                      rart_stream_t stream) {
  RART_CHECK_ARG(in && out && n > 0 && h > 0 && w > 0 && resize_h > 0 && resize_w > 0, "rart_cv_resize_u8: bad arguments");
  RART_CHECK_ARG(interpolation >= 0 && interpolation <= 4, "rart_cv_resize_u8: interpolation must be a cv2 constant 0..4 "
                                                           "(INTER_NEAREST, LINEAR, CUBIC, AREA, LANCZOS4)");
  RART_CHECK_ARG(crop_y >= 0 && crop_x >= 0 && crop_h > 0 && crop_w > 0 && crop_y + crop_h <= resize_h &&
                     crop_x + crop_w <= resize_w, "rart_cv_resize_u8: crop window outside the resized image");
  hipStream_t st = (hipStream_t)stream;
  const CvPlan p = make_plan(n, h, w, resize_h, resize_w, interpolation, crop_y, crop_h, crop_w);
  const double scale_x = 1.0 / ((double)resize_w / w), scale_y = 1.0 / ((double)resize_h / h);
  const size_t out_items = (size_t)n * crop_h * crop_w * 3;
  if (p.mode == 0) {
    hipLaunchKernelGGL(k_cv_nearest, dim3(grid_for(out_items / 3)), dim3(kBlock), 0, st, in, out, n, h, w, scale_x, scale_y,
                       crop_y, crop_x, crop_h, crop_w);
    RART_CHECK_LAUNCH("rart_cv_resize_u8 (nearest)");
    return RART_OK;
  }
  if (p.mode == 3) {
    const int isx = (int)lrint(scale_x), isy = (int)lrint(scale_y);
    hipLaunchKernelGGL(k_cv_area_fast, dim3(grid_for(out_items)), dim3(kBlock), 0, st, in, out, n, h, w, isx, isy,
                       1.f / (float)(isx * isy), crop_y, crop_x, crop_h, crop_w);
    RART_CHECK_LAUNCH("rart_cv_resize_u8 (area, integer scale)");
    return RART_OK;
  }
  const size_t need = p.tabx_bytes + p.taby_bytes + p.tmp_bytes;
  if (!workspace || workspace_bytes < need) {
    rart_set_error("rart_cv_resize_u8: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return RART_ERR_WORKSPACE;
  }
  int* tabx = (int*)workspace;
  int* taby = (int*)((uint8_t*)workspace + p.tabx_bytes);
  const std::vector<int>& dx = p.mode == 2 ? p.ax->data : p.tx->data;
  const std::vector<int>& dy = p.mode == 2 ? p.ay->data : p.ty->data;
  if (hipMemcpyAsync(tabx, dx.data(), dx.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(taby, dy.data(), dy.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess) {
    rart_set_error("rart_cv_resize_u8: coefficient upload failed");
    return RART_ERR_HIP;
  }
  if (p.mode == 2) {
    hipLaunchKernelGGL(k_cv_area, dim3(grid_for(out_items)), dim3(kBlock), 0, st, in, out, (const int*)tabx, p.ax->max_count,
                       (const int*)taby, p.ay->max_count, n, h, w, crop_y, crop_x, crop_h, crop_w);
  } else {
    int* tmp = (int*)((uint8_t*)workspace + p.tabx_bytes + p.taby_bytes);
    hipLaunchKernelGGL(k_cv_h, dim3(grid_for((size_t)n * p.y_count * crop_w * 3)), dim3(kBlock), 0, st, in, tmp,
                       (const int*)tabx, p.tx->ksize, n, h, w, crop_x, crop_w, p.y_first, p.y_count);
    hipLaunchKernelGGL(k_cv_v, dim3(grid_for(out_items)), dim3(kBlock), 0, st, (const int*)tmp, out, (const int*)taby,
                       p.ty->ksize, n, h, p.y_first, p.y_count, crop_y, crop_h, crop_w);
  }
  RART_CHECK_LAUNCH("rart_cv_resize_u8");
  return RART_OK;
}

}  // extern "C"
