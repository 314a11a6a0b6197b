// ImageNet-S resize operators on gfx950: Pillow's Image.resize for 8-bit RGB, bit-exact.
// Reference: RobustART/noise/utils/imagenet_s_gen.py:19-34,127-166 (pil_resize_mode_dict, 'val' transform:
// resize to 8/7 * 224 then centre crop).  Arithmetic: Pillow libImaging/Resample.c (22-bit fixed-point
// coefficients, horizontal pass to a uint8 intermediate, then vertical) and Geometry.c (NEAREST, 16.16 affine).
// Coefficient tables are built on the host with the C library's sin/cos exactly as Pillow builds them (device
// transcendental functions differ in the last ulp, which would flip a rounded integer coefficient now and then),
// cached for the process lifetime and uploaded asynchronously.  The crop is fused: only the rows / columns of
// the crop window are ever computed.
#include "rart_common.h"
#include <math.h>
#include <map>
#include <tuple>
#include <vector>

namespace {
constexpr int kBlock = 256;

double sinc_f(double x) {
  if (x == 0.0) return 1.0;
  x = x * M_PI;
  return sin(x) / x;
}
double filter_value(int f, double x) {
  switch (f) {
    case 3: return (x > -0.5 && x <= 0.5) ? 1.0 : 0.0;                      // box
    case 1: x = fabs(x); return x < 1.0 ? 1.0 - x : 0.0;                    // bilinear
    case 4: {                                                                // hamming
      x = fabs(x);
      if (x == 0.0) return 1.0;
      if (x >= 1.0) return 0.0;
      x = x * M_PI;
      return sin(x) / x * (0.54 + 0.46 * cos(x));
    }
    case 2: {                                                                // bicubic, a = -0.5
      const double a = -0.5;
      x = fabs(x);
      if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
      if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
      return 0.0;
    }
    case 5: return (x >= -3.0 && x < 3.0) ? sinc_f(x) * sinc_f(x / 3) : 0.0;  // lanczos
  }
  return 0.0;
}
double filter_support(int f) {
  switch (f) { case 3: return 0.5; case 1: case 4: return 1.0; case 2: return 2.0; case 5: return 3.0; }
  return 0.0;
}

// table: per output index [xmin, n, k0 .. k_{ksize-1}] as int32
struct CoeffTable {
  int ksize;
  std::vector<int> data;
};
const CoeffTable& coeff_table(int in_size, int out_size, int f) {
  static std::map<std::tuple<int, int, int>, CoeffTable> cache;
  std::lock_guard<std::mutex> lk(rart_host_table_mutex());      // (loader threads: see rart_common.h)
  auto key = std::make_tuple(in_size, out_size, f);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const double scale = (double)in_size / (double)out_size;
  const double fs = scale < 1.0 ? 1.0 : scale;
  const double support = filter_support(f) * fs;
  const int ksize = (int)ceil(support) * 2 + 1;
  CoeffTable t;
  t.ksize = ksize;
  t.data.assign((size_t)out_size * (ksize + 2), 0);
  const double ss = 1.0 / fs;
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = ((double)xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      k[x] = filter_value(f, ((double)(x + xmin) - center + 0.5) * ss);
      ww += k[x];
    }
    int* row = &t.data[(size_t)xx * (ksize + 2)];
    row[0] = xmin;
    row[1] = xmax;
    for (int x = 0; x < xmax; ++x) {
      double v = k[x];
      if (ww != 0.0) v = v / ww;
      row[2 + x] = v < 0 ? (int)(-0.5 + v * 4194304.0) : (int)(0.5 + v * 4194304.0);
    }
  }
  return cache.emplace(key, std::move(t)).first->second;
}

// horizontal pass over the crop's columns: tmp[n][h][cw][3]
__global__ __launch_bounds__(kBlock) void k_resample_h(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp,
                                                       const int* __restrict__ tab, int ksize, int n, int h, int w,
                                                       int cx, int cw, int y_first, int y_count) {
  const size_t total = (size_t)n * y_count * cw * 3;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int c = (int)(i % 3);
    size_t p = i / 3;
    const int xo = (int)(p % cw);
    p /= cw;
    const int yo = (int)(p % y_count), img = (int)(p / y_count);
    const int* row = tab + (size_t)(cx + xo) * (ksize + 2);
    const int xmin = row[0], cnt = row[1];
    const uint8_t* src = in + (((size_t)img * h + (y_first + yo)) * w + xmin) * 3 + c;
    int acc = 1 << 21;
    for (int k = 0; k < cnt; ++k) acc += (int)src[k * 3] * row[2 + k];
    acc >>= 22;
    tmp[i] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
  }
}
// vertical pass over the crop's rows: out[n][ch][cw][3] from tmp[n][y_count][cw][3]
__global__ __launch_bounds__(kBlock) void k_resample_v(const uint8_t* __restrict__ tmp, uint8_t* __restrict__ out,
                                                       const int* __restrict__ tab, int ksize, int n, int y_first,
                                                       int y_count, int cy, int ch, int cw) {
  const size_t total = (size_t)n * ch * cw * 3;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const size_t col = i % ((size_t)cw * 3);
    size_t p = i / ((size_t)cw * 3);
    const int yo = (int)(p % ch), img = (int)(p / ch);
    const int* row = tab + (size_t)(cy + yo) * (ksize + 2);
    const int ymin = row[0], cnt = row[1];
    const uint8_t* src = tmp + ((size_t)img * y_count + (ymin - y_first)) * cw * 3 + col;
    int acc = 1 << 21;
    for (int k = 0; k < cnt; ++k) acc += (int)src[(size_t)k * cw * 3] * row[2 + k];
    acc >>= 22;
    out[i] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
  }
}
__global__ __launch_bounds__(kBlock) void k_resize_nearest(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           int n, int h, int w, int fx0, int fxs, int fy0, int fys,
                                                           int cy, int cx, int ch, int cw) {
  const size_t total = (size_t)n * ch * cw;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int xo = (int)(i % cw), yo = (int)((i / cw) % ch), img = (int)(i / ((size_t)cw * ch));
    int sx = (int)(((long long)fx0 + (long long)(cx + xo) * fxs) >> 16);
    int sy = (int)(((long long)fy0 + (long long)(cy + yo) * fys) >> 16);
    sx = sx < 0 ? 0 : (sx > w - 1 ? w - 1 : sx);
    sy = sy < 0 ? 0 : (sy > h - 1 ? h - 1 : sy);
    const uint8_t* s = in + (((size_t)img * h + sy) * w + sx) * 3;
    out[i * 3] = s[0];
    out[i * 3 + 1] = s[1];
    out[i * 3 + 2] = s[2];
  }
}
int grid_for(size_t items) { return rart_grid_for(items, kBlock, 256 * 16); }

struct Plan {
  int y_first, y_count, ks_h, ks_v;
  size_t tab_h_bytes, tab_v_bytes, tmp_bytes;
};
Plan make_plan(int n, int h, int w, int rh, int rw, int f, int cy, int cx, int ch, int cw) {
  Plan p{};
  if (f == 0) return p;
  const CoeffTable& tv = coeff_table(h, rh, f);
  const CoeffTable& th = coeff_table(w, rw, f);
  int first = h, last = 0;
  for (int yy = cy; yy < cy + ch; ++yy) {
    const int* row = &tv.data[(size_t)yy * (tv.ksize + 2)];
    if (row[0] < first) first = row[0];
    if (row[0] + row[1] > last) last = row[0] + row[1];
  }
  p.y_first = first;
  p.y_count = last - first;
  p.ks_h = th.ksize;
  p.ks_v = tv.ksize;
  p.tab_h_bytes = rart_align_up(th.data.size() * sizeof(int), 256);
  p.tab_v_bytes = rart_align_up(tv.data.size() * sizeof(int), 256);
  p.tmp_bytes = rart_align_up((size_t)n * p.y_count * cw * 3, 256);
  (void)cx;
  return p;
}
}  // namespace

extern "C" {

size_t rart_pil_resize_workspace_bytes(int n, int h, int w, int resize_h, int resize_w, int filter, int crop_y, int crop_x,
                                       int crop_h, int crop_w) {
  if (n <= 0 || h <= 0 || w <= 0 || resize_h <= 0 || resize_w <= 0 || filter < 0 || filter > 5) return 0;
  if (crop_y < 0 || crop_x < 0 || crop_h <= 0 || crop_w <= 0 || crop_y + crop_h > resize_h || crop_x + crop_w > resize_w)
    return 0;
  const Plan p = make_plan(n, h, w, resize_h, resize_w, filter, crop_y, crop_x, crop_h, crop_w);
  return p.tab_h_bytes + p.tab_v_bytes + p.tmp_bytes;
}

int rart_pil_resize_u8(const uint8_t* in, uint8_t* out, int n, int h, int w, int resize_h, int resize_w, int filter,
                       int crop_y, int crop_x, int crop_h, int crop_w, void* workspace, size_t workspace_bytes,
                       rart_stream_t stream) {
  RART_CHECK_ARG(in && out && n > 0 && h > 0 && w > 0 && resize_h > 0 && resize_w > 0, "rart_pil_resize_u8: bad arguments");
  RART_CHECK_ARG(filter >= 0 && filter <= 5, "rart_pil_resize_u8: filter must be 0..5 (nearest, bilinear, bicubic, box, "
                                             "hamming, lanczos = PIL.Image constants)");
  RART_CHECK_ARG(crop_y >= 0 && crop_x >= 0 && crop_h > 0 && crop_w > 0 && crop_y + crop_h <= resize_h &&
                     crop_x + crop_w <= resize_w, "rart_pil_resize_u8: crop window outside the resized image");
  hipStream_t st = (hipStream_t)stream;
  if (filter == 0) {
    auto fix = [](double v) { return (int)floor(v * 65536.0 + 0.5); };
    const double a0 = (double)w / resize_w, a4 = (double)h / resize_h;
    hipLaunchKernelGGL(k_resize_nearest, dim3(grid_for((size_t)n * crop_h * crop_w)), dim3(kBlock), 0, st, in, out, n, h,
                       w, fix(a0 * 0.5), fix(a0), fix(a4 * 0.5), fix(a4), crop_y, crop_x, crop_h, crop_w);
    RART_CHECK_LAUNCH("rart_pil_resize_u8 (nearest)");
    return RART_OK;
  }
  const Plan p = make_plan(n, h, w, resize_h, resize_w, filter, crop_y, crop_x, crop_h, crop_w);
  const size_t need = p.tab_h_bytes + p.tab_v_bytes + p.tmp_bytes;
  if (!workspace || workspace_bytes < need) {
    rart_set_error("rart_pil_resize_u8: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return RART_ERR_WORKSPACE;
  }
  const CoeffTable& tv = coeff_table(h, resize_h, filter);
  const CoeffTable& th = coeff_table(w, resize_w, filter);
  int* tab_h = (int*)workspace;
  int* tab_v = (int*)((uint8_t*)workspace + p.tab_h_bytes);
  uint8_t* tmp = (uint8_t*)workspace + p.tab_h_bytes + p.tab_v_bytes;
  if (hipMemcpyAsync(tab_h, th.data.data(), th.data.size() * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(tab_v, tv.data.data(), tv.data.size() * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) {
    rart_set_error("rart_pil_resize_u8: coefficient upload failed");
    return RART_ERR_HIP;
  }
  hipLaunchKernelGGL(k_resample_h, dim3(grid_for((size_t)n * p.y_count * crop_w * 3)), dim3(kBlock), 0, st, in, tmp,
                     (const int*)tab_h, p.ks_h, n, h, w, crop_x, crop_w, p.y_first, p.y_count);
  hipLaunchKernelGGL(k_resample_v, dim3(grid_for((size_t)n * crop_h * crop_w * 3)), dim3(kBlock), 0, st,
                     (const uint8_t*)tmp, out, (const int*)tab_v, p.ks_v, n, p.y_first, p.y_count, crop_y, crop_h, crop_w);
  RART_CHECK_LAUNCH("rart_pil_resize_u8");
  return RART_OK;
}

}  // extern "C"
