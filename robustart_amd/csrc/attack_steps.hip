// Adversarial step kernels for gfx950: PGD-Linf / FGSM, PGD-L2, MIM, APGD (Linf, L2),
// random starts, per-sample row selection and the row-wise logit losses.
// Reference: RobustART/noise/utils/adv/attack.py:20-42,
//            RobustART/noise/utils/adv/Attacks/imfgsm_attack.py:62-93,
//            RobustART/noise/utils/adv/Attacks/autoattack/autopgd_base.py:167-448,599-604.
//
// Every kernel is HBM-bound elementwise work (16 B/element for a PGD step); per-sample norms
// use a deterministic two-level reduction: RCH partial sums per sample written to the caller's
// workspace, re-summed in fixed order by the consuming kernel (no float atomics, so results
// do not depend on scheduling).  FMA contraction is off so each kernel reproduces the
// op-by-op fp32 rounding of the reference's unfused torch expressions.
#include "rart_common.h"

#pragma clang fp contract(off)

namespace {
constexpr int kBlock = 256;
constexpr int RCH = 32;  // reduction chunks per sample

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ float signf(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }
__device__ __forceinline__ float u01(uint32_t w) { return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = rart_wave_sum(v);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) {
    for (int i = 0; i < kBlock / 64; ++i) t += sh[i];
  }
  __syncthreads();
  return t;  // valid on thread 0
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = rart_wave_max(v);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) {
    t = sh[0];
    for (int i = 1; i < kBlock / 64; ++i) t = fmaxf(t, sh[i]);
  }
  __syncthreads();
  return t;
}
__device__ __forceinline__ float sum_partials(const float* part) {
  float t = 0.f;
  for (int i = 0; i < RCH; ++i) t += part[i];
  return t;
}
__device__ __forceinline__ float max_partials(const float* part) {
  float t = part[0];
  for (int i = 1; i < RCH; ++i) t = fmaxf(t, part[i]);
  return t;
}

// chunk geometry: grid = (RCH, batch); chunk c of sample b covers [c*len, min((c+1)*len, n))
struct Chunk {
  size_t begin, end;
};
__device__ __forceinline__ Chunk chunk_of(size_t n) {
  const size_t len = (n + RCH - 1) / RCH;
  Chunk c;
  c.begin = (size_t)blockIdx.x * len;
  c.end = c.begin + len < n ? c.begin + len : n;
  if (c.begin > n) c.begin = n;
  return c;
}

// uniform in (-1, 1) for element e of a sample (2 elements per Threefry call)
__device__ __forceinline__ float native_pm1(uint32_t k0, uint32_t k1, size_t e, uint32_t sample) {
  const uint2 w = threefry2x32(k0, k1, rart_ctr0((uint32_t)(e >> 1), 1), sample);
  return 2.0f * u01((e & 1) ? w.y : w.x) - 1.0f;
}
__device__ __forceinline__ float native_normal(uint32_t k0, uint32_t k1, size_t e, uint32_t sample) {
  const float4 z = rart_normal4(k0, k1, (uint32_t)(e >> 2), 2, sample);
  const int j = (int)(e & 3);
  return j == 0 ? z.x : (j == 1 ? z.y : (j == 2 ? z.z : z.w));
}

// Global sample index of batch row b: rows[b] when the caller passes the per-row index tensor (a still-robust SUBSET of a batch inside
// AutoAttack: row k of the subset draws at its own sample's index, so the draws do not depend on how a dataset is batched or sharded),
// else the contiguous sample_offset + b.
__device__ __forceinline__ uint32_t row_sample(const int64_t* __restrict__ rows, uint32_t sbase, uint32_t b) {
  return rows ? (uint32_t)rows[b] : sbase + b;
}

// ---- random starts -------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_init_linf(float* __restrict__ x, const float* __restrict__ x0,
                                                      size_t nps, float eps, float lo, float hi, uint32_t k0,
                                                      uint32_t k1, uint32_t sbase, const float* __restrict__ inj,
                                                      const int64_t* __restrict__ rows) {
  const uint32_t b = blockIdx.y;
  const size_t base = (size_t)b * nps;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < nps; e += (size_t)gridDim.x * kBlock) {
    const float u = inj ? inj[base + e] : eps * native_pm1(k0, k1, e, row_sample(rows, sbase, b));
    float v = x0[base + e] + u;
    if (lo <= hi) v = clampf(v, lo, hi);
    x[base + e] = v;
  }
}

// APGD start, pass 1: per-sample max|t| (Linf) or sum t^2 (L2) partials
template <int NORM>
__global__ __launch_bounds__(kBlock) void k_apgd_init_reduce(float* __restrict__ part, size_t nps, uint32_t k0,
                                                             uint32_t k1, uint32_t sbase,
                                                             const float* __restrict__ inj, const int64_t* __restrict__ rows) {
  __shared__ float sh[kBlock / 64];
  const uint32_t b = blockIdx.y;
  const Chunk c = chunk_of(nps);
  float acc = 0.f;
  for (size_t e = c.begin + threadIdx.x; e < c.end; e += kBlock) {
    const float t = inj ? inj[(size_t)b * nps + e]
                        : (NORM == 0 ? native_pm1(k0, k1, e, row_sample(rows, sbase, b)) : native_normal(k0, k1, e, row_sample(rows, sbase, b)));
    acc = NORM == 0 ? fmaxf(acc, fabsf(t)) : (NORM == 1 ? acc + t * t : acc + fabsf(t));      // NORM 2: FAB's L1 start, sum |t|
  }
  const float r = NORM == 0 ? block_max(acc, sh) : block_sum(acc, sh);
  if (threadIdx.x == 0) part[(size_t)b * RCH + blockIdx.x] = r;
}
template <int NORM>
__global__ __launch_bounds__(kBlock) void k_apgd_init_apply(float* __restrict__ x, const float* __restrict__ x0,
                                                            const float* __restrict__ part, size_t nps, float eps,
                                                            uint32_t k0, uint32_t k1, uint32_t sbase,
                                                            const float* __restrict__ inj, const int64_t* __restrict__ rows) {
  const uint32_t b = blockIdx.y;
  const float red = NORM == 0 ? max_partials(part + (size_t)b * RCH)
                              : (NORM == 1 ? sqrtf(sum_partials(part + (size_t)b * RCH)) : sum_partials(part + (size_t)b * RCH));
  const float denom = red + 1e-12f;
  const size_t base = (size_t)b * nps;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < nps; e += (size_t)gridDim.x * kBlock) {
    const float t = inj ? inj[base + e]
                        : (NORM == 0 ? native_pm1(k0, k1, e, row_sample(rows, sbase, b)) : native_normal(k0, k1, e, row_sample(rows, sbase, b)));
    const float tn = t / denom;
    const float v = x0[base + e] + eps * tn;  // eps * ones_like(x) * normalize(t)
    x[base + e] = clampf(v, 0.f, 1.f);
  }
}

// ---- PGD-Linf / FGSM step: 16 B/element, float4 lanes ----------------------------------
__global__ __launch_bounds__(kBlock) void k_pgd_linf(float4* __restrict__ x, const float4* __restrict__ g,
                                                     const float4* __restrict__ x0, size_t nvec, float eps,
                                                     float alpha) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += (size_t)gridDim.x * kBlock) {
    const float4 xv = x[i], gv = g[i], ov = x0[i];
    float4 r;
#define STEP(c)                                              \
  {                                                          \
    const float s = alpha * signf(gv.c);                     \
    const float x1 = xv.c + s;                               \
    const float d = clampf(x1 - ov.c, -eps, eps);            \
    r.c = clampf(ov.c + d, 0.f, 1.f);                        \
  }
    STEP(x) STEP(y) STEP(z) STEP(w)
#undef STEP
    x[i] = r;
  }
}
__global__ __launch_bounds__(kBlock) void k_pgd_linf_tail(float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ x0, size_t begin, size_t n,
                                                          float eps, float alpha) {
  const size_t i = begin + (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    const float x1 = x[i] + alpha * signf(g[i]);
    const float d = clampf(x1 - x0[i], -eps, eps);
    x[i] = clampf(x0[i] + d, 0.f, 1.f);
  }
}

// ---- generic per-sample sum-of-squares / sum-abs partials -----------------------------
// MODE 0: sum g^2     MODE 1: sum |g|
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_reduce_rows(const float* __restrict__ g, float* __restrict__ part,
                                                        size_t nps) {
  __shared__ float sh[kBlock / 64];
  const uint32_t b = blockIdx.y;
  const Chunk c = chunk_of(nps);
  float acc = 0.f;
  for (size_t e = c.begin + threadIdx.x; e < c.end; e += kBlock) {
    const float v = g[(size_t)b * nps + e];
    acc += MODE == 0 ? v * v : fabsf(v);
  }
  const float r = block_sum(acc, sh);
  if (threadIdx.x == 0) part[(size_t)b * RCH + blockIdx.x] = r;
}

// ---- PGD-L2 (foolbox) ------------------------------------------------------------------
// pass B: x <- x + alpha * g / max(|g|,1e-12)   and partial sums of (x - x0)^2
__global__ __launch_bounds__(kBlock) void k_pgd_l2_move(float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ x0,
                                                        const float* __restrict__ gpart, float* __restrict__ dpart,
                                                        size_t nps, float alpha) {
  __shared__ float sh[kBlock / 64];
  const uint32_t b = blockIdx.y;
  const float gn = fmaxf(sqrtf(sum_partials(gpart + (size_t)b * RCH)), 1e-12f);
  const Chunk c = chunk_of(nps);
  float acc = 0.f;
  for (size_t e = c.begin + threadIdx.x; e < c.end; e += kBlock) {
    const size_t i = (size_t)b * nps + e;
    const float gg = g[i] / gn;
    const float x1 = x[i] + alpha * gg;
    x[i] = x1;
    const float d = x1 - x0[i];
    acc += d * d;
  }
  const float r = block_sum(acc, sh);
  if (threadIdx.x == 0) dpart[(size_t)b * RCH + blockIdx.x] = r;
}
__global__ __launch_bounds__(kBlock) void k_pgd_l2_project(float* __restrict__ x, const float* __restrict__ x0,
                                                           const float* __restrict__ dpart, size_t nps, float eps) {
  const uint32_t b = blockIdx.y;
  const float dn = fmaxf(sqrtf(sum_partials(dpart + (size_t)b * RCH)), 1e-12f);
  const float factor = fminf(eps / dn, 1.0f);
  const size_t base = (size_t)b * nps;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < nps; e += (size_t)gridDim.x * kBlock) {
    const float d = x[base + e] - x0[base + e];
    x[base + e] = clampf(x0[base + e] + d * factor, 0.f, 1.f);
  }
}

// ---- PGD-L1 (ART ProjectedGradientDescentPyTorch, norm = 1; adv/attack.py:44-49) ---------------------------
// ART (unpinned in requirements.txt:25; algorithm of the 1.x series): perturbation = g / (sum|g| + tol),
// x <- clip(x + eps_step * perturbation, 0, 1), then delta <- delta * min(1, eps / (sum|delta| + tol)), tol = 10e-8.
constexpr float kArtTol = 10e-8f;
__global__ __launch_bounds__(kBlock) void k_pgd_l1_move(float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ x0,
                                                        const float* __restrict__ gpart, float* __restrict__ dpart,
                                                        size_t nps, float eps_step) {
  __shared__ float sh[kBlock / 64];
  const uint32_t b = blockIdx.y;
  const float gn = sum_partials(gpart + (size_t)b * RCH) + kArtTol;
  const Chunk c = chunk_of(nps);
  float acc = 0.f;
  for (size_t e = c.begin + threadIdx.x; e < c.end; e += kBlock) {
    const size_t i = (size_t)b * nps + e;
    const float x1 = clampf(x[i] + eps_step * (g[i] / gn), 0.f, 1.f);
    x[i] = x1;
    acc += fabsf(x1 - x0[i]);
  }
  const float r = block_sum(acc, sh);
  if (threadIdx.x == 0) dpart[(size_t)b * RCH + blockIdx.x] = r;
}
__global__ __launch_bounds__(kBlock) void k_pgd_l1_project(float* __restrict__ x, const float* __restrict__ x0,
                                                           const float* __restrict__ dpart, size_t nps, float eps) {
  const uint32_t b = blockIdx.y;
  const float factor = fminf(1.0f, eps / (sum_partials(dpart + (size_t)b * RCH) + kArtTol));
  const size_t base = (size_t)b * nps;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < nps; e += (size_t)gridDim.x * kBlock)
    x[base + e] = (x[base + e] - x0[base + e]) * factor + x0[base + e];
}
// ART random_sphere(norm=1): a uniform point of the L1 sphere of radius r = sqrt(U(0, eps^2)): spacings of
// sorted uniforms x random signs.  Spacings of n-1 sorted uniforms are Dirichlet(1,..,1) = normalised Exp(1)
// draws, which is how they are formed here (no 150 527-key sort per sample).  injected: [batch][nps] signed
// exponentials (sign * e_i) and injected_r[batch] radii for the parity tests.
__global__ __launch_bounds__(kBlock) void k_l1_start_reduce(float* __restrict__ part, size_t nps, uint32_t k0, uint32_t k1,
                                                            uint32_t sbase, const float* __restrict__ inj, const int64_t* __restrict__ rows) {
  __shared__ float sh[kBlock / 64];
  const uint32_t b = blockIdx.y;
  const Chunk c = chunk_of(nps);
  float acc = 0.f;
  for (size_t e = c.begin + threadIdx.x; e < c.end; e += kBlock) {
    float v;
    if (inj) v = fabsf(inj[(size_t)b * nps + e]);
    else {
      const uint2 w = threefry2x32(k0, k1, rart_ctr0((uint32_t)e, 5), row_sample(rows, sbase, b));
      v = -logf(u01(w.x));
    }
    acc += v;
  }
  const float r = block_sum(acc, sh);
  if (threadIdx.x == 0) part[(size_t)b * RCH + blockIdx.x] = r;
}
__global__ __launch_bounds__(kBlock) void k_l1_start_apply(float* __restrict__ x, const float* __restrict__ x0,
                                                           const float* __restrict__ part, size_t nps, float eps,
                                                           uint32_t k0, uint32_t k1, uint32_t sbase,
                                                           const float* __restrict__ inj, const float* __restrict__ inj_r,
                                                           const int64_t* __restrict__ rows) {
  const uint32_t b = blockIdx.y;
  float r;
  if (inj_r) r = inj_r[b];
  else {
    const uint2 w = threefry2x32(k0, k1, rart_ctr0(0xFFFFFFFu, 5), row_sample(rows, sbase, b));
    r = sqrtf(u01(w.y) * eps * eps);
  }
  const float scale = r / sum_partials(part + (size_t)b * RCH);
  const size_t base = (size_t)b * nps;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < nps; e += (size_t)gridDim.x * kBlock) {
    float v;
    if (inj) v = inj[base + e];
    else {
      const uint2 w = threefry2x32(k0, k1, rart_ctr0((uint32_t)e, 5), row_sample(rows, sbase, b));
      v = -logf(u01(w.x));
      if (w.y & 1u) v = -v;
    }
    x[base + e] = clampf(x0[base + e] + v * scale, 0.f, 1.f);
  }
}

// ---- MIM -------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_mim_apply(float* __restrict__ x, float* __restrict__ m,
                                                      const float* __restrict__ g, const float* __restrict__ x0,
                                                      const float* __restrict__ part, size_t nps, float eps,
                                                      float step, float decay) {
  const uint32_t b = blockIdx.y;
  const float mean_abs = sum_partials(part + (size_t)b * RCH) / (float)nps;
  const size_t base = (size_t)b * nps;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < nps; e += (size_t)gridDim.x * kBlock) {
    const size_t i = base + e;
    const float gn = g[i] / mean_abs;
    const float mm = decay * m[i] + gn;
    m[i] = mm;
    const float x1 = x[i] + step * signf(mm);
    const float eta = clampf(x1 - x0[i], -eps, eps);
    x[i] = clampf(x0[i] + eta, 0.f, 1.f);
  }
}

// ---- APGD step -------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_apgd_linf(float* __restrict__ xa, float* __restrict__ xold,
                                                      const float* __restrict__ grad, const float* __restrict__ x0,
                                                      const float* __restrict__ step, size_t nps, float eps, float a) {
  const uint32_t b = blockIdx.y;
  const float ss = step[b];
  const float oma = 1.0f - a;
  const size_t base = (size_t)b * nps;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < nps; e += (size_t)gridDim.x * kBlock) {
    const size_t i = base + e;
    const float xv = xa[i], xo = xold[i], x = x0[i];
    const float grad2 = xv - xo;
    float x1 = xv + ss * signf(grad[i]);
    x1 = clampf(fminf(fmaxf(x1, x - eps), x + eps), 0.f, 1.f);
    const float t1 = (x1 - xv) * a;
    const float t2 = grad2 * oma;
    float x2 = (xv + t1) + t2;
    x2 = clampf(fminf(fmaxf(x2, x - eps), x + eps), 0.f, 1.f);
    xold[i] = xv;
    xa[i] = x2;
  }
}

// L2 variant, 4 passes.  STAGE 0: partial |x1' - x|^2 ; STAGE 1: partial |x2' - x|^2 ; STAGE 2: apply.
__device__ __forceinline__ float apgd_l2_x1(float xv, float g, float x, float ss, float gnorm, float n1, float eps) {
  // x_adv_1 = clamp(x + normalize(x1' - x) * min(eps, |x1' - x|), 0, 1), x1' = x_adv + step*g/(|g|+1e-12)
  const float x1p = xv + ss * (g / (gnorm + 1e-12f));
  const float d = x1p - x;
  return clampf(x + (d / (n1 + 1e-12f)) * fminf(eps, n1), 0.f, 1.f);
}
template <int STAGE>
__global__ __launch_bounds__(kBlock) void k_apgd_l2(float* __restrict__ xa, float* __restrict__ xold,
                                                    const float* __restrict__ grad, const float* __restrict__ x0,
                                                    const float* __restrict__ step, float* __restrict__ ws,
                                                    int batch, size_t nps, float eps, float a) {
  __shared__ float sh[kBlock / 64];
  const uint32_t b = blockIdx.y;
  const float ss = step[b];
  const float* gpart = ws;
  float* n1part = ws + (size_t)batch * RCH;
  float* n2part = ws + (size_t)2 * batch * RCH;
  const float gnorm = sqrtf(sum_partials(gpart + (size_t)b * RCH));
  const float n1 = STAGE >= 1 ? sqrtf(sum_partials(n1part + (size_t)b * RCH)) : 0.f;
  const float n2 = STAGE >= 2 ? sqrtf(sum_partials(n2part + (size_t)b * RCH)) : 0.f;
  const float oma = 1.0f - a;
  const Chunk c = chunk_of(nps);
  float acc = 0.f;
  for (size_t e = c.begin + threadIdx.x; e < c.end; e += kBlock) {
    const size_t i = (size_t)b * nps + e;
    const float xv = xa[i], x = x0[i], g = grad[i];
    if (STAGE == 0) {
      const float x1p = xv + ss * (g / (gnorm + 1e-12f));
      const float d = x1p - x;
      acc += d * d;
    } else {
      const float x1 = apgd_l2_x1(xv, g, x, ss, gnorm, n1, eps);
      const float grad2 = xv - xold[i];
      const float x2p = (xv + (x1 - xv) * a) + grad2 * oma;
      const float d = x2p - x;
      if (STAGE == 1) {
        acc += d * d;
      } else {
        xold[i] = xv;
        xa[i] = clampf(x + (d / (n2 + 1e-12f)) * fminf(eps, n2), 0.f, 1.f);
      }
    }
  }
  if (STAGE < 2) {
    const float r = block_sum(acc, sh);
    if (threadIdx.x == 0) (STAGE == 0 ? n1part : n2part)[(size_t)b * RCH + blockIdx.x] = r;
  }
}

// ---- per-sample masked row copy ----------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_select_rows(float* __restrict__ dst, const float* __restrict__ src,
                                                        const uint8_t* __restrict__ mask, size_t nps) {
  const uint32_t b = blockIdx.y;
  if (!mask[b]) return;
  const size_t base = (size_t)b * nps;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < nps; e += (size_t)gridDim.x * kBlock)
    dst[base + e] = src[base + e];
}

// ---- Square attack, Linf (Attacks/autoattack/square.py:228-258) ---------------------------------
// start: x_best = clamp(x + eps * sign_stripes[b][c][w], 0, 1) (one sign per image column and channel)
__global__ __launch_bounds__(kBlock) void k_square_init(float* __restrict__ xb, const float* __restrict__ x0, int C,
                                                        int H, int W, float eps, uint32_t k0, uint32_t k1,
                                                        uint32_t sbase, const float* __restrict__ inj, const int64_t* __restrict__ rows) {
  const uint32_t b = blockIdx.y;
  const size_t nps = (size_t)C * H * W;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < nps; e += (size_t)gridDim.x * kBlock) {
    const int w = (int)(e % W), c = (int)(e / ((size_t)H * W));
    float sg;
    if (inj) {
      sg = inj[((size_t)b * C + c) * W + w];
    } else {
      const uint2 wv = threefry2x32(k0, k1, rart_ctr0((uint32_t)(c * W + w), 3), row_sample(rows, sbase, b));
      sg = (wv.x & 1u) ? 1.f : -1.f;
    }
    xb[b * nps + e] = clampf(x0[b * nps + e] + eps * sg, 0.f, 1.f);
  }
}
// proposal: x_new = clamp(min(max(x_best + delta, x - eps), x + eps), 0, 1), delta = 2*eps*sign[c] inside the
// s x s window at (vh, vw) -- the window and signs are shared by the whole batch (square.py:248-252)
struct SquareSigns {
  float s[8];
};
__global__ __launch_bounds__(kBlock) void k_square_propose(float* __restrict__ xn, const float* __restrict__ xb,
                                                           const float* __restrict__ x0, size_t total, int C, int H,
                                                           int W, float eps, int vh, int vw, int s, SquareSigns sg) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const int w = (int)(i % W), h = (int)((i / W) % H), c = (int)((i / ((size_t)W * H)) % C);
    const bool in = h >= vh && h < vh + s && w >= vw && w < vw + s;
    const float two_eps = 2.f * eps;
    const float d = in ? two_eps * sg.s[c] : 0.f;
    float v = xb[i] + d;
    const float x = x0[i];
    v = fminf(fmaxf(v, x - eps), x + eps);
    xn[i] = clampf(v, 0.f, 1.f);
  }
}

// ---- FAB (Linf) ---------------------------------------------------------------------------------
// Box-constrained Linf projection of a point t onto the hyperplane {x : w.x = b}
// (Attacks/autoattack/fab_projections.py:7-59).  With sign chosen so that w.t - b >= 0, a_i = [w_i < 0],
// p_i = room of coordinate i towards its helpful bound (1 - t_i or t_i), the step is
//     d_i = (2 a_i - 1) * min(lambda, p_i) * [w_i != 0]
// where lambda solves F(lambda) = sum_i |w_i| min(lambda, p_i) = |w.t - b| (F is monotone, piecewise linear).
// The reference finds the segment by argsort(p) + cumsum + binary search over 150 528 keys per row; here one
// workgroup per row brackets lambda by bisection on F (each evaluation is a row reduction; the 1.2 MB row stays
// in L2) and then solves the active segment exactly -- no sort.  Also returns max_i |d_i| (fab_base.py:196).
constexpr int kFabThreads = 512;
__device__ __forceinline__ double fab_block_sum(double v, double* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < kFabThreads / 64; ++i) t += sh[i];
  return t;   // valid on every thread
}
__device__ __forceinline__ float fab_block_min(float v, double* sh, bool want_max) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off, 64);
    v = want_max ? fmaxf(v, o) : fminf(v, o);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = (double)v;
  __syncthreads();
  float t = (float)sh[0];
  for (int i = 1; i < kFabThreads / 64; ++i) t = want_max ? fmaxf(t, (float)sh[i]) : fminf(t, (float)sh[i]);
  return t;
}

__global__ __launch_bounds__(kFabThreads) void k_fab_project_linf(const float* __restrict__ pts,
                                                                  const float* __restrict__ wv,
                                                                  const float* __restrict__ bv, float* __restrict__ dout,
                                                                  float* __restrict__ rowmax, size_t n) {
  __shared__ double sh[kFabThreads / 64];
  const size_t row = blockIdx.x;
  const float* t = pts + row * n;
  const float* w = wv + row * n;
  float* d = dout + row * n;
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) acc += (double)(w[i] * t[i]);
  const double wt = fab_block_sum(acc, sh);
  const float sgn = (wt - (double)bv[row] >= 0.0) ? 1.f : -1.f;
  const double target = fabs(wt - (double)bv[row]);
  double wsum = 0.0, ftot = 0.0;
  float pmin = INFINITY;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    const float wi = sgn * w[i];
    const float p = wi < 0.f ? 1.f - t[i] : t[i];
    pmin = fminf(pmin, p);
    wsum += (double)fabsf(wi);
    ftot += (double)(fabsf(wi) * p);
  }
  wsum = fab_block_sum(wsum, sh);
  ftot = fab_block_sum(ftot, sh);
  pmin = fab_block_min(pmin, sh, false);
  float lam;
  if (wsum * (double)pmin > target) {          // c_l: no coordinate saturates
    lam = (float)fmax(target / wsum, 0.0);
  } else if (ftot > target) {                   // c2: some coordinates saturate at their bound
    float lo = 0.f, hi = 1.f;
    for (int it = 0; it < 26; ++it) {
      const float mid = 0.5f * (lo + hi);
      double f = 0.0;
      for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
        const float wi = sgn * w[i];
        const float p = wi < 0.f ? 1.f - t[i] : t[i];
        f += (double)(fabsf(wi) * fminf(mid, p));
      }
      f = fab_block_sum(f, sh);
      if (f > target) hi = mid; else lo = mid;
    }
    double sat = 0.0, act = 0.0;                // exact solve of the segment: saturated p <= lo, active p > lo
    for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
      const float wi = sgn * w[i];
      const float p = wi < 0.f ? 1.f - t[i] : t[i];
      if (p <= lo) sat += (double)(fabsf(wi) * p); else act += (double)fabsf(wi);
    }
    sat = fab_block_sum(sat, sh);
    act = fab_block_sum(act, sh);
    lam = act > 0.0 ? (float)fmax((target - sat) / act, 0.0) : hi;
  } else {
    lam = INFINITY;                             // hyperplane out of reach inside the box: go to the bounds
  }
  float mx = 0.f;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    const float wi = sgn * w[i];
    const float p = wi < 0.f ? 1.f - t[i] : t[i];
    const float dv = wi == 0.f ? 0.f : (wi < 0.f ? fminf(lam, p) : -fminf(lam, p));
    d[i] = dv;
    mx = fmaxf(mx, fabsf(dv));
  }
  mx = fab_block_min(mx, sh, true);
  if (threadIdx.x == 0 && rowmax) rowmax[row] = mx;
}

// ---- FAB (L2, L1) ----------------------------------------------------------------------------------
// Box-constrained L2 projection (fab_projections.py:62-117).  With the sign chosen so that c = w.t - b >= 0, coordinate i can
// follow the steepest direction d_i = -alpha w_i until alpha reaches r_i = max(t_i / w_i, (t_i - 1) / w_i), where it meets the
// box; the step is d_i = -min(alpha, r_i) w_i and alpha solves F(alpha) = sum_i w_i^2 min(alpha, r_i) = c (monotone, piecewise
// linear).  The reference sorts r, builds two cumulative sums and binary-searches the segment; here one workgroup per row
// bisects on the BIT PATTERN of alpha (non-negative floats order like their bits, r spans 0 .. 1e8 so a linear bisection
// would lose the small roots), then solves the segment exactly.  Coordinates with |w_i| <= 1e-8 do not move (:67,:117).
// Also returns ||d||_2 (fab_base.py:198-200).
__device__ __forceinline__ float fab_l2_reach(float t, float w) {          // r_i, for |w| > 1e-8 (finite, >= 0)
  return fmaxf(__fdiv_rn(t, w), __fdiv_rn(t - 1.f, w));
}
__global__ __launch_bounds__(kFabThreads) void k_fab_project_l2(const float* __restrict__ pts, const float* __restrict__ wv,
                                                                const float* __restrict__ bv, float* __restrict__ dout,
                                                                float* __restrict__ rownorm, size_t n) {
  __shared__ double sh[kFabThreads / 64];
  const size_t row = blockIdx.x;
  const float* t = pts + row * n;
  const float* w = wv + row * n;
  float* d = dout + row * n;
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) acc += (double)(w[i] * t[i]);
  const double wt = fab_block_sum(acc, sh);
  const float sgn = (wt - (double)bv[row] >= 0.0) ? 1.f : -1.f;
  const double c = fabs(wt - (double)bv[row]);
  double w5 = 0.0, fmax_ = 0.0;
  float rmin = INFINITY, rmax = 0.f;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    const float wi = sgn * w[i];
    w5 += (double)(wi * wi);
    if (fabsf(wi) > 1e-8f) {
      const float r = fab_l2_reach(t[i], wi);
      rmin = fminf(rmin, r);
      rmax = fmaxf(rmax, r);
      fmax_ += (double)(wi * wi) * (double)r;
    }
  }
  w5 = fab_block_sum(w5, sh);
  fmax_ = fab_block_sum(fmax_, sh);
  rmin = fab_block_min(rmin, sh, false);
  rmax = fab_block_min(rmax, sh, true);
  float alpha;
  if (rmin == INFINITY) {
    alpha = 0.f;                                      // no coordinate can move
  } else if (c < w5 * (double)rmin) {                 // c4 (:87,:104-106): nobody reaches the box
    alpha = (float)(c / w5);
  } else if (c > fmax_) {                             // c3 (:88): out of reach, every coordinate goes to its bound
    alpha = INFINITY;
  } else {                                            // c2 (:89-102,:108-112)
    uint32_t lo = __float_as_uint(rmin), hi = __float_as_uint(rmax);      // F(lo) <= c <= F(hi)
    while (hi - lo > 1u) {
      const uint32_t midb = lo + ((hi - lo) >> 1);
      const float mid = __uint_as_float(midb);
      double f = 0.0;
      for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
        const float wi = sgn * w[i];
        if (fabsf(wi) > 1e-8f) f += (double)(wi * wi) * (double)fminf(mid, fab_l2_reach(t[i], wi));
      }
      f = fab_block_sum(f, sh);
      if (f > c) hi = midb; else lo = midb;
    }
    const float lof = __uint_as_float(lo);
    double sat = 0.0, act = 0.0;                      // the segment [lo, hi] holds no breakpoint inside: solve it exactly
    for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
      const float wi = sgn * w[i];
      if (fabsf(wi) > 1e-8f) {
        const float r = fab_l2_reach(t[i], wi);
        if (r <= lof) sat += (double)(wi * wi) * (double)r; else act += (double)(wi * wi);
      }
    }
    sat = fab_block_sum(sat, sh);
    act = fab_block_sum(act, sh);
    alpha = act > 0.0 ? fminf(fmaxf((float)((c - sat) / act), lof), __uint_as_float(hi)) : lof;
  }
  double nn = 0.0;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    const float wi = sgn * w[i];
    float dv = 0.f;
    if (fabsf(wi) > 1e-8f) dv = -fminf(alpha, fab_l2_reach(t[i], wi)) * wi;
    d[i] = dv;
    nn += (double)dv * (double)dv;
  }
  nn = fab_block_sum(nn, sh);
  if (threadIdx.x == 0 && rownorm) rownorm[row] = (float)sqrt(nn);
}

// Box-constrained L1 projection (fab_projections.py:120-166).  With c = w.t - b >= 0, sending coordinate i to its helpful
// bound (0 when w_i > 0, 1 when w_i < 0) lowers w.x by gain_i = |w_i| * room_i; the L1-cheapest way to remove c is greedy in
// decreasing |w_i|: the first coordinates go all the way, one goes part of the way, the rest stay.  The reference argsorts
// 1 / |w| and binary-searches the cumulative gains; here one workgroup per row finds the |w| of the partial coordinate by
// bisection on its bit pattern (kappa = the largest key with sum_{|w_i| >= kappa} gain_i >= c), ties between equal |w| are
// taken in index order (a second bisection, on the index, only when there are ties).  Also returns ||d||_1 (fab_base.py:201-203).
__device__ __forceinline__ float fab_l1_gain(float t, float wi) { return wi > 0.f ? wi * t : -wi * (1.f - t); }
__global__ __launch_bounds__(kFabThreads) void k_fab_project_l1(const float* __restrict__ pts, const float* __restrict__ wv,
                                                                const float* __restrict__ bv, float* __restrict__ dout,
                                                                float* __restrict__ rownorm, size_t n) {
  __shared__ double sh[kFabThreads / 64];
  const size_t row = blockIdx.x;
  const float* t = pts + row * n;
  const float* w = wv + row * n;
  float* d = dout + row * n;
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) acc += (double)(w[i] * t[i]);
  const double wt = fab_block_sum(acc, sh);
  const float sgn = (wt - (double)bv[row] >= 0.0) ? 1.f : -1.f;
  const double c = fabs(wt - (double)bv[row]);
  double total = 0.0;
  float wmax = 0.f;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    const float wi = sgn * w[i];
    total += (double)fab_l1_gain(t[i], wi);
    wmax = fmaxf(wmax, fabsf(wi));
  }
  total = fab_block_sum(total, sh);
  wmax = fab_block_min(wmax, sh, true);
  uint32_t kappa = 0xFFFFFFFFu;                       // unreachable (:135 c2 false): everything with w != 0 goes to its bound
  size_t tie_end = 0;                                 // with ties: tied coordinates of index < tie_end go all the way, tie_end part of it
  bool have_ties = false;
  double before = 0.0;                                // gain of everything taken all the way
  if (total > c) {
    uint32_t lo = 0u, hi = __float_as_uint(wmax) + 1u;                   // G(lo) >= c > G(hi) = 0, G(k) = sum_{key >= k} gain
    while (hi - lo > 1u) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      double g = 0.0;
      for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
        const float wi = sgn * w[i];
        if ((__float_as_uint(wi) & 0x7FFFFFFFu) >= mid) g += (double)fab_l1_gain(t[i], wi);
      }
      g = fab_block_sum(g, sh);
      if (g >= c && g > 0.0) lo = mid; else hi = mid;
    }
    kappa = lo;
    double above = 0.0, ties = 0.0;
    for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
      const float wi = sgn * w[i];
      const uint32_t key = __float_as_uint(wi) & 0x7FFFFFFFu;
      if (key > kappa) above += (double)fab_l1_gain(t[i], wi);
      ties += key == kappa ? 1.0 : 0.0;
    }
    above = fab_block_sum(above, sh);
    ties = fab_block_sum(ties, sh);
    before = above;
    have_ties = ties > 1.0;
    if (have_ties) {                                  // the smallest m with above + sum_{tied, i < m} gain_i >= c is one past the partial coordinate
      size_t mlo = 0, mhi = n;                        // P(mlo) < c <= P(mhi)
      while (mhi - mlo > 1) {
        const size_t mid = mlo + ((mhi - mlo) >> 1);
        double g = 0.0;
        for (size_t i = threadIdx.x; i < mid; i += kFabThreads) {
          const float wi = sgn * w[i];
          if ((__float_as_uint(wi) & 0x7FFFFFFFu) == kappa) g += (double)fab_l1_gain(t[i], wi);
        }
        g = fab_block_sum(g, sh);
        if (above + g >= c) mhi = mid; else mlo = mid;
      }
      tie_end = mhi - 1;                              // index of the partial coordinate (tied or not; non-tied ones add nothing)
      double g = 0.0;
      for (size_t i = threadIdx.x; i < tie_end; i += kFabThreads) {
        const float wi = sgn * w[i];
        if ((__float_as_uint(wi) & 0x7FFFFFFFu) == kappa) g += (double)fab_l1_gain(t[i], wi);
      }
      before = above + fab_block_sum(g, sh);
    }
  }
  double nn = 0.0;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    const float wi = sgn * w[i];
    const uint32_t key = __float_as_uint(wi) & 0x7FFFFFFFu;
    float dv = 0.f;
    if (fabsf(wi) > 1e-8f) {                          // (:166; also w != 0 of :131)
      const float full = wi < 0.f ? 1.f - t[i] : -t[i];
      if (key > kappa || kappa == 0xFFFFFFFFu) dv = full;
      else if (key == kappa) {
        if (have_ties && i < tie_end) dv = full;
        else if (!have_ties || i == tie_end) dv = (float)(-(c - before) / (double)wi);   // alpha of :158: one coordinate only
      }
    }
    d[i] = dv;
    nn += (double)fabsf(dv);
  }
  nn = fab_block_sum(nn, sh);
  if (threadIdx.x == 0 && rownorm) rownorm[row] = (float)nn;
}

// ---- APGD, L1 threat model (autopgd_base.py:19-83, 222-226, 351-364, 431-441) -----------------------------------------
// L1_projection: delta with ||y + delta||_1 <= eps and 0 <= x + y + delta <= 1.  Per coordinate |y_i| shrinks by
// clip(alpha, lo_i, a_i), a_i = |y_i|, lo_i = -min(min(1 - x_i - y_i, x_i + y_i), 0) (what the box alone demands), and alpha
// is the root of G(alpha) = sum_i clip(alpha, lo_i, a_i) = sum_i a_i - eps (only when G(0) = sum lo_i is below that
// target).  The reference sorts the 2n breakpoints and binary-searches a cumulative sum; here one workgroup per row brackets
// alpha by bisection on G (a row reduction per evaluation, fp64 accumulation) and then solves the active linear segment
// exactly -- no sort, the same structure as k_fab_project_linf.  In place: out = x + y + delta (the projected point) when
// `point_out`, else out = delta.
__device__ __forceinline__ void l1_lo_hi(float xv, float yv, float& lo, float& a) {
  a = fabsf(yv);
  const float s = xv + yv;
  lo = -fminf(fminf(1.f - s, s), 0.f);
}
__global__ __launch_bounds__(kFabThreads) void k_l1_project(const float* __restrict__ xs, const float* __restrict__ ys,
                                                            float* __restrict__ out, size_t n, float eps, int point_out,
                                                            int clamp01) {
  __shared__ double sh[kFabThreads / 64];
  const size_t row = blockIdx.x;
  const float* x = xs + row * n;
  const float* y = ys + row * n;
  float* o = out + row * n;
  double sa = 0.0, slo = 0.0;
  float amax = 0.f;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    float lo, a;
    l1_lo_hi(x[i], y[i], lo, a);
    sa += (double)a;
    slo += (double)lo;
    amax = fmaxf(amax, a);
  }
  sa = fab_block_sum(sa, sh);
  slo = fab_block_sum(slo, sh);
  amax = fab_block_min(amax, sh, true);
  const double target = sa - (double)eps;           // = -c of the reference
  float alpha = 0.f;                                  // s1 + c >= 0: the box shrink alone is inside the ball (d = u)
  if (slo < target) {
    float blo = 0.f, bhi = amax;
    for (int it = 0; it < 30; ++it) {
      const float mid = 0.5f * (blo + bhi);
      double g = 0.0;
      for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
        float lo, a;
        l1_lo_hi(x[i], y[i], lo, a);
        g += (double)fminf(fmaxf(mid, lo), a);
      }
      g = fab_block_sum(g, sh);
      if (g > target) bhi = mid; else blo = mid;
    }
    // exact solve on the segment containing blo: coordinates with lo_i < blo < a_i move with alpha, the others are fixed
    double fixed = 0.0, act = 0.0;
    for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
      float lo, a;
      l1_lo_hi(x[i], y[i], lo, a);
      if (lo <= blo && blo < a) act += 1.0; else fixed += (double)fminf(fmaxf(blo, lo), a);
    }
    fixed = fab_block_sum(fixed, sh);
    act = fab_block_sum(act, sh);
    alpha = act > 0.0 ? (float)((target - fixed) / act) : bhi;
    alpha = fminf(fmaxf(alpha, blo), bhi);
  }
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    const float xv = x[i], yv = y[i];
    float lo, a;
    l1_lo_hi(xv, yv, lo, a);
    const float shrink = (slo < target) ? fminf(fmaxf(alpha, lo), a) : lo;      // d = -shrink
    const float sg = yv > 0.f ? 1.f : (yv < 0.f ? -1.f : 0.f);
    const float delta = -sg * shrink;
    float v = point_out ? (xv + yv) + delta : delta;
    if (point_out && clamp01) v = clampf(v, 0.f, 1.f);
    o[i] = v;
  }
}

// thr[r] = the k[r]-th smallest |g[r][.]| (0-based), EXACT: radix select on the float bit patterns, 4 passes of 8 bits,
// one workgroup per row (autopgd_base.py:352-354: grad.abs().sort()[..., topk_curr]).
__global__ __launch_bounds__(kFabThreads) void k_row_kth_abs(const float* __restrict__ gs, const int64_t* __restrict__ ks,
                                                             float* __restrict__ thr, size_t n) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_prefix, s_rank;
  const size_t row = blockIdx.x;
  const float* g = gs + row * n;
  if (threadIdx.x == 0) {
    long long k = ks[row];
    k = k < 0 ? 0 : (k > (long long)n - 1 ? (long long)n - 1 : k);
    s_prefix = 0u;
    s_rank = (uint32_t)k;
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t prefix = s_prefix, hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
      const uint32_t b = __float_as_uint(g[i]) & 0x7FFFFFFFu;      // |g|: non-negative floats order like their bits
      if ((b & hi_mask) == prefix) atomicAdd(&hist[(b >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t r = s_rank, d = 0;
      for (; d < 255; ++d) {
        if (r < hist[d]) break;
        r -= hist[d];
      }
      s_rank = r;
      s_prefix = prefix | (d << shift);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) thr[row] = __uint_as_float(s_prefix);
}

// The sparse sign step of APGD-L1 (autopgd_base.py:355-358): delta_u = x_adv + step * sign(sparse) / (count + 1e-10) - x0 with
// sparse = g * [|g| >= thr]; count = number of non-zero signs of the row.  Writes delta_u (the projection follows).
__global__ __launch_bounds__(kFabThreads) void k_apgd_l1_move(const float* __restrict__ xa, const float* __restrict__ gs,
                                                              const float* __restrict__ x0, const float* __restrict__ thr,
                                                              const float* __restrict__ step, float* __restrict__ du, size_t n) {
  __shared__ double sh[kFabThreads / 64];
  const size_t row = blockIdx.x;
  const float t = thr[row];
  double cnt = 0.0;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    const float g = gs[row * n + i];
    cnt += (fabsf(g) >= t && g != 0.f) ? 1.0 : 0.0;
  }
  cnt = fab_block_sum(cnt, sh);
  const float denom = (float)cnt + 1e-10f;
  const float ss = step[row];
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    const float g = gs[row * n + i];
    const float sg = (fabsf(g) >= t) ? signf(g) : 0.f;
    const float q = ss * sg;
    const float x1 = xa[row * n + i] + q / denom;
    du[row * n + i] = x1 - x0[row * n + i];
  }
}

// out[r] = number of i with a[r][i] != b[r][i]  (L0_norm(x_best - x), other_utils.py:42-43)
__global__ __launch_bounds__(kFabThreads) void k_row_count_diff(const float* __restrict__ a, const float* __restrict__ b,
                                                                float* __restrict__ out, size_t n) {
  __shared__ double sh[kFabThreads / 64];
  const size_t row = blockIdx.x;
  double c = 0.0;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) c += ((a[row * n + i] - b[row * n + i]) != 0.f) ? 1.0 : 0.0;
  c = fab_block_sum(c, sh);
  if (threadIdx.x == 0) out[row] = (float)c;
}

// out[r] = sum_i a[r][i]*b[r][i]  (b of the hyperplane: -df + <grad, x1>, fab_base.py:170-171)
__global__ __launch_bounds__(kFabThreads) void k_row_dot(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, size_t n) {
  __shared__ double sh[kFabThreads / 64];
  const size_t row = blockIdx.x;
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) acc += (double)(a[row * n + i] * b[row * n + i]);
  acc = fab_block_sum(acc, sh);
  if (threadIdx.x == 0) out[row] = (float)acc;
}
// out[r] = max_i |a[r][i] - b[r][i]|
__global__ __launch_bounds__(kFabThreads) void k_row_absmax_diff(const float* __restrict__ a, const float* __restrict__ b,
                                                                 float* __restrict__ out, size_t n) {
  __shared__ double sh[kFabThreads / 64];
  const size_t row = blockIdx.x;
  float m = 0.f;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) m = fmaxf(m, fabsf(a[row * n + i] - b[row * n + i]));
  m = fab_block_min(m, sh, true);
  if (threadIdx.x == 0) out[row] = m;
}
// out[r] = ||a[r] - b[r]||_p, NORM 0 = Linf, 1 = L1, 2 = L2  (fab_base.py:226-236, :296-301)
template <int NORM>
__global__ __launch_bounds__(kFabThreads) void k_row_norm_diff(const float* __restrict__ a, const float* __restrict__ b,
                                                               float* __restrict__ out, size_t n) {
  __shared__ double sh[kFabThreads / 64];
  const size_t row = blockIdx.x;
  double acc = 0.0;
  float m = 0.f;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    const float v = a[row * n + i] - b[row * n + i];
    if (NORM == 0) m = fmaxf(m, fabsf(v));
    else acc += NORM == 1 ? (double)fabsf(v) : (double)v * (double)v;
  }
  if (NORM == 0) {
    m = fab_block_min(m, sh, true);
    if (threadIdx.x == 0) out[row] = m;
  } else {
    acc = fab_block_sum(acc, sh);
    if (threadIdx.x == 0) out[row] = NORM == 1 ? (float)acc : (float)sqrt(acc);
  }
}
// ---- expectation over transformation (autopgd_base.py:271-289): acc += g, and after the last pass acc /= eot_iter ----------------
// mode 0: acc[i] += g[i]     mode 1: acc[i] = acc[i] / divisor   (a division, as the reference's `grad /= float(eot_iter)`)
__global__ __launch_bounds__(kBlock) void k_eot_accumulate(float* __restrict__ acc, const float* __restrict__ g, size_t n, int mode,
                                                           float divisor) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    acc[i] = mode == 0 ? acc[i] + g[i] : acc[i] / divisor;
}

// ---- Square attack, L2 / L1 (Attacks/autoattack/square.py:296-530) ---------------------------------------------------------
// One workgroup per image (the norms couple all channels of an image).  NORM 2: sums of squares and square roots; NORM 1: sums
// of absolute values.  kSqC = channels supported (images are RGB).
constexpr int kSqC = 4;
template <int NORM>
__device__ __forceinline__ float sq_mag(float v) { return NORM == 2 ? v * v : fabsf(v); }
template <int NORM>
__device__ __forceinline__ float sq_root(double v) { return NORM == 2 ? sqrtf((float)v) : (float)v; }

// Start point (:297-312 / :410-426): delta_init = on each tile of the tiles_h x tiles_w grid (side s, origin sp) the pattern eta(s)
// (transposed when the tile's draw says so) times the tile's per-(image, channel) sign.  NORM 2: out = clamp(x0 + delta /
// (||delta||_2 + 1e-12) * eps, 0, 1).  NORM 1: out = delta_init (the caller adds L1_projection(x0, delta, eps (1 - 1e-6))).
template <int NORM>
__global__ __launch_bounds__(kFabThreads) void k_square_init_lp(float* __restrict__ out, const float* __restrict__ x0, int B, int C,
                                                                int H, int W, float eps, int s, int sp, int tiles_h, int tiles_w,
                                                                const float* __restrict__ eta2, const uint8_t* __restrict__ transposed,
                                                                const float* __restrict__ signs) {
  __shared__ double sh[kFabThreads / 64];
  const int b = blockIdx.x;
  const size_t plane = (size_t)H * W, n = (size_t)C * plane, base = (size_t)b * n;
  auto delta_at = [&](int c, int y, int x) -> float {
    const int ry = y - sp, rx = x - sp;
    if (ry < 0 || rx < 0) return 0.f;
    const int ty = ry / s, tx = rx / s;
    if (ty >= tiles_h || tx >= tiles_w) return 0.f;
    const int t = ty * tiles_w + tx;
    const float e = eta2[(size_t)(transposed[t] ? 1 : 0) * s * s + (size_t)(ry - ty * s) * s + (rx - tx * s)];
    return e * signs[((size_t)t * B + b) * C + c];
  };
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    const int c = (int)(i / plane), r = (int)(i - (size_t)c * plane);
    const float d = delta_at(c, r / W, r % W);
    if (NORM == 1) out[base + i] = d;
    acc += (double)sq_mag<NORM>(d);
  }
  if (NORM == 1) return;
  const float t = sq_root<NORM>(fab_block_sum(acc, sh));
  for (size_t i = threadIdx.x; i < n; i += kFabThreads) {
    const int c = (int)(i / plane), r = (int)(i - (size_t)c * plane);
    const float d = delta_at(c, r / W, r % W);
    out[base + i] = clampf(x0[base + i] + d / (t + 1e-12f) * eps, 0.f, 1.f);
  }
}

// One query (:314-377 / :428-486).  delta = x_best - x0; window 1 at (vh, vw) receives eta * sign + delta / (1e-12 + |delta|_win1),
// rescaled to the budget the image has left plus what the two windows held; window 2 at (vh2, vw2) is cleared (where it does not
// overlap window 1).  NORM 2: out = clamp(x0 + delta' / (||delta'||_2 + 1e-12) * eps, 0, 1); NORM 1: out = delta'.
template <int NORM>
__global__ __launch_bounds__(kFabThreads) void k_square_propose_lp(float* __restrict__ out, const float* __restrict__ xb,
                                                                   const float* __restrict__ x0, int C, int H, int W, float eps, int vh,
                                                                   int vw, int vh2, int vw2, int s, const float* __restrict__ eta,
                                                                   const float* __restrict__ signs) {
  __shared__ double sh[kFabThreads / 64];
  const int b = blockIdx.x;
  const size_t plane = (size_t)H * W, n = (size_t)C * plane, base = (size_t)b * n;
  double s_all = 0.0, s_w1[kSqC], s_un[kSqC];
#pragma unroll
  for (int c = 0; c < kSqC; ++c) s_w1[c] = s_un[c] = 0.0;
#pragma unroll
  for (int c = 0; c < kSqC; ++c) {
    if (c >= C) break;
    for (size_t r = threadIdx.x; r < plane; r += kFabThreads) {
      const int y = (int)(r / W), x = (int)(r - (size_t)y * W);
      const size_t i = base + (size_t)c * plane + r;
      const double v = (double)sq_mag<NORM>(xb[i] - x0[i]);
      const bool in1 = (unsigned)(y - vh) < (unsigned)s && (unsigned)(x - vw) < (unsigned)s;
      const bool in2 = (unsigned)(y - vh2) < (unsigned)s && (unsigned)(x - vw2) < (unsigned)s;
      s_all += v;
      if (in1) s_w1[c] += v;
      if (in1 || in2) s_un[c] += v;
    }
  }
  const float n_img = sq_root<NORM>(fab_block_sum(s_all, sh));
  float n_w1[kSqC], n_un[kSqC], fac[kSqC], n_new[kSqC];
#pragma unroll
  for (int c = 0; c < kSqC; ++c) {
    n_w1[c] = n_un[c] = fac[c] = n_new[c] = 0.f;
    if (c < C) {
      n_w1[c] = sq_root<NORM>(fab_block_sum(s_w1[c], sh));
      n_un[c] = sq_root<NORM>(fab_block_sum(s_un[c], sh));
    }
  }
  auto fresh = [&](int c, int wy, int wx) -> float {           // eta * sign + delta / (1e-12 + |delta|_win1), before the rescaling
    const size_t i = base + (size_t)c * plane + (size_t)(vh + wy) * W + (vw + wx);
    return eta[wy * s + wx] * signs[(size_t)b * C + c] + (xb[i] - x0[i]) / (1e-12f + n_w1[c]);
  };
#pragma unroll
  for (int c = 0; c < kSqC; ++c) {
    if (c >= C) break;
    double acc = 0.0;
    for (int i = threadIdx.x; i < s * s; i += kFabThreads) acc += (double)sq_mag<NORM>(fresh(c, i / s, i % s));
    n_new[c] = sq_root<NORM>(fab_block_sum(acc, sh));
    if (NORM == 2) fac[c] = sqrtf(fmaxf(eps * eps - n_img * n_img, 0.f) / (float)C + n_un[c] * n_un[c]);
    else fac[c] = fmaxf(eps - n_img, 0.f) / (float)C + n_un[c];
  }
  auto delta_new = [&](int c, int y, int x, size_t i) -> float {
    if ((unsigned)(y - vh) < (unsigned)s && (unsigned)(x - vw) < (unsigned)s) {
      const float v = fresh(c, y - vh, x - vw) / (1e-12f + n_new[c]) * fac[c];
      return NORM == 2 ? v : v * (float)C;
    }
    if ((unsigned)(y - vh2) < (unsigned)s && (unsigned)(x - vw2) < (unsigned)s) return 0.f;
    return xb[i] - x0[i];
  };
  double tot = 0.0;
#pragma unroll
  for (int c = 0; c < kSqC; ++c) {
    if (c >= C) break;
    for (size_t r = threadIdx.x; r < plane; r += kFabThreads) {
      const int y = (int)(r / W), x = (int)(r - (size_t)y * W);
      const size_t i = base + (size_t)c * plane + r;
      const float d = delta_new(c, y, x, i);
      if (NORM == 1) out[i] = d;
      tot += (double)sq_mag<NORM>(d);
    }
  }
  if (NORM == 1) return;
  const float t = sq_root<NORM>(fab_block_sum(tot, sh));
#pragma unroll
  for (int c = 0; c < kSqC; ++c) {
    if (c >= C) break;
    for (size_t r = threadIdx.x; r < plane; r += kFabThreads) {
      const int y = (int)(r / W), x = (int)(r - (size_t)y * W);
      const size_t i = base + (size_t)c * plane + r;
      out[i] = clampf(x0[i] + delta_new(c, y, x, i) / (t + 1e-12f) * eps, 0.f, 1.f);
    }
  }
}

// x1 = clamp((x1 + eta*d1)*(1 - alpha) + (x0 + eta*d2)*alpha, 0, 1)   (fab_base.py:218-219)
__global__ __launch_bounds__(kBlock) void k_fab_update(float* __restrict__ x1, const float* __restrict__ x0,
                                                       const float* __restrict__ d1, const float* __restrict__ d2,
                                                       const float* __restrict__ alpha, size_t nps, float eta) {
  const uint32_t b = blockIdx.y;
  const float al = alpha[b], om = 1.0f - al;
  const size_t base = (size_t)b * nps;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < nps; e += (size_t)gridDim.x * kBlock) {
    const size_t i = base + e;
    const float u = (x1[i] + eta * d1[i]) * om;
    const float v = (x0[i] + d2[i] * eta) * al;
    x1[i] = clampf(u + v, 0.f, 1.f);
  }
}
// rows with mask: x1 = x0 + (x1 - x0)*beta   (backward step after a success, fab_base.py:244-245)
__global__ __launch_bounds__(kBlock) void k_fab_backoff(float* __restrict__ x1, const float* __restrict__ x0,
                                                        const uint8_t* __restrict__ mask, size_t nps, float beta) {
  const uint32_t b = blockIdx.y;
  if (!mask[b]) return;
  const size_t base = (size_t)b * nps;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < nps; e += (size_t)gridDim.x * kBlock) {
    const size_t i = base + e;
    x1[i] = x0[i] + (x1[i] - x0[i]) * beta;
  }
}

// ---- row-wise logit losses: one wave per row ------------------------------------------------
struct Top {
  float v;
  int i;
};
__device__ __forceinline__ Top wave_argmax(Top t) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(t.v, off, 64);
    const int oi = __shfl_xor(t.i, off, 64);
    if (ov > t.v || (ov == t.v && oi < t.i)) {
      t.v = ov;
      t.i = oi;
    }
  }
  return t;
}

// kind 0 CE, 1 DLR, 2 targeted DLR
__global__ __launch_bounds__(kBlock) void k_logit_loss(const float* __restrict__ logits, const int64_t* __restrict__ y,
                                                       const int64_t* __restrict__ yt, int batch, int classes,
                                                       int kind, float scale, float* __restrict__ loss_out,
                                                       float* __restrict__ dl, int32_t* __restrict__ pred_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= batch) return;
  const float* z = logits + (size_t)row * classes;
  const int yy = (int)y[row];
  // top-4 of the row by repeated wave argmax with exclusion (classes ~1000: 4 cheap passes)
  Top top[4];
  int excl[4] = {-1, -1, -1, -1};
  for (int r = 0; r < 4; ++r) {
    Top best{-INFINITY, 0x7fffffff};
    for (int c = lane; c < classes; c += 64) {
      if (c == excl[0] || c == excl[1] || c == excl[2]) continue;
      const float v = z[c];
      if (v > best.v || (v == best.v && c < best.i)) {
        best.v = v;
        best.i = c;
      }
    }
    best = wave_argmax(best);
    top[r] = best;
    excl[r] = best.i;
    if (kind == 0 || kind == 4) break;  // CE / targeted difference only need the max (for pred)
    if (kind == 3 && r == 1) break;     // margin needs top-2
    if (kind == 1 && r == 2) break;     // DLR needs top-3
  }
  if (pred_out && lane == 0) pred_out[row] = top[0].i;
  const float zy = z[yy];
  if (kind == 0) {
    // log-sum-exp as max + log1p(sum over the OTHER classes): for a confident row (1 - p_y ~ 1e-6) forming lse = log(s) + max first
    // drops log(s) below the ulp of the maximum logit (2e-6 at 16..32), p_y becomes exactly 1 and dl no longer sums to zero -- the
    // common-mode part of the logit Jacobian then flips the sign of the input gradient (measured on the fitted ResNet-50 of
    // tests/test_outcome_gpu.py: cosine -0.997 against fp32 AND fp64 autograd on 5 of 64 images).  torch's log_softmax subtracts the
    // maximum first for the same reason.
    const float mx = top[0].v;
    const int imax = top[0].i;
    float s1 = 0.f;
    for (int c = lane; c < classes; c += 64)
      if (c != imax) s1 += expf(z[c] - mx);
    s1 = rart_wave_sum(s1);
    const float ls = log1pf(s1);                       // lse - max
    if (loss_out && lane == 0) loss_out[row] = (mx - zy) + ls;
    if (dl) {
      for (int c = lane; c < classes; c += 64) {
        const float t = (z[c] - mx) - ls;               // log p_c
        dl[(size_t)row * classes + c] = scale * (c == yy ? expm1f(t) : expf(t));      // p - onehot; p_y - 1 without cancellation
      }
    }
    return;
  }
  if (kind == 4) {
    // FAB targeted difference (fab_pt.py:102-117): -(z_y - z_t); d/dz = -e_y + e_t
    const int tt = (int)yt[row];
    if (loss_out && lane == 0) loss_out[row] = -(zy - z[tt]);
    if (dl)
      for (int c = lane; c < classes; c += 64)
        dl[(size_t)row * classes + c] = scale * ((c == tt ? 1.f : 0.f) - (c == yy ? 1.f : 0.f));
    return;
  }
  if (kind == 3) {
    // margin (Attacks/autoattack/square.py:68-86): z_y - max over the other classes; d/dz = e_y - e_other
    const bool ismax = top[0].i == yy;
    const int other = ismax ? top[1].i : top[0].i;
    if (loss_out && lane == 0) loss_out[row] = zy - (ismax ? top[1].v : top[0].v);
    if (dl)
      for (int c = lane; c < classes; c += 64)
        dl[(size_t)row * classes + c] = scale * ((c == yy ? 1.f : 0.f) - (c == other ? 1.f : 0.f));
    return;
  }
  float N, D, loss;
  // gradient contributions as (index, weight) pairs, applied additively
  int gi[5];
  float gw[5];
  int ng = 0;
  if (kind == 1) {
    const bool ind = top[0].i == yy;
    const float other = ind ? top[1].v : top[0].v;
    N = zy - other;
    D = top[0].v - top[2].v + 1e-12f;
    loss = -N / D;
    gi[ng] = yy; gw[ng++] = -1.f / D;
    gi[ng] = ind ? top[1].i : top[0].i; gw[ng++] = 1.f / D;
    gi[ng] = top[0].i; gw[ng++] = N / (D * D);
    gi[ng] = top[2].i; gw[ng++] = -N / (D * D);
  } else {
    const int tt = (int)yt[row];
    N = zy - z[tt];
    D = top[0].v - 0.5f * (top[2].v + top[3].v) + 1e-12f;
    loss = -N / D;
    gi[ng] = yy; gw[ng++] = -1.f / D;
    gi[ng] = tt; gw[ng++] = 1.f / D;
    gi[ng] = top[0].i; gw[ng++] = N / (D * D);
    gi[ng] = top[2].i; gw[ng++] = -0.5f * N / (D * D);
    gi[ng] = top[3].i; gw[ng++] = -0.5f * N / (D * D);
  }
  if (loss_out && lane == 0) loss_out[row] = loss;
  if (dl) {
    for (int c = lane; c < classes; c += 64) {
      float w = 0.f;
      for (int k = 0; k < ng; ++k)
        if (gi[k] == c) w += gw[k];
      dl[(size_t)row * classes + c] = scale * w;
    }
  }
}

dim3 grid_rows(size_t nps, int batch) {
  size_t gx = (nps + kBlock - 1) / kBlock;
  size_t cap = 4096 / (batch < 1 ? 1 : batch);
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return dim3((uint32_t)gx, (uint32_t)batch, 1);
}

int need_ws(const char* who, void* ws, size_t have, size_t floats) {
  if (!ws || have < floats * sizeof(float)) {
    rart_set_error("%s: workspace of %zu bytes required, got %zu", who, floats * sizeof(float), have);
    return RART_ERR_WORKSPACE;
  }
  return RART_OK;
}
}  // namespace

namespace {
// out[b][j] = +1 / -1 from bit 31 of the first Threefry word of counter (index_base + j, stream_id) of row b's sample: the value
// noise/rng.py's host_uniform(seed, sample, stream_id, index_base + j) >= 0.5 gives, generated where it is consumed (Square's per-image
// sign rows: 5 000 queries used to draw them in numpy and copy them to the device every query)
__global__ __launch_bounds__(kBlock) void k_rng_signs(float* __restrict__ out, int batch, int n, uint32_t k0, uint32_t k1, uint32_t sbase,
                                                      const int64_t* __restrict__ rows, int stream_id, uint32_t index_base) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= batch * n) return;
  const uint32_t b = (uint32_t)(i / n), j = (uint32_t)(i - (int)b * n);
  const uint2 w = threefry2x32(k0, k1, rart_ctr0(index_base + j, stream_id), rows ? (uint32_t)rows[b] : sbase + b);
  out[i] = (w.x >> 31) ? 1.f : -1.f;
}
// standard normals for explicit rows: out[b][e] = the value rart_rng_normal_f32 gives sample rows[b] (four per Threefry pair)
__global__ __launch_bounds__(kBlock) void k_rng_normal_rows(float* __restrict__ out, uint32_t elems, uint32_t k0, uint32_t k1,
                                                            const int64_t* __restrict__ rows, int stream_id) {
  const uint32_t b = blockIdx.y, smp = (uint32_t)rows[b];
  float* o = out + (size_t)b * elems;
  for (uint32_t v = blockIdx.x * kBlock + threadIdx.x; v * 4 < elems; v += gridDim.x * kBlock) {
    const float4 z = rart_normal4(k0, k1, v, stream_id, smp);
    const float zz[4] = {z.x, z.y, z.z, z.w};
    for (int j = 0; j < 4; ++j)
      if (v * 4 + j < elems) o[v * 4 + j] = zz[j];
  }
}
}  // namespace

extern "C" {

size_t rart_attack_workspace_bytes(int batch) { return (size_t)(batch < 1 ? 1 : batch) * RCH * 4 * sizeof(float); }

int rart_rng_signs_f32(float* out, int batch, int n_per_row, uint64_t seed, uint64_t sample_offset, const int64_t* row_samples,
                       int stream_id, uint32_t index_base, rart_stream_t stream) {
  RART_CHECK_ARG(out && batch > 0 && n_per_row > 0 && (long long)batch * n_per_row < (1ll << 31) && stream_id >= 0 && stream_id < 16,
                 "rart_rng_signs_f32: bad arguments");
  hipLaunchKernelGGL(k_rng_signs, dim3((batch * n_per_row + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, out, batch,
                     n_per_row, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)sample_offset, row_samples, stream_id, index_base);
  RART_CHECK_LAUNCH("rart_rng_signs_f32");
  return RART_OK;
}

int rart_rng_normal_rows_f32(float* out, int batch, size_t elems, uint64_t seed, const int64_t* row_samples, int stream_id,
                             rart_stream_t stream) {
  RART_CHECK_ARG(out && row_samples && batch > 0 && batch <= 65535 && elems > 0 && elems < (1ull << 30) && stream_id >= 0 && stream_id < 16,
                 "rart_rng_normal_rows_f32: bad arguments");
  uint32_t gx = (uint32_t)((elems + 3) / 4 + kBlock - 1) / kBlock;
  const uint32_t cap = (uint32_t)(2048 / batch) < 1 ? 1 : (uint32_t)(2048 / batch);
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(k_rng_normal_rows, dim3(gx, batch), dim3(kBlock), 0, (hipStream_t)stream, out, (uint32_t)elems, (uint32_t)seed,
                     (uint32_t)(seed >> 32), row_samples, stream_id);
  RART_CHECK_LAUNCH("rart_rng_normal_rows_f32");
  return RART_OK;
}

int rart_attack_init_linf(float* x, const float* x0, int batch, size_t nps, float eps, float lo, float hi,
                          uint64_t seed, uint64_t sample_offset, const int64_t* row_samples, const float* injected_u,
                          rart_stream_t stream) {
  RART_CHECK_ARG(x && x0 && batch > 0 && nps > 0 && nps < (1ull << 29), "rart_attack_init_linf: bad arguments");
  hipLaunchKernelGGL(k_init_linf, grid_rows(nps, batch), dim3(kBlock), 0, (hipStream_t)stream, x, x0, nps, eps, lo,
                     hi, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)sample_offset, injected_u, row_samples);
  RART_CHECK_LAUNCH("rart_attack_init_linf");
  return RART_OK;
}

int rart_pgd_step_linf(float* x, const float* g, const float* x0, size_t n, float eps, float alpha,
                       rart_stream_t stream) {
  RART_CHECK_ARG(x && g && x0, "rart_pgd_step_linf: null pointer");
  if (n == 0) return RART_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool al = (((uintptr_t)x | (uintptr_t)g | (uintptr_t)x0) & 15) == 0;
  const size_t nvec = al ? n / 4 : 0;
  if (nvec)
    hipLaunchKernelGGL(k_pgd_linf, dim3(rart_grid_for(nvec)), dim3(kBlock), 0, s, (float4*)x, (const float4*)g,
                       (const float4*)x0, nvec, eps, alpha);
  const size_t rest = n - nvec * 4;
  if (rest)
    hipLaunchKernelGGL(k_pgd_linf_tail, dim3((uint32_t)((rest + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, x, g, x0,
                       nvec * 4, n, eps, alpha);
  RART_CHECK_LAUNCH("rart_pgd_step_linf");
  return RART_OK;
}

int rart_pgd_step_l2(float* x, const float* g, const float* x0, int batch, size_t nps, float eps, float alpha,
                     void* ws, size_t ws_bytes, rart_stream_t stream) {
  RART_CHECK_ARG(x && g && x0 && batch > 0 && nps > 0, "rart_pgd_step_l2: bad arguments");
  if (int e = need_ws("rart_pgd_step_l2", ws, ws_bytes, (size_t)batch * RCH * 2)) return e;
  hipStream_t s = (hipStream_t)stream;
  float* gpart = (float*)ws;
  float* dpart = gpart + (size_t)batch * RCH;
  hipLaunchKernelGGL(k_reduce_rows<0>, dim3(RCH, batch), dim3(kBlock), 0, s, g, gpart, nps);
  hipLaunchKernelGGL(k_pgd_l2_move, dim3(RCH, batch), dim3(kBlock), 0, s, x, g, x0, gpart, dpart, nps, alpha);
  hipLaunchKernelGGL(k_pgd_l2_project, grid_rows(nps, batch), dim3(kBlock), 0, s, x, x0, dpart, nps, eps);
  RART_CHECK_LAUNCH("rart_pgd_step_l2");
  return RART_OK;
}

int rart_pgd_step_l1(float* x, const float* g, const float* x0, int batch, size_t nps, float eps, float eps_step,
                     void* ws, size_t ws_bytes, rart_stream_t stream) {
  RART_CHECK_ARG(x && g && x0 && batch > 0 && nps > 0, "rart_pgd_step_l1: bad arguments");
  if (int e = need_ws("rart_pgd_step_l1", ws, ws_bytes, (size_t)batch * RCH * 2)) return e;
  hipStream_t s = (hipStream_t)stream;
  float* gpart = (float*)ws;
  float* dpart = gpart + (size_t)batch * RCH;
  hipLaunchKernelGGL(k_reduce_rows<1>, dim3(RCH, batch), dim3(kBlock), 0, s, g, gpart, nps);
  hipLaunchKernelGGL(k_pgd_l1_move, dim3(RCH, batch), dim3(kBlock), 0, s, x, g, x0, gpart, dpart, nps, eps_step);
  hipLaunchKernelGGL(k_pgd_l1_project, grid_rows(nps, batch), dim3(kBlock), 0, s, x, x0, dpart, nps, eps);
  RART_CHECK_LAUNCH("rart_pgd_step_l1");
  return RART_OK;
}

int rart_random_start_l1(float* x, const float* x0, int batch, size_t nps, float eps, uint64_t seed, uint64_t sample_offset,
                         const int64_t* row_samples, const float* injected_signed_exp, const float* injected_radius, void* ws, size_t ws_bytes,
                         rart_stream_t stream) {
  RART_CHECK_ARG(x && x0 && batch > 0 && nps > 0 && nps < (1ull << 28) - 1, "rart_random_start_l1: bad arguments");
  RART_CHECK_ARG((injected_signed_exp == nullptr) == (injected_radius == nullptr),
                 "rart_random_start_l1: inject both the signed exponentials and the radii, or neither");
  if (int e = need_ws("rart_random_start_l1", ws, ws_bytes, (size_t)batch * RCH)) return e;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), sb = (uint32_t)sample_offset;
  hipLaunchKernelGGL(k_l1_start_reduce, dim3(RCH, batch), dim3(kBlock), 0, s, (float*)ws, nps, k0, k1, sb, injected_signed_exp,
                     row_samples);
  hipLaunchKernelGGL(k_l1_start_apply, grid_rows(nps, batch), dim3(kBlock), 0, s, x, x0, (const float*)ws, nps, eps, k0, k1,
                     sb, injected_signed_exp, injected_radius, row_samples);
  RART_CHECK_LAUNCH("rart_random_start_l1");
  return RART_OK;
}

int rart_mim_step(float* x, float* m, const float* g, const float* x0, int batch, size_t nps, float eps,
                  float step, float decay, void* ws, size_t ws_bytes, rart_stream_t stream) {
  RART_CHECK_ARG(x && m && g && x0 && batch > 0 && nps > 0, "rart_mim_step: bad arguments");
  if (int e = need_ws("rart_mim_step", ws, ws_bytes, (size_t)batch * RCH)) return e;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_reduce_rows<1>, dim3(RCH, batch), dim3(kBlock), 0, s, g, (float*)ws, nps);
  hipLaunchKernelGGL(k_mim_apply, grid_rows(nps, batch), dim3(kBlock), 0, s, x, m, g, x0, (const float*)ws, nps,
                     eps, step, decay);
  RART_CHECK_LAUNCH("rart_mim_step");
  return RART_OK;
}

int rart_apgd_init(float* x, const float* x0, int batch, size_t nps, int norm, float eps, uint64_t seed,
                   uint64_t sample_offset, const int64_t* row_samples, const float* inj, void* ws, size_t ws_bytes,
                   rart_stream_t stream) {
  RART_CHECK_ARG(x && x0 && batch > 0 && nps > 0 && nps < (1ull << 29), "rart_apgd_init: bad arguments");
  RART_CHECK_ARG(norm >= 0 && norm <= 2, "rart_apgd_init: norm must be 0 (Linf), 1 (L2) or 2 (L1: normal draws divided by their L1 norm)");
  if (int e = need_ws("rart_apgd_init", ws, ws_bytes, (size_t)batch * RCH)) return e;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), sb = (uint32_t)sample_offset;
  float* part = (float*)ws;
  if (norm == 0) {
    hipLaunchKernelGGL(k_apgd_init_reduce<0>, dim3(RCH, batch), dim3(kBlock), 0, s, part, nps, k0, k1, sb, inj, row_samples);
    hipLaunchKernelGGL(k_apgd_init_apply<0>, grid_rows(nps, batch), dim3(kBlock), 0, s, x, x0, part, nps, eps, k0,
                       k1, sb, inj, row_samples);
  } else if (norm == 1) {
    hipLaunchKernelGGL(k_apgd_init_reduce<1>, dim3(RCH, batch), dim3(kBlock), 0, s, part, nps, k0, k1, sb, inj, row_samples);
    hipLaunchKernelGGL(k_apgd_init_apply<1>, grid_rows(nps, batch), dim3(kBlock), 0, s, x, x0, part, nps, eps, k0,
                       k1, sb, inj, row_samples);
  } else {
    hipLaunchKernelGGL(k_apgd_init_reduce<2>, dim3(RCH, batch), dim3(kBlock), 0, s, part, nps, k0, k1, sb, inj, row_samples);
    hipLaunchKernelGGL(k_apgd_init_apply<2>, grid_rows(nps, batch), dim3(kBlock), 0, s, x, x0, part, nps, eps, k0,
                       k1, sb, inj, row_samples);
  }
  RART_CHECK_LAUNCH("rart_apgd_init");
  return RART_OK;
}

int rart_apgd_step(float* xa, float* xold, const float* grad, const float* x0, const float* step, int batch,
                   size_t nps, int norm, float eps, float a, void* ws, size_t ws_bytes, rart_stream_t stream) {
  RART_CHECK_ARG(xa && xold && grad && x0 && step && batch > 0 && nps > 0, "rart_apgd_step: bad arguments");
  RART_CHECK_ARG(norm == 0 || norm == 1, "rart_apgd_step: norm must be 0 (Linf) or 1 (L2)");
  hipStream_t s = (hipStream_t)stream;
  if (norm == 0) {
    hipLaunchKernelGGL(k_apgd_linf, grid_rows(nps, batch), dim3(kBlock), 0, s, xa, xold, grad, x0, step, nps, eps, a);
  } else {
    if (int e = need_ws("rart_apgd_step", ws, ws_bytes, (size_t)batch * RCH * 3)) return e;
    float* w = (float*)ws;
    hipLaunchKernelGGL(k_reduce_rows<0>, dim3(RCH, batch), dim3(kBlock), 0, s, grad, w, nps);
    hipLaunchKernelGGL(k_apgd_l2<0>, dim3(RCH, batch), dim3(kBlock), 0, s, xa, xold, grad, x0, step, w, batch, nps, eps, a);
    hipLaunchKernelGGL(k_apgd_l2<1>, dim3(RCH, batch), dim3(kBlock), 0, s, xa, xold, grad, x0, step, w, batch, nps, eps, a);
    hipLaunchKernelGGL(k_apgd_l2<2>, dim3(RCH, batch), dim3(kBlock), 0, s, xa, xold, grad, x0, step, w, batch, nps, eps, a);
  }
  RART_CHECK_LAUNCH("rart_apgd_step");
  return RART_OK;
}

int rart_square_init_linf(float* x_best, const float* x0, int batch, int c, int h, int w, float eps, uint64_t seed,
                          uint64_t sample_offset, const int64_t* row_samples, const float* injected_sign, rart_stream_t stream) {
  RART_CHECK_ARG(x_best && x0 && batch > 0 && c > 0 && h > 0 && w > 0, "rart_square_init_linf: bad arguments");
  hipLaunchKernelGGL(k_square_init, grid_rows((size_t)c * h * w, batch), dim3(kBlock), 0, (hipStream_t)stream, x_best,
                     x0, c, h, w, eps, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)sample_offset, injected_sign, row_samples);
  RART_CHECK_LAUNCH("rart_square_init_linf");
  return RART_OK;
}

int rart_eot_accumulate(float* acc, const float* g, size_t n, int mode, float divisor, rart_stream_t stream) {
  RART_CHECK_ARG(acc && n > 0 && (mode == 0 ? g != nullptr : (mode == 1 && divisor != 0.f)), "rart_eot_accumulate: bad arguments");
  hipLaunchKernelGGL(k_eot_accumulate, dim3(rart_grid_for(n, kBlock, 256 * 16)), dim3(kBlock), 0, (hipStream_t)stream, acc, g, n, mode,
                     divisor);
  RART_CHECK_LAUNCH("rart_eot_accumulate");
  return RART_OK;
}
int rart_square_init_lp(float* out, const float* x0, int batch, int c, int h, int w, float eps, int norm, int s, int sp, int tiles_h,
                        int tiles_w, const float* eta2, const uint8_t* transposed, const float* signs, rart_stream_t stream) {
  RART_CHECK_ARG(out && x0 && eta2 && transposed && signs && batch > 0 && c > 0 && c <= kSqC && h > 0 && w > 0 && (norm == 1 || norm == 2),
                 "rart_square_init_lp: bad arguments");
  RART_CHECK_ARG(s > 0 && sp >= 0 && tiles_h > 0 && tiles_w > 0 && sp + tiles_h * s <= h && sp + tiles_w * s <= w,
                 "rart_square_init_lp: the tile grid must lie inside the image");
  auto kern = norm == 2 ? k_square_init_lp<2> : k_square_init_lp<1>;
  hipLaunchKernelGGL(kern, dim3(batch), dim3(kFabThreads), 0, (hipStream_t)stream, out, x0, batch, c, h, w, eps, s, sp, tiles_h, tiles_w,
                     eta2, transposed, signs);
  RART_CHECK_LAUNCH("rart_square_init_lp");
  return RART_OK;
}
int rart_square_propose_lp(float* out, const float* x_best, const float* x0, int batch, int c, int h, int w, float eps, int norm, int vh,
                           int vw, int vh2, int vw2, int s, const float* eta, const float* signs, rart_stream_t stream) {
  RART_CHECK_ARG(out && x_best && x0 && eta && signs && batch > 0 && c > 0 && c <= kSqC && h > 0 && w > 0 && (norm == 1 || norm == 2),
                 "rart_square_propose_lp: bad arguments");
  RART_CHECK_ARG(s > 0 && vh >= 0 && vw >= 0 && vh + s <= h && vw + s <= w && vh2 >= 0 && vw2 >= 0 && vh2 + s <= h && vw2 + s <= w,
                 "rart_square_propose_lp: window outside the image");
  auto kern = norm == 2 ? k_square_propose_lp<2> : k_square_propose_lp<1>;
  hipLaunchKernelGGL(kern, dim3(batch), dim3(kFabThreads), 0, (hipStream_t)stream, out, x_best, x0, c, h, w, eps, vh, vw, vh2, vw2, s,
                     eta, signs);
  RART_CHECK_LAUNCH("rart_square_propose_lp");
  return RART_OK;
}
int rart_square_propose_linf(float* x_new, const float* x_best, const float* x0, int batch, int c, int h, int w,
                             float eps, int vh, int vw, int s, const float* sign_host, rart_stream_t stream) {
  RART_CHECK_ARG(x_new && x_best && x0 && sign_host && batch > 0 && c > 0 && c <= 8 && h > 0 && w > 0 && s > 0,
                 "rart_square_propose_linf: bad arguments");
  RART_CHECK_ARG(vh >= 0 && vw >= 0 && vh + s <= h && vw + s <= w, "rart_square_propose_linf: window outside the image");
  SquareSigns sg;
  for (int i = 0; i < 8; ++i) sg.s[i] = i < c ? sign_host[i] : 0.f;
  const size_t total = (size_t)batch * c * h * w;
  hipLaunchKernelGGL(k_square_propose, dim3(rart_grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, x_new, x_best,
                     x0, total, c, h, w, eps, vh, vw, s, sg);
  RART_CHECK_LAUNCH("rart_square_propose_linf");
  return RART_OK;
}

int rart_fab_project_linf(const float* points, const float* w, const float* b, float* d_out, float* rowmax_out, int rows,
                          size_t n, rart_stream_t stream) {
  RART_CHECK_ARG(points && w && b && d_out && rows > 0 && n > 0, "rart_fab_project_linf: bad arguments");
  hipLaunchKernelGGL(k_fab_project_linf, dim3(rows), dim3(kFabThreads), 0, (hipStream_t)stream, points, w, b, d_out,
                     rowmax_out, n);
  RART_CHECK_LAUNCH("rart_fab_project_linf");
  return RART_OK;
}
int rart_fab_project(const float* points, const float* w, const float* b, float* d_out, float* rownorm_out, int rows, size_t n,
                     int norm, rart_stream_t stream) {
  RART_CHECK_ARG(points && w && b && d_out && rows > 0 && n > 0 && norm >= 0 && norm <= 2, "rart_fab_project: bad arguments");
  auto kern = norm == 0 ? k_fab_project_linf : (norm == 2 ? k_fab_project_l2 : k_fab_project_l1);
  hipLaunchKernelGGL(kern, dim3(rows), dim3(kFabThreads), 0, (hipStream_t)stream, points, w, b, d_out, rownorm_out, n);
  RART_CHECK_LAUNCH("rart_fab_project");
  return RART_OK;
}
int rart_row_norm_diff(const float* a, const float* b, float* out, int rows, size_t n, int norm, rart_stream_t stream) {
  RART_CHECK_ARG(a && b && out && rows > 0 && n > 0 && norm >= 0 && norm <= 2, "rart_row_norm_diff: bad arguments");
  auto kern = norm == 0 ? k_row_norm_diff<0> : (norm == 1 ? k_row_norm_diff<1> : k_row_norm_diff<2>);
  hipLaunchKernelGGL(kern, dim3(rows), dim3(kFabThreads), 0, (hipStream_t)stream, a, b, out, n);
  RART_CHECK_LAUNCH("rart_row_norm_diff");
  return RART_OK;
}
int rart_row_dot(const float* a, const float* b, float* out, int rows, size_t n, rart_stream_t stream) {
  RART_CHECK_ARG(a && b && out && rows > 0 && n > 0, "rart_row_dot: bad arguments");
  hipLaunchKernelGGL(k_row_dot, dim3(rows), dim3(kFabThreads), 0, (hipStream_t)stream, a, b, out, n);
  RART_CHECK_LAUNCH("rart_row_dot");
  return RART_OK;
}
int rart_row_absmax_diff(const float* a, const float* b, float* out, int rows, size_t n, rart_stream_t stream) {
  RART_CHECK_ARG(a && b && out && rows > 0 && n > 0, "rart_row_absmax_diff: bad arguments");
  hipLaunchKernelGGL(k_row_absmax_diff, dim3(rows), dim3(kFabThreads), 0, (hipStream_t)stream, a, b, out, n);
  RART_CHECK_LAUNCH("rart_row_absmax_diff");
  return RART_OK;
}
int rart_fab_update(float* x1, const float* x0, const float* d1, const float* d2, const float* alpha, int batch,
                    size_t nps, float eta, rart_stream_t stream) {
  RART_CHECK_ARG(x1 && x0 && d1 && d2 && alpha && batch > 0 && nps > 0, "rart_fab_update: bad arguments");
  hipLaunchKernelGGL(k_fab_update, grid_rows(nps, batch), dim3(kBlock), 0, (hipStream_t)stream, x1, x0, d1, d2, alpha, nps,
                     eta);
  RART_CHECK_LAUNCH("rart_fab_update");
  return RART_OK;
}
int rart_fab_backoff(float* x1, const float* x0, const uint8_t* mask, int batch, size_t nps, float beta,
                     rart_stream_t stream) {
  RART_CHECK_ARG(x1 && x0 && mask && batch > 0 && nps > 0, "rart_fab_backoff: bad arguments");
  hipLaunchKernelGGL(k_fab_backoff, grid_rows(nps, batch), dim3(kBlock), 0, (hipStream_t)stream, x1, x0, mask, nps, beta);
  RART_CHECK_LAUNCH("rart_fab_backoff");
  return RART_OK;
}

int rart_select_rows(float* dst, const float* src, const uint8_t* mask, int batch, size_t nps, rart_stream_t stream) {
  RART_CHECK_ARG(dst && src && mask && batch > 0 && nps > 0, "rart_select_rows: bad arguments");
  hipLaunchKernelGGL(k_select_rows, grid_rows(nps, batch), dim3(kBlock), 0, (hipStream_t)stream, dst, src, mask, nps);
  RART_CHECK_LAUNCH("rart_select_rows");
  return RART_OK;
}

int rart_logit_loss(const float* logits, const int64_t* y, const int64_t* yt, int batch, int classes, int kind,
                    float scale, float* loss_out, float* dl, int32_t* pred_out, rart_stream_t stream) {
  RART_CHECK_ARG(logits && y && batch > 0 && classes > 0, "rart_logit_loss: bad arguments");
  RART_CHECK_ARG(kind >= 0 && kind <= 4, "rart_logit_loss: kind must be 0 (CE), 1 (DLR), 2 (targeted DLR), 3 (margin), 4 (targeted diff)");
  RART_CHECK_ARG((kind != 2 && kind != 4) || yt != nullptr, "rart_logit_loss: targeted losses need y_target");
  RART_CHECK_ARG(kind == 0 || classes >= ((kind == 3 || kind == 4) ? 2 : (kind == 1 ? 3 : 4)), "rart_logit_loss: too few classes");
  const int rows_per_block = kBlock / 64;
  hipLaunchKernelGGL(k_logit_loss, dim3((batch + rows_per_block - 1) / rows_per_block), dim3(kBlock), 0,
                     (hipStream_t)stream, logits, y, yt, batch, classes, kind, scale, loss_out, dl, pred_out);
  RART_CHECK_LAUNCH("rart_logit_loss");
  return RART_OK;
}


int rart_l1_project(const float* x, const float* y, float* out, int rows, size_t n, float eps, int point_out, int clamp01,
                    rart_stream_t stream) {
  RART_CHECK_ARG(x && y && out && rows > 0 && n > 0 && eps >= 0.f, "rart_l1_project: bad arguments");
  hipLaunchKernelGGL(k_l1_project, dim3(rows), dim3(kFabThreads), 0, (hipStream_t)stream, x, y, out, n, eps, point_out, clamp01);
  RART_CHECK_LAUNCH("rart_l1_project");
  return RART_OK;
}

int rart_row_kth_abs(const float* g, const int64_t* k, float* thr, int rows, size_t n, rart_stream_t stream) {
  RART_CHECK_ARG(g && k && thr && rows > 0 && n > 0 && n < (1ull << 32), "rart_row_kth_abs: bad arguments");
  hipLaunchKernelGGL(k_row_kth_abs, dim3(rows), dim3(kFabThreads), 0, (hipStream_t)stream, g, k, thr, n);
  RART_CHECK_LAUNCH("rart_row_kth_abs");
  return RART_OK;
}

int rart_apgd_l1_move(const float* x_adv, const float* grad, const float* x0, const float* thr, const float* step_size,
                      float* delta_u, int rows, size_t n, rart_stream_t stream) {
  RART_CHECK_ARG(x_adv && grad && x0 && thr && step_size && delta_u && rows > 0 && n > 0, "rart_apgd_l1_move: bad arguments");
  hipLaunchKernelGGL(k_apgd_l1_move, dim3(rows), dim3(kFabThreads), 0, (hipStream_t)stream, x_adv, grad, x0, thr, step_size,
                     delta_u, n);
  RART_CHECK_LAUNCH("rart_apgd_l1_move");
  return RART_OK;
}

int rart_row_count_diff(const float* a, const float* b, float* out, int rows, size_t n, rart_stream_t stream) {
  RART_CHECK_ARG(a && b && out && rows > 0 && n > 0, "rart_row_count_diff: bad arguments");
  hipLaunchKernelGGL(k_row_count_diff, dim3(rows), dim3(kFabThreads), 0, (hipStream_t)stream, a, b, out, n);
  RART_CHECK_LAUNCH("rart_row_count_diff");
  return RART_OK;
}

}  // extern "C"
