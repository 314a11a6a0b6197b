// Training-side step kernels for gfx950: label-smoothed cross entropy with its
// gradient, SGD-Nesterov / AdamW parameter updates fused with the EMA update and the gradient reset.
// Reference: the solver configuration the adversarial-training experiments run with
//   exprs/nips_benchmark/pgd_adv_train/resnet50/config.yaml:11-33  (SGD nesterov, momentum 0.9, wd 1e-4,
//       label_smooth 0.1, EMA 0.9999, cosine schedule with warm-up)
//   exprs/nips_benchmark/new_adv_train/vit_base/config.yaml:11-38  (AdamW)
// whose arithmetic is torch.optim.SGD / torch.optim.AdamW / F.cross_entropy(label_smoothing=...).
//
// Parameters, gradients, momenta and the EMA copy live in flat fp32 arenas (one allocation each, module
// parameters are views), so one launch updates the whole model: per element 16 B read (p, g, m, ema) and
// 16 B written (p, m, ema, g = 0) -- HBM-bound, 16-byte vectorised, no atomics, no reductions.
// FMA contraction is off so the update reproduces the op-by-op fp32 rounding of the unfused expressions.
#include "rart_common.h"

#pragma clang fp contract(off)

namespace {
constexpr int kBlock = 256;

struct SgdArgs {
  float lr, momentum, weight_decay, grad_scale, ema_decay, one_minus_ema;   // 1 - x constants: double on the host, rounded once
  int nesterov, zero_grad;
};

__device__ __forceinline__ void sgd_one(float& p, float& g, float& m, float* ema, const SgdArgs& a) {
  float d = g * a.grad_scale;
  if (a.weight_decay != 0.f) d = d + a.weight_decay * p;   // torch: grad.add(param, alpha=wd)
  m = a.momentum * m + d;                                   // buf.mul_(momentum).add_(grad); first step: buf = grad (m starts 0)
  if (a.nesterov) d = d + a.momentum * m;                   // grad.add(buf, alpha=momentum)
  else d = m;
  p = p - a.lr * d;                                         // param.add_(grad, alpha=-lr)
  if (ema) *ema = a.ema_decay * (*ema) + a.one_minus_ema * p;
  if (a.zero_grad) g = 0.f;
}

__global__ __launch_bounds__(kBlock) void k_sgd_step(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ ema, size_t n, SgdArgs a) {
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i], gv = reinterpret_cast<float4*>(g)[i], mv = reinterpret_cast<float4*>(m)[i];
    float4 ev = ema ? reinterpret_cast<float4*>(ema)[i] : make_float4(0, 0, 0, 0);
    sgd_one(pv.x, gv.x, mv.x, ema ? &ev.x : nullptr, a);
    sgd_one(pv.y, gv.y, mv.y, ema ? &ev.y : nullptr, a);
    sgd_one(pv.z, gv.z, mv.z, ema ? &ev.z : nullptr, a);
    sgd_one(pv.w, gv.w, mv.w, ema ? &ev.w : nullptr, a);
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    if (ema) reinterpret_cast<float4*>(ema)[i] = ev;
    if (a.zero_grad) reinterpret_cast<float4*>(g)[i] = gv;
  }
  // tail (n % 4 elements)
  const size_t t = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (t < n) sgd_one(p[t], g[t], m[t], ema ? ema + t : nullptr, a);
}

struct AdamArgs {
  float lr, beta1, beta2, eps, grad_scale, ema_decay;
  // formed in double on the host and rounded once, as torch does with its Python-float scalars
  float decay_mul, one_minus_beta1, one_minus_beta2, one_minus_ema, bias1, sqrt_bias2;
  int zero_grad;
};

__device__ __forceinline__ void adamw_one(float& p, float& g, float& m, float& v, float* ema, const AdamArgs& a) {
  const float gg = g * a.grad_scale;
  p = p * a.decay_mul;                                      // param.mul_(1 - lr * wd)
  m = m + a.one_minus_beta1 * (gg - m);                     // exp_avg.lerp_(grad, 1 - beta1)
  v = a.beta2 * v + a.one_minus_beta2 * gg * gg;             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
  const float denom = sqrtf(v) / a.sqrt_bias2 + a.eps;      // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
  p = p - (a.lr / a.bias1) * (m / denom);                   // param.addcdiv_(exp_avg, denom, value=-step_size)
  if (ema) *ema = a.ema_decay * (*ema) + a.one_minus_ema * p;
  if (a.zero_grad) g = 0.f;
}

__global__ __launch_bounds__(kBlock) void k_adamw_step(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ ema, size_t n, AdamArgs a) {
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i], gv = reinterpret_cast<float4*>(g)[i], mv = reinterpret_cast<float4*>(m)[i],
           vv = reinterpret_cast<float4*>(v)[i];
    float4 ev = ema ? reinterpret_cast<float4*>(ema)[i] : make_float4(0, 0, 0, 0);
    adamw_one(pv.x, gv.x, mv.x, vv.x, ema ? &ev.x : nullptr, a);
    adamw_one(pv.y, gv.y, mv.y, vv.y, ema ? &ev.y : nullptr, a);
    adamw_one(pv.z, gv.z, mv.z, vv.z, ema ? &ev.z : nullptr, a);
    adamw_one(pv.w, gv.w, mv.w, vv.w, ema ? &ev.w : nullptr, a);
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (ema) reinterpret_cast<float4*>(ema)[i] = ev;
    if (a.zero_grad) reinterpret_cast<float4*>(g)[i] = gv;
  }
  const size_t t = n4 * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (t < n) adamw_one(p[t], g[t], m[t], v[t], ema ? ema + t : nullptr, a);
}

__global__ __launch_bounds__(kBlock) void k_ema_update(float* __restrict__ ema, const float* __restrict__ p, size_t n, float decay,
                                                       float one_minus) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
    ema[i] = decay * ema[i] + one_minus * p[i];
}

// One wave per row of logits: log-softmax, the smoothed loss and d(scale * loss)/dlogits.
//   loss = (1 - s) * (-logp[y]) + s * mean_c(-logp[c])          (F.cross_entropy(label_smoothing=s), reduction none)
//   dlogits[c] = scale * (softmax[c] - (1 - s) * [c == y] - s / C)
__global__ __launch_bounds__(kBlock) void k_label_smooth_ce(const float* __restrict__ logits, const int64_t* __restrict__ y,
                                                            int batch, int classes, float smoothing, float scale,
                                                            float* __restrict__ loss_out, float* __restrict__ dlogits) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= batch) return;
  const float* z = logits + (size_t)row * classes;
  float mx = -3.402823466e38f;
  for (int c = lane; c < classes; c += 64) mx = fmaxf(mx, z[c]);
  mx = rart_wave_max(mx);
  float se = 0.f, sz = 0.f;
  for (int c = lane; c < classes; c += 64) {
    se += expf(z[c] - mx);
    sz += z[c] - mx;
  }
  se = rart_wave_sum(se);
  sz = rart_wave_sum(sz);
  const float lse = logf(se);
  const int yy = (int)y[row];
  if (loss_out && lane == 0) {
    const float nll = lse - (z[yy] - mx);                       // -logp[y]
    const float mean_nlp = lse - sz / (float)classes;           // mean_c(-logp[c])
    loss_out[row] = (1.0f - smoothing) * nll + smoothing * mean_nlp;
  }
  if (dlogits) {
    float* d = dlogits + (size_t)row * classes;
    const float inv = 1.0f / se, off = smoothing / (float)classes;
    for (int c = lane; c < classes; c += 64) {
      float t = expf(z[c] - mx) * inv - off;
      if (c == yy) t -= (1.0f - smoothing);
      d[c] = scale * t;
    }
  }
}

inline unsigned grid_for(size_t n4) {
  size_t b = (n4 + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > 256u * 32u) b = 256u * 32u;   // 32 workgroups per CU, grid-stride beyond that
  return (unsigned)b;
}
}  // namespace

// hyper-parameters arrive as doubles (they are Python floats in the reference's configs): 1 - x constants are
// formed in double and rounded to fp32 once, exactly as torch does with its scalar arguments
extern "C" int rart_sgd_step_f32(float* param, float* grad, float* momentum_buf, float* ema, size_t n, double lr,
                                 double momentum, double weight_decay, int nesterov, double grad_scale, double ema_decay,
                                 int zero_grad, rart_stream_t stream) {
  RART_CHECK_ARG(param && grad && momentum_buf && n > 0, "rart_sgd_step_f32: null buffer or empty arena");
  RART_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf | (uintptr_t)ema) % 16 == 0,
                 "rart_sgd_step_f32: arenas must be 16-byte aligned");
  RART_CHECK_ARG(!nesterov || momentum > 0.0, "rart_sgd_step_f32: nesterov needs momentum > 0 (torch.optim.SGD)");
  SgdArgs a{(float)lr, (float)momentum, (float)weight_decay, (float)grad_scale, (float)ema_decay, (float)(1.0 - ema_decay),
            nesterov, zero_grad};
  hipLaunchKernelGGL(k_sgd_step, dim3(grid_for(n / 4)), dim3(kBlock), 0, (hipStream_t)stream, param, grad, momentum_buf,
                     ema, n, a);
  RART_CHECK_LAUNCH("rart_sgd_step_f32");
  return RART_OK;
}

extern "C" int rart_adamw_step_f32(float* param, float* grad, float* exp_avg, float* exp_avg_sq, float* ema, size_t n,
                                   double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                                   double grad_scale, double ema_decay, int zero_grad, rart_stream_t stream) {
  RART_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0, "rart_adamw_step_f32: null buffer or empty arena");
  RART_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)ema) % 16 == 0,
                 "rart_adamw_step_f32: arenas must be 16-byte aligned");
  RART_CHECK_ARG(step >= 1, "rart_adamw_step_f32: step counts from 1");
  AdamArgs a;
  a.lr = (float)lr; a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps; a.grad_scale = (float)grad_scale;
  a.ema_decay = (float)ema_decay; a.zero_grad = zero_grad;
  a.decay_mul = (float)(1.0 - lr * weight_decay);
  a.one_minus_beta1 = (float)(1.0 - beta1);
  a.one_minus_beta2 = (float)(1.0 - beta2);
  a.one_minus_ema = (float)(1.0 - ema_decay);
  a.bias1 = (float)(1.0 - pow(beta1, (double)step));
  a.sqrt_bias2 = (float)sqrt(1.0 - pow(beta2, (double)step));
  hipLaunchKernelGGL(k_adamw_step, dim3(grid_for(n / 4)), dim3(kBlock), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, ema, n, a);
  RART_CHECK_LAUNCH("rart_adamw_step_f32");
  return RART_OK;
}

extern "C" int rart_ema_update_f32(float* ema, const float* param, size_t n, double decay, rart_stream_t stream) {
  RART_CHECK_ARG(ema && param && n > 0, "rart_ema_update_f32: null buffer or empty arena");
  hipLaunchKernelGGL(k_ema_update, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, ema, param, n, (float)decay,
                     (float)(1.0 - decay));
  RART_CHECK_LAUNCH("rart_ema_update_f32");
  return RART_OK;
}

extern "C" int rart_label_smooth_ce_f32(const float* logits, const int64_t* labels, int batch, int classes, double smoothing,
                                        double scale, float* loss_out, float* dlogits_out, rart_stream_t stream) {
  RART_CHECK_ARG(logits && labels && batch > 0 && classes > 0, "rart_label_smooth_ce_f32: bad arguments");
  RART_CHECK_ARG(smoothing >= 0.0 && smoothing <= 1.0, "rart_label_smooth_ce_f32: label_smoothing must be in [0, 1]");
  RART_CHECK_ARG(loss_out || dlogits_out, "rart_label_smooth_ce_f32: nothing to compute");
  const int rows_per_block = kBlock / 64;
  hipLaunchKernelGGL(k_label_smooth_ce, dim3((batch + rows_per_block - 1) / rows_per_block), dim3(kBlock), 0,
                     (hipStream_t)stream, logits, labels, batch, classes, (float)smoothing, (float)scale, loss_out, dlogits_out);
  RART_CHECK_LAUNCH("rart_label_smooth_ce_f32");
  return RART_OK;
}
