"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, fp32 op by op) of the training-side step arithmetic.

The reference's solver (RobustART/train/__init__.py:1 -> absent `prototype` submodule) is configured by
exprs/nips_benchmark/pgd_adv_train/resnet50/config.yaml:11-33 (SGD nesterov, momentum 0.9, weight_decay 1e-4,
label_smooth 0.1, EMA decay 0.9999, cosine schedule with warm-up) and new_adv_train/vit_base/config.yaml:11-38
(AdamW); its arithmetic is torch.optim.SGD / torch.optim.AdamW / F.cross_entropy(label_smoothing=...).
Pinned in tests/test_train_steps_cpu.py against those torch implementations run on the CPU (same container,
same torch the GPU box has).  Only tests/ may import this module.
"""
import numpy as np

f32 = np.float32


def sgd_step(p, g, m, lr, momentum=0.9, weight_decay=1e-4, nesterov=True, grad_scale=1.0):
    """torch.optim.SGD single-tensor step (dampening 0; momentum buffer starts at 0 == torch's clone at step 1).
    Returns (p_new, m_new)."""
    p, g, m = p.astype(f32), g.astype(f32), m.astype(f32)
    d = g * f32(grad_scale)
    if weight_decay != 0:
        d = d + f32(weight_decay) * p
    m = f32(momentum) * m + d
    d = d + f32(momentum) * m if nesterov else m
    p = p - f32(lr) * d
    return p.astype(f32), m.astype(f32)


def adamw_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.05, grad_scale=1.0):
    """torch.optim.AdamW single-tensor step (amsgrad False, maximize False).  Returns (p, m, v)."""
    p, g, m, v = p.astype(f32), g.astype(f32) * f32(grad_scale), m.astype(f32), v.astype(f32)
    # torch forms 1 - lr*wd, 1 - beta1, 1 - beta2 in Python doubles and rounds them once
    p = p * f32(1.0 - float(lr) * float(weight_decay))
    m = m + f32(1.0 - float(beta1)) * (g - m)
    v = f32(beta2) * v + f32(1.0 - float(beta2)) * g * g
    bias1 = f32(1.0 - float(beta1) ** step)
    sqrt_bias2 = f32(np.sqrt(1.0 - float(beta2) ** step))
    denom = np.sqrt(v) / sqrt_bias2 + f32(eps)
    p = p - (f32(lr) / bias1) * (m / denom)
    return p.astype(f32), m.astype(f32), v.astype(f32)


def ema_update(ema, p, decay=0.9999):
    return (f32(decay) * ema.astype(f32) + f32(1.0 - float(decay)) * p.astype(f32)).astype(f32)


def label_smooth_ce(logits, y, smoothing=0.1, scale=1.0):
    """F.cross_entropy(logits, y, label_smoothing=s, reduction='none') and d(scale*loss)/dlogits.
    float64 internally (the kernel's expf/logf differ from numpy's in the last ulp; tests use 1e-6)."""
    z = logits.astype(np.float64)
    zc = z - z.max(axis=1, keepdims=True)
    lse = np.log(np.exp(zc).sum(axis=1))
    logp = zc - lse[:, None]
    n, c = z.shape
    nll = -logp[np.arange(n), y]
    loss = (1.0 - smoothing) * nll + smoothing * (-logp.mean(axis=1))
    target = np.full((n, c), smoothing / c)
    target[np.arange(n), y] += 1.0 - smoothing
    grad = scale * (np.exp(logp) - target)
    return loss, grad


def cosine_lr(step, total, base_lr, warmup_lr, warmup_steps, min_lr=0.0):
    """Linear warm-up base_lr -> warmup_lr over warmup_steps, then cosine warmup_lr -> min_lr
    (pgd_adv_train/resnet50/config.yaml:17-25: base_lr 0.1, warmup_lr 0.4, warmup 2 epochs)."""
    if warmup_steps > 0 and step < warmup_steps:
        return base_lr + (warmup_lr - base_lr) * step / warmup_steps
    t = (step - warmup_steps) / max(total - warmup_steps, 1)
    return min_lr + 0.5 * (warmup_lr - min_lr) * (1.0 + np.cos(np.pi * min(max(t, 0.0), 1.0)))
