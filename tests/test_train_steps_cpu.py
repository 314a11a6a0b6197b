"""Oracle pin for the training-side step arithmetic: oracle/train_ref.py vs torch.optim / F.cross_entropy on CPU."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import train_ref as T


def _rand(n, seed, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(n) * scale).astype(np.float32)


def test_sgd_nesterov_matches_torch_optim():
    p0, n = _rand(4099, 0), 4099
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.SGD([tp], lr=0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    p, m = p0.copy(), np.zeros(n, np.float32)
    for step in range(4):
        g = _rand(n, 10 + step, 0.01)
        lr = 0.1 + 0.05 * step
        for gp in opt.param_groups:
            gp['lr'] = lr
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        p, m = T.sgd_step(p, g, m, lr, 0.9, 1e-4, True)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(m, opt.state[tp]['momentum_buffer'].numpy(), rtol=2e-6, atol=1e-8)


def test_sgd_plain_momentum_and_no_decay():
    p0, n = _rand(1000, 1), 1000
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.SGD([tp], lr=0.05, momentum=0.8, weight_decay=0.0, nesterov=False)
    p, m = p0.copy(), np.zeros(n, np.float32)
    for step in range(3):
        g = _rand(n, 20 + step, 0.1)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        p, m = T.sgd_step(p, g, m, 0.05, 0.8, 0.0, False)
    np.testing.assert_allclose(p, tp.detach().numpy(), rtol=2e-6, atol=1e-7)


def test_adamw_matches_torch_optim():
    p0, n = _rand(3001, 2), 3001
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.AdamW([tp], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    p, m, v = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    for step in range(1, 6):
        g = _rand(n, 30 + step, 0.02)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        p, m, v = T.adamw_step(p, g, m, v, 1e-3, step, 0.9, 0.999, 1e-8, 0.05)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=3e-6, atol=2e-7)
    np.testing.assert_allclose(v, opt.state[tp]['exp_avg_sq'].numpy(), rtol=3e-6, atol=1e-12)


def test_label_smooth_ce_matches_torch():
    rs = np.random.RandomState(3)
    z = (rs.standard_normal((7, 1000)) * 3).astype(np.float32)
    y = rs.randint(0, 1000, 7)
    for s in (0.0, 0.1, 0.3):
        zt = torch.from_numpy(z.copy()).requires_grad_(True)
        lt = F.cross_entropy(zt, torch.from_numpy(y), label_smoothing=s, reduction='none')
        (lt.sum() / 7).backward()
        loss, grad = T.label_smooth_ce(z, y, s, scale=1.0 / 7)
        np.testing.assert_allclose(loss, lt.detach().numpy(), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(grad, zt.grad.numpy(), rtol=1e-5, atol=1e-8)


def test_ema_and_cosine_schedule():
    e, p = _rand(100, 4), _rand(100, 5)
    np.testing.assert_allclose(T.ema_update(e, p, 0.9999), 0.9999 * e + (1 - 0.9999) * p, rtol=1e-5, atol=1e-7)
    assert T.cosine_lr(0, 100, 0.1, 0.4, 10) == 0.1
    assert abs(T.cosine_lr(10, 100, 0.1, 0.4, 10) - 0.4) < 1e-12
    assert abs(T.cosine_lr(100, 100, 0.1, 0.4, 10)) < 1e-12
    assert T.cosine_lr(55, 100, 0.1, 0.4, 10) == 0.5 * 0.4 * (1 + np.cos(np.pi * 0.5))
