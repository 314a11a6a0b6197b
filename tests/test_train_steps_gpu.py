"""HIP training-side step kernels vs the oracle (oracle/train_ref.py, pinned to torch.optim in the CPU suite)."""
import numpy as np
import pytest
import torch

from oracle import train_ref as T

pytestmark = pytest.mark.gpu


def _lib():
    from robustart_amd import _lib
    return _lib, _lib.load()


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _rand(n, seed, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(n) * scale).astype(np.float32)


@pytest.mark.parametrize('n', [4, 1027, 262147])
@pytest.mark.parametrize('nesterov,wd,ema', [(1, 1e-4, True), (0, 0.0, False)])
def test_sgd_step_kernel_matches_oracle(n, nesterov, wd, ema):
    L, lib = _lib()
    p, m = _rand(n, 0), np.zeros(n, np.float32)
    e = p.copy()
    dp, dm, de = _dev(p), _dev(m), _dev(e)
    for step in range(3):
        g = _rand(n, 10 + step, 0.05)
        dg = _dev(g)
        lr, scale = 0.1 + 0.1 * step, 0.5
        L.check(lib.rart_sgd_step_f32(dp.data_ptr(), dg.data_ptr(), dm.data_ptr(), de.data_ptr() if ema else None, n, lr,
                                      0.9, wd, nesterov, scale, 0.999, 1, L.stream_ptr()))
        p, m = T.sgd_step(p, g, m, lr, 0.9, wd, bool(nesterov), grad_scale=scale)
        e = T.ema_update(e, p, 0.999)
        torch.cuda.synchronize()
        assert torch.count_nonzero(dg).item() == 0                      # fused gradient reset
        np.testing.assert_array_equal(dp.cpu().numpy(), p)              # op-by-op fp32: bit exact
        np.testing.assert_array_equal(dm.cpu().numpy(), m)
        if ema:
            np.testing.assert_array_equal(de.cpu().numpy(), e)


@pytest.mark.parametrize('n', [5, 40003])
def test_adamw_step_kernel_matches_oracle(n):
    L, lib = _lib()
    p, m, v = _rand(n, 1), np.zeros(n, np.float32), np.zeros(n, np.float32)
    dp, dm, dv = _dev(p), _dev(m), _dev(v)
    for step in range(1, 5):
        g = _rand(n, 20 + step, 0.02)
        dg = _dev(g)
        L.check(lib.rart_adamw_step_f32(dp.data_ptr(), dg.data_ptr(), dm.data_ptr(), dv.data_ptr(), None, n, 1e-3, 0.9,
                                        0.999, 1e-8, 0.05, step, 1.0, 0.0, 0, L.stream_ptr()))
        p, m, v = T.adamw_step(p, g, m, v, 1e-3, step, 0.9, 0.999, 1e-8, 0.05)
        torch.cuda.synchronize()
        assert torch.equal(dg.cpu(), torch.from_numpy(g))               # zero_grad = 0 leaves the gradient alone
        np.testing.assert_allclose(dp.cpu().numpy(), p, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(dv.cpu().numpy(), v, rtol=1e-6, atol=1e-14)


def test_optimizer_argument_checks():
    L, lib = _lib()
    t = torch.zeros(16, device='cuda')
    assert lib.rart_sgd_step_f32(None, t.data_ptr(), t.data_ptr(), None, 16, 0.1, 0.9, 0.0, 1, 1.0, 0.0, 1, None) != 0
    assert lib.rart_sgd_step_f32(t.data_ptr(), t.data_ptr(), t.data_ptr(), None, 16, 0.1, 0.0, 0.0, 1, 1.0, 0.0, 1,
                                 None) != 0                                # nesterov without momentum (torch raises too)
    assert lib.rart_adamw_step_f32(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), None, 16, 1e-3, 0.9, 0.999,
                                   1e-8, 0.0, 0, 1.0, 0.0, 0, None) != 0   # step counts from 1


@pytest.mark.parametrize('batch,classes,s', [(1, 10, 0.0), (7, 1000, 0.1), (256, 1000, 0.1), (5, 63, 0.3)])
def test_label_smooth_ce_kernel(batch, classes, s):
    from robustart_amd.train.arena import label_smooth_ce
    rs = np.random.RandomState(batch)
    z = (rs.standard_normal((batch, classes)) * 4).astype(np.float32)
    y = rs.randint(0, classes, batch)
    loss, dl = label_smooth_ce(_dev(z), _dev(y.astype(np.int64)), s, 1.0 / batch)
    ref_loss, ref_grad = T.label_smooth_ce(z, y, s, 1.0 / batch)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(dl.cpu().numpy(), ref_grad, rtol=2e-5, atol=2e-9)
    # and against torch on the GPU
    zt = _dev(z).requires_grad_(True)
    lt = torch.nn.functional.cross_entropy(zt, _dev(y.astype(np.int64)), label_smoothing=s, reduction='none')
    (lt.sum() / batch).backward()
    np.testing.assert_allclose(loss.cpu().numpy(), lt.detach().cpu().numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(dl.cpu().numpy(), zt.grad.cpu().numpy(), rtol=2e-5, atol=2e-9)


def test_solver_train_hip_optimizer_tracks_torch_scaffold():
    """3 iterations of cls_solver.train on a tiny model: HIP loss/optimizer/EMA path vs torch.optim on the same arena."""
    import robustart_amd.model as M
    from robustart_amd.train import cls_solver as S

    def tiny(**kw):
        torch.manual_seed(0)
        return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, stride=4), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                                   torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(8, 1000))
    M._REGISTRY['tiny_bn_test'] = tiny

    class Args:
        pass
    outs = {}
    for engine in ('hip', 'torch'):
        args = Args()
        args.engine, args.max_iter = engine, 3
        cfg = {'model': {'type': 'tiny_bn_test'}, 'data': {'fake_size': 16, 'batch_size': 8, 'input_size': 32},
               'label_smooth': 0.1, 'ema': {'enable': True, 'kwargs': {'decay': 0.9}}, 'max_iter': 3, 'bf16': False,
               'optimizer': {'type': 'SGD', 'no_wd': {'norm': True},
                             'kwargs': {'nesterov': True, 'momentum': 0.9, 'weight_decay': 1e-2}},
               'lr_scheduler': {'kwargs': {'base_lr': 0.05, 'warmup_lr': 0.1}}}
        loss, model = S.train(cfg, args, 0, 1, torch.device('cuda'))
        outs[engine] = (loss, torch.cat([p.detach().flatten() for p in model.parameters()]).cpu())
    assert abs(outs['hip'][0] - outs['torch'][0]) < 1e-4
    np.testing.assert_allclose(outs['hip'][1].numpy(), outs['torch'][1].numpy(), rtol=1e-4, atol=1e-6)
