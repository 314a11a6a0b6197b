"""cls_solver-shaped evaluation on the GPU: clean / ImageNet-C / PGD evaluation through the HIP engines."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Args:
    engine = 'hip'
    corruption = None
    attack = None
    eps = '2/255'
    steps = 2
    severity = 3
    seed = 0
    max_iter = 2


def _cfg(mtype, n=12, bs=6):
    return {'model': {'type': mtype, 'kwargs': {'num_classes': 1000}},
            'data': {'fake_size': n, 'batch_size': bs, 'input_size': 224, 'read_from': 'fake'}}


@pytest.mark.parametrize('mtype', ['resnet50_official', 'vit_base'])
def test_evaluate_clean_and_corrupted(mtype):
    from robustart_amd.train import cls_solver as S
    rank, world, device = S.init_dist()
    a = _Args()
    res = S.evaluate(_cfg(mtype), a, rank, world, device)
    assert res['count'] == 12 and 0.0 <= res['top1'] <= res['top5'] <= 1.0
    a.corruption = 'gaussian_noise'
    res2 = S.evaluate(_cfg(mtype), a, rank, world, device)
    assert res2['count'] == 12 and res2['noise'] == 'gaussian_noise'


def test_evaluate_under_pgd_and_adv_train_step():
    from robustart_amd.train import cls_solver as S
    rank, world, device = S.init_dist()
    a = _Args()
    a.attack = 'pgd_linf'
    res = S.evaluate(_cfg('resnet50_official', n=6, bs=6), a, rank, world, device)
    assert res['count'] == 6 and res['noise'] == 'pgd_linf'
    cfg = _cfg('resnet50_official', n=8, bs=4)
    cfg.update({'adv_train': {'eps': '4/255', 'steps': 2, 'rel_stepsize': 0.5}, 'label_smooth': 0.1, 'max_iter': 2,
                'ema': {'enable': True, 'kwargs': {'decay': 0.99}}})
    loss, _ = S.train(cfg, a, rank, world, device)
    assert loss == loss and loss > 0
