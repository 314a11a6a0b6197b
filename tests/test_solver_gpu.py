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


def test_vit_attacked_eval_transfer_results_and_adv_train(tmp_path):
    """ViT-B/16 through the same solver paths: PGD evaluation on the HIP engine, result files + AR metric, a transfer
    run (adversarial examples crafted on ViT, scored on ResNet-50) and two AdamW adversarial-training iterations on the
    HIP train engine."""
    import os
    from robustart_amd import metrics as M
    from robustart_amd.train import cls_solver as S
    rank, world, device = S.init_dist()
    a = _Args()
    a.save_dir, a.src_name, a.tgt_name, a.tgt_type = str(tmp_path), 'vitA', None, None
    clean = S.evaluate(_cfg('vit_base', n=6, bs=6), a, rank, world, device)
    a.attack = 'pgd_linf'
    adv = S.evaluate(_cfg('vit_base', n=6, bs=6), a, rank, world, device)
    # (each evaluate() builds its own randomly initialised model, so accuracies are not comparable across calls)
    assert clean['count'] == 6 and adv['count'] == 6 and adv['noise'] == 'pgd_linf'
    p_clean = os.path.join(str(tmp_path), 'vitA', 'none_0', 'results.txt.all')
    p_adv = os.path.join(str(tmp_path), 'vitA', 'pgd_linf_%.3f' % (2 / 255), 'results.txt.all')
    assert len(open(p_clean).readlines()) == 6 == len(open(p_adv).readlines())
    n_clean_ok = sum(1 for ln in open(p_clean) if M.parse_line(ln)[0] == M.parse_line(ln)[1])
    if n_clean_ok:
        ar = M.AdvRobustEvaluator().eval(p_clean, p_adv)
        assert 0.0 <= ar <= 100.0
    a.tgt_name, a.tgt_type = 'r50B', 'resnet50_official'
    tr = S.evaluate(_cfg('vit_base', n=6, bs=6), a, rank, world, device)
    assert tr['count'] == 6
    assert os.path.exists(os.path.join(str(tmp_path), 'vitA_To_r50B', 'pgd_linf_%.3f' % (2 / 255), 'results.txt.all'))
    a2 = _Args()
    cfg = _cfg('vit_base', n=8, bs=4)
    cfg.update({'adv_train': {'eps': '4/255', 'steps': 2, 'rel_stepsize': 0.5}, 'label_smooth': 0.1, 'max_iter': 2,
                'optimizer': {'type': 'AdamW', 'no_wd': {'norm': True, 'fc': True}, 'kwargs': {'weight_decay': 0.05}},
                'lr_scheduler': {'kwargs': {'base_lr': 1e-5, 'warmup_lr': 5e-4}},
                'ema': {'enable': True, 'kwargs': {'decay': 0.99}}})
    loss, _ = S.train(cfg, a2, rank, world, device)
    assert loss == loss and loss > 0
