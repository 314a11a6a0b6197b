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


def _write_jpegs(root, n=16):
    import os
    import numpy as np
    from PIL import Image
    rs = np.random.RandomState(4)
    os.makedirs(os.path.join(root, 'val'), exist_ok=True)
    lines = []
    for i in range(n):
        h, w = 230 + 13 * (i % 5), 250 + 17 * (i % 4)
        # smooth content + noise so the JPEG is a plausible photograph-like array
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 100 * np.sin(xx / (9.0 + i)) * np.cos(yy / 17.0), 127 + 90 * np.cos(xx / 23.0 + i),
                         127 + 80 * np.sin((xx + yy) / (11.0 + i))], -1)
        arr = np.clip(base + rs.randn(h, w, 3) * 12, 0, 255).astype(np.uint8)
        name = 'img_%02d.jpg' % i
        Image.fromarray(arr).save(os.path.join(root, 'val', name), quality=92)
        lines.append((name, (i * 37) % 1000))
    meta = os.path.join(root, 'val.txt')
    with open(meta, 'w') as f:
        f.write(''.join('%s %d\n' % ln for ln in lines))
    return os.path.join(root, 'val'), meta, lines


def test_evaluate_from_files_onecrop_matches_pillow_and_the_torch_module_path(tmp_path):
    """`data.read_from: fs` with the reference's test block (root_dir / meta_file / image_reader pil / ONECROP,
    exp/imagenet_c_loop_mini/config_vit_base.yaml:80-104): 16 JPEGs on disk -> PIL decode on the host -> resize 256 + centre
    crop 224 on the GPU (bit-exact with Pillow) -> clean / gaussian_noise / pgd_linf evaluation on the HIP engine; the result
    files equal those of the fp32 torch module fed by Pillow's own resize (predictions identical, scores within 1e-4 of
    the logit scale for the reference-precision engine)."""
    import json
    import os
    import numpy as np
    from PIL import Image
    from robustart_amd.model import get_model
    from robustart_amd.noise import AddNoise, imagenet_c as C, rng
    from robustart_amd.train import cls_solver as S
    rank, world, device = S.init_dist()
    root, meta, lines = _write_jpegs(str(tmp_path))
    cfg = {'model': {'type': 'resnet50_official'},
           'data': {'type': 'imagenet', 'read_from': 'fs', 'batch_size': 8, 'input_size': 224, 'test_resize': 256,
                    'test': {'root_dir': root, 'meta_file': meta, 'image_reader': {'type': 'pil'},
                             'sampler': {'type': 'distributed'}, 'transforms': {'type': 'ONECROP'},
                             'evaluator': {'type': 'imagenet', 'kwargs': {'topk': [1, 5]}}}}}
    ds = S.make_dataset(cfg['data'], 0, 224, 'test')
    got, labs = ds.batch(list(range(16)), 'cuda')
    want = []
    for name, _ in lines:
        with Image.open(os.path.join(root, name)) as im:
            r = im.convert('RGB').resize((256, 256), Image.BILINEAR)
        want.append(np.asarray(r)[16:240, 16:240])
    assert np.array_equal(got.cpu().numpy(), np.stack(want))                 # torchvision Resize([256,256]) + CenterCrop(224)
    assert labs.tolist() == [lab for _, lab in lines]
    torch.manual_seed(11)
    from robustart_amd.model.resnet_torch import randomize_bn_stats
    model = randomize_bn_stats(get_model(cfg['model']), 11).eval()
    for p_ in model.parameters():
        p_.requires_grad_(False)
    mean = torch.tensor(S.IMAGENET_MEAN, device='cuda').view(1, 3, 1, 1)
    std = torch.tensor(S.IMAGENET_STD, device='cuda').view(1, 3, 1, 1)

    def read(path):
        return [json.loads(ln) for ln in open(path)]
    a = _Args()
    a.precision, a.save_dir, a.src_name, a.tgt_name, a.tgt_type = 'fp32x', str(tmp_path / 'res'), 'r50', None, None
    # ---- clean
    res = S.evaluate(cfg, a, rank, world, device, model=model)
    assert res['count'] == 16
    recs = read(os.path.join(a.save_dir, 'r50', 'none_0', 'results.txt.all'))
    x = torch.from_numpy(np.stack(want)).cuda()
    f32 = lambda z: model((z - mean) / std)      # noqa: E731
    lt = f32(x.permute(0, 3, 1, 2).float() / 255.0)
    sc = torch.tensor([r['score'] for r in recs], device='cuda')
    assert [r['index'] for r in recs] == list(range(16)) and [r['label'] for r in recs] == labs.tolist()
    assert (sc - lt).abs().max().item() <= 1e-4 * lt.abs().max().item()
    assert [r['prediction'] for r in recs] == lt.argmax(1).tolist()
    # ---- gaussian_noise severity 3 (the kernel's own native draws: same corrupted pixels for both model paths)
    a.corruption = 'gaussian_noise'
    S.evaluate(cfg, a, rank, world, device, model=model)
    recs = read(os.path.join(a.save_dir, 'r50', 'gaussian_noise_3', 'results.txt.all'))
    xc = x.clone()
    for s in (0, 8):
        part = xc[s:s + 8].contiguous()
        C.corrupt_batch_(part, C.CORRUPTION_NAMES.index('gaussian_noise'), 3, seed=a.seed, sample_offset=s)
        xc[s:s + 8] = part
    assert not torch.equal(xc, x)
    lt = f32(xc.permute(0, 3, 1, 2).float() / 255.0)
    sc = torch.tensor([r['score'] for r in recs], device='cuda')
    assert (sc - lt).abs().max().item() <= 1e-4 * lt.abs().max().item()
    # ---- pgd_linf: the attack driven by the HIP engine vs by torch autograd through the fp32 module, same random starts
    a.corruption, a.attack, a.steps = None, 'pgd_linf', 3
    S.evaluate(cfg, a, rank, world, device, model=model)
    recs = read(os.path.join(a.save_dir, 'r50', 'pgd_linf_%.3f' % (2 / 255), 'results.txt.all'))
    att = AddNoise('pgd_linf')
    att.config.update(f_model=f32, eps=2 / 255, steps=3)
    preds, scores = [], []
    for s in (0, 8):
        rng.manual_seed(a.seed, s)
        xa = att.add_noise((x[s:s + 8].permute(0, 3, 1, 2).float() / 255.0).contiguous(), labs[s:s + 8])
        with torch.no_grad():
            la = f32(xa)
        preds += la.argmax(1).tolist()
        scores.append(la)
    scores = torch.cat(scores)
    sc = torch.tensor([r['score'] for r in recs], device='cuda')
    agree = sum(int(p == r['prediction']) for p, r in zip(preds, recs))
    err = (sc - scores).abs().max().item() / scores.abs().max().item()
    print('pgd_linf from files: prediction agreement %d / 16, score error %.2e of scale' % (agree, err))
    assert agree >= 15 and err <= 2e-2          # sign steps may differ where the gradient is ~0: not a 1e-4 quantity


def test_hip_training_resumes_bit_identically(tmp_path):
    """--recover is a RESUME (optimizer arena, EMA, schedule position), on the HIP train engine + HIP optimizer:
    6 iterations straight == 3 iterations, checkpoint, 3 more."""
    import os
    from robustart_amd.train import cls_solver as S
    rank, world, device = S.init_dist()

    class A(_Args):
        train_engine = 'hip'
        recover = None
        ckpt_dir = None

    def cfg(save_dir, **saver):
        return {'model': {'type': 'resnet50_official'}, 'data': {'read_from': 'fake', 'fake_size': 64, 'batch_size': 8, 'input_size': 64},
                'label_smooth': 0.1, 'max_iter': 6, 'ema': {'enable': True, 'kwargs': {'decay': 0.9}},
                'adv_train': {'eps': '4/255', 'steps': 1, 'rel_stepsize': 1.0},
                'lr_scheduler': {'kwargs': {'base_lr': 0.01, 'warmup_lr': 0.02, 'warmup_steps': 2}},
                'saver': dict(save_dir=save_dir, print_freq=100, **saver)}
    torch.manual_seed(5)
    _, m_full = S.train(cfg(str(tmp_path / 'full')), A(), rank, world, device)
    torch.manual_seed(5)
    S.train(cfg(str(tmp_path / 'part'), val_freq=3, save_many=True), A(), rank, world, device)
    a = A()
    a.recover = os.path.join(str(tmp_path / 'part'), 'ckpt_3.pth.tar')
    torch.manual_seed(99)                            # the initial weights come from the checkpoint, not from the seed
    _, m_res = S.train(cfg(str(tmp_path / 'res')), a, rank, world, device)
    assert S.train.start_iter == 3
    worst = 0.0
    for (k, v), (_, w) in zip(m_full.state_dict().items(), m_res.state_dict().items()):
        worst = max(worst, (v.float() - w.float()).abs().max().item() / (v.float().abs().max().item() + 1e-12))
    print('resume vs straight run: worst relative parameter difference %.3g' % worst)
    assert worst <= 1e-6
    ck_a = torch.load(os.path.join(str(tmp_path / 'full'), 'ckpt.pth.tar'), weights_only=True)
    ck_b = torch.load(os.path.join(str(tmp_path / 'res'), 'ckpt.pth.tar'), weights_only=True)
    assert ck_a['last_iter'] == ck_b['last_iter'] == 6
    assert (ck_a['optimizer']['m'] - ck_b['optimizer']['m']).abs().max().item() <= 1e-6 * ck_a['optimizer']['m'].abs().max().item()
    for k in ck_a['ema']:
        assert torch.allclose(ck_a['ema'][k].float(), ck_b['ema'][k].float(), rtol=1e-5, atol=1e-7), k
