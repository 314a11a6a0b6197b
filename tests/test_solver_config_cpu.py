"""cls_solver host logic on the CPU: the reference's YAML key set (schedule in epochs, `saver` keys), checkpoint resume
(`--recover`: optimizer state, EMA, schedule position), safe checkpoint loading, and the file-backed dataset's host side
(meta files, PIL decode, crop boxes).  Reference: exprs/nips_benchmark/pgd_adv_train/resnet50/config.yaml,
exprs/exp/imagenet_c_loop_mini/config_vit_base.yaml:80-104 (the dict below is rebuilt key by key, not copied)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from robustart_amd.train import cls_solver as S


def reference_shaped_config(**over):
    """A dict with exactly the key set of the reference's pgd_adv_train/resnet50/config.yaml."""
    cfg = {
        'model': {'type': 'resnet50_official', 'kwargs': {'bn': {'use_sync_bn': False, 'kwargs': {}}}},
        'dist': {'sync': True},
        'optimizer': {'type': 'SGD', 'kwargs': {'nesterov': True, 'momentum': 0.9, 'weight_decay': 0.0001}},
        'lr_scheduler': {'type': 'CosineEpoch',
                         'kwargs': {'base_lr': 0.1, 'warmup_lr': 0.4, 'min_lr': 0.0, 'warmup_epoch': 2, 'max_epoch': 100}},
        'label_smooth': 0.1,
        'ema': {'enable': True, 'kwargs': {'decay': 0.9999}},
        'data': {'type': 'imagenet', 'read_from': 'fake', 'use_dali': True, 'batch_size': 32, 'num_workers': 4,
                 'pin_memory': True, 'input_size': 224, 'test_resize': 256,
                 'train': {'root_dir': '/mnt/lustre/share/images/train/', 'meta_file': '/mnt/lustre/share/images/meta/train.txt',
                           'image_reader': {'type': 'pil'}, 'sampler': {'type': 'distributed_iteration'},
                           'transforms': {'type': 'STANDARD'}},
                 'test': {'root_dir': '/mnt/lustre/share/images/val/', 'meta_file': '/mnt/lustre/share/images/meta/val.txt',
                          'image_reader': {'type': 'pil'}, 'sampler': {'type': 'distributed'},
                          'transforms': {'type': 'ONECROP'},
                          'evaluator': {'type': 'imagenet', 'kwargs': {'topk': [1, 5]}}}},
        'saver': {'print_freq': 10, 'val_freq': 5000, 'save_many': False},
    }
    for k, v in over.items():
        node = cfg
        *path, leaf = k.split('.')
        for p in path:
            node = node[p]
        node[leaf] = v
    return cfg


def test_epoch_keys_of_the_reference_yaml_resolve_to_iterations():
    cfg = reference_shaped_config()
    # ImageNet-1k, the reference's launch of this config: 16 GPUs x 32 images (run.sh:2)
    max_iter, warm = S.resolve_schedule(cfg, 1281167, 32, 16)
    per_epoch = -(-1281167 // 512)
    assert (max_iter, warm) == (100 * per_epoch, 2 * per_epoch) == (250300, 5006)
    # the iteration keys (commented in the reference file) win over the epoch keys; this build's top-level max_iter over both
    cfg['lr_scheduler']['kwargs'].update(max_iter=125000, warmup_steps=2500)
    assert S.resolve_schedule(cfg, 1281167, 32, 16) == (125000, 2500)
    cfg['max_iter'] = 7
    assert S.resolve_schedule(cfg, 1281167, 32, 16) == (7, 7)
    # neither: the caller's default, warm-up a twentieth of it
    assert S.resolve_schedule({}, 100, 10, 1, default_max_iter=40) == (40, 2)
    # the schedule those numbers drive: linear warm-up base_lr -> warmup_lr, cosine to min_lr
    lr = [S.cosine_lr(i, 250300, 0.1, 0.4, 5006, 0.0) for i in (0, 2503, 5006, 250299)]
    assert lr[0] == 0.1 and abs(lr[1] - 0.25) < 1e-12 and abs(lr[2] - 0.4) < 1e-12 and lr[3] < 1e-9


class _Args:
    engine = 'torch'
    train_engine = 'torch'
    corruption = None
    attack = None
    eps = '2/255'
    steps = 0
    severity = 3
    seed = 0
    max_iter = 20
    recover = None
    ckpt_dir = None


def _tiny_registered():
    import robustart_amd.model as M

    def tiny(**kw):
        torch.manual_seed(0)
        return torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, stride=4), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                                   torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(4, 1000))
    M._REGISTRY['tiny_cfg_test'] = tiny


def test_train_with_the_reference_key_set_and_resume(tmp_path):
    """The loop accepts the reference's dict as is (epochs -> iterations from the dataset length), saves at val_freq, and a
    run interrupted at the saved iteration and resumed with --recover ends bit-identical to the uninterrupted run:
    parameters, EMA (parameters and BatchNorm buffers), momentum, schedule position."""
    _tiny_registered()
    dev = torch.device('cpu')
    over = {'model.type': 'tiny_cfg_test', 'data.batch_size': 4, 'data.input_size': 32, 'lr_scheduler.kwargs.max_epoch': 3,
            'lr_scheduler.kwargs.warmup_epoch': 1, 'lr_scheduler.kwargs.base_lr': 0.01, 'lr_scheduler.kwargs.warmup_lr': 0.02,
            'ema.kwargs.decay': 0.9, 'saver.val_freq': 4}
    cfg = reference_shaped_config(**over)
    cfg['data']['fake_size'] = 12                                     # 3 iterations per epoch -> 9 iterations, warm-up 3
    cfg['bf16'] = False
    full_dir, part_dir = str(tmp_path / 'full'), str(tmp_path / 'part')
    cfg['saver']['save_dir'] = full_dir
    loss_full, m_full = S.train(cfg, _Args(), 0, 1, dev)
    assert S.train.start_iter == 0
    ck_full = torch.load(os.path.join(full_dir, 'ckpt.pth.tar'), weights_only=True)      # tensors and numbers only
    assert ck_full['last_iter'] == 9 and set(ck_full) >= {'model', 'ema', 'optimizer', 'last_iter'}
    # "interrupted": the same config with save_many, so the checkpoint of iteration 4 survives the later saves
    cfg_i = reference_shaped_config(**over)
    cfg_i['data']['fake_size'] = 12
    cfg_i['bf16'] = False
    cfg_i['saver'].update(save_dir=part_dir, save_many=True)
    S.train(cfg_i, _Args(), 0, 1, dev)
    mid = os.path.join(part_dir, 'ckpt_4.pth.tar')                  # written at iteration 4 and 8 (save_many keeps both)
    assert os.path.exists(mid) and os.path.exists(os.path.join(part_dir, 'ckpt_8.pth.tar'))
    ck_mid = torch.load(mid, weights_only=True)
    assert ck_mid['last_iter'] == 4 and 'torch' in ck_mid['optimizer']
    a = _Args()
    a.recover = mid
    cfg_r = reference_shaped_config(**over)
    cfg_r['data']['fake_size'] = 12
    cfg_r['bf16'] = False
    cfg_r['saver']['save_dir'] = str(tmp_path / 'resumed')
    loss_res, m_res = S.train(cfg_r, a, 0, 1, dev)
    assert S.train.start_iter == 4
    for (k, v), (_, w) in zip(m_full.state_dict().items(), m_res.state_dict().items()):
        assert torch.equal(v, w), k
    ck_res = torch.load(os.path.join(str(tmp_path / 'resumed'), 'ckpt.pth.tar'), weights_only=True)
    for k in ck_full['ema']:
        assert torch.equal(ck_full['ema'][k], ck_res['ema'][k]), k
    assert loss_full == loss_res
    # saver.pretrain.ignore.key: [optimizer, last_iter] -> fine-tuning: weights only, the schedule starts at 0
    cfg_f = reference_shaped_config(**over)
    cfg_f['data']['fake_size'] = 12
    cfg_f['bf16'] = False
    cfg_f['saver']['pretrain'] = {'path': mid, 'ignore': {'key': ['optimizer', 'last_iter']}}
    S.train(cfg_f, _Args(), 0, 1, dev)
    assert S.train.start_iter == 0


def test_checkpoints_load_without_unpickling_objects(tmp_path):
    class Evil:
        def __reduce__(self):
            return (os.system, ('true',))
    p = str(tmp_path / 'evil.pth')
    with open(p, 'wb') as f:
        pickle.dump({'model': {}, 'x': Evil()}, f)
    m = torch.nn.Linear(2, 2)
    with pytest.raises(RuntimeError, match='weights_only'):
        S.load_pretrain(m, p)
    good = str(tmp_path / 'good.pth')
    torch.save({'model': {('module.' + k): v for k, v in m.state_dict().items()}, 'last_iter': 3}, good)
    m2 = torch.nn.Linear(2, 2)
    ck = S.load_pretrain(m2, good)
    assert ck['last_iter'] == 3 and torch.equal(m2.weight, m.weight)
    with pytest.warns(RuntimeWarning, match='missing keys'):
        S.load_pretrain(torch.nn.Linear(2, 2), _partial(tmp_path, m), strict=False)


def _partial(tmp_path, m):
    p = str(tmp_path / 'partial.pth')
    torch.save({'weight': m.weight.detach()}, p)
    return p


def _write_images(root, n=6):
    from PIL import Image
    rs = np.random.RandomState(0)
    os.makedirs(os.path.join(root, 'val', 'n01'), exist_ok=True)
    lines, arrays = [], []
    for i in range(n):
        h, w = 40 + 7 * i, 60 + 5 * (i % 3)
        arr = (rs.rand(h, w, 3) * 255).astype(np.uint8)
        name = 'n01/img_%d.%s' % (i, 'png' if i % 2 else 'jpg')
        Image.fromarray(arr).save(os.path.join(root, 'val', name), **({} if i % 2 else {'quality': 95}))
        lines.append((name, i % 5))
        arrays.append(arr)
    return lines, arrays


def test_file_dataset_host_side(tmp_path):
    root = str(tmp_path)
    lines, arrays = _write_images(root)
    meta_txt = os.path.join(root, 'val.txt')
    with open(meta_txt, 'w') as f:
        f.write('\n'.join('%s %d' % ln for ln in lines) + '\n\n')
    meta_json = os.path.join(root, 'val.json')
    with open(meta_json, 'w') as f:
        for name, lab in lines:
            f.write(json.dumps({'filename': name, 'label': lab, 'label_name': 'x'}) + '\n')
    assert S.read_meta_file(meta_txt) == lines == S.read_meta_file(meta_json)
    dcfg = {'read_from': 'fs', 'input_size': 32, 'test_resize': 36,
            'test': {'root_dir': os.path.join(root, 'val'), 'meta_file': meta_txt, 'image_reader': {'type': 'pil'},
                     'transforms': {'type': 'ONECROP'}}}
    ds = S.make_dataset(dcfg, 0, 32, 'test')
    assert len(ds) == 6 and ds.transform == 'ONECROP' and ds.test_resize == 36
    arr, lab = ds.decode(1)                                         # PNG: lossless
    assert lab == 1 and np.array_equal(arr, arrays[1])
    arr0, _ = ds.decode(0)                                          # JPEG: the PIL decoder's output, same shape
    assert arr0.shape == arrays[0].shape and arr0.dtype == np.uint8
    y, x, h, w, flip = ds.box(3, arrays[3].shape[:2])
    assert (y, x, h, w, flip) == ds.box(3, arrays[3].shape[:2])     # a pure function of (seed, index)
    assert 0 <= y and y + h <= arrays[3].shape[0] and 0 <= x and x + w <= arrays[3].shape[1]
    with pytest.raises(RuntimeError, match='GPU'):
        ds.batch([0], 'cpu')                                        # the transform's arithmetic has no CPU fallback
    with pytest.raises(ValueError):
        S.make_dataset({'read_from': 'fs', 'test': {}}, 0, 32, 'test')
    with pytest.raises(NotImplementedError):
        S.make_dataset(dict(dcfg, test=dict(dcfg['test'], image_reader={'type': 'opencv'})), 0, 32, 'test')
    # the reference configs keep read_from: fake next to cluster paths: still the synthetic set
    fake = S.make_dataset(reference_shaped_config()['data'], 8, 32, 'test')
    assert isinstance(fake, S.FakeImageNet) and len(fake) == 8


def test_epoch_sampler_shuffles_a_class_sorted_list_and_is_a_pure_function_of_seed_and_epoch():
    """ImageNet's train list is sorted by class; file-order batches from contiguous per-rank ranges hold one class
    each.  The `distributed_iteration` sampler of the reference configs (pgd_adv_train/resnet50/config.yaml:39-41) shuffles:
    here a seeded permutation per epoch, dealt to the ranks by stride."""
    from robustart_amd.train import cls_solver as S
    n, bs, world = 1000, 10, 4
    labels = [i // 100 for i in range(n)]                                  # class-sorted meta file: 10 classes x 100
    samplers = [S.EpochSampler(n, bs, r, world, seed=5) for r in range(world)]
    assert samplers[0].per_epoch == 25
    seen = []
    for it in range(25):
        for sm in samplers:
            sel, ep = sm.batch(it)
            assert ep == 0 and len(sel) == bs
            seen += sel
            if it == 0:
                assert len({labels[i] for i in sel}) >= 4                  # mixed classes inside a batch
    assert sorted(seen) == list(range(n))                                  # one epoch visits every sample exactly once
    # epoch 1 is another permutation; both are reproducible from (seed, epoch) alone -- a resumed run sees the same batches
    e1 = samplers[0].batch(25)
    assert e1[1] == 1 and e1[0] != samplers[0].batch(0)[0]
    fresh = S.EpochSampler(n, bs, 0, world, seed=5)
    assert fresh.batch(25) == e1 and fresh.batch(3) == samplers[0].batch(3)
    assert S.EpochSampler(n, bs, 0, world, seed=6).batch(0)[0] != samplers[0].batch(0)[0]
    # the ranks of a larger world see the same GLOBAL batch as one rank with world x the batch
    one = S.EpochSampler(n, bs * world, 0, 1, seed=5)
    assert sorted(one.batch(7)[0]) == sorted(i for sm in samplers for i in sm.batch(7)[0])
    # ragged tail: wraps to the head of the permutation (DistributedSampler's padding): full batches always
    rag = S.EpochSampler(23, 4, 1, 2, seed=0)
    assert rag.per_epoch == 3 and all(len(rag.batch(k)[0]) == 4 for k in range(6))
    # shuffle=False keeps file order, still strided
    assert S.EpochSampler(100, 4, 1, 2, shuffle=False).batch(0)[0] == [1, 3, 5, 7]


def test_standard_transform_redraws_the_crop_every_epoch(tmp_path):
    """RandomResizedCrop / RandomHorizontalFlip redraw on every visit; the box is a pure function of
    (seed, epoch, index)."""
    from robustart_amd.train import cls_solver as S
    root = str(tmp_path)
    lines, arrays = _write_images(root)
    meta = os.path.join(root, 'train.txt')
    with open(meta, 'w') as f:
        f.write('\n'.join('%s %d' % ln for ln in lines) + '\n')
    ds = S.FileImageNet(os.path.join(root, 'val'), meta, 32, 36, 'STANDARD', seed=3)
    hw = arrays[2].shape[:2]
    boxes = [ds.box(2, hw, e) for e in range(8)]
    assert len(set(boxes)) > 1                                             # not frozen over the epochs
    assert boxes == [ds.box(2, hw, e) for e in range(8)]                   # reproducible (resume-safe)
    assert ds.box(2, hw) == boxes[0]                                       # default epoch 0


def test_missing_checkpoint_file_raises_oserror_not_a_pickle_message(tmp_path):
    from robustart_amd.train import cls_solver as S
    with pytest.raises(OSError):
        S.load_checkpoint_file(str(tmp_path / 'nope.pth.tar'))


def _timm_vit_b16_state_dict(num_classes=1000, seed=0):
    """A state dict with the 152 key names and shapes of timm's jx_vit_base_p16_224-80ecf9dd.pth -- the file the reference's
    ViT configs point saver.pretrain.path at (exprs/nips_benchmark/new_adv_train/vit_base/config.yaml:78-79).  Names and shapes
    are written out here, NOT taken from the repository's module, so the test notices if the module drifts from timm's layout."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) * 0.02      # noqa: E731
    sd = {'cls_token': r(1, 1, 768), 'pos_embed': r(1, 197, 768),
          'patch_embed.proj.weight': r(768, 3, 16, 16), 'patch_embed.proj.bias': r(768)}
    for i in range(12):
        b = 'blocks.%d.' % i
        sd.update({b + 'norm1.weight': r(768), b + 'norm1.bias': r(768),
                   b + 'attn.qkv.weight': r(2304, 768), b + 'attn.qkv.bias': r(2304),
                   b + 'attn.proj.weight': r(768, 768), b + 'attn.proj.bias': r(768),
                   b + 'norm2.weight': r(768), b + 'norm2.bias': r(768),
                   b + 'mlp.fc1.weight': r(3072, 768), b + 'mlp.fc1.bias': r(3072),
                   b + 'mlp.fc2.weight': r(768, 3072), b + 'mlp.fc2.bias': r(768)})
    sd.update({'norm.weight': r(768), 'norm.bias': r(768), 'head.weight': r(num_classes, 768), 'head.bias': r(num_classes)})
    assert len(sd) == 152
    return sd


def test_vit_loads_a_timm_layout_checkpoint_strictly_and_honours_ignore_model(tmp_path):
    """`get_model({'type': 'vit_base'})` must take timm's published checkpoint with strict=True, and
    `saver.pretrain.ignore.model` (vit_base/config.yaml:80-87, config_vit_base.yaml:106-118) pops the listed keys first."""
    sd = _timm_vit_b16_state_dict()
    bare = str(tmp_path / 'jx_vit_base_p16_224.pth')
    torch.save(sd, bare)                                              # timm ships a BARE state dict
    cfg = {'model': {'type': 'vit_base', 'kwargs': {'num_classes': 1000}}, 'saver': {'pretrain': {'path': bare}}}
    m = S.build_model(cfg)
    got = m.state_dict()
    assert list(got) == list(sd)                                      # same names, same order as timm's module tree
    for k in sd:
        assert torch.equal(got[k], sd[k]), k
    x = torch.rand(1, 3, 224, 224)
    with torch.no_grad():
        assert torch.isfinite(m(x)).all()
    # the same file under DistributedDataParallel's prefix inside the solver's checkpoint shape
    wrapped = str(tmp_path / 'ckpt.pth.tar')
    torch.save({'model': {'module.' + k: v for k, v in sd.items()}, 'last_iter': 7}, wrapped)
    cfg['saver']['pretrain']['path'] = wrapped
    m2 = S.build_model(cfg)
    assert torch.equal(m2.blocks[3].mlp.fc1.weight, sd['blocks.3.mlp.fc1.weight'])
    # a checkpoint written by rounds 1-4 of this repository (patch_embed.weight, blocks.N.fc1.*) still loads
    from robustart_amd.model.vit_torch import timm_to_legacy_keys
    legacy = str(tmp_path / 'legacy.pth')
    old = timm_to_legacy_keys(sd)
    assert 'patch_embed.weight' in old and 'blocks.0.fc1.weight' in old and 'blocks.0.mlp.fc1.weight' not in old
    torch.save({'model': old}, legacy)
    cfg['saver']['pretrain']['path'] = legacy
    m3 = S.build_model(cfg)
    assert torch.equal(m3.patch_embed.proj.weight, sd['patch_embed.proj.weight'])
    # fine-tuning with another label space: head popped from the checkpoint, everything else strict
    cfg10 = {'model': {'type': 'vit_base', 'kwargs': {'num_classes': 10}},
             'saver': {'pretrain': {'path': bare, 'ignore': {'model': ['module.head.weight', 'module.head.bias']}}}}
    torch.manual_seed(3)
    m10 = S.build_model(cfg10)
    assert S.load_pretrain.last_ignored == ['head.weight', 'head.bias']
    assert m10.head.weight.shape == (10, 768) and torch.equal(m10.norm.weight, sd['norm.weight'])
    # without the ignore list the 1000-class head does not fit a 10-class model: strict loading refuses
    cfg10['saver']['pretrain'].pop('ignore')
    with pytest.raises(RuntimeError):
        S.build_model(cfg10)
    # a key that is not in the file is a configuration error, not a silent no-op
    cfg10['saver']['pretrain']['ignore'] = {'model': ['module.fc.weight']}
    with pytest.raises(KeyError, match='fc.weight'):
        S.build_model(cfg10)
    # a missing key is still refused when strict
    short = dict(sd)
    del short['blocks.11.mlp.fc2.bias']
    p = str(tmp_path / 'short.pth')
    torch.save(short, p)
    with pytest.raises(RuntimeError, match='missing keys'):
        S.load_pretrain(m, p)
    # save side: timm's names by default, the old names behind the flag
    out = S.save_checkpoint(str(tmp_path / 'o' / 'a.pth'), m)
    assert 'patch_embed.proj.weight' in torch.load(out, weights_only=True)['model']
    out = S.save_checkpoint(str(tmp_path / 'o' / 'b.pth'), m, legacy_vit_keys=True)
    assert 'patch_embed.weight' in torch.load(out, weights_only=True)['model']


def test_resume_of_a_legacy_named_vit_checkpoint_continues_the_whole_ema(tmp_path):
    """a ViT checkpoint written by rounds 1-4 (patch_embed.weight, blocks.N.fc{1,2}.*) must resume with its WHOLE EMA: the renamed
    tensors used to keep the freshly loaded live weights as their average while the moments and last_iter continued, silently.  Here the
    checkpoint of iteration 2 is rewritten with the old names ('model' and 'ema'), resumed, and must end bit-identical to the uninterrupted
    run; an 'ema' dict that lacks a parameter is refused with a message that names the way out."""
    import robustart_amd.model as M
    from robustart_amd.model.vit_torch import VisionTransformer, timm_to_legacy_keys

    def tiny_vit(**kw):
        torch.manual_seed(0)
        return VisionTransformer(img_size=32, patch_size=16, num_classes=1000, embed_dim=32, depth=2, num_heads=2)
    M._REGISTRY['tiny_vit_cfg_test'] = tiny_vit
    dev = torch.device('cpu')
    over = {'model.type': 'tiny_vit_cfg_test', 'data.batch_size': 4, 'data.input_size': 32, 'lr_scheduler.kwargs.max_epoch': 2,
            'lr_scheduler.kwargs.warmup_epoch': 1, 'lr_scheduler.kwargs.base_lr': 0.01, 'lr_scheduler.kwargs.warmup_lr': 0.02,
            'ema.kwargs.decay': 0.5, 'saver.val_freq': 2}

    def config(save_dir, **saver):
        cfg = reference_shaped_config(**over)
        cfg['data']['fake_size'] = 8                                   # 2 iterations per epoch -> 4 iterations
        cfg['bf16'] = False
        cfg['saver'].update(save_dir=save_dir, **saver)
        return cfg
    full_dir, part_dir = str(tmp_path / 'full'), str(tmp_path / 'part')
    S.train(config(full_dir), _Args(), 0, 1, dev)
    ck_full = torch.load(os.path.join(full_dir, 'ckpt.pth.tar'), weights_only=True)
    assert ck_full['last_iter'] == 4 and 'blocks.0.mlp.fc1.weight' in ck_full['ema']
    S.train(config(part_dir, save_many=True), _Args(), 0, 1, dev)
    mid = os.path.join(part_dir, 'ckpt_2.pth.tar')
    ck = torch.load(mid, weights_only=True)
    live, avg = ck['model']['blocks.0.mlp.fc1.weight'], ck['ema']['blocks.0.mlp.fc1.weight']
    assert not torch.equal(live, avg)                                  # otherwise a dropped average would go unnoticed
    ck['model'], ck['ema'] = timm_to_legacy_keys(ck['model']), timm_to_legacy_keys(ck['ema'])
    assert 'blocks.0.fc1.weight' in ck['ema'] and 'blocks.0.mlp.fc1.weight' not in ck['ema']
    legacy = str(tmp_path / 'legacy_2.pth.tar')
    torch.save(ck, legacy)
    a = _Args()
    a.recover = legacy
    S.train(config(str(tmp_path / 'resumed')), a, 0, 1, dev)
    assert S.train.start_iter == 2
    ck_res = torch.load(os.path.join(str(tmp_path / 'resumed'), 'ckpt.pth.tar'), weights_only=True)
    for k in ck_full['ema']:
        assert torch.equal(ck_full['ema'][k], ck_res['ema'][k]), k
    for k in ck_full['model']:
        assert torch.equal(ck_full['model'][k], ck_res['model'][k]), k
    del ck['ema']['blocks.1.fc2.bias']
    short = str(tmp_path / 'short_2.pth.tar')
    torch.save(ck, short)
    a.recover = short
    with pytest.raises(RuntimeError, match="lacks 1 of the model's"):
        S.train(config(str(tmp_path / 'refused')), a, 0, 1, dev)
