"""cls_solver-shaped train / eval entry for the hot path.

Reference: RobustART/train/__init__.py:1 re-exports `prototype.prototype.solver.cls_solver`, whose source is
absent; behaviour is taken from the YAML schema (exprs/nips_benchmark/pgd_adv_train/resnet50/config.yaml:1-61),
the launchers (`python -m ...solver.X --config cfg [--evaluate] [--attack NAME --eps E]`,
exprs/nips_benchmark/batch_eval_adv/eval.sh:43) and the only readable instance of the loop,
cifar10/code/train.py:96-127.

    python -m robustart_amd.train.cls_solver --config cfg.yaml --evaluate [--attack pgd_linf --eps 2/255]
    python -m torch.distributed.run --nproc-per-node 8 -m robustart_amd.train.cls_solver --config cfg.yaml

One process per GPU.  Evaluation: samples shard over ranks (distributed, non-repeating sampler), noise comes from
AddNoise (HIP kernels), the forward from the HIP engine, and the ONLY collective is one all-reduce of the
(top-1, top-5, count) counters.  Adversarial training (PGD-k inner loop, BN in eval mode during the attack,
cifar10/code/train.py:105-111): the attack runs on the HIP engine with the current weights re-folded; the
train-mode forward/backward of ResNet-50 runs on the HIP train engine (model/train_engine.py: igemm contractions,
batch-statistics BatchNorm kernels, split-K weight gradients; other architectures use torch autograd as scaffold); the
loss + its gradient (label-smoothed CE), the optimizer (SGD-Nesterov / AdamW), the EMA and the gradient reset are HIP
kernels over flat parameter arenas (train/arena.py), and the gradient exchange is a few large all-reduces on arena
slices over RCCL/xGMI, overlapped with backward unless `dist.sync: True`.
"""
import argparse
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def parse_eps(s):
    """'8/255' or '0.031' -> float (the launchers pass fractions, batch_eval_adv/eval.sh:9-10)."""
    if isinstance(s, (int, float)):
        return float(s)
    if '/' in s:
        a, b = s.split('/')
        return float(a) / float(b)
    return float(s)


def load_config(path):
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)


class FakeImageNet(torch.utils.data.Dataset):
    """`data.read_from: fake` (pgd_adv_train/resnet50/config.yaml:37): synthetic uint8 NHWC images whose content
    is a pure function of the global index, so every world size sees the same dataset."""

    def __init__(self, n, size=224, classes=1000):
        self.n, self.size, self.classes = n, size, classes

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        img, lab = self.batch([i], 'cpu')
        return img[0], int(lab[0]), i

    def batch(self, indices, device):
        """(uint8 [b, size, size, 3], int64 [b]) generated ON `device` from the global indices alone: an integer
        hash of (index, element), identical on CPU and GPU and for every world size (no host loop, no H2D copy)."""
        idx = torch.as_tensor(list(indices), dtype=torch.int64, device=device).view(-1, 1)
        e = torch.arange(self.size * self.size * 3, dtype=torch.int64, device=device).view(1, -1)
        h = (idx * 1000003 + e) * 2654435761 % 4294967296
        h = (h ^ (h >> 15)) * 2246822519 % 4294967296
        h = h ^ (h >> 13)
        img = (h & 255).to(torch.uint8).view(-1, self.size, self.size, 3)
        lab = ((idx.view(-1) * 2654435761 % 4294967296) >> 7) % self.classes
        return img, lab


class StructuredFakeImageNet(FakeImageNet):
    """`data.read_from: structured`: a LEARNABLE synthetic set (still a pure function of the global index): class c of
    `classes` (default 16) owns a fixed +-1 pattern on an 8 x 8 x 3 grid; an image is 128 + contrast * pattern(upsampled)
    + the hash noise of FakeImageNet scaled to +-noise.  A network fitted to it has real decision margins, which is what
    the bf16-engine-vs-fp32 attack-outcome test needs (random-pixel images with hash labels can only be memorised)."""

    def __init__(self, n, size=224, classes=16, contrast=48.0, noise=40.0):
        super().__init__(n, size, classes)
        self.contrast, self.noise = float(contrast), float(noise)

    def batch(self, indices, device):
        raw, _ = super().batch(indices, device)
        idx = torch.as_tensor(list(indices), dtype=torch.int64, device=device)
        lab = ((idx * 2654435761 % 4294967296) >> 7) % self.classes
        cell = torch.arange(self.size, device=device) * 8 // self.size                      # pixel -> grid cell
        cy, cx, ch = cell.view(1, -1, 1, 1), cell.view(1, 1, -1, 1), torch.arange(3, device=device).view(1, 1, 1, 3)
        h = (lab.view(-1, 1, 1, 1) * 7919 + cy * 131 + cx * 17 + ch) * 2654435761 % 4294967296
        sign = (((h ^ (h >> 13)) >> 5) & 1).float() * 2.0 - 1.0
        img = 128.0 + self.contrast * sign + (raw.float() - 127.5) * (self.noise / 127.5)
        return img.clamp_(0, 255).to(torch.uint8), lab


def read_meta_file(path):
    """`data.{train,test}.meta_file`: one sample per line, either "relative/path.JPEG label" (the ImageNet lists of the
    reference configs, pgd_adv_train/resnet50/config.yaml:45,55) or a JSON object {"filename": ..., "label": ...} (the
    imagenet-c / imagenet-s lists, exp/imagenet_c_loop_mini/config_vit_base.yaml:82).  -> [(relpath, label)]"""
    out = []
    with open(path) as f:
        for ln in f:
            ln = ln.strip()
            if not ln:
                continue
            if ln.startswith('{'):
                o = json.loads(ln)
                out.append((o['filename'], int(o['label'])))
            else:
                name, lab = ln.rsplit(None, 1)
                out.append((name, int(lab)))
    return out


class FileImageNet(torch.utils.data.Dataset):
    """`data.read_from: fs` with `data.<split>.{root_dir, meta_file, image_reader.type: pil, transforms.type}`
    (exp/imagenet_c_loop_mini/config_vit_base.yaml:80-104): files are decoded on the host by PIL (the reference's 'pil'
    reader) and the transform's arithmetic runs on the GPU:
      ONECROP  = Resize([test_resize, test_resize]) (PIL bilinear, what torchvision applies to a PIL image) + CenterCrop(
                 input_size) -> rart_pil_resize_u8 with the fused centre crop, bit-exact with Pillow;
      STANDARD = RandomResizedCrop(input_size) + RandomHorizontalFlip: the crop box / flip are drawn on the host from
                 (seed, global index) (a pure function of the sample, like every draw of this path), the resize runs in
                 the same kernel.  ColorJitter of the reference's commented variant is not applied.
    Output is the uint8 NHWC batch the corruption kernels and the engines' u8 entry consume; normalisation happens in
    the engine's input kernel."""

    def __init__(self, root_dir, meta_file, size=224, test_resize=256, transform='ONECROP', reader='pil', limit=None, seed=0):
        if reader != 'pil':
            raise NotImplementedError("image_reader.type %r: only 'pil' is available in this build" % (reader,))
        if transform not in ('ONECROP', 'STANDARD'):
            raise NotImplementedError('transforms.type %r (ONECROP / STANDARD)' % (transform,))
        self.root, self.items = root_dir, read_meta_file(meta_file)
        if limit:
            self.items = self.items[:int(limit)]
        self.size, self.test_resize, self.transform, self.seed = int(size), int(test_resize), transform, int(seed)
        self.n = len(self.items)

    def __len__(self):
        return self.n

    def decode(self, i):
        """host side: (HxWx3 uint8 ndarray, label) of sample i"""
        from ..noise.imagenet_s import decode
        name, lab = self.items[i]
        return decode(os.path.join(self.root, name), 'pil'), lab

    def __getitem__(self, i):
        img, lab = self.batch([i], 'cuda')
        return img[0], int(lab[0]), i

    def box(self, i, hw, epoch=0):
        """STANDARD: (y, x, h, w, flip) of sample i in `epoch`, drawn from (seed, epoch, i) with the reference's
        get_params loop: torchvision's RandomResizedCrop / RandomHorizontalFlip redraw every time a sample is visited,
        so the epoch is part of the key -- still a pure function of its arguments (resume-safe, world-size invariant)."""
        import random as _random
        from ..noise.imagenet_s import _train_params
        r = _random.Random((self.seed * 1000003 + i) * 1000033 + int(epoch))
        y, x, h, w = _train_params(hw, r)
        return y, x, h, w, r.random() < 0.5

    def batch(self, indices, device, epoch=0):
        from ..noise.imagenet_s import pil_resize
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise RuntimeError('FileImageNet: the resize / crop of the transform runs on the GPU (no CPU fallback)')
        out = torch.empty(len(indices), self.size, self.size, 3, dtype=torch.uint8, device=dev)
        labs = []
        for k, i in enumerate(indices):
            arr, lab = self.decode(i)
            labs.append(lab)
            if self.transform == 'ONECROP':
                r, t = self.test_resize, self.size
                o = int(round((r - t) / 2.0))                         # torchvision CenterCrop
                src = torch.from_numpy(arr).to(dev, non_blocking=True)[None]
                out[k] = pil_resize(src, (r, r), 1, crop=(o, o, t, t))[0]
            else:
                y, x, h, w, flip = self.box(i, arr.shape[:2], epoch)
                src = torch.from_numpy(arr[y:y + h, x:x + w].copy()).to(dev, non_blocking=True)[None]
                img = pil_resize(src, (self.size, self.size), 1)[0]
                out[k] = img.flip(1) if flip else img
        return out, torch.tensor(labs, dtype=torch.int64, device=dev)


def make_dataset(dcfg, n, size, split='test'):
    """`data.read_from`: 'fake' (the reference configs' own setting) / 'structured' -> synthetic sets of n samples;
    'fs' (or 'file' / 'folder') -> FileImageNet over data.<split>.{root_dir, meta_file} (n = limit, 0 / None = all)."""
    rf = dcfg.get('read_from', 'fake')
    if rf == 'structured':
        return StructuredFakeImageNet(n, size, int(dcfg.get('structured_classes', 16)),
                                      float(dcfg.get('structured_contrast', 48.0)), float(dcfg.get('structured_noise', 40.0)))
    if rf in ('fs', 'file', 'folder'):
        sec = dcfg.get(split) or {}
        if not sec.get('root_dir') or not sec.get('meta_file'):
            raise ValueError('data.read_from: %s needs data.%s.root_dir and data.%s.meta_file' % (rf, split, split))
        return FileImageNet(sec['root_dir'], sec['meta_file'], size, int(dcfg.get('test_resize', 256)),
                            (sec.get('transforms') or {}).get('type', 'ONECROP' if split == 'test' else 'STANDARD'),
                            (sec.get('image_reader') or {}).get('type', 'pil'), dcfg.get('limit_samples'),
                            int(dcfg.get('seed', 0)))
    if rf != 'fake':
        raise NotImplementedError("data.read_from %r ('fake', 'structured', 'fs')" % (rf,))
    return FakeImageNet(n, size)


def resolve_schedule(cfg, n_train, batch_size, world, default_max_iter=20):
    """-> (max_iter, warmup_steps) from the reference's key set (pgd_adv_train/resnet50/config.yaml:18-29): the
    iteration keys `lr_scheduler.kwargs.{max_iter, warmup_steps}` when present, else the epoch keys `max_epoch` /
    `warmup_epoch` times the iterations of one epoch of the distributed_iteration sampler,
    ceil(len(train set) / (batch_size * world_size)).  A top-level `max_iter` (this build's own shortcut, used by the
    tests) wins over both."""
    lk = (cfg.get('lr_scheduler') or {}).get('kwargs') or {}
    per_epoch = max(1, -(-int(n_train) // (int(batch_size) * int(world))))
    if cfg.get('max_iter') is not None:
        max_iter = int(cfg['max_iter'])
    elif lk.get('max_iter') is not None:
        max_iter = int(lk['max_iter'])
    elif lk.get('max_epoch') is not None:
        max_iter = int(round(float(lk['max_epoch']) * per_epoch))
    else:
        max_iter = int(default_max_iter)
    if lk.get('warmup_steps') is not None:
        warm = int(lk['warmup_steps'])
    elif lk.get('warmup_epoch') is not None:
        warm = int(round(float(lk['warmup_epoch']) * per_epoch))
    else:
        warm = max(max_iter // 20, 1)
    return max_iter, min(warm, max_iter)


def load_checkpoint_file(path):
    """torch.load restricted to tensors / containers / numbers (weights_only=True): a checkpoint is data, and a pickle from
    an untrusted source would execute code.  Checkpoints of this solver and plain state dicts load this way; a legacy
    reference checkpoint that pickles other objects needs the explicit opt-in RART_ALLOW_PICKLE_CHECKPOINT=1."""
    import pickle
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:      # a missing / unreadable file (OSError) propagates as itself
        if os.environ.get('RART_ALLOW_PICKLE_CHECKPOINT') == '1':
            return torch.load(path, map_location='cpu', weights_only=False)
        raise RuntimeError('%s does not load with weights_only=True (%s); set RART_ALLOW_PICKLE_CHECKPOINT=1 to unpickle a '
                           'TRUSTED legacy checkpoint' % (path, e)) from e


def load_pretrain(model, path, prefer='model', strict=True, ignore_model=()):
    """`saver.pretrain.path` / --recover: load a checkpoint written by this solver or by the reference's
    (torch.save of {'model': state_dict, 'ema': {...}, ...} or a bare state_dict such as timm's
    jx_vit_base_p16_224-80ecf9dd.pth, exprs/nips_benchmark/new_adv_train/vit_base/config.yaml:78-79; DistributedDataParallel's
    'module.' prefix stripped).  prefer = 'ema' picks the EMA weights when the file has them.
    `ignore_model` = `saver.pretrain.ignore.model` (same file :80-87, exp/imagenet_c_loop_mini/config_vit_base.yaml:106-118): keys
    (with or without 'module.') popped from the checkpoint before loading -- fine-tuning with another number of classes; the
    model keeps its own initialisation for them and every OTHER key still has to match when strict.
    A ViT checkpoint in the key layout rounds 1-4 of this repository wrote (`patch_embed.weight`, `blocks.N.fc1.*`) is renamed to
    timm's layout, which the module tree now carries."""
    ck = load_checkpoint_file(path)
    sd = ck
    if isinstance(ck, dict):
        for key in ((prefer, 'model', 'state_dict', 'ema') if prefer else ('model', 'state_dict')):
            if key in ck and isinstance(ck[key], dict):
                sd = ck[key]
                break
    if isinstance(sd, dict) and 'ema_state_dict' in sd:          # reference EMA wrapper
        sd = sd['ema_state_dict']
    strip = lambda k: k[7:] if k.startswith('module.') else k    # noqa: E731
    sd = {strip(k): v for k, v in sd.items()}
    from ..model.vit_torch import VisionTransformer, legacy_vit_keys
    if isinstance(model, VisionTransformer):
        sd = legacy_vit_keys(sd)
    dropped = []
    for k in ignore_model or ():
        k = strip(str(k))
        if isinstance(model, VisionTransformer):
            k = next(iter(legacy_vit_keys({k: None})))
        if k in sd:
            del sd[k]
            dropped.append(k)
        else:
            raise KeyError('saver.pretrain.ignore.model: %r is not a key of %s' % (k, path))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [k for k in missing if k not in dropped]
    if strict and (missing or unexpected):
        raise RuntimeError('load_pretrain(%s): missing keys %s, unexpected keys %s' % (path, list(missing), list(unexpected)))
    if missing or unexpected:
        import warnings
        warnings.warn('load_pretrain(%s): missing keys %s, unexpected keys %s' % (path, list(missing), list(unexpected)),
                      RuntimeWarning)
    load_pretrain.last_ignored = dropped
    return ck if isinstance(ck, dict) else {}


def save_checkpoint(path, model, ema_state=None, optimizer_state=None, step=0, extra=None, legacy_vit_keys=False):
    """The reference solver's checkpoint shape: {'model', 'ema', 'optimizer', 'last_iter'} (state dicts on the CPU).
    legacy_vit_keys (`saver.legacy_vit_keys: True`): write a ViT's parameters under the names rounds 1-4 of this repository used
    instead of timm's (for tools that read those files)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ren = (lambda d: d)
    if legacy_vit_keys:
        from ..model.vit_torch import timm_to_legacy_keys as ren
    ck = {'model': ren({k: v.detach().cpu() for k, v in model.state_dict().items()}), 'last_iter': int(step)}
    if ema_state is not None:
        ck['ema'] = ren({k: v.detach().cpu() for k, v in ema_state.items()})
    if optimizer_state is not None:
        ck['optimizer'] = optimizer_state
    if extra:
        ck.update(extra)
    torch.save(ck, path)
    return path


def shard_indices(n, rank, world):
    """`sampler.type: distributed` (non-repeating): contiguous ranges, the last ranks may get one fewer."""
    per = (n + world - 1) // world
    return list(range(min(rank * per, n), min((rank + 1) * per, n)))


class EpochSampler:
    """`sampler.type: distributed_iteration` of the training configs (pgd_adv_train/resnet50/config.yaml:39-41): every
    epoch visits a fresh random permutation of the train set, dealt to the ranks by stride.  ImageNet's train list is
    sorted by class, so file-order batches from contiguous per-rank ranges would hold one class each (degenerate
    BatchNorm statistics and SGD).  The permutation is a pure function of (seed, epoch) -- numpy's RandomState stream --
    so a resumed run and every world size of the same global batch see the same samples; the tail of an epoch wraps to
    the permutation's head (DistributedSampler's padding), so every iteration has a full batch.
    shuffle=False keeps file order (still strided over the ranks)."""

    def __init__(self, n, batch_size, rank, world, seed=0, shuffle=True):
        self.n, self.bs, self.rank, self.world = int(n), int(batch_size), int(rank), int(world)
        self.seed, self.shuffle = int(seed), bool(shuffle)
        self.per_epoch = max(1, -(-self.n // (self.bs * self.world)))
        self._epoch, self._mine = None, None

    def epoch_of(self, it):
        return int(it) // self.per_epoch

    def _order(self, epoch):
        import numpy as np
        if self._epoch != epoch:
            if self.shuffle:
                perm = np.random.RandomState((self.seed * 1000003 + epoch) % (2 ** 32)).permutation(self.n)
            else:
                perm = np.arange(self.n)
            total = self.per_epoch * self.bs * self.world
            perm = np.resize(perm, total)                      # repeats from the head when total > n
            self._epoch, self._mine = epoch, perm[self.rank::self.world]
        return self._mine

    def batch(self, it):
        """-> (global indices of this rank's batch at iteration `it`, epoch)"""
        epoch = self.epoch_of(it)
        k = int(it) % self.per_epoch
        return [int(v) for v in self._order(epoch)[k * self.bs:(k + 1) * self.bs]], epoch

    def batch_rows(self, it, device):
        """The same indices as an int64 tensor on `device`, a VIEW of this rank's share of the epoch's permutation, which is uploaded ONCE per
        epoch (the attack kernels read it as every row's global sample index: round 5 built a tensor from a Python list per step).  The
        range [0, 2^32) the counter generator needs is checked here, on the host, once per epoch -- the attack entry does not look at a
        device tensor's values (that was a blocking device-to-host copy per iteration)."""
        import torch
        epoch = self.epoch_of(it)
        mine = self._order(epoch)
        if getattr(self, '_dev_epoch', None) != (epoch, str(device)):
            if len(mine) and (int(mine.min()) < 0 or int(mine.max()) >= 1 << 32):
                raise ValueError('sample indices must lie in [0, 2^32)')
            self._dev = torch.as_tensor(mine, dtype=torch.int64).to(device)
            self._dev_epoch = (epoch, str(device))
        k = int(it) % self.per_epoch
        return self._dev[k * self.bs:(k + 1) * self.bs]


def cosine_lr(step, total, base_lr, warmup_lr, warmup_steps, min_lr=0.0):
    """lr_scheduler.type CosineEpoch with linear warm-up (config.yaml:21-29)."""
    if step < warmup_steps:
        return base_lr + (warmup_lr - base_lr) * step / max(warmup_steps, 1)
    t = (step - warmup_steps) / max(total - warmup_steps, 1)
    return min_lr + 0.5 * (warmup_lr - min_lr) * (1 + math.cos(math.pi * t))


def topk_correct(logits, labels, ks=(1, 5)):
    _, pred = logits.topk(max(ks), dim=1)
    hit = pred.eq(labels.view(-1, 1))
    return [int(hit[:, :k].any(1).sum()) for k in ks]


def init_dist():
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local)
    # RART_FORCE_DIST=1: join the process group even as a single rank (a one-GPU box then creates the RCCL communicator and runs
    # the same collectives the 8-GPU launch does)
    if (world > 1 or os.environ.get('RART_FORCE_DIST') == '1') and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl' if use_cuda else 'gloo')       # "nccl" == RCCL on ROCm
    return rank, world, torch.device('cuda', local) if use_cuda else torch.device('cpu')


def all_reduce_counters(values, device):
    """The eval path's only collective: sum of a few int64 counters."""
    t = torch.tensor(values, dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t)
    return [int(v) for v in t.tolist()]


def build_model(cfg, args=None):
    """get_model + `saver.pretrain.path` (or --recover): the weights an evaluation / fine-tuning run starts from."""
    from ..model import get_model
    model = get_model(cfg['model'])
    pre = (cfg.get('saver', {}) or {}).get('pretrain', {}) or {}
    path = getattr(args, 'recover', None) or pre.get('path')
    build_model.last_checkpoint = None
    if path:
        ck = load_pretrain(model, path, prefer='ema' if pre.get('use_ema', False) else 'model',
                           ignore_model=((pre.get('ignore') or {}).get('model')) or ())
        # `saver.pretrain.ignore.key: [optimizer, last_iter]` (pgd_adv_train/resnet50/config.yaml:67-70): fine-tuning from a
        # checkpoint instead of resuming it
        drop = set(((pre.get('ignore') or {}).get('key')) or [])
        build_model.last_checkpoint = {k: v for k, v in ck.items() if k not in drop and k != 'model'}
    return model


def evaluate(cfg, args, rank, world, device, model=None):
    """Clean / corrupted / attacked evaluation over the sharded dataset; returns the reduced metrics dict."""
    dcfg = cfg.get('data', {})
    n = int(dcfg.get('fake_size', dcfg.get('limit_samples', 256)))
    bs = int(dcfg.get('batch_size', 64))
    size = int(dcfg.get('input_size', 224))
    ds = make_dataset(dcfg, n, size, 'test')
    n = len(ds)                                              # a file-backed set brings its own length
    idx = shard_indices(n, rank, world)
    model = model or build_model(cfg, args)
    model = model.to(device).eval()
    use_hip = device.type == 'cuda' and args.engine == 'hip'
    noise = None
    if use_hip:
        from ..model.engine import EngineModel
        # `engine_precision: fp32x` (or --precision fp32x): the reference-precision engine mode (logits within 1e-4 of the
        # fp32 network, ~3x the MFMA work); default bf16
        prec = getattr(args, 'precision', None) or cfg.get('engine_precision', 'bf16')
        f_model = EngineModel(model, takes_normalized=False, precision=prec)
        n_model = EngineModel(None, takes_normalized=True, engine=f_model.rart_engine)
    if args.corruption:
        from ..noise import AddNoise
        noise = AddNoise('imagenet-c')
        noise.config.update(corruption_name=args.corruption, severity=args.severity)
    attack = None
    if args.attack and args.attack != 'none':
        from ..noise import AddNoise
        attack = AddNoise(args.attack)
        key = 'f_model' if 'f_model' in attack.config else 'model'
        if not use_hip:
            raise RuntimeError('attacks need the GPU path (no CPU fallback)')
        attack.config[key] = f_model if key == 'f_model' else n_model      # ResNet-50 and ViT-B/16: HIP forward + backward
        attack.config['eps'] = parse_eps(args.eps)
        if 'steps' in attack.config and args.steps:
            attack.config['steps'] = args.steps
    mean = torch.tensor(IMAGENET_MEAN, device=device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=device).view(1, 3, 1, 1)
    # transfer harness: adversarial examples crafted on cfg.model, scored on the target model
    tgt_model = None
    if getattr(args, 'tgt_type', None):
        from ..model import get_model
        tgt = get_model({'type': args.tgt_type}).to(device).eval()
        if use_hip:
            from ..model.engine import EngineModel
            try:
                tgt_model = EngineModel(tgt, takes_normalized=False)
            except Exception:                                   # no HIP engine for this architecture: torch module
                tgt_model = lambda z, _t=tgt: _t((z - mean) / std)   # noqa: E731
        else:
            tgt_model = lambda z, _t=tgt: _t((z - mean) / std)       # noqa: E731
    writer = None
    if getattr(args, 'save_dir', None):
        from ..metrics import ResultWriter, result_dir
        noise_name = args.corruption or (args.attack if args.attack and args.attack != 'none' else 'none')
        eps_name = ('%d' % args.severity) if args.corruption else (
            ('%.3f' % parse_eps(args.eps)) if noise_name != 'none' else '0')
        writer = ResultWriter(result_dir(args.save_dir, getattr(args, 'src_name', None) or cfg['model']['type'],
                                         noise_name, eps_name, tgt_name=getattr(args, 'tgt_name', None)), rank, world)
    c1 = c5 = cnt = 0
    t0 = time.time()
    for s in range(0, len(idx), bs):
        items = idx[s:s + bs]
        imgs, labels = ds.batch(items, device)
        first = items[0]                                       # global index of the batch's first sample
        if noise is not None:
            from ..noise import imagenet_c as C
            C.corrupt_batch_(imgs, C.CORRUPTION_NAMES.index(args.corruption), args.severity, seed=args.seed,
                             sample_offset=first)
        if attack is not None:
            # the attack's random starts are a function of (seed, dataset index): set the process-wide counter to the
            # batch's first global index, which every attack reads when no explicit sample_offset is passed
            from ..noise import rng
            rng.manual_seed(args.seed, first)
            x01 = imgs.permute(0, 3, 1, 2).float().div(255.0).contiguous()
            adv_x = attack.add_noise(x01, labels)
            with torch.no_grad():
                logits = (tgt_model or f_model)(adv_x)
        elif use_hip:
            logits = f_model.rart_engine.logits_from_u8(imgs, IMAGENET_MEAN, IMAGENET_STD)
        else:
            with torch.no_grad():
                logits = model((imgs.permute(0, 3, 1, 2).float() / 255.0 - mean) / std)
        a, b = topk_correct(logits.float(), labels)
        c1, c5, cnt = c1 + a, c5 + b, cnt + len(items)
        if writer is not None:
            writer.write_batch(logits, labels, idx[s:s + bs])
    if writer is not None:
        writer.close(barrier=dist.barrier if dist.is_available() and dist.is_initialized() else None)
    c1, c5, cnt = all_reduce_counters([c1, c5, cnt], device)
    res = {'top1': c1 / max(cnt, 1), 'top5': c5 / max(cnt, 1), 'count': cnt, 'world_size': world,
           'noise': args.corruption or args.attack or 'none', 'seconds': time.time() - t0}
    if rank == 0:
        print(json.dumps(res))
    return res


def train(cfg, args, rank, world, device):
    """(Adversarial) training loop; returns the last loss (float) for tests."""
    dcfg = cfg.get('data', {})
    n = int(dcfg.get('fake_size', 512))
    bs = int(dcfg.get('batch_size', 32))
    size = int(dcfg.get('input_size', 224))
    ds = make_dataset(dcfg, n, size, 'train')
    n = len(ds)
    max_iter, warmup_steps = resolve_schedule(cfg, n, bs, world, getattr(args, 'max_iter', 20))
    model = build_model(cfg, args).to(device)
    resume = getattr(build_model, 'last_checkpoint', None) or {}
    ocfg = cfg.get('optimizer', {'type': 'SGD', 'kwargs': {'nesterov': True, 'momentum': 0.9, 'weight_decay': 1e-4}})
    okw = dict(ocfg.get('kwargs', {}))
    lcfg = cfg.get('lr_scheduler', {}).get('kwargs', {})
    base_lr, warmup_lr = float(lcfg.get('base_lr', 0.1)), float(lcfg.get('warmup_lr', 0.4))
    min_lr = float(lcfg.get('min_lr', 0.0))
    kind = ocfg.get('type', 'SGD')
    # flat arenas: parameters / gradients are views; the exchange is a few large all-reduces on arena slices,
    # overlapped with backward unless the config asks for `dist.sync: True`
    from .arena import HipOptimizer, ParamArena, label_smooth_ce
    no_wd = ocfg.get('no_wd', {}) or {}
    norm_names, fc_bias_names = set(), set()
    for mn, mod in model.named_modules():
        if isinstance(mod, (torch.nn.modules.batchnorm._BatchNorm, torch.nn.LayerNorm, torch.nn.GroupNorm)):
            norm_names.update(mn + '.' + pn for pn, _ in mod.named_parameters(recurse=False))
        if isinstance(mod, torch.nn.Linear) and mod.bias is not None:
            fc_bias_names.add(mn + '.bias')

    def no_decay(name, p):
        return (bool(no_wd.get('norm', False)) and name in norm_names) or \
               (bool(no_wd.get('fc', False)) and name in fc_bias_names)

    arena = ParamArena(model, bucket_bytes=int(cfg.get('dist', {}).get('bucket_mb', 48)) << 20, no_decay=no_decay,
                       overlap=not bool(cfg.get('dist', {}).get('sync', False)))
    ema_on = bool(cfg.get('ema', {}).get('enable', False))
    decay = float(cfg.get('ema', {}).get('kwargs', {}).get('decay', 0.9999))
    use_hip_opt = device.type == 'cuda' and args.engine == 'hip'
    ema = None
    if use_hip_opt:
        # SGD-Nesterov / AdamW + EMA + gradient reset: one HIP launch over the arena (rart_sgd_step_f32 / rart_adamw_step_f32)
        opt = HipOptimizer(arena, kind=kind, lr=base_lr, momentum=float(okw.get('momentum', 0.9)),
                           nesterov=bool(okw.get('nesterov', False)), weight_decay=float(okw.get('weight_decay', 0.0)),
                           betas=tuple(okw.get('betas', (0.9, 0.999))), eps=float(okw.get('eps', 1e-8)),
                           ema_decay=decay if ema_on else None)
    else:
        # CPU scaffold for the multi-process tests (gloo): torch's optimizer on the same arena views
        groups = [{'params': [p for n, p in zip(arena.names, arena.params) if not no_decay(n, p)]},
                  {'params': [p for n, p in zip(arena.names, arena.params) if no_decay(n, p)], 'weight_decay': 0.0}]
        groups = [g for g in groups if g['params']]
        opt = torch.optim.AdamW(groups, lr=base_lr, **okw) if kind == 'AdamW' else torch.optim.SGD(groups, lr=base_lr, **okw)
        if ema_on:
            ema = arena.flat_p.clone()
    ema_buffers = {k: v.detach().clone() for k, v in model.named_buffers() if v.dtype.is_floating_point} if ema_on else {}
    train_engine = None
    if use_hip_opt and getattr(args, 'train_engine', 'hip') == 'hip':
        from ..model.resnet_torch import ResNet
        from ..model.vit_torch import VisionTransformer
        model.train()
        if isinstance(model, ResNet) and size % 32 == 0:
            from ..model.train_engine import ResNet50TrainEngine
            train_engine = ResNet50TrainEngine(model, device, on_grad_ready=arena.grad_ready)
        elif isinstance(model, VisionTransformer):
            from ..model.vit_train_engine import ViTTrainEngine
            train_engine = ViTTrainEngine(model, device, on_grad_ready=arena.grad_ready)
    ls = float(cfg.get('label_smooth', 0.0))
    adv = cfg.get('adv_train', None)                        # {'eps': '4/255', 'steps': 3, 'rel_stepsize': 0.4}
    mean = torch.tensor(IMAGENET_MEAN, device=device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=device).view(1, 3, 1, 1)
    use_amp = device.type == 'cuda' and cfg.get('bf16', True)
    sampler = EpochSampler(n, bs, rank, world, int(dcfg.get('seed', 0)), bool(dcfg.get('shuffle', True)))
    loss_v = float('nan')
    attack_model = None
    scfg = cfg.get('saver', {}) or {}
    save_dir = scfg.get('save_dir') or getattr(args, 'ckpt_dir', None)
    save_freq = int(scfg.get('save_freq', scfg.get('val_freq', 0)) or 0)     # the reference solver saves at val_freq
    # ---- resume (--recover / saver.pretrain.path with 'optimizer' + 'last_iter' in the file and not ignored): the
    # momentum / Adam moments, the EMA (parameters and buffers) and the schedule position continue where they stopped
    start_iter = 0
    if resume.get('last_iter') is not None and resume.get('optimizer') is not None:
        start_iter = min(int(resume['last_iter']), max_iter)
        ost = resume['optimizer']
        # the two optimizer paths keep different state (flat arenas of the HIP step vs torch.optim's per-parameter
        # dicts): resuming across them would either fail deep inside load_state_dict or silently restart the moments
        # while the schedule continues -- refuse with a message that says so
        is_torch_state = isinstance(ost, dict) and ost.get('torch') is not None
        if use_hip_opt:
            if is_torch_state or not isinstance(ost, dict) or 'kind' not in ost:
                raise RuntimeError("checkpoint's optimizer state was written by the torch scaffold (--engine torch); it cannot "
                                   "resume the HIP optimizer. Resume with --engine torch, or drop it with "
                                   "saver.pretrain.ignore.key: ['optimizer', 'last_iter'] to fine-tune from the weights")
            if ost.get('kind') != kind:
                raise RuntimeError('checkpoint optimizer kind %r != configured optimizer %r' % (ost.get('kind'), kind))
            opt.load_state_dict(ost)
        else:
            if not is_torch_state:
                raise RuntimeError("checkpoint's optimizer state was written by the HIP optimizer; it cannot resume the torch "
                                   "scaffold (--engine torch / CPU). Resume on the GPU with --engine hip, or ignore "
                                   "['optimizer', 'last_iter'] to fine-tune from the weights")
            opt.load_state_dict(ost['torch'])
        if ema_on and resume.get('ema') is not None:
            esd = {(k[7:] if k.startswith('module.') else k): v for k, v in resume['ema'].items()}
            from ..model.vit_torch import VisionTransformer, legacy_vit_keys
            if isinstance(model, VisionTransformer):
                # a ViT checkpoint written by rounds 1-4 names patch_embed.{weight,bias} and blocks.N.fc{1,2}.*: load_pretrain renames the
                # MODEL's dict only, so without this the 50 renamed tensors kept the freshly loaded live weights as their average while the
                # moments and last_iter continued
                esd = legacy_vit_keys(esd)
            absent = [nm for nm in arena.names if nm not in esd]
            if absent:
                raise RuntimeError("resume: the checkpoint's 'ema' dict lacks %d of the model's %d parameters (first: %s); the average of "
                                   "those tensors cannot continue -- drop it with saver.pretrain.ignore.key: ['ema'] to restart the average "
                                   "from the loaded weights" % (len(absent), len(arena.names), absent[:3]))
            for nm, p_, o in zip(arena.names, arena.params, arena.offsets):
                tgt = opt.ema if use_hip_opt else ema
                tgt[o:o + p_.numel()].copy_(esd[nm].reshape(-1).to(tgt.device))
            for k in ema_buffers:
                if k in esd:
                    ema_buffers[k].copy_(esd[k].to(ema_buffers[k].device))
        if train_engine is not None:
            train_engine.repack()
    train.start_iter = start_iter

    def ema_state():
        sd = None
        if ema_on:
            if use_hip_opt:
                sd = opt.ema_state_dict(model)
            elif ema is not None:
                sd = {nm: ema[o:o + p_.numel()].view_as(p_).clone() for nm, p_, o in zip(arena.names, arena.params, arena.offsets)}
            if sd is not None:
                sd.update({k: v.clone() for k, v in ema_buffers.items()})
                for k, v in model.named_buffers():              # integer buffers (num_batches_tracked) are not averaged
                    if k not in sd:
                        sd[k] = v.detach().clone()
        return sd

    def save(step):
        if save_dir and rank == 0:
            ost = opt.state_dict() if use_hip_opt else {'torch': opt.state_dict()}
            name = 'ckpt_%d.pth.tar' % step if scfg.get('save_many', False) and step != max_iter else 'ckpt.pth.tar'
            save_checkpoint(os.path.join(save_dir, name), model, ema_state(), ost, step,
                            legacy_vit_keys=bool(scfg.get('legacy_vit_keys', False)))

    for it in range(start_iter, max_iter):
        lr = cosine_lr(it, max_iter, base_lr, warmup_lr, warmup_steps, min_lr)
        sel, epoch = sampler.batch(it)
        items = sel
        imgs, labels = ds.batch(sel, device, epoch) if isinstance(ds, FileImageNet) else ds.batch(sel, device)
        x01 = imgs.permute(0, 3, 1, 2).float().div(255.0)
        if adv and device.type == 'cuda':
            # inner maximisation on the HIP eval engine with the CURRENT weights, BN in inference mode: the engine is
            # built once and re-folded on the GPU from the live parameters / running statistics every iteration
            from ..model.engine import EngineModel
            from ..noise import adv as A
            if attack_model is None:
                model.eval()
                attack_model = EngineModel(model, takes_normalized=False)
            else:
                attack_model.rart_refold(model)
            x01 = A.pgd_linf(x01.contiguous(), labels, attack_model, parse_eps(adv['eps']),
                             float(adv.get('rel_stepsize', 3 / 40)), int(adv.get('steps', 3)), seed=it,
                             # every row draws its random start at ITS dataset index (sel is a strided slice of the epoch's
                             # permutation, not a contiguous range): no two images of an iteration share a field, and the
                             # draws do not depend on the world size
                             sample_offset=sampler.batch_rows(it, device))
        model.train()
        if train_engine is not None:
            # train-mode forward (batch statistics), label-smoothed CE, backward to every parameter: all HIP
            logits = train_engine.forward(x01.contiguous(), False, IMAGENET_MEAN, IMAGENET_STD)
            loss_rows, dlogits = label_smooth_ce(logits, labels, ls, 1.0 / len(items))
            train_engine.backward(dlogits)                  # gradients land in the arena; buckets reduce as they fill
            loss = loss_rows.mean()
        else:
            xin = ((x01 - mean) / std).contiguous(memory_format=torch.channels_last)
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=use_amp):
                out = model(xin)
            if use_hip_opt:
                # label-smoothed CE and its gradient in one HIP kernel; autograd continues from dlogits
                loss_rows, dlogits = label_smooth_ce(out, labels, ls, 1.0 / len(items))
                out.backward(dlogits.to(out.dtype))         # bucket all-reduces start from the grad hooks
                loss = loss_rows.mean()
            else:
                loss = F.cross_entropy(out.float(), labels, label_smoothing=ls)
                loss.backward()
        scale = arena.finish_grad_exchange()                # waits for the buckets; returns 1 / world_size
        if use_hip_opt:
            opt.lr = lr
            opt.step(grad_scale=scale)                      # update + EMA + grad reset, one launch per decay range
            if train_engine is not None:
                train_engine.repack()                       # fp32 master weights -> bf16 igemm tables
        else:
            for gp in opt.param_groups:
                gp['lr'] = lr
            if scale != 1.0:
                arena.flat_g.mul_(scale)
            opt.step()
            arena.flat_g.zero_()
            if ema is not None:
                ema.mul_(decay).add_(arena.flat_p, alpha=1 - decay)
        if ema_buffers:
            with torch.no_grad():
                for k, v in model.named_buffers():
                    if k in ema_buffers:
                        ema_buffers[k].mul_(decay).add_(v.detach(), alpha=1 - decay)
        loss_v = float(loss.detach())
        if it % int(scfg.get('print_freq', 10)) == 0 or it == max_iter - 1:
            xs = arena.exchange_stats()
            if xs is not None:
                # every rank, on stderr: the gradient exchange of this step -- buckets, bytes, how many all-reduces started during backward,
                # and how long the compute stream stalled behind them (a first N-rank run is debugged from these lines)
                print('[cls_solver rank %d] %s' % (rank, json.dumps(dict(iter=it, grad_exchange=xs))), file=sys.stderr, flush=True)
            if rank == 0:
                rec = {'iter': it, 'loss': loss_v, 'lr': lr}
                if xs is not None:
                    rec['grad_exchange'] = {k: xs[k] for k in ('buckets', 'bytes', 'launched_during_backward', 'stream_wait_s', 'host_wait_s') if k in xs}
                print(json.dumps(rec))
        if save_freq and (it + 1) % save_freq == 0 and it + 1 < max_iter:
            save(it + 1)                                     # an interrupted run resumes from here with --recover
    # checkpoint (rank 0): model, EMA (parameters from the optimizer's arena + the EMA'd buffers), optimizer, last iteration
    train.last_ema_state = ema_state()
    save(max_iter)
    return loss_v, model


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', required=True)
    ap.add_argument('--evaluate', action='store_true')
    ap.add_argument('--attack', default=None)
    ap.add_argument('--eps', default='8/255')
    ap.add_argument('--steps', type=int, default=0)
    ap.add_argument('--corruption', default=None)
    ap.add_argument('--severity', type=int, default=3)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--engine', choices=['hip', 'torch'], default='hip')
    ap.add_argument('--precision', choices=['bf16', 'bf16x3', 'fp32x'], default=None,
                    help="evaluation engine arithmetic: bf16 (default) or the reference-precision mode 'fp32x' (= 'bf16x3')")
    ap.add_argument('--max-iter', type=int, default=20)
    ap.add_argument('--train-engine', choices=['hip', 'torch'], default='hip', dest='train_engine',
                    help='train-mode forward/backward: hip = ResNet50TrainEngine, torch = autograd scaffold')
    ap.add_argument('--recover', default=None, help='checkpoint to start from (overrides saver.pretrain.path)')
    ap.add_argument('--ckpt-dir', default=None, dest='ckpt_dir', help='write <dir>/ckpt.pth.tar at the end of training')
    ap.add_argument('--save-dir', default=None, help='root of <model>/<noise>_<eps>/results.txt.all (robustart_amd.metrics)')
    ap.add_argument('--src_name', default=None, help='name of the attacked (source) model in the result path')
    ap.add_argument('--tgt_name', default=None, help='transfer: name of the target model (new_transfer/eval.sh:42-44)')
    ap.add_argument('--tgt-type', default=None, help='transfer: model.type of the target model; adversarial examples '
                    'are crafted on cfg.model and scored on this one')
    args = ap.parse_args(argv)
    cfg = load_config(args.config)
    rank, world, device = init_dist()
    try:
        if args.evaluate:
            return evaluate(cfg, args, rank, world, device)
        return train(cfg, args, rank, world, device)[0]
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
