"""Diagnostic: logit error of the reference-precision engine on the fitted network of tests/test_outcome_gpu.py against fp64 and against the
fp32 torch module (which of the two is off when |x3 - fp32| approaches 1e-4), and where it arises (per-stage error against an fp64 forward)."""
import os, sys, copy, tempfile
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from test_outcome_gpu import _Args, MEAN, STD
from robustart_amd.train import cls_solver as S
from robustart_amd.model.engine import EngineModel
rank, world, device = S.init_dist()
torch.manual_seed(20260927)
d = tempfile.mkdtemp()
cfg = {'model': {'type': 'resnet50_official', 'kwargs': {'num_classes': 1000}},
       'data': {'read_from': 'structured', 'fake_size': 4096, 'batch_size': 64, 'input_size': 224},
       'label_smooth': 0.1, 'max_iter': 400, 'ema': {'enable': True, 'kwargs': {'decay': 0.9}},
       'lr_scheduler': {'kwargs': {'base_lr': 0.02, 'warmup_lr': 0.08, 'warmup_steps': 10}},
       'saver': {'save_dir': d, 'print_freq': 1000}}
loss, model = S.train(cfg, _Args(), rank, world, device)
model = model.cuda().eval()
for p_ in model.parameters(): p_.requires_grad_(False)
ds = S.make_dataset(cfg['data'], 4096, 224)
mean = torch.tensor(MEAN, device='cuda').view(1, 3, 1, 1); std = torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
m64 = copy.deepcopy(model).double()
eng = EngineModel(model, takes_normalized=False, precision='fp32x')
e3, e32, e332 = [], [], []
for s in range(0, 256, 64):
    imgs, y = ds.batch(list(range(8192 + s, 8192 + s + 64)), 'cuda')
    x = imgs.permute(0, 3, 1, 2).float().div(255.0).contiguous()
    with torch.no_grad():
        l3 = eng(x).double(); l32 = model((x - mean) / std).double(); l64 = m64((x.double() - mean.double()) / std.double())
    sc = l64.abs().max(1)[0]
    e3.append(((l3 - l64).abs().max(1)[0] / sc).cpu()); e32.append(((l32 - l64).abs().max(1)[0] / sc).cpu()); e332.append(((l3 - l32).abs().max(1)[0] / sc).cpu())
e3, e32, e332 = torch.cat(e3), torch.cat(e32), torch.cat(e332)
print('final loss %.4f; logit scale median %.2f' % (loss, float(sc.median())))
print('|x3 - fp64|   max %.2e median %.2e' % (e3.max(), e3.median()))
print('|fp32 - fp64| max %.2e median %.2e' % (e32.max(), e32.median()))
print('|x3 - fp32|   max %.2e median %.2e' % (e332.max(), e332.median()))
# BatchNorm conditioning: folded scale gamma / sqrt(var + eps) per layer
import math
worst = []
for n, mod in model.named_modules():
    if isinstance(mod, torch.nn.BatchNorm2d):
        sc_ = (mod.weight / (mod.running_var + mod.eps).sqrt()).abs()
        worst.append((float(sc_.max()), float(mod.running_var.min()), n))
worst.sort(reverse=True)
print('largest folded BatchNorm scales (scale, min running_var, layer):', [(round(a, 1), float('%.2e' % b), n) for a, b, n in worst[:5]])
