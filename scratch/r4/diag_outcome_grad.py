"""Diagnostic: per-image gradient cosine of the engines vs fp32 autograd on the fitted network of tests/test_outcome_gpu.py, with the
round-4 fused pair kernels switched off one at a time."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch, tempfile
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
       'saver': {'save_dir': d, 'print_freq': 100}}
a = _Args()
loss, model = S.train(cfg, a, rank, world, device)
model = model.cuda().eval()
for p_ in model.parameters():
    p_.requires_grad_(False)
ds = S.make_dataset(cfg['data'], 4096, 224)
mean = torch.tensor(MEAN, device='cuda').view(1, 3, 1, 1); std = torch.tensor(STD, device='cuda').view(1, 3, 1, 1)
f32 = lambda z: model((z - mean) / std)
imgs, y = ds.batch(list(range(20000, 20064)), 'cuda')
x = imgs.permute(0, 3, 1, 2).float().div(255.0).contiguous()
xr = x.clone().requires_grad_(True)
lt = f32(xr)
g_t, = torch.autograd.grad(torch.nn.functional.cross_entropy(lt, y, reduction='sum'), xr)
x64 = x.double().clone().requires_grad_(True)
m64 = __import__('copy').deepcopy(model).double()
l64 = m64((x64 - mean.double()) / std.double())
g_64, = torch.autograd.grad(torch.nn.functional.cross_entropy(l64, y, reduction='sum'), x64)
def cosv(g, ref):
    a_, b_ = g.flatten(1).double(), ref.flatten(1).double()
    return ((a_ * b_).sum(1) / (a_.norm(dim=1) * b_.norm(dim=1))).cpu()
p = torch.softmax(lt.detach().double(), 1)
py = p.gather(1, y.view(-1, 1)).squeeze(1)
print('torch fp32 vs fp64 autograd: cos min %.6f' % cosv(g_t, g_64).min().item())
print('1 - p_y (fp64 softmax of fp32 logits): min %.3e  median %.3e' % ((1 - py).min().item(), (1 - py).median().item()))
print('|g_t| per image: min %.3e median %.3e' % (g_t.flatten(1).norm(dim=1).min().item(), g_t.flatten(1).norm(dim=1).median().item()))
for prec in ('bf16', 'fp32x'):
    eng = EngineModel(model, takes_normalized=False, precision=prec).rart_engine
    variants = [('default', {})]
    if prec == 'fp32x':
        variants += [('no fused stem bwd', {'fused_stem_bwd': False}), ('no fused stem fwd', {'fused_stem_fwd': False}),
                     ('no tail', {'fused_tail_pair': False}), ('no next', {'fused_next_pair': False})]
    for name, kw in variants:
        for k, v in kw.items(): setattr(eng, k, v)
        lg, _, g_e, _ = eng.forward_backward(x, MEAN, STD, y, 0)
        c32, c64 = cosv(g_e, g_t), cosv(g_e, g_64)
        bad = (c64 < 0.99).nonzero().flatten().tolist()
        print('%-6s %-18s cos vs fp32 autograd: min %.6f | vs fp64 autograd: min %.6f median %.8f  bad images %s  1-p_y there %s' %
              (prec, name, c32.min().item(), c64.min().item(), c64.median().item(), bad, [float('%.2e' % (1 - py[i]).item()) for i in bad]))
        for k, v in kw.items(): setattr(eng, k, True)
