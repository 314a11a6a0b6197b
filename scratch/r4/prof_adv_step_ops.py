"""Round 4: which torch ops (tiny elementwise launches) one adversarial-training step issues beside the HIP kernels.
gpurun -- python scratch/r4/prof_adv_step_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from robustart_amd.model import get_model
from robustart_amd.model.engine import EngineModel
from robustart_amd.model.train_engine import ResNet50TrainEngine
from robustart_amd.noise import adv as A
from robustart_amd.train.arena import HipOptimizer, ParamArena, label_smooth_ce
B = 64
dev = torch.device('cuda')
x01 = torch.rand(B, 3, 224, 224, device=dev); labels = torch.randint(0, 1000, (B,), device=dev)
model = get_model({'type': 'resnet50_official'}).to(dev)
arena = ParamArena(model, bucket_bytes=48 << 20)
opt = HipOptimizer(arena, 'SGD', lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4, ema_decay=0.9999)
model.eval(); attack = EngineModel(model, takes_normalized=False); model.train()
eng = ResNet50TrainEngine(model, dev, on_grad_ready=arena.grad_ready)
mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
phases = {}
def step(k, prof=None):
    def ph(name, fn):
        if prof is None: return fn()
        with torch.profiler.record_function(name): return fn()
    ph('refold', lambda: attack.rart_engine.refold(model))
    xa = ph('pgd', lambda: A.pgd_linf(x01, labels, attack, 4 / 255, 0.4, 3, seed=k))
    logits = ph('train_fwd', lambda: eng.forward(xa, False, mean, std))
    loss_rows, dl = ph('loss', lambda: label_smooth_ce(logits, labels, 0.1, 1.0 / B))
    ph('train_bwd', lambda: eng.backward(dl))
    ph('opt', lambda: opt.step(grad_scale=arena.finish_grad_exchange()))
    ph('repack', lambda: eng.repack())
for i in range(2): step(i)
torch.cuda.synchronize()
k = 5
def run(name, fn):
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        out = fn()
    torch.cuda.synchronize()
    c = {e.key: e.count for e in prof.key_averages()}
    print('%-10s copy_ %4d  select %4d  add_ %3d  to %3d  fill_ %3d  cat %2d' % (name, c.get('aten::copy_', 0), c.get('aten::select', 0), c.get('aten::add_', 0),
                                                                         c.get('aten::to', 0), c.get('aten::fill_', 0), c.get('aten::cat', 0)))
    return out
run('refold', lambda: attack.rart_engine.refold(model))
xa = run('pgd', lambda: A.pgd_linf(x01, labels, attack, 4 / 255, 0.4, 3, seed=k))
logits = run('train_fwd', lambda: eng.forward(xa, False, mean, std))
loss_rows, dl = run('loss', lambda: label_smooth_ce(logits, labels, 0.1, 1.0 / B))
run('train_bwd', lambda: eng.backward(dl))
run('opt', lambda: opt.step(grad_scale=arena.finish_grad_exchange()))
run('repack', lambda: eng.repack())
