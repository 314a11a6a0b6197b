"""Round 6 (copy of scratch/r5/ab_engine_x3.py): A/B of two builds of the library on the reference-precision ResNet-50 engine at B = 256: forward and forward + backward-to-input,
alternating the builds in child processes so that box-to-box and warm-up effects cancel.
    gpurun -- python scratch/r5/ab_engine_x3.py scratch/r5/ko/lib_OLD.so product [rounds]
    gpurun -- python scratch/r5/ab_engine_x3.py product product:RART_PAIR_WFRAG=1 [rounds]        (a switch instead of a second build)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    from robustart_amd import _lib
    if sys.argv[2] != 'product':
        _lib.LIB_PATH = os.path.abspath(sys.argv[2])
    import torch
    from robustart_amd.model import get_model
    from robustart_amd.model.engine import ResNet50Engine
    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    torch.manual_seed(0)
    eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda', 'bf16')
    x = torch.rand(256, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (256,), device='cuda')
    def t(fn, n=8):
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    fb = t(lambda: eng.forward_backward(x, MEAN, STD, y, 0))
    f = t(lambda: eng.logits(x, MEAN, STD))
    print(json.dumps({'fwd_ms': round(f, 3), 'fwd_bwd_ms': round(fb, 3)}))
    sys.exit(0)
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ)
        for kv in l.split(':')[1:]:                       # 'product:RART_PAIR_WFRAG=1' = the product library with a switch set
            env[kv.split('=', 1)[0]] = kv.split('=', 1)[1]
        o = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', l.split(':')[0]], capture_output=True, text=True, timeout=600, env=env)
        res[l].append(json.loads([ln for ln in o.stdout.splitlines() if ln.startswith('{')][-1]))
        print(l, res[l][-1], flush=True)
for l in libs:
    print(l, 'median fwd+bwd %.3f ms, fwd %.3f ms' % (sorted(q['fwd_bwd_ms'] for q in res[l])[len(res[l]) // 2], sorted(q['fwd_ms'] for q in res[l])[len(res[l]) // 2]))
