"""Round 6: the ping-pong schedule of k_gemm_pair (csrc/gemm_pair_pp.hip) against round 4's two-stage loop, in ONE process (the schedule is a
library switch): every rart_gemm_pair_bf16 launch shape of a reference-precision ResNet-50 gradient evaluation at B = 256 replayed under
both, interleaved, outputs compared bit for bit; then the whole forward and forward + backward under both.
    gpurun -- python scratch/r6/time_pair_pp.py [--tiles]   ->  gpurun_out/r06_pair_pp.json
--tiles additionally replays every shape with the 256 x 256 and 256 x 128 tiles forced (tile policy for the new schedule)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from robustart_amd import _lib  # noqa: E402
from robustart_amd.model import get_model  # noqa: E402
from robustart_amd.model.engine import ResNet50Engine  # noqa: E402

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
lib = _lib.load()
torch.manual_seed(0)
B = int(os.environ.get('B', 256))
eng = ResNet50Engine(get_model({'type': 'resnet50_official'}).eval(), 'cuda', 'fp32x')
x = torch.rand(B, 3, 224, 224, device='cuda')
y = torch.randint(0, 1000, (B,), device='cuda')


def t_ms(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def record(fused_tail):
    eng.fused_tail_pair = fused_tail
    calls = {}
    orig = eng._gemm_pair

    def rec(*a):
        src, wgt, dst, batch, grid, src_hw, src_pix, k_per_tap, taps, n_cols = a[:10]
        key = (batch * grid[0] * grid[1], k_per_tap * len(taps), n_cols, len(taps), a[13] is not None)
        calls.setdefault(key, [0, a])[0] += 1
        return orig(*a)
    eng._gemm_pair = rec
    eng.forward_backward(x, MEAN, STD, y, 0)
    torch.cuda.synchronize()
    eng._gemm_pair = orig
    return calls


out = {'shapes': {}, 'engine': {}}
calls = record(True)
print('%d distinct launch shapes in the fused-tail engine' % len(calls), flush=True)
tiles = [(0, 0)] + ([(256, 256), (256, 128)] if '--tiles' in sys.argv else [])
tot = {0: 0.0, 1: 0.0, 2: 0.0}
for key, (cnt, a) in sorted(calls.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1][0]):
    dst = a[2]
    row = {'count': cnt}
    for tile in tiles:
        eng.pair_tile = tile
        lib.rart_gemm_pair_set_schedule(0)
        eng._gemm_pair(*a); torch.cuda.synchronize()
        want = dst.clone()
        lib.rart_gemm_pair_set_schedule(2)
        dst.zero_()
        eng._gemm_pair(*a); torch.cuda.synchronize()
        same = bool(torch.equal(dst.view(torch.int16) if dst.dtype == torch.bfloat16 else dst.view(torch.int32),
                                want.view(torch.int16) if want.dtype == torch.bfloat16 else want.view(torch.int32)))
        ts = {0: [], 1: [], 2: []}
        for _ in range(3):
            for s in (0, 1, 2):
                lib.rart_gemm_pair_set_schedule(s)
                ts[s].append(t_ms(lambda: eng._gemm_pair(*a)) * 1e3)
        name = 'auto' if tile == (0, 0) else '%dx%d' % tile
        row[name] = {'two_stage_us': round(min(ts[0]), 1), 'pingpong_us': round(min(ts[1]), 1), 'persistent_us': round(min(ts[2]), 1),
                     'bit_identical': same}
        if tile == (0, 0):
            for q in (0, 1, 2):
                tot[q] += min(ts[q]) * cnt
    eng.pair_tile = (0, 0)
    out['shapes']['%d_%d_%d_%d_%s' % (key[0], key[1], key[2], key[3], 'res' if key[4] else 'nores')] = row
    print(key, row, flush=True)
print('sum over the launches of a gradient evaluation: two-stage %.1f us, ping-pong %.1f us, + persistent %.1f us' % (tot[0], tot[1], tot[2]), flush=True)
out['sum_us'] = {'two_stage': round(tot[0], 1), 'pingpong': round(tot[1], 1), 'persistent': round(tot[2], 1)}
eng.fused_tail_pair = True
for rnd in range(3):
    for s in (0, 1, 2):
        lib.rart_gemm_pair_set_schedule(s)
        fb = t_ms(lambda: eng.forward_backward(x, MEAN, STD, y, 0), 5)
        f = t_ms(lambda: eng.logits(x, MEAN, STD), 5)
        nm = ('two_stage', 'pingpong', 'persistent')[s]
        out['engine'].setdefault(nm, []).append({'fwd_ms': round(f, 3), 'fwd_bwd_ms': round(fb, 3)})
        print('schedule', s, out['engine'][nm][-1], flush=True)
# the whole engine under both schedules: logits and input gradient must be equal bit for bit
lib.rart_gemm_pair_set_schedule(0)
l0, _, g0, _ = eng.forward_backward(x, MEAN, STD, y, 0)
l0, g0 = l0.clone(), g0.clone()
lib.rart_gemm_pair_set_schedule(2)
l1, _, g1, _ = eng.forward_backward(x, MEAN, STD, y, 0)
out['engine']['bit_identical_logits'] = bool(torch.equal(l0, l1))
out['engine']['bit_identical_gradient'] = bool(torch.equal(g0, g1))
print('engine bit-identical:', out['engine']['bit_identical_logits'], out['engine']['bit_identical_gradient'])
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'r06_pair_pp.json'), 'w'), indent=1)
