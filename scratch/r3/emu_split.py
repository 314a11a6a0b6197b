"""CPU emulation: how accurate is a split-bf16 (x3 / x6) ResNet-50 forward vs fp64, next to torch fp32?
Products computed in fp64 on exactly-representable bf16 pieces => isolates the representational error."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from robustart_amd.model.resnet_torch import resnet50, randomize_bn_stats
torch.manual_seed(0)
m = randomize_bn_stats(resnet50().eval())
B = 4
x = torch.rand(B, 3, 224, 224)
mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1); std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
xn = (x - mean) / std

def split(t, n):
    parts, r = [], t.double()
    for _ in range(n):
        p = r.float().bfloat16().double(); parts.append(p); r = r - p
    return parts

def fold(conv, bn):
    w = conv.weight.detach().double()
    inv = (bn.running_var.double() + bn.eps).rsqrt() * bn.weight.detach().double()
    return w * inv.view(-1, 1, 1, 1), bn.bias.detach().double() - bn.running_mean.double() * inv

def conv_mode(x, w, b, stride, pad, mode):
    if mode == 'f64':
        return F.conv2d(x, w, b, stride, pad)
    if mode == 'f32':
        return F.conv2d(x.float(), w.float(), b.float(), stride, pad).double()
    n = {'x1': 1, 'x3': 2, 'x6': 3}[mode]
    xs, ws = split(x, n), split(w.float(), n)      # weights are fp32 masters; activations: fp32 acc
    out = None
    for i in range(n):
        for j in range(n):
            if i + j >= n: continue
            o = F.conv2d(xs[i], ws[j], None, stride, pad)
            out = o if out is None else out + o
    out = out + b.view(1, -1, 1, 1)
    return out.float().double()                      # accumulator is fp32

def fwd(mode):
    c = lambda x, conv, bn: conv_mode(x, *fold(conv, bn), conv.stride[0], conv.padding[0], mode)
    h = F.relu(c(xn.double(), m.conv1, m.bn1))
    h = F.max_pool2d(h, 3, 2, 1)
    for layer in (m.layer1, m.layer2, m.layer3, m.layer4):
        for blk in layer:
            idt = h if blk.downsample is None else c(h, blk.downsample[0], blk.downsample[1])
            o = F.relu(c(h, blk.conv1, blk.bn1)); o = F.relu(c(o, blk.conv2, blk.bn2)); o = c(o, blk.conv3, blk.bn3)
            h = F.relu(o + idt)
            if mode != 'f64': h = h.float().double()
    h = h.mean((2, 3))
    w = m.fc.weight.detach().double(); b = m.fc.bias.detach().double()
    return conv_mode(h.view(B, -1, 1, 1), w.view(*w.shape, 1, 1), b, 1, 0, mode).view(B, -1)

with torch.no_grad():
    ref = fwd('f64')
    scale = ref.abs().max().item()
    print('logit scale', scale, 'std', ref.std().item())
    for mode in ('f32', 'x6', 'x3', 'x1'):
        o = fwd(mode)
        e = (o - ref).abs()
        print(mode, 'max abs err / scale = %.3e' % (e.max().item() / scale), ' median rel = %.3e' % ((e / ref.abs().clamp_min(1e-9)).median().item()),
              ' rms/std %.3e' % ((e.pow(2).mean().sqrt() / ref.std()).item()))
