#!/usr/bin/env python
"""bench.py -- headline benchmark of the AddNoise hot path on MI355X.

Metric (BASELINE.json): corrupted + attacked images/sec/node (ResNet-50, PGD-7 + IN-C x5).
One "step" = one pass of the hot path over one resident batch of B = 256 uint8 NHWC images:

  * IN-C x5 : gaussian_noise at severities 1..5 (5*B corrupted images), each normalised and
              evaluated by ResNet-50 (forward, top-1);
  * PGD-7   : PGD-Linf (eps 2/255, rel_stepsize 3/40, 7 steps, random start) on the B clean images
              = 7 x (forward + backward-to-input + fused sign/project kernel) + 1 final forward.

images per step = 6*B.  Inputs are resident in HBM before the timed region.  Weak scaling: every
rank processes its own B images (global sample indices keep the noise field rank-invariant); the
only collective is the 3-scalar metric all-reduce after the timed region.

The TOP-LEVEL value / ms_per_step / dtype / roofline are those of the tolerance-meeting path: the reference-precision
engine (`--precision bf16x3`, the default: logits within 1e-4 of the fp32 network, which is what the reference computes,
noise/utils/adv/attack.py:20-23).  The bf16 engine's figure for the same step follows as `fast_mode` with its stated
error; BASELINE configs 4 and 5 (ViT-B/16 x ImageNet-C, ResNet-50 adversarial training) follow as `secondary`, three
steps each; the oracle on the host cores as `cpu_baseline`.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_DEFAULT = 256
H = W = 224
ELEMS = H * W * 3
BYTES_PER_IMAGE = 2 * ELEMS            # uint8 in + uint8 out
FLOP_FWD = 8.2e9                        # ResNet-50 forward, 2 x 4.09 GMAC
HBM_PEAK = 8.0e12
MFMA_BF16_PEAK = 2.5e15


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=B_DEFAULT)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', choices=['bf16x3', 'bf16'], default='bf16x3',
                    help="engine of the timed step: 'bf16x3' = the reference-precision engine (hi + lo bf16 pairs, three MFMA products per "
                         "contraction, logits within 1e-4 of the fp32 network; default) or 'bf16' (fast mode, ~3e-3)")
    ap.add_argument('--no-fast-mode', '--no-reference-precision', dest='no_other_engine', action='store_true',
                    help='skip the block that repeats the step on the OTHER engine (fast_mode under bf16x3, reference_precision under bf16)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the `secondary` block (adv_train and vit_inc, three steps each)')
    ap.add_argument('--secondary-steps', type=int, default=3)
    ap.add_argument('--cpu-sample', type=int, default=64, help='images of the bounded model leg of the CPU baseline')
    ap.add_argument('--cpu-sweep', action='store_true', help='only run the CPU 14 corruptions x 5 severities sweep (BASELINE.md 3b)')
    ap.add_argument('--workload', choices=['headline', 'vit_inc', 'vit_pgd', 'adv_train'], default='headline',
                    help="'headline' = the BASELINE.json metric (default); 'vit_inc' = BASELINE config 4: ViT-B/16 evaluated "
                         "on all 15 ImageNet-C corruptions x 5 severities generated on the GPU (frost on synthetic textures)")
    ap.add_argument('--one-stream', action='store_true', help='run the two halves of a step back to back on one stream (A/B)')
    ap.add_argument('--spawn-check', action='store_true',
                    help='launch-path self test (no GPU work): every rank joins a gloo group, rank 0 prints one JSON line '
                         'with the world size it saw -- covers the --gpus N re-launch on a CPU-only box')
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_multi_rank(n):
    """`python bench.py --gpus N` started as ONE plain process (no RANK in the environment): re-launch this very
    command line as N ranks, one per GPU, under torch.distributed.run on 127.0.0.1 -- the launch shape of the reference's
    own evaluation (exprs/exp/imagenet_c_loop_mini/eval.sh:21-23, torchrun --nproc_per_node=8) -- and hand back its exit
    code.  Rank 0 of the child job prints the one JSON line; stdout / stderr pass straight through."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def spawn_check(args):
    """Body of --spawn-check: the rendezvous of the multi-rank launch without any GPU work (gloo)."""
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({'spawn_check': True, 'n_gpus': world, 'requested_gpus': args.gpus,
                          'rank_sum': seen, 'expected_rank_sum': world * (world + 1) // 2}))


def build_workload(B, device, rank):
    from robustart_amd.model import get_model
    g = torch.Generator().manual_seed(1234 + rank)
    images = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8).to(device)
    labels = torch.randint(0, 1000, (B,), generator=g).to(device)
    torch.manual_seed(0)
    model = get_model({'type': 'resnet50_official', 'kwargs': {'num_classes': 1000}}).eval()
    for p in model.parameters():
        p.requires_grad_(False)
    return images, labels, model


class HipEngine:
    """The product path: the hand-written MFMA engine (robustart_amd/model/engine.py), in either precision.
    Two engine instances (own activation buffers, same folded weights) let the step run its two independent halves on two
    HIP streams: the corrupted evaluations (HBM-bound layer1 / layer2 most of the time) overlap the PGD chain's
    MFMA-bound deep layers and fill each other's grid tails."""
    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

    def __init__(self, model, device, two_streams=True, precision='bf16x3'):
        from robustart_amd.model.engine import ResNet50Engine, EngineModel
        self.precision = precision
        self.name = 'hip-igemm-' + precision
        self.eng = ResNet50Engine(model, device, precision)
        self.f_model = EngineModel(None, takes_normalized=False, mean=self.MEAN, std=self.STD, engine=self.eng)
        n_side = int(os.environ.get('RART_BENCH_SIDE_STREAMS', '1')) if two_streams else 0
        self.eval_engs = [ResNet50Engine(model, device, precision) for _ in range(n_side)] or [self.eng]
        self.sides = [torch.cuda.Stream(device=device) for _ in range(n_side)]

    def logits_from_u8(self, u8, k=0):
        return self.eval_engs[k % len(self.eval_engs)].logits_from_u8(u8, self.MEAN, self.STD)


def one_step(images, labels, path, step_idx, rank, B):
    from robustart_amd.noise import imagenet_c as C, adv
    base = (step_idx * 1_000_003 + rank * B)          # global sample index of this rank's first image
    main = torch.cuda.current_stream()
    sides = path.sides or [main]
    for sd in sides:
        if sd is not main:
            sd.wait_stream(main)                       # the previous step's consumers of the scratch buffers are done
    parts = []
    # IN-C x5: ONE launch reads the batch once and writes gaussian_noise at severities 1..5, each with its own field (seed = severity:
    # the reference's generation loop draws fresh noise per (image, severity)); the five evaluations then alternate over the streams
    bufs = one_step.extra.setdefault(('sev', images.shape), [torch.empty_like(images) for _ in range(5)])
    with torch.cuda.stream(sides[0]):
        C.noise_severities_(images, bufs, (1, 2, 3, 4, 5), (1, 2, 3, 4, 5), sample_offset=base, corruption_id=0)
    for sd in sides[1:]:
        sd.wait_stream(sides[0])
    for sev in range(1, 6):
        k = (sev - 1) % len(sides)
        with torch.cuda.stream(sides[k]):
            logits = path.logits_from_u8(bufs[sev - 1], k)
            _, _, pred = adv.logit_loss(logits, labels, 0, None, 1.0, want_grad=False)
            parts.append((pred.long() == labels).sum())
    x01 = C.to_unit_nchw(images, one_step.extra.setdefault(('x01', images.shape), torch.empty(B, 3, H, W, device=images.device)))
    x_adv = adv.pgd_linf(x01, labels, path.f_model, 2 / 255, 3 / 40, 7, seed=1, sample_offset=base)
    with torch.no_grad():
        logits = path.f_model(x_adv).float()
    _, _, pred = adv.logit_loss(logits, labels, 0, None, 1.0, want_grad=False)
    correct_adv = (pred.long() == labels).sum()
    for sd in sides:
        if sd is not main:
            main.wait_stream(sd)                       # join: the step is complete when every stream's part is
    return sum(parts), correct_adv


one_step.extra = {}
# Round 6, measured and not kept (one box, alternating runs): the same step as N independent chains, each over a contiguous 1 / N of the batch on
# its own stream and engine, so that every stream carries the same mix of work for the whole step (the form above overlaps its two streams
# only while the five evaluations last, a quarter of the step): N = 2 / 3 / 4 gave 184.0 / 184.2 / 184.2 ms per step against 183.5-184.1 for
# this form and 191.7 on one stream -- the step is bound by what its kernels need from the chip, not by gaps between them.



def measure_gaussian_roofline(B, device, launches=40, npairs=9):
    """HBM roofline of the dominant hand-written kernel of the corruption half (k_normal_noise_mfma<0>): rotate > 600 MB of
    distinct buffer pairs so the 256 MiB Infinity Cache cannot serve the stream.  The launch duration is the time between two
    events on the launch stream around `launches` back-to-back launches, divided by their number (it contains the dispatch gap
    between consecutive kernels, ~0.6 us, and agrees with rocprofv3's per-kernel average); bracketing EVERY launch with its own
    event pair adds ~2.5 us of event-record time to a 16 us kernel -- that figure is returned too, as a diagnostic.
    -> (avg_s, per_launch_bracket_avg_s, copy_avg_s) ; copy = torch's device copy of the same bytes over the same pairs."""
    from robustart_amd.noise import imagenet_c as C
    g = torch.Generator().manual_seed(7)
    src = [torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8).to(device) for _ in range(npairs)]
    dst = [torch.empty_like(s) for s in src]

    def timed(fn):
        for i in range(npairs):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for _ in range(3):                                   # three passes, the median one
            e0.record()
            for i in range(launches):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            best = sorted((best or []) + [e0.elapsed_time(e1) / launches])
        return best[len(best) // 2] * 1e-3

    noise = lambda i: C.corrupt_batch_(src[i % npairs], 0, 3, seed=0, sample_offset=i * B, out=dst[i % npairs])  # noqa: E731
    avg = timed(noise)
    measure_gaussian_roofline.extra = {}
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for i in range(launches):
        ev[i][0].record()
        noise(i)
        ev[i][1].record()
    torch.cuda.synchronize()
    bracket = sum(a.elapsed_time(b) for a, b in ev) / launches * 1e-3
    # calibration beside it: a plain device copy of the SAME bytes over the same rotating pairs (PyTorch's copy kernel): what a
    # read + write stream of this size sustains on this part, next to the 8 TB/s spec the fraction is quoted against
    copy_s = timed(lambda i: dst[i % npairs].copy_(src[i % npairs]))
    measure_gaussian_roofline.extra = extra = {}
    # and a PLAIN HIP copy kernel of the same bytes (rart_copy_calibration): in the noise kernel's own geometry (one wave per 1 KiB, 16 B per
    # lane) and as the fastest plain copy found at this size (grid-stride, 1 024 workgroups) -- the ceiling a kernel that does nothing but move
    # the launch's bytes reaches at B = 256
    from robustart_amd import _lib
    lib, nbytes = _lib.load(), src[0].numel()
    for variant, key in ((0, 'plain_copy_same_geometry_s'), (1, 'plain_copy_grid_stride_s')):
        extra[key] = timed(lambda i, v=variant: _lib.check(lib.rart_copy_calibration(src[i % npairs].data_ptr(), dst[i % npairs].data_ptr(),
                                                                                      nbytes, v, _lib.stream_ptr())))
    if launches >= 20:
        # (a) the same single-severity launches alternating over TWO streams: the tail of one launch overlaps the ramp of the next
        #     (the five severities of the workload are independent); time = main-stream events around the fork / join
        main = torch.cuda.current_stream(device)
        st = [torch.cuda.Stream(device=device) for _ in range(2)]

        def two_stream_pass():
            for q in st:
                q.wait_stream(main)
            for i in range(launches):
                with torch.cuda.stream(st[i & 1]):
                    noise(i)
            for q in st:
                main.wait_stream(q)
        two_stream_pass()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            two_stream_pass()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / launches * 1e-3)
        extra['two_streams_s'] = sorted(ts)[1]
        # (b) the workload's own form: ONE launch per source batch writes severities 1..5 (rart_noise_multi_u8: the chunk is read once,
        #     five fields): 6 x 38.5 MB moved for 5 launch-equivalents of 2 x 38.5 MB algorithmic bytes each
        m_pairs = max(2, min(npairs, 5))
        outs = [[torch.empty_like(src[0]) for _ in range(5)] for _ in range(m_pairs)]
        multi = lambda i: C.noise_severities_(src[i % npairs], outs[i % m_pairs], (1, 2, 3, 4, 5), (1, 2, 3, 4, 5),   # noqa: E731
                                              sample_offset=i * B, corruption_id=0)
        extra['multi_launch_s'] = timed(multi)
        extra['multi_buffers_mb'] = (npairs + 5 * m_pairs) * src[0].numel() / 1e6
        del outs
    return avg, bracket, copy_s


def gaussian_noise_block(B, device):
    """`hbm_roofline_gaussian_noise`: the corruption half's kernel against the HBM roof.  achieved / frac = the bytes the launch MOVES
    (source read once + five severity outputs = 6 x 38.5 MB) / its duration / 8 TB/s -- a physical roofline fraction, always < 1.  The
    throughput figure "five corrupted batches per launch, each worth the 2 x 150 528 B per image of a single-severity pass" is NOT a
    roofline fraction (the launch avoids 4 of those 10 byte-units) and is reported under `launch_equivalent` with its own names.  The
    single-severity launch -- what AddNoise.add_noise(batch, severity=s) issues -- follows with the same definition."""
    avg, bracket, copy_s = measure_gaussian_roofline(B, device)
    ex = dict(getattr(measure_gaussian_roofline, 'extra', {}) or {})
    algo = BYTES_PER_IMAGE * B
    single = {'kernel': 'k_normal_noise_mfma<0> (gaussian_noise, one severity, B=%d, u8 NHWC in/out)' % B,
              'bound': 'hbm', 'achieved': algo / avg / 1e9, 'peak': HBM_PEAK / 1e9,
              'unit': 'GB/s', 'frac': algo / avg / HBM_PEAK, 'traffic': pmc_traffic('k_normal_noise_mfmaI'),
              'traffic_note': 'HBM bytes per launch from the committed PMC pass profiles/%s '
                              '(FETCH_SIZE x2 + WRITE_SIZE), not re-measured in this run' % getattr(pmc_traffic, 'source', '?'),
              'avg_launch_us': avg * 1e6, 'per_launch_event_bracket_us': bracket * 1e6,
              'timing': 'two events on the launch stream around 40 back-to-back launches / 40 (median of 3 passes); '
                        'per_launch_event_bracket_us = every launch between its own event pair (adds the event records)',
              'algorithmic_bytes_per_launch': algo,
              'device_copy_same_bytes': {'avg_launch_us': copy_s * 1e6, 'achieved': algo / copy_s / 1e9, 'unit': 'GB/s',
                                         'frac_of_peak': algo / copy_s / HBM_PEAK, 'kernel_vs_copy': copy_s / avg,
                                         'note': 'torch Tensor.copy_ over the same 9 rotating buffer pairs: the read+write '
                                                 'rate this part sustains at this size'}}
    if 'plain_copy_same_geometry_s' in ex:
        cg, cs = ex['plain_copy_same_geometry_s'], ex['plain_copy_grid_stride_s']
        single['plain_copy_kernel'] = {
            'same_geometry': {'avg_launch_us': cg * 1e6, 'frac_of_peak': algo / cg / HBM_PEAK, 'kernel_vs_copy': cg / avg},
            'grid_stride_1024_workgroups': {'avg_launch_us': cs * 1e6, 'frac_of_peak': algo / cs / HBM_PEAK, 'kernel_vs_copy': cs / avg},
            'note': 'rart_copy_calibration: a hand-written HIP kernel that only copies the launch\'s bytes (16 B per lane), in the noise '
                    'kernel\'s launch geometry and as the fastest plain copy of this size; SQ counters of the noise kernels beside it: '
                    'profiles/r06_noise_counters.csv (single severity: 52 % of wave cycles waiting on memory, 32 % issue stalls; five '
                    'severities: 64 % issue stalls = VALU-bound)'}
    # the same kernel on a 4x larger launch (B = 1024 by default, 3 rotating pairs = 925 MB): how much of the B = 256 gap to the peak is
    # launch ramp / tail of a 18 us kernel rather than the steady-state rate
    if os.environ.get('RART_BENCH_NO_4X') != '1':        # (the PMC passes set this: their per-kernel averages must hold B = 256 launches only)
        avg4, _, copy4 = measure_gaussian_roofline(4 * B, device, launches=12, npairs=3)
        single['at_4x_batch'] = {'batch': 4 * B, 'avg_launch_us': avg4 * 1e6, 'achieved': 4 * algo / avg4 / 1e9,
                                 'unit': 'GB/s', 'frac': 4 * algo / avg4 / HBM_PEAK,
                                 'device_copy_frac_of_peak': 4 * algo / copy4 / HBM_PEAK}
    if 'multi_launch_s' not in ex:
        return single
    t5 = ex['multi_launch_s']
    moved = (1 + 5) * ELEMS * B
    return {'kernel': 'k_normal_noise_mfma_multi<0> (gaussian_noise, severities 1..5 of B=%d u8 NHWC images in one launch, one field per severity: '
                      'the launch the bench step issues)' % B,
            'bound': 'hbm', 'achieved': moved / t5 / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': moved / t5 / HBM_PEAK,
            'basis': 'bytes the launch moves (the source once + five outputs) / its duration',
            'avg_launch_us': t5 * 1e6, 'algorithmic_bytes_per_launch': moved, 'bytes_moved_per_launch': moved,
            'traffic': pmc_traffic('k_normal_noise_mfma_multi'),
            'launch_equivalent': {'launch_equivalent_us': t5 / 5 * 1e6, 'corrupted_images_per_s': 5 * B / t5,
                                  'throughput_equivalent_gb_per_s': 5 * algo / t5 / 1e9, 'throughput_equivalent_frac': 5 * algo / t5 / HBM_PEAK,
                                  'single_severity_bytes_x5': 5 * algo,
                                  'note': 'NOT a roofline fraction: 5 x the algorithmic bytes of a single-severity pass (2 x 150528 B per image) / '
                                          'the duration of the five-severity launch, which moves only 6/10 of those bytes -- the speed-up of '
                                          'reading the source once, for comparison with SURVEY 8(d)\'s per-severity 13.8 us target'},
            'timing': 'two events on the launch stream around 40 back-to-back launches / 40 (median of 3 passes), %d MB of rotating buffers' % ex['multi_buffers_mb'],
            'single_severity_launch_one_stream': single,
            'single_severity_launches_two_streams': {'avg_launch_us': ex['two_streams_s'] * 1e6, 'achieved': algo / ex['two_streams_s'] / 1e9,
                                                     'unit': 'GB/s', 'frac': algo / ex['two_streams_s'] / HBM_PEAK}}


def pmc_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/r06_pmc_traffic.json, else r05 / r04 / ...; produced by profiles/summarize_pmc.py); None if absent."""
    try:
        path = next((q for q in (os.path.join(ROOT, 'profiles', 'r0%d_pmc_traffic.json' % r) for r in (6, 5, 4, 3, 2)) if os.path.exists(q)), None)
        pmc_traffic.source = os.path.basename(path)
        with open(path) as f:
            d = json.load(f)
        if key == 'igemm':
            return d['k_conv_igemm_bf16_all']['hbm_bytes']
        # a kernel family: the launch-weighted mean over its template instances (forward / backward, identity / first block)
        hit = [v for k, v in d['kernels'].items() if key in k]
        if hit:
            return sum(v['hbm_bytes'] * v.get('calls', 1) for v in hit) / sum(v.get('calls', 1) for v in hit)
    except Exception:
        pass
    return None


KERNEL_NAMES = {
    'igemm': 'k_conv_igemm_bf16 (layer4 first block, the unfused remainder)',
    'halo3x3': 'k_conv3x3_halo / k_conv3x3_image256 (3x3, input tile resident in LDS)',
    'bottleneck': 'k_bottleneck56 (layer1 blocks fused: 1x1 + 3x3 + 1x1 + shortcut in one launch)',
    'bottleneck14': 'k_bottleneck14 (layer3 identity blocks fused, one image per workgroup)',
    'bottleneck28': 'k_bottleneck28 (layer2 identity blocks fused, a quarter image per workgroup)',
    'bottleneck7': 'k_bottleneck7 (layer4 identity blocks fused, one image per workgroup)',
    'bottleneck_s2': 'k_bottleneck_s2 (stride-2 first blocks of layer2 / layer3, forward: 1x1 + 3x3/2 + 1x1 + projection in one launch)',
    'bottleneck_s2_bwd': 'k_bottleneck_s2_bwd (the same blocks, backward-to-input in one launch)',
    'fc_small_m': 'k_gemm_small_m (classifier head and its backward: one 32 x 32 tile per workgroup, waves split K)',
    'gemm_pair': 'k_gemm_pair<TM, TN, conv> / k_gemm_pair_pp<TN, conv> (split-bf16 implicit-GEMM convolution of the reference-precision engine: '
                 'four operand planes staged once per K step, three MFMAs per fragment pair; the 256-row tiles of layers at least 256 wide '
                 'and deep on the ping-pong schedule of round 6, csrc/gemm_pair_pp.hip)',
    'conv_tail_pair': 'k_conv3x3_tail_pair<C, NEXT> (reference-precision 3x3 + 1x1 expansion [+ the neighbouring block\'s 1x1 reduction] in one launch)',
    'conv_tail256_pair': 'k_conv3x3_tail256_pair (reference-precision 3x3 + 1x1 expansion of layer3 / layer4 in one launch)',
    'stem_pair': 'k_stem_fwd_pair / k_stem_bwd_pair (reference-precision stem: normalise + 7x7/2 + ReLU + max pool, and its backward, one launch each)'}
PMC_KEYS = {'bottleneck': 'k_bottleneck56', 'bottleneck14': 'k_bottleneck14', 'bottleneck28': 'k_bottleneck28', 'bottleneck7': 'k_bottleneck7',
            'bottleneck_s2': '15k_bottleneck_s2I', 'bottleneck_s2_bwd': 'k_bottleneck_s2_bwd', 'halo3x3': 'k_conv3x3', 'igemm': 'igemm',
            'gemm_pair': 'k_gemm_pair', 'conv_tail_pair': 'k_conv3x3_tail_pairI', 'conv_tail256_pair': 'k_conv3x3_tail256_pair',
            'stem_pair': 'k_stem_'}


def measure_engine_roofline(path, images, labels):
    """`roofline` = the kernel family with the LARGEST share of one PGD gradient evaluation (forward + backward-to-input at B = 256,
    every launch timed with events on the launch stream) of the engine the step ran on.  A family is priced against its BINDING roof --
    max(FLOPs / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s), where the algorithmic bytes are every operand once (x in + out + weight tables
    + 1-bit sign tensors; the residual re-read and the halo rows are implementation traffic: they show up in `traffic`, the PMC bytes per
    launch, and in `overfetch` = traffic / algorithmic bytes) -- with the MFMA fraction beside it.  On the reference-precision engine the
    FLOPs are the bf16 MFMA FLOPs ISSUED (three per algorithmic product; `achieved_fp32_equivalent` = a third of it) and the bytes are
    those of the hi + lo pair tensors (4 B per element).  Every other family follows under `other_mfma_kernels`, the total under
    `all_conv_launches`."""
    eng = path.eng
    x3 = path.precision != 'bf16'
    x01 = images.permute(0, 3, 1, 2).float().div_(255.0).contiguous()
    eng.forward_backward(x01, path.MEAN, path.STD, labels, 0)          # warm
    eng.profile = []
    eng.forward_backward(x01, path.MEAN, path.STD, labels, 0)
    torch.cuda.synchronize()
    prof, eng.profile = eng.profile, None
    fam = {}
    for p in prof:
        f = fam.setdefault(p[3], {'n': 0, 's': 0.0, 'flops': 0.0, 'bytes': 0.0, 'has_bytes': True})
        f['n'] += 1
        f['s'] += p[1].elapsed_time(p[2]) * 1e-3
        f['flops'] += p[0]
        if len(p) > 4 and p[4] is not None:
            f['bytes'] += p[4]
        else:
            f['has_bytes'] = False
    tot_s = sum(f['s'] for f in fam.values())
    tot_f = sum(f['flops'] for f in fam.values())

    def block(kind, f):
        o = {'achieved': f['flops'] / f['s'] / 1e12, 'unit': 'TFLOP/s', 'frac': f['flops'] / f['s'] / MFMA_BF16_PEAK,
             'avg_launch_us': f['s'] / f['n'] * 1e6, 'launches': f['n'], 'share_of_gradient_evaluation': f['s'] / tot_s}
        if x3:
            o['achieved_fp32_equivalent'] = o['achieved'] / 3.0
        if f['has_bytes']:
            floor = max(f['flops'] / MFMA_BF16_PEAK, f['bytes'] / HBM_PEAK)
            o['algorithmic_bytes_per_launch'] = f['bytes'] / f['n']
            o['binding_roof'] = 'hbm' if f['bytes'] / HBM_PEAK > f['flops'] / MFMA_BF16_PEAK else 'mfma'
            o['frac_of_binding_roof'] = floor / f['s']
            o['hbm_algorithmic'] = {'achieved': f['bytes'] / f['s'] / 1e9, 'unit': 'GB/s', 'frac': f['bytes'] / f['s'] / HBM_PEAK}
        tr = pmc_traffic(PMC_KEYS.get(kind, kind))
        if tr:
            o['traffic'] = tr
            if f['has_bytes']:
                o['overfetch'] = tr / (f['bytes'] / f['n'])
        return o
    dom = max(fam, key=lambda k: fam[k]['s'])
    d = block(dom, fam[dom])
    out = {'kernel': '%s, %d launches of one gradient evaluation (ResNet-50 forward + backward-to-input, B=%d, engine precision %s)'
                     % (KERNEL_NAMES.get(dom, dom), fam[dom]['n'], images.shape[0], path.precision),
           'selection': 'the kernel family with the largest share of the gradient evaluation (%.0f %% of the kernel time)' % (100 * fam[dom]['s'] / tot_s),
           'bound': d.get('binding_roof', 'mfma')}
    if d.get('binding_roof') == 'hbm':
        out.update({'achieved': d['hbm_algorithmic']['achieved'], 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': d['hbm_algorithmic']['frac'],
                    'mfma': {'achieved': d['achieved'], 'peak': MFMA_BF16_PEAK / 1e12, 'unit': 'TFLOP/s', 'frac': d['frac']}})
    else:
        out.update({'achieved': d['achieved'], 'peak': MFMA_BF16_PEAK / 1e12, 'unit': 'TFLOP/s', 'frac': d['frac']})
        if 'hbm_algorithmic' in d:
            out['hbm_algorithmic'] = d['hbm_algorithmic']
    if x3:
        out['achieved_fp32_equivalent'] = d['achieved_fp32_equivalent']
        out['note'] = ('achieved counts the bf16 MFMA FLOPs actually issued (3 per algorithmic product); the fp32-equivalent rate is a third of it '
                       '(the fp32 MFMA peak of this part is %.1f TFLOP/s)' % (MFMA_F32_PEAK / 1e12))
    out.update({'traffic': d.get('traffic'), 'overfetch': d.get('overfetch'),
                'algorithmic_bytes_per_launch': d.get('algorithmic_bytes_per_launch'),
                'issued_flops_per_launch': fam[dom]['flops'] / fam[dom]['n'],
                'algorithmic_flops_per_launch': fam[dom]['flops'] / fam[dom]['n'] / (3.0 if x3 else 1.0),
                'avg_launch_us': d['avg_launch_us'], 'launches': fam[dom]['n'], 'share_of_gradient_evaluation': d['share_of_gradient_evaluation'],
                'traffic_note': 'HBM bytes per launch (launch-weighted mean over the family\'s template instances) from the committed PMC pass '
                                'profiles/%s (rocprofv3 FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB units), not re-measured in this run'
                                % getattr(pmc_traffic, 'source', '?')})
    if x3:
        out['clock_note'] = ('peaks are priced at the 2.4 GHz boost clock; this workload holds the part at its 1.32 kW package-power cap, where the '
                             'shader clock reads 2.25 GHz (ResNet-50) / 1.91 GHz (ViT-B/16): profiles/r06_clock_under_load.txt, not re-measured in this run')
    out['other_mfma_kernels'] = {'%s, %d launches' % (KERNEL_NAMES.get(k, k), f['n']): block(k, f) for k, f in sorted(fam.items()) if k != dom}
    out['all_conv_launches'] = {'achieved': tot_f / tot_s / 1e12, 'unit': 'TFLOP/s', 'frac': tot_f / tot_s / MFMA_BF16_PEAK,
                                'seconds_per_fwd_bwd': tot_s, 'launches': len(prof)}
    return out


MFMA_F32_PEAK = 157.3e12               # v_mfma_f32_32x32x2_f32, the rate fp32 operands would get

ARITHMETIC = {
    'bf16x3': 'activations / gradients / weights as hi + lo bf16 pairs (16 significand bits), x.w = lo.hi + hi.lo + hi.hi on '
              'v_mfma_f32_32x32x16_bf16 with fp32 accumulation (rart_gemm_pair_bf16 and the fused pair kernels: the four operand planes of a '
              'K step staged once in LDS, three MFMAs per fragment pair); logits within 1e-4 of the fp32 network, PGD / AutoAttack outcomes '
              'identical to the fp32 module (tests/test_engine_x3_gpu.py, tests/test_outcome_gpu.py, profiles/r04_outcome_x3_vs_fp32.json)',
    'bf16': 'activations / gradients / weights stored as bf16, fp32 MFMA accumulation: logits ~2.6e-3 (median) from the fp32 network, PGD-7 '
            'per-image outcome agreement 99.5 %, gradient sign agreement 97.4 % (profiles/r04_outcome_bf16_vs_fp32.json) -- narrower than the '
            'reference\'s fp32 arithmetic and outside the north star\'s 1e-4; offered as a fast mode, never as the headline'}


def measure_other_engine(model, device, images, labels, rank, B, steps, two_streams, precision):
    """The SAME step on the other engine: `fast_mode` (bf16) beside the reference-precision headline, or `reference_precision` when the
    step was run with --precision bf16.  One warm-up step, `steps` timed steps, then the roofline block of that engine."""
    path = HipEngine(model, device, two_streams=two_streams, precision=precision)
    one_step(images, labels, path, 0, rank, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        acc = one_step(images, labels, path, 1 + i, rank, B)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_flops = (5 + 15) * B * FLOP_FWD
    return {'value': 6 * B * steps / dt, 'unit': 'images/s', 'ms_per_step': dt / steps * 1e3, 'steps': steps, 'dtype': precision,
            'arithmetic': ARITHMETIC[precision], 'model_path': path.name, 'correct_corrupted': int(acc[0]), 'correct_adv': int(acc[1]),
            'step_algorithmic': {'achieved': step_flops * steps / dt / 1e12, 'unit': 'TFLOP/s',
                                 'note': 'algorithmic FLOPs of the step (20 forward-equivalents) / time'},
            'roofline': measure_engine_roofline(path, images, labels)}


def _cpu_model_name():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


_CPU_IMGS = None      # the 256-image batch of BASELINE.md section 3, created before the pool forks (workers inherit it)


def _cpu_batch():
    global _CPU_IMGS
    if _CPU_IMGS is None:
        g = torch.Generator().manual_seed(1234)
        _CPU_IMGS = torch.randint(0, 256, (256, H, W, 3), generator=g, dtype=torch.uint8).numpy()
    return _CPU_IMGS


def _cpu_corrupt_chunk(job):
    """Pool worker: the oracle's per-image loop (add_noise_utils.py:27-31) over one chunk of the 256-image batch."""
    import numpy as np
    from oracle import corruptions_np as O
    name, sev, seed, lo, hi = job
    rs = np.random.RandomState(seed + lo)
    imgs = _cpu_batch()[lo:hi]
    t0 = time.perf_counter()
    O.corrupt_batch(name, imgs, sev, rs)
    return time.perf_counter() - t0


def _cpu_corrupt_rate(pool, procs, name, sev, reps):
    """images/s of one corruption at one severity over the 256-image batch of BASELINE.md section 3, all cores
    (multiprocessing.Pool): median wall time of `reps` repetitions after one warm-up."""
    n = 256
    per = max(1, (n + procs - 1) // procs)
    jobs = [(name, sev, 0, lo, min(lo + per, n)) for lo in range(0, n, per)]
    walls = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        pool.map(_cpu_corrupt_chunk, jobs)
        walls.append(time.perf_counter() - t0)
    walls = sorted(walls[1:])
    return n / walls[len(walls) // 2]


def cpu_baseline(model_sample, model_fp32, reps=5):
    """the oracle (numpy / scipy / Pillow restatement, pinned to the reference by the golden
    fixtures) + a PyTorch-CPU fp32 ResNet-50 on the host cores, on the same workload as the GPU line:
      * gaussian_noise at severities 1..5 on the 256-image batch (seed 1234): one process and Pool(os.cpu_count()),
        median of `reps` repetitions after a warm-up;
      * ResNet-50 fp32 evaluation (batch 32) and PGD-Linf-7 on a bounded sample of `model_sample` images, all cores;
    combined into images/s of the step (5 corrupted evaluations + 1 attacked evaluation per source image)."""
    import multiprocessing as mp
    import numpy as np
    from oracle import corruptions_np as O
    from oracle import attacks_ref as A
    procs = os.cpu_count() or 1
    t_all = time.time()
    imgs = _cpu_batch()
    torch.set_num_threads(torch.get_num_threads())      # (no-op; the pool below must fork BEFORE torch spins up its workers)
    with mp.get_context('fork').Pool(procs) as pool:
        pool_rate = {sev: _cpu_corrupt_rate(pool, procs, 'gaussian_noise', sev, reps) for sev in range(1, 6)}
    # single process, severity 3 (the configuration the survey container timed: 297 images/s on one Xeon core)
    rs = np.random.RandomState(0)
    one = []
    for r in range(3):
        t0 = time.perf_counter()
        O.corrupt_batch('gaussian_noise', imgs[:64], 3, rs)
        one.append(64 / (time.perf_counter() - t0))
    m = model_fp32.float().eval()
    f = lambda z: m(A.normalize(z))  # noqa: E731
    n = model_sample
    y = torch.from_numpy(rs.randint(0, 1000, (n,)))
    x = torch.from_numpy(imgs[:n]).permute(0, 3, 1, 2).float() / 255
    with torch.no_grad():
        f(x[:32]).argmax(1)                                      # warm-up (thread pool, MKL-DNN primitives)
    t0 = time.perf_counter()
    with torch.no_grad():
        for s in range(0, n, 32):
            f(x[s:s + 32]).argmax(1)
    t_eval = (time.perf_counter() - t0) / n                      # seconds per evaluated image
    t0 = time.perf_counter()
    for s in range(0, n, 32):
        u = (torch.rand(x[s:s + 32].shape) * 2 - 1) * (2 / 255)
        adv_x = A.pgd_linf(f, x[s:s + 32], y[s:s + 32], 2 / 255, 3 / 40, 7, init_u=u)
        with torch.no_grad():
            f(adv_x).argmax(1)
    t_pgd = (time.perf_counter() - t0) / n                       # seconds per attacked + evaluated image
    t_corrupt = sum(1.0 / pool_rate[sev] for sev in range(1, 6))  # seconds per source image, 5 severities, all cores
    per_source_image = t_corrupt + 5 * t_eval + t_pgd
    return {'value': 6.0 / per_source_image, 'unit': 'images/s', 'cores': procs, 'kind': 'port',
            'cpu_model': _cpu_model_name(), 'torch_threads': torch.get_num_threads(),
            'sample': 'oracle gaussian_noise sev 1..5 on 256 images (Pool(%d), median of %d reps after a warm-up) + torch-CPU '
                      'fp32 ResNet-50 evaluation and PGD-Linf-7 on %d images at batch 32 (%.1f s in total)'
                      % (procs, reps, n, time.time() - t_all),
            'gaussian_noise_sev3_images_per_s': {'pool_all_cores': pool_rate[3], 'one_process': sorted(one)[1]},
            'resnet50_fp32_eval_images_per_s': 1.0 / t_eval, 'pgd7_images_per_s': 1.0 / t_pgd,
            'libs': {'numpy': np.__version__, 'torch': torch.__version__}}


def cpu_sweep(reps=5):
    """every ImageNet-C corruption the oracle restates x 5 severities on the 256-image batch,
    all cores (frost needs the reference's absent textures and is skipped) -> profiles/r02_cpu_sweep.json.
        python bench.py --cpu-sweep"""
    import multiprocessing as mp
    from oracle import corruptions_np as O
    procs = os.cpu_count() or 1
    names = [n for n in O.CORRUPTION_NAMES[:15] if n != 'frost']
    _cpu_batch()
    res = {'cpu_model': _cpu_model_name(), 'cores': procs, 'images': 256, 'reps': reps, 'unit': 'images/s', 'rates': {}}
    with mp.get_context('fork').Pool(procs) as pool:
        for nm in names:
            res['rates'][nm] = {}
            for sev in range(1, 6):
                r = reps if nm not in ('glass_blur', 'zoom_blur') else 1       # python pixel loops: one repetition
                res['rates'][nm][str(sev)] = _cpu_corrupt_rate(pool, procs, nm, sev, r)
            print(nm, res['rates'][nm], flush=True)
    tot = sum(1.0 / v for d in res['rates'].values() for v in d.values())
    res['all_14x5_images_per_s'] = len(names) * 5 / tot
    out = os.path.join(ROOT, 'gpurun_out', 'cpu_sweep.json')
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps({'cpu_sweep': res['all_14x5_images_per_s'], 'unit': 'corrupted images/s', 'cores': procs}))


def run_vit_inc(args, device, rank, world, dist):
    """BASELINE config 4 (secondary mode, not the headline line): per step, one resident batch of 256 uint8 images is corrupted by each of
    the 15 benchmark corruptions at 5 severities (75 corrupted batches; frost blends SYNTHETIC textures -- the reference's six photographs
    are not in its repository, imagenet_c/corruptions.py:251-256 -- drawn and cropped on the device exactly as for real ones) and each is
    evaluated by ViT-B/16 on the HIP engine.  A `reference_precision` block repeats the step on the 'fp32x' engine (logits within 1e-4 of
    the fp32 network; the reference evaluates in fp32: exprs/exp/imagenet_c_loop_mini/config_vit_base.yaml:1-9)."""
    import numpy as np
    from robustart_amd.model import get_model
    from robustart_amd.model.vit_engine import ViTEngine
    from robustart_amd.noise import imagenet_c as C, adv
    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    images = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8).to(device)
    labels = torch.randint(0, 1000, (B,), generator=g).to(device)
    torch.manual_seed(0)
    rs = np.random.RandomState(77)
    C.set_frost_textures([rs.randint(0, 256, (240 + 16 * i, 300 + 8 * i, 3)).astype(np.uint8) for i in range(6)])
    # the 75 (corruption, severity) items of a step are independent: they alternate between two streams, each with its own engine
    # workspace, so the ragged last round of one launch (a 256 x 256-tile GEMM with N = 768 has 591 workgroups for 256 CUs) is
    # filled by the other stream's work.  --one-stream keeps a single queue.
    n_q = 1 if args.one_stream else 2
    model = get_model({'type': 'vit_base'}).eval()
    scratches = [torch.empty_like(images) for _ in range(n_q)]
    queues = [torch.cuda.Stream(device=device) for _ in range(n_q)] if n_q > 1 else [torch.cuda.current_stream(device)]
    ids = list(range(15))
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

    def make_step(engs):
        def step(k):
            main = torch.cuda.current_stream(device)
            tots = [torch.zeros((), dtype=torch.long, device=device) for _ in range(n_q)]
            for q in queues:
                q.wait_stream(main)
            j = 0
            for cid in ids:
                for sev in range(1, 6):
                    qi = j % n_q
                    j += 1
                    with torch.cuda.stream(queues[qi]):
                        C.corrupt_batch_(images, cid, sev, seed=0, sample_offset=k * 1_000_003 + rank * B, out=scratches[qi])
                        logits = engs[qi].logits_from_u8(scratches[qi], mean, std)
                        _, _, pred = adv.logit_loss(logits, labels, 0, None, 1.0, want_grad=False)
                        tots[qi] += (pred.long() == labels).sum()
            for q in queues:
                main.wait_stream(q)
            return sum(tots)
        return step

    def timed(step, warmup, steps):
        for i in range(warmup):
            step(i)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt
    engs = [ViTEngine(model, device) for _ in range(n_q)]
    dt = timed(make_step(engs), args.warmup, args.steps)
    n_img = len(ids) * 5 * B * world
    out = {'metric': 'corrupted images/sec/node (ViT-B/16, ImageNet-C 15 corruptions x 5 severities, on-GPU noise)',
           'value': n_img * args.steps / dt, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
           'config': {'workload': 'BASELINE config 4 (secondary): 75 corrupted batches of %d per step -> ViT-B/16 eval' % B,
                      'corruptions': len(ids), 'frost_textures': 'synthetic (6 random uint8 photographs, cropped on the device)',
                      'global_batch': B * world, 'parallelism': 'dp%d' % world, 'streams': n_q}}
    if not args.no_other_engine:
        del engs
        torch.cuda.empty_cache()
        engs = [ViTEngine(model, device, precision='fp32x') for _ in range(n_q)]
        rsteps = max(1, min(args.steps, 2))
        rdt = timed(make_step(engs), 1, rsteps)
        out['reference_precision'] = {
            'value': n_img * rsteps / rdt, 'unit': 'images/s', 'ms_per_step': rdt / rsteps * 1e3, 'steps': rsteps, 'dtype': 'bf16x3',
            'arithmetic': 'activations / weights as hi + lo bf16 pairs, x.w = lo.hi + hi.lo + hi.hi on v_mfma_f32_32x32x16_bf16 with fp32 '
                          'accumulation (rart_gemm_pair_bf16), LayerNorm / soft-max / GELU in fp32; logits within 1e-4 of the fp32 network '
                          '(tests/test_vit_x3_gpu.py)',
            'step_algorithmic': {'achieved': n_img * rsteps * 35.1e9 / rdt / 1e12, 'unit': 'TFLOP/s',
                                 'vs_fp32_mfma_peak': n_img * rsteps * 35.1e9 / rdt / MFMA_F32_PEAK,
                                 'note': 'fp32-equivalent FLOPs (35.1 GFLOP per ViT-B/16 forward) / time; 157.3 TFLOP/s is what fp32 MFMA operands peak at'}}
    del engs
    torch.cuda.empty_cache()
    return out


def run_vit_pgd(args, device, rank, world, dist):
    """Secondary mode: PGD-Linf-7 (eps 2/255) evaluation of ViT-B/16, forward and backward-to-input on the HIP engine
    (fused attention forward / backward kernels, every GEMM on the igemm kernel)."""
    from robustart_amd.model import get_model
    from robustart_amd.model.engine import EngineModel
    from robustart_amd.model.vit_engine import ViTEngine
    from robustart_amd.noise import adv as A
    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    x01 = (torch.randint(0, 256, (B, 3, H, W), generator=g, dtype=torch.uint8).float() / 255.0).to(device)
    labels = torch.randint(0, 1000, (B,), generator=g).to(device)
    torch.manual_seed(0)
    f_model = EngineModel(None, takes_normalized=False, engine=ViTEngine(get_model({'type': 'vit_base'}).eval(), device))

    def step(k):          # (splitting the batch into two half-batch chains on two streams was measured: 1 086 vs 1 077 images/s, not kept)
        xa = A.pgd_linf(x01, labels, f_model, 2 / 255, 3 / 40, 7, seed=k, sample_offset=rank * B)
        return (f_model(xa).argmax(1) == labels).sum()
    for i in range(args.warmup):
        step(i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    flops = 15 * 35.1e9 * B * world                       # (2k + 1) F, F(ViT-B/16) = 35.1 GFLOP
    return ({'metric': 'attacked images/sec/node (ViT-B/16, PGD-Linf-7 eval)', 'value': B * world * args.steps / dt,
                          'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
                          'step_mfma': {'achieved': flops * args.steps / dt / 1e12, 'unit': 'TFLOP/s',
                                        'note': '15 forward-equivalents per image'},
                          'config': {'workload': 'secondary: PGD-Linf-7 eps 2/255 + final forward, ViT-B/16, batch 256 per GPU',
                                     'global_batch': B * world, 'parallelism': 'dp%d' % world}})


def run_adv_train(args, device, rank, world, dist):
    """BASELINE config 5 (secondary mode): ResNet-50 adversarial training, PGD-3 inner loop on the HIP eval engine
    (re-folded from the live weights every step), train-mode forward/backward on the HIP train engine, label-smoothed
    CE / SGD-Nesterov + EMA kernels over flat arenas, bucketed gradient all-reduce over RCCL overlapped with backward."""
    from robustart_amd.model import get_model
    from robustart_amd.model.engine import EngineModel
    from robustart_amd.model.train_engine import ResNet50TrainEngine
    from robustart_amd.noise import adv as A
    from robustart_amd.train.arena import HipOptimizer, ParamArena, label_smooth_ce
    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    x01 = (torch.randint(0, 256, (B, 3, H, W), generator=g, dtype=torch.uint8).float() / 255.0).to(device)
    labels = torch.randint(0, 1000, (B,), generator=g).to(device)
    torch.manual_seed(0)                                   # identical initial weights on every rank
    model = get_model({'type': 'resnet50_official'}).to(device)
    arena = ParamArena(model, bucket_bytes=48 << 20)
    opt = HipOptimizer(arena, 'SGD', lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4, ema_decay=0.9999)
    model.eval()
    attack = EngineModel(model, takes_normalized=False)
    model.train()
    eng = ResNet50TrainEngine(model, device, on_grad_ready=arena.grad_ready)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

    def step(k):
        attack.rart_engine.refold(model)
        xa = A.pgd_linf(x01, labels, attack, 4 / 255, 0.4, 3, seed=k, sample_offset=rank * B)
        logits = eng.forward(xa, False, mean, std)
        loss_rows, dl = label_smooth_ce(logits, labels, 0.1, 1.0 / B)
        eng.backward(dl)
        opt.lr = 0.1
        opt.step(grad_scale=arena.finish_grad_exchange())
        eng.repack()
        return loss_rows
    for i in range(args.warmup):
        step(i)
    tb = time.perf_counter()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss_rows = step(args.warmup + i)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    with_barrier = dt
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the same per-rank diagnostics as the headline workload, plus what the gradient exchange of the last step did:
        # the first real 8-GPU run of config 5 reads communicator / barrier / exchange times apart from the step time
        _diag(rank, phase='timed_region', opening_barrier_s=t0 - tb, own_steps_s=own, with_closing_barrier_s=with_barrier, max_over_ranks_s=dt)
        xs = arena.exchange_stats()
        if xs is not None:
            _diag(rank, phase='grad_exchange', **{k: v for k, v in xs.items() if isinstance(v, (int, float))})
    flops = (3 * 2 + 3) * FLOP_FWD * B * world          # PGD-3: 3 x (fwd + bwd-to-input); train step: fwd + 2 x bwd
    final_loss = float(loss_rows.mean())
    del eng, attack, opt, arena, model
    torch.cuda.empty_cache()
    return ({'metric': 'adversarially trained images/sec/node (ResNet-50, cls_solver step, PGD-3 inner loop)',
                          'value': B * world * args.steps / dt, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
                          'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
                          'final_loss': final_loss,
                          'step_mfma': {'achieved': flops * args.steps / dt / 1e12, 'unit': 'TFLOP/s',
                                        'note': '9 forward-equivalents per image (SURVEY.md 8d)'},
                          'config': {'workload': 'BASELINE config 5 (secondary): PGD-Linf-3 eps 4/255 on the eval engine + '
                                                 'train-mode fwd/bwd + SGD-Nesterov/EMA, batch 256 per GPU',
                                     'global_batch': B * world, 'parallelism': 'dp%d' % world}})


def measure_secondary(args, device, B):
    """`secondary`: BASELINE configs 4 and 5 timed by the same process right after the headline (one rank, `--secondary-steps` steps each
    after one warm-up step): ResNet-50 adversarial training (cls_solver step, PGD-3 inner loop, bf16 as config 5 names it) and the ViT-B/16
    ImageNet-C sweep (15 corruptions x 5 severities generated on the GPU) on the bf16 engine and on the reference-precision engine."""
    import types
    a = types.SimpleNamespace(steps=max(1, args.secondary_steps), warmup=1, batch=B, one_stream=args.one_stream, no_other_engine=False)
    out = {'note': 'one warm-up + %d timed steps each, batch %d, run after the headline inside the same process' % (a.steps, B)}
    t0 = time.time()
    r = run_adv_train(a, device, 0, 1, None)
    out['adv_train'] = {k: r[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'dtype', 'final_loss', 'step_mfma', 'config')}
    r = run_vit_inc(a, device, 0, 1, None)
    out['vit_inc'] = {k: r[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'dtype', 'config')}
    rp = r['reference_precision']
    out['vit_inc_reference_precision'] = {k: rp[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'dtype', 'step_algorithmic')}
    out['seconds'] = time.time() - t0
    return out


def _diag(rank, **kv):
    """Per-rank launch diagnostics on STDERR (stdout carries the one JSON line): communicator creation, barrier, all-reduce and step
    times separately, so that the first real N-rank run is cheap to debug (exprs/exp/imagenet_c_loop_mini/eval.sh:21-23 launch shape)."""
    print('[bench rank %d] %s' % (rank, json.dumps(kv)), file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        # started as a plain process: become N ranks (the driver's own torchrun launch sets RANK and skips this)
        sys.exit(relaunch_multi_rank(args.gpus))
    if args.spawn_check:
        return spawn_check(args)
    if args.cpu_sweep:
        return cpu_sweep()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the hot path has no CPU fallback')
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but the launcher started %d rank(s)' % (args.gpus, world))
    if local >= torch.cuda.device_count():
        raise SystemExit('bench.py: rank %d has no GPU (%d visible)' % (local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    dist = None
    comm = {}
    if world > 1 or os.environ.get('RART_FORCE_DIST') == '1':     # (forced: a one-rank RCCL group, tests/test_rccl_gpu.py)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        t0 = time.perf_counter()
        dist.init_process_group('nccl', device_id=device)       # "nccl" == RCCL on ROCm
        comm['init_process_group_s'] = time.perf_counter() - t0
        # the first collective creates the RCCL communicator (ring / tree setup over xGMI): timed apart from every later one
        t0 = time.perf_counter()
        warm = torch.ones(1, device=device)
        dist.all_reduce(warm)
        torch.cuda.synchronize()
        comm['first_all_reduce_s'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        dist.barrier()
        torch.cuda.synchronize()
        comm['barrier_s'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        dist.all_reduce(warm)
        torch.cuda.synchronize()
        comm['all_reduce_s'] = time.perf_counter() - t0
        _diag(rank, phase='communicator', world=world, local_rank=local, device=torch.cuda.get_device_name(local),
              ipc_mode_legacy=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'), **comm)
    if args.workload in ('vit_inc', 'vit_pgd', 'adv_train'):
        out = {'vit_inc': run_vit_inc, 'vit_pgd': run_vit_pgd, 'adv_train': run_adv_train}[args.workload](args, device, rank, world, dist)
        if rank == 0:
            print(json.dumps(out))
        if dist is not None:
            dist.destroy_process_group()
        return
    B = args.batch
    images, labels, model = build_workload(B, device, rank)
    import copy
    model_cpu = copy.deepcopy(model) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    path = HipEngine(model, device, two_streams=not args.one_stream, precision=args.precision)

    for i in range(args.warmup):
        one_step(images, labels, path, i, rank, B)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    tb = time.perf_counter()
    barrier()
    t0 = time.perf_counter()
    acc = None
    for i in range(args.steps):
        acc = one_step(images, labels, path, args.warmup + i, rank, B)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        ta = time.perf_counter()
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_max = float(t.item())
        stats = torch.stack([acc[0], acc[1], torch.tensor(B, device=device), torch.tensor(rank + 1, device=device)]).to(torch.int64)
        dist.all_reduce(stats)                                   # the eval metric exchange + a rank checksum
        torch.cuda.synchronize()
        _diag(rank, phase='timed_region', steps=args.steps, opening_barrier_s=t0 - tb, own_steps_s=t_local, with_closing_barrier_s=dt,
              max_over_ranks_s=dt_max, metric_all_reduce_s=time.perf_counter() - ta, ms_per_step_own=t_local / args.steps * 1e3)
        dt = dt_max
    imgs_per_step = 6 * B * world
    value = imgs_per_step * args.steps / dt

    out = {
        'metric': 'corrupted+attacked images/sec/node (ResNet-50, PGD-7 + IN-C x5)',
        'value': value, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': 'per step and GPU: gaussian_noise sev 1..5 on 256 u8 224x224 images -> normalise -> '
                               'ResNet-50 eval (1280 img) + PGD-Linf-7 eps 2/255 ResNet-50 eval (256 img)',
                   'global_batch': B * world, 'images_per_step': imgs_per_step,
                   'model_path': path.name, 'parallelism': 'dp%d' % world,
                   'streams': 1 + len(path.sides), 'arithmetic': ARITHMETIC[args.precision]},
    }
    if dist is not None:
        # evidence in the driver's SCALE record that the collective really spanned N ranks: world size as torch.distributed saw
        # it, the all-reduced sum of (rank + 1) against N (N + 1) / 2, and the images the ranks reported together
        out['world_size_seen'] = dist.get_world_size()
        out['rccl_rank_sum'] = int(stats[3].item())
        out['rccl_rank_sum_expected'] = world * (world + 1) // 2
        out['images_per_step_all_ranks'] = int(stats[2].item()) * 6
        out['rank0_communicator'] = comm
    if rank == 0:
        step_flops = (5 + 15) * B * world * FLOP_FWD
        out['step_algorithmic'] = {'achieved': step_flops * args.steps / dt / 1e12, 'unit': 'TFLOP/s',
                                   'note': 'whole-step algorithmic FLOPs (20 forward-equivalents per image batch) / step time'
                                           + ('; against the %.1f TFLOP/s fp32 MFMA peak of the arithmetic the reference uses: %.2f x'
                                              % (MFMA_F32_PEAK / 1e12, step_flops * args.steps / dt / MFMA_F32_PEAK / world)
                                              if args.precision != 'bf16' else '')}
        if world == 1:
            out['roofline'] = measure_engine_roofline(path, images, labels)
            out['hbm_roofline_gaussian_noise'] = gaussian_noise_block(B, device)
            other = 'bf16' if args.precision != 'bf16' else 'bf16x3'
            if not args.no_other_engine:
                del path
                torch.cuda.empty_cache()
                out['fast_mode' if other == 'bf16' else 'reference_precision'] = measure_other_engine(
                    model, device, images, labels, rank, B, max(2, min(args.steps, 5)), not args.one_stream, other)
            if not args.no_secondary:
                one_step.extra.clear()
                torch.cuda.empty_cache()
                out['secondary'] = measure_secondary(args, device, B)
            if model_cpu is not None:
                out['cpu_baseline'] = cpu_baseline(args.cpu_sample, model_cpu)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
