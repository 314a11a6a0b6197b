"""Round 5: the 95 (corruption, severity) launches of the ImageNet-C generator dealt over N HIP streams (per-stream workspaces: _lib.workspace):
many of the kernels are one workgroup per image with a sequential chain inside (glass, spatter, plasma, jpeg, contrast, pixelate), i.e. 256 workgroups
of a few waves on 256 CUs -- what does co-scheduling independent corruptions buy?  ms per 256 source images for N = 1, 2, 3, 4, 6."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robustart_amd.noise import imagenet_c as C
B = 256
g = torch.Generator().manual_seed(11)
C.set_frost_textures(list(np.random.RandomState(0).randint(0, 256, (6, 300, 300, 3)).astype(np.uint8)))
jobs = [(cid, sev) for cid in range(len(C.CORRUPTION_NAMES)) for sev in range(1, 6)]
# longest first, so that the streams end together (times of the single-stream sweep)
try:
    t1 = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'profiles', 'r05_all_severities.json')))
    jobs.sort(key=lambda j: -t1[C.CORRUPTION_NAMES[j[0]]][j[1] - 1])
except Exception:
    pass
res = {}
for ns in (1, 2, 3, 4, 6):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    src = [torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8).cuda() for _ in range(ns)]
    dst = [[torch.empty_like(src[0]) for _ in range(2)] for _ in range(ns)]
    def one_pass(off):
        main = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(main)
        load = [0.0] * ns
        for k, (cid, sev) in enumerate(jobs):
            i = min(range(ns), key=lambda q: load[q])                    # greedy list scheduling on the single-stream times
            load[i] += t1[C.CORRUPTION_NAMES[cid]][sev - 1] if 't1' in globals() else 1.0
            with torch.cuda.stream(streams[i]):
                C.corrupt_batch_(src[i], cid, sev, seed=0, sample_offset=off, out=dst[i][k & 1])
        for s in streams:
            main.wait_stream(s)
    one_pass(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(3):
        one_pass((r + 1) * B)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    res[ns] = ms
    print('%d stream(s): %.2f ms per 256 source images x 95 = %.0f corrupted images/s' % (ns, ms, 95 * B / (ms * 1e-3)), flush=True)
    del src, dst, streams
    torch.cuda.empty_cache()
json.dump(res, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'gpurun_out', 'r05_sweep_streams.json'), 'w'), indent=1)
