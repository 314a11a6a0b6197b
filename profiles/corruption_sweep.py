"""Launch every ImageNet-C corruption (19, severity 3) on B=256 uint8 224x224 images, rotating over buffer pairs whose
total footprint exceeds the 256 MiB Infinity Cache, so that a rocprofv3 --kernel-trace / --pmc pass of this command
gives per-kernel time and HBM bytes at BASELINE's batch size.

    rocprofv3 --kernel-trace --stats -d gpurun_out/corr_kt -o sweep -- python profiles/corruption_sweep.py
    python profiles/corruption_sweep.py --events      # in-process event timing per corruption (whole launch sequence)
"""
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from robustart_amd.noise import imagenet_c as C

B, NP, REPS = 256, 8, 4
g = torch.Generator().manual_seed(11)
src = [torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8).cuda() for _ in range(NP)]
dst = [torch.empty_like(s) for s in src]
tex = np.random.RandomState(0).randint(0, 256, (6, 300, 300, 3)).astype(np.uint8)
C.set_frost_textures(list(tex))
names = C.CORRUPTION_NAMES
only = [a for a in sys.argv[1:] if not a.startswith('--')]
events = '--events' in sys.argv
rows = []
for cid, nm in enumerate(names):
    if only and nm not in only:
        continue
    sev = 3 if nm != 'spatter' else int(os.environ.get('RART_SPATTER_SEV', '4'))
    try:
        C.corrupt_batch_(src[0], cid, sev, seed=0, sample_offset=0, out=dst[0])      # warm (tables, workspace)
        torch.cuda.synchronize()
        ev = []
        for r in range(REPS):
            i = (r + 1) % NP
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            C.corrupt_batch_(src[i], cid, sev, seed=0, sample_offset=r * B, out=dst[i])
            e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        med = us[len(us) // 2]
        rows.append({'corruption': nm, 'severity': sev, 'us_per_batch': med, 'algorithmic_GBps': 2 * B * 150528 / med / 1e3,
                     'frac_hbm_8TBps': 2 * B * 150528 / (med * 1e-6) / 8e12})
        print('%-18s sev %d  %10.1f us/batch  %8.1f GB/s algorithmic (%.3f of 8 TB/s)' %
              (nm, sev, med, rows[-1]['algorithmic_GBps'], rows[-1]['frac_hbm_8TBps']), flush=True)
    except Exception as e:      # noqa: BLE001
        print('%-18s FAILED: %s' % (nm, e), flush=True)
if events:
    json.dump(rows, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'corruption_events.json'), 'w'), indent=1)
