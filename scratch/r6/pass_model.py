"""From a trace_list.py listing: per dispatch, workgroups per XCD (workgroup i -> XCD i % 8), the resident workgroups per CU that LDS / VGPRs /
threads allow, and the launch's pass efficiency  (wgs / 8) / (ceil(wgs / 8 / (32 resident)) * 32 resident).
    python scratch/r6/pass_model.py <listing.txt>"""
import math, re, sys
tot = 0.0; lost = 0.0
for l in open(sys.argv[1]):
    m = re.match(r'(.{70})\s+wgs\s+(\d+) x\s+(\d+) thr\s+lds\s+(\d+) vgpr\s+(\d+)\s+([\d.]+) us', l)
    if not m: continue
    name, wgs, thr, lds, vg, us = m.group(1).strip(), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), float(m.group(6))
    waves = (thr + 63) // 64
    vg_alloc = max(8, (vg + 7) // 8 * 8)
    by_v = (512 // vg_alloc) * 4 // waves if waves <= (512 // vg_alloc) * 4 else 0
    by_l = (160 * 1024) // lds if lds else 99
    by_w = 32 // waves                      # 8 waves per SIMD x 4
    res = max(1, min(by_v, by_l, by_w))
    per_xcd = math.ceil(wgs / 8)
    passes = per_xcd / (32 * res)
    eff = passes / math.ceil(passes)
    tot += us; lost += us * (1 - eff)
    if us > 20 and eff < 0.93:
        print('%-60s wgs %6d  resident/CU %2d  passes %6.2f  eff %.2f  %8.1f us' % (name[:60], wgs, res, passes, eff, us))
print('kernel time %.1f us, time in the empty part of last passes (upper bound) %.1f us = %.1f %%' % (tot, lost, 100 * lost / max(tot, 1)))
