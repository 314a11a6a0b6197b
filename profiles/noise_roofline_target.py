"""Target for `rocprofv3 --kernel-trace --stats`: ONLY bench.py's gaussian_noise roofline measurement (B = 256, nine rotating buffer
pairs, 3 x 40 back-to-back launches + 40 bracketed ones), so that the per-kernel average of k_normal_noise_mfma in the trace can be
set beside the live figure of the bench line (inside a full bench.py trace the kernel's average also holds the launches of the timed
steps, which overlap the attack stream's kernels)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
avg, bracket, copy_s = bench.measure_gaussian_roofline(256, torch.device('cuda:0'))
ex = bench.measure_gaussian_roofline.extra          # round 4: the five-severity launch the workload issues + the two-stream variant
out = {'avg_launch_us': avg * 1e6, 'per_launch_event_bracket_us': bracket * 1e6, 'device_copy_us': copy_s * 1e6,
       'frac_of_8TBps': 2 * 256 * 150528 / avg / 8e12}
if 'multi_launch_s' in ex:
    out['five_severity_launch_us'] = ex['multi_launch_s'] * 1e6
    out['five_severity_throughput_equivalent_frac_NOT_a_roofline_fraction'] = 5 * 2 * 256 * 150528 / ex['multi_launch_s'] / 8e12
    out['five_severity_frac_of_bytes_moved'] = 6 * 256 * 150528 / ex['multi_launch_s'] / 8e12
    out['two_streams_launch_us'] = ex['two_streams_s'] * 1e6
print(json.dumps(out))
