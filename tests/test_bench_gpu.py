"""The driver's contract for bench.py, checked on the GPU: ONE JSON line on stdout with the metric / timing fields, the `roofline` object
(dominant MFMA kernel) and, on the default run, `cpu_baseline` and `reference_precision`; here a short run without the two slow legs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
                        '--no-reference-precision'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # exactly one line on stdout
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['unit'] == 'images/s' and d['value'] > 0 and d['vs_baseline'] is None and d['data'] == 'synthetic' and d['dtype'] == 'bf16'
    assert 'workload' in d['config'] and 'model' not in d['config']
    # value = images of the step / time: 6 x 256 images per step
    assert abs(d['value'] - d['config']['images_per_step'] / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    rf = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9 and 0 < rf['frac'] < 1
    h = d['hbm_roofline_gaussian_noise']
    assert h['bound'] == 'hbm' and h['unit'] == 'GB/s' and abs(h['frac'] - h['achieved'] / h['peak']) < 1e-9 and 0.2 < h['frac'] < 1
    assert abs(h['achieved'] * 1e9 - h['algorithmic_bytes_per_launch'] / (h['avg_launch_us'] * 1e-6)) <= 1e-6 * h['achieved'] * 1e9
