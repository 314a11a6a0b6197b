"""The driver's contract for bench.py, checked on the GPU: ONE JSON line on stdout with the metric / timing fields, the `roofline` object
(the kernel family with the largest share of a gradient evaluation, priced against its binding roof) and, on the default run, `cpu_baseline`
and `reference_precision`; here a short run without the two slow legs.  The secondary workloads (BASELINE configs 4 and 5) run two steps
each so that the driver's test pass exercises them as workloads."""
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
    assert (rf['bound'], rf['unit']) in (('mfma', 'TFLOP/s'), ('hbm', 'GB/s'))
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9 and 0 < rf['frac'] < 1
    assert 0 < rf['share_of_gradient_evaluation'] <= 1 and rf['launches'] > 0 and 'kernel' in rf
    # the dominant family really is the largest: no other family has a larger share
    assert all(o['share_of_gradient_evaluation'] <= rf['share_of_gradient_evaluation'] + 1e-9 for o in rf['other_mfma_kernels'].values())
    if rf['bound'] == 'hbm':
        assert rf['algorithmic_bytes_per_launch'] > 0 and 0 < rf['mfma']['frac'] < 1
    h = d['hbm_roofline_gaussian_noise']
    assert h['bound'] == 'hbm' and h['unit'] == 'GB/s' and abs(h['frac'] - h['achieved'] / h['peak']) < 1e-9 and 0.2 < h['frac'] < 1
    assert abs(h['achieved'] * 1e9 - h['algorithmic_bytes_per_launch'] / (h['avg_launch_us'] * 1e-6)) <= 1e-6 * h['achieved'] * 1e9
    # the block's headline is the five-severity launch the workload issues, per launch-equivalent; the single-severity figures sit beside it
    assert h['bytes_moved_per_launch'] == 6 * 256 * 150528 and h['algorithmic_bytes_per_launch'] == 10 * 256 * 150528
    assert 0.2 < h['against_bytes_moved']['frac'] < h['frac']
    one = h['single_severity_launch_one_stream']
    assert one['bound'] == 'hbm' and 0.2 < one['frac'] < 1 and 0.2 < h['single_severity_launches_two_streams']['frac'] < 1


@pytest.mark.parametrize('workload,extra', [('vit_inc', ['--no-reference-precision']), ('adv_train', []), ('vit_pgd', [])])
def test_secondary_workloads_run_and_print_one_line(workload, extra):
    """BASELINE configs 4 (ViT-B/16 x all 15 ImageNet-C corruptions, on-GPU noise) and 5 (ResNet-50 adversarial training) and the ViT
    PGD evaluation as bench workloads, two steps each at a reduced batch."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', workload, '--steps', '2', '--warmup', '1', '--batch', '32']
                       + extra, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['value'] > 0 and d['unit'] == 'images/s' and d['steps'] == 2 and d['n_gpus'] == 1 and 'workload' in d['config']
    if workload == 'vit_inc':
        assert d['config']['corruptions'] == 15            # frost included (synthetic textures)
