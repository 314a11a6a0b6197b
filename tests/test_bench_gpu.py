"""The driver's contract for bench.py, checked on the GPU: ONE JSON line on stdout with the metric / timing fields led by the
tolerance-meeting path (dtype bf16x3: the reference-precision engine), the `roofline` object (the kernel family with the largest share of
a gradient evaluation of THAT engine, priced against its binding roof) and, on the default run, `fast_mode` (the bf16 engine with its
stated error), `secondary` (BASELINE configs 4 and 5, driver-timed) and `cpu_baseline`; here short runs.  The secondary workloads also run
two steps each as stand-alone workloads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + extra, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # exactly one line on stdout
    return json.loads(lines[0])


def _check_roofline(rf):
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert (rf['bound'], rf['unit']) in (('mfma', 'TFLOP/s'), ('hbm', 'GB/s'))
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9 and 0 < rf['frac'] < 1
    assert 0 < rf['share_of_gradient_evaluation'] <= 1 and rf['launches'] > 0 and 'kernel' in rf
    # the dominant family really is the largest: no other family has a larger share
    assert all(o['share_of_gradient_evaluation'] <= rf['share_of_gradient_evaluation'] + 1e-9 for o in rf['other_mfma_kernels'].values())
    if rf['bound'] == 'hbm':
        assert rf['algorithmic_bytes_per_launch'] > 0 and 0 < rf['mfma']['frac'] < 1


def test_bench_leads_with_the_reference_precision_engine_and_carries_the_contract_fields():
    d = _run(['--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-secondary'])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['unit'] == 'images/s' and d['value'] > 0 and d['vs_baseline'] is None and d['data'] == 'synthetic'
    # the top-level number is the tolerance-meeting path
    assert d['dtype'] == 'bf16x3' and d['config']['model_path'] == 'hip-igemm-bf16x3' and '1e-4' in d['config']['arithmetic']
    assert 'workload' in d['config'] and 'model' not in d['config']
    # value = images of the step / time: 6 x 256 images per step
    assert abs(d['value'] - d['config']['images_per_step'] / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    rf = d['roofline']
    _check_roofline(rf)
    assert 'pair' in rf['kernel'] and rf['achieved_fp32_equivalent'] * 3 == pytest.approx(rf['mfma']['achieved'] if rf['bound'] == 'hbm' else rf['achieved'])
    # the bf16 engine follows as a fast mode with its error stated, slower-is-impossible sanity: it is the faster one
    fm = d['fast_mode']
    assert fm['dtype'] == 'bf16' and fm['value'] > d['value'] and 'outside the north star' in fm['arithmetic']
    _check_roofline(fm['roofline'])
    # gaussian_noise: the block's frac is a physical roofline fraction (bytes moved / time / peak) ...
    h = d['hbm_roofline_gaussian_noise']
    assert h['bound'] == 'hbm' and h['unit'] == 'GB/s' and abs(h['frac'] - h['achieved'] / h['peak']) < 1e-9 and 0.2 < h['frac'] < 0.8
    assert h['bytes_moved_per_launch'] == 6 * 256 * 150528 == h['algorithmic_bytes_per_launch']
    assert abs(h['achieved'] * 1e9 - h['bytes_moved_per_launch'] / (h['avg_launch_us'] * 1e-6)) <= 1e-6 * h['achieved'] * 1e9
    # ... and the per-launch-equivalent figure is named as what it is (it can exceed a physical roof)
    le = h['launch_equivalent']
    assert le['single_severity_bytes_x5'] == 10 * 256 * 150528 and le['throughput_equivalent_frac'] > h['frac'] and 'NOT a roofline' in le['note']
    assert 'frac' not in le
    one = h['single_severity_launch_one_stream']
    assert one['bound'] == 'hbm' and 0.2 < one['frac'] < 1 and 0.2 < h['single_severity_launches_two_streams']['frac'] < 1


def test_bench_secondary_block_times_configs_4_and_5():
    """the default run appends `secondary` -- adv_train (config 5) and vit_inc on both engines (config 4) -- so the
    driver's own BENCH record carries them.  Here at batch 32, two steps."""
    d = _run(['--steps', '2', '--warmup', '1', '--batch', '32', '--no-cpu-baseline', '--no-fast-mode', '--secondary-steps', '2'])
    sec = d['secondary']
    for k in ('adv_train', 'vit_inc', 'vit_inc_reference_precision'):
        assert sec[k]['value'] > 0 and sec[k]['unit'] == 'images/s' and sec[k]['steps'] == 2, k
    assert sec['adv_train']['dtype'] == 'bf16' and sec['adv_train']['final_loss'] > 0
    assert sec['vit_inc']['config']['corruptions'] == 15 and sec['vit_inc_reference_precision']['dtype'] == 'bf16x3'
    assert sec['vit_inc']['value'] > sec['vit_inc_reference_precision']['value']


def test_bench_fast_engine_can_still_be_the_timed_one():
    d = _run(['--steps', '2', '--warmup', '1', '--batch', '64', '--precision', 'bf16', '--no-cpu-baseline', '--no-secondary', '--no-reference-precision'])
    assert d['dtype'] == 'bf16' and 'fast_mode' not in d and 'reference_precision' not in d
    _check_roofline(d['roofline'])


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
