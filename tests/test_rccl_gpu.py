"""RCCL on device memory before the driver's 8-GPU run: `dist.init_process_group('nccl')`, the barrier, the
metric all-reduce of bench.py and the bucketed gradient all_reduce on flat-arena slices of cls_solver all execute on a ONE-rank
process group under torch.distributed.run (RART_FORCE_DIST=1 turns the exchange on although there is nothing to exchange with), so
communicator creation and every collective call of the N-rank path have run on this box's GPU.  The N > 1 arithmetic (sharding,
averaging, replicas bit-identical) is covered on CPU by tests/test_distributed_cpu.py (gloo, world size 2).
Reference launch shape: exprs/exp/imagenet_c_loop_mini/eval.sh:21-23 (torchrun, one process per GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(args, timeout=900):
    env = dict(os.environ, RART_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_port())] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_bench_headline_under_a_one_rank_rccl_group():
    r = _torchrun([os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '64', '--no-cpu-baseline',
                   '--no-fast-mode', '--no-secondary'])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    # the process group existed and the all-reduce over it returned what one rank contributes
    assert d['world_size_seen'] == 1 and d['rccl_rank_sum'] == 1 == d['rccl_rank_sum_expected']
    assert d['images_per_step_all_ranks'] == 6 * 64 and d['n_gpus'] == 1 and d['value'] > 0 and d['dtype'] == 'bf16x3'
    # per-rank diagnostics on stderr -- communicator creation, barrier and all-reduce apart from the step time
    diag = {}
    for ln in r.stderr.splitlines():
        if ln.startswith('[bench rank 0] '):
            q = json.loads(ln[len('[bench rank 0] '):])
            diag[q['phase']] = q
    c, t = diag['communicator'], diag['timed_region']
    for k in ('init_process_group_s', 'first_all_reduce_s', 'barrier_s', 'all_reduce_s'):
        assert c[k] >= 0 and d['rank0_communicator'][k] == c[k], k
    assert c['world'] == 1 and c['ipc_mode_legacy'] == '0'
    for k in ('opening_barrier_s', 'own_steps_s', 'with_closing_barrier_s', 'max_over_ranks_s', 'metric_all_reduce_s'):
        assert t[k] >= 0, k
    assert t['own_steps_s'] <= t['with_closing_barrier_s'] <= t['max_over_ranks_s'] + 1e-9


def test_bench_adv_train_workload_under_a_one_rank_rccl_group():
    """`bench.py --workload adv_train` (BASELINE config 5) prints the same per-rank diagnostics as the headline workload --
    communicator creation, barriers, step time -- plus the gradient exchange of the last step, so the first real 8-GPU run of config 5 is
    as cheap to read as config 3's."""
    r = _torchrun([os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--workload', 'adv_train', '--steps', '2', '--warmup', '1', '--batch', '32',
                   '--no-cpu-baseline'])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['dtype'] == 'bf16' and d['final_loss'] == d['final_loss']
    diag = {}
    for ln in r.stderr.splitlines():
        if ln.startswith('[bench rank 0] '):
            q = json.loads(ln[len('[bench rank 0] '):])
            diag[q['phase']] = q
    c, t, gx = diag['communicator'], diag['timed_region'], diag['grad_exchange']
    for k in ('init_process_group_s', 'first_all_reduce_s', 'barrier_s', 'all_reduce_s'):
        assert c[k] >= 0, k
    assert c['world'] == 1 and c['ipc_mode_legacy'] == '0'
    for k in ('opening_barrier_s', 'own_steps_s', 'with_closing_barrier_s', 'max_over_ranks_s'):
        assert t[k] >= 0, k
    assert t['own_steps_s'] <= t['with_closing_barrier_s'] <= t['max_over_ranks_s'] + 1e-9
    assert gx['buckets'] >= 2 and gx['bytes'] == 4 * 25557032 and 0 <= gx['launched_during_backward'] <= gx['buckets']
    assert gx['stream_wait_s'] >= 0 and gx['host_wait_s'] >= 0


def test_cls_solver_adv_train_steps_under_a_one_rank_rccl_group(tmp_path):
    """Two adversarial-training iterations of cls_solver (HIP train engine, bucketed all_reduce of the gradient arena overlapped with
    backward, optimizer + EMA kernels) and a clean evaluation (metric all-reduce) in a one-rank RCCL group."""
    cfg = tmp_path / 'cfg.yaml'
    cfg.write_text('''
model: {type: resnet50_official, kwargs: {num_classes: 1000}}
data: {read_from: fake, fake_size: 64, batch_size: 16, input_size: 64}
optimizer: {type: SGD, kwargs: {nesterov: true, momentum: 0.9, weight_decay: 0.0001}}
lr_scheduler: {type: CosineEpoch, kwargs: {base_lr: 0.01, warmup_lr: 0.02, warmup_steps: 1, max_iter: 2}}
label_smooth: 0.1
ema: {enable: true, kwargs: {decay: 0.999}}
adv_train: {eps: 4/255, steps: 1, rel_stepsize: 1.0}
dist: {sync: false, bucket_mb: 8}
saver: {print_freq: 1}
''')
    r = _torchrun(['-m', 'robustart_amd.train.cls_solver', '--config', str(cfg), '--max-iter', '2'])
    assert r.returncode == 0, r.stderr[-3000:]
    recs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip().startswith('{"iter"')]
    assert [q['iter'] for q in recs] == [0, 1] and all(q['loss'] == q['loss'] and q['loss'] > 0 for q in recs)
    # bucket count / bytes / wait time of the gradient exchange, per step, in the record and per rank on stderr
    for q in recs:
        gx = q['grad_exchange']
        assert gx['buckets'] >= 2 and gx['bytes'] == 4 * 25557032 and 0 <= gx['launched_during_backward'] <= gx['buckets']
        assert gx['stream_wait_s'] >= 0 and gx['host_wait_s'] >= 0
    assert sum(1 for ln in r.stderr.splitlines() if ln.startswith('[cls_solver rank 0] ')) == 2
    r = _torchrun(['-m', 'robustart_amd.train.cls_solver', '--config', str(cfg), '--evaluate'])
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip().startswith('{')]
    assert res and res[-1]['count'] == 64 and res[-1]['world_size'] == 1 and 0 <= res[-1]['top1'] <= 1
