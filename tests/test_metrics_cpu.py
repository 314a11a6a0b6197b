"""robustart_amd.metrics against golden vectors produced by running the reference's evaluators
(tests/golden/make_metrics_golden.py), plus the sharded result writer."""
import json
import os

import numpy as np
import torch

from robustart_amd import metrics as M

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'metrics_ref.json')))


def _write(tmp_path, key):
    d = tmp_path / key
    d.mkdir()
    p = d / 'results.txt.all'
    p.write_text('\n'.join(GOLD['files'][key]) + '\n')
    return str(p)


def test_parse_line_matches_reference():
    for k, lines in GOLD['files'].items():
        assert [list(M.parse_line(ln)) for ln in lines] == GOLD['parsed'][k]
        assert [list(M.AdvRobustEvaluator.parse_line(ln)) for ln in lines] == GOLD['parsed'][k]


def test_topk_matches_reference(tmp_path):
    for k in GOLD['files']:
        p = _write(tmp_path, k)
        m = M.ImageNetCEvaluator(topk=[1, 5]).eval(p)
        assert abs(m.metric['top1'] - GOLD['topk'][k]['top1']) < 1e-9
        assert abs(m.metric['top5'] - GOLD['topk'][k]['top5']) < 1e-9
        assert m.cmp_key == 'top1' and m.v == m.metric['top1']
        assert json.load(open(p.replace('results.txt.all', 'metric'))) == m.metric


def test_ar_wcar_transfer(tmp_path, capsys):
    paths = {k: _write(tmp_path, k) for k in GOLD['files']}
    for a in ('adv1', 'adv2'):
        assert abs(M.AdvRobustEvaluator().eval(paths['clean_a'], paths[a]) - GOLD['AR'][a]) < 1e-9
    assert 'Adversarial Robustness' in capsys.readouterr().out
    w = M.WorstCaseAdvRobustEvaluator().eval(paths['clean_a'], [paths['adv1'], paths['adv2']])
    assert abs(w - GOLD['WCAR']) < 1e-9
    assert w <= min(GOLD['AR'].values()) + 1e-9                      # worst case never exceeds a single attack
    assert abs(M.transfer_rate(paths['clean_a'], paths['clean_b'], paths['trans']) - GOLD['transfer']) < 1e-12


def test_imagenet_s_evaluator_mean_std(tmp_path):
    ev = M.ImageNetSEvaluator()
    accs = []
    for k, (dec, rs) in zip(('clean_a', 'clean_b', 'adv1'), (('pil', 'pil-bilinear'), ('pil', 'pil-nearest'),
                                                             ('opencv', 'opencv-area'))):
        out = ev.eval(_write(tmp_path, k), dec, rs)
        assert list(out.keys()) == [(dec, rs)]
        accs.append(out[(dec, rs)])
        assert abs(accs[-1] - GOLD['topk'][k]['top1']) < 1e-9
    assert abs(ev.get_mean()['Mean'] - np.mean(accs)) < 1e-12 and abs(ev.get_std()['Std.'] - np.std(accs)) < 1e-12
    ev.clear()
    assert ev.metric.metric == {}


def test_result_writer_merges_shards_in_global_order(tmp_path):
    d = M.result_dir(str(tmp_path), 'resnet50', 'pgd_linf', '0.031')
    assert d.endswith(os.path.join('resnet50', 'pgd_linf_0.031'))
    assert M.result_dir('r', 'A', 'fgsm', '0.031', tgt_name='B') == os.path.join('r', 'A_To_B', 'fgsm_0.031')
    torch.manual_seed(0)
    logits = torch.randn(10, 7)
    labels = torch.randint(0, 7, (10,))
    # two "ranks" with contiguous shards, written out of order
    w1 = M.ResultWriter(d, rank=1, world=2)
    w1.write_batch(logits[5:], labels[5:], list(range(5, 10)))
    w1.close()
    w0 = M.ResultWriter(d, rank=0, world=2)
    w0.write_batch(logits[:5], labels[:5], list(range(0, 5)))
    out = w0.close()
    recs = [json.loads(ln) for ln in open(out)]
    assert [r['index'] for r in recs] == list(range(10))
    assert [r['prediction'] for r in recs] == logits.argmax(1).tolist()
    assert [r['label'] for r in recs] == labels.tolist()
    assert list(recs[0].keys())[:2] == ['prediction', 'label']       # AR / WCAR read the first two fields
    m = M.ImageNetCEvaluator(topk=[1]).eval(out)
    assert abs(m.metric['top1'] - (logits.argmax(1) == labels).float().mean().item() * 100) < 1e-5
