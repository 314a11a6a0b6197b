"""Evaluators over `results.txt.all` files (see package docstring for the reference anchors)."""
import json

import numpy as np


def parse_line(line):
    """First two `: value` fields of a result line, as strings (AR_evaluator.py:13-21: scan for ':' and cut the text
    from two characters later up to the next ',' or '}')."""
    res = []
    for i in range(len(line)):
        if line[i] == ':':
            for j in range(i + 2, len(line)):
                if line[j] == ',' or line[j] == '}':
                    res.append(line[i + 2:j])
                    break
    return res[0], res[1]


def _lines(path):
    with open(path) as f:
        return f.readlines()


class Metric(object):
    """RobustART/metrics/base_evaluator.py:7-24."""

    def __init__(self, metric_dict=None):
        self.metric = dict(metric_dict or {})
        self.cmp_key, self.v = None, None

    def __repr__(self):
        return 'metric=%s key=%s' % (self.metric, self.cmp_key)

    __str__ = __repr__

    def update(self, up_dict=None):
        self.metric.update(up_dict or {})

    def set_cmp_key(self, key):
        self.cmp_key = key
        self.v = self.metric[key]


def _load_res(res_file):
    res = {}
    for line in _lines(res_file):
        info = json.loads(line)
        for k, v in info.items():
            res.setdefault(k, []).append(v)
    return res


def _topk_acc(score, label, topk):
    """imagenetc_evaluator.py:55-66: torch.topk(maxk, largest, sorted) on the scores, compare with the label."""
    score = np.asarray(score, dtype=np.float64)
    label = np.asarray(label).reshape(-1)
    maxk = max(topk)
    # stable descending order (ties resolve to the lower class index, as torch.topk does for sorted output on CPU)
    order = np.argsort(-score, axis=1, kind='stable')[:, :maxk]
    correct = order == label[:, None]
    return {k: float(correct[:, :k].any(axis=1).sum() * (100.0 / len(label))) for k in topk}


class ImageNetCEvaluator(object):
    def __init__(self, topk=(1, 5)):
        self.topk = list(topk)

    def load_res(self, res_file):
        return _load_res(res_file)

    def eval(self, res_file):
        res = self.load_res(res_file)
        acc = _topk_acc(res['score'], res['label'], self.topk)
        metric = Metric({'top%d' % k: acc[k] for k in self.topk})
        metric.set_cmp_key('top%d' % self.topk[0])
        with open(res_file.replace('results.txt.all', 'metric'), 'w') as f:
            json.dump(metric.metric, f)
        return metric


class ImageNetSEvaluator(object):
    def __init__(self):
        self.metric = Metric()

    def load_res(self, res_file):
        return _load_res(res_file)

    def eval(self, res_file, decoder_type='pil', resize_type='pil-bilinear'):
        res = self.load_res(res_file)
        acc = _topk_acc(res['score'], res['label'], [1])[1]
        out = {(decoder_type, resize_type): acc}
        self.metric.update(out)
        return out

    def get_mean(self):
        return {'Mean': float(np.mean(list(self.metric.metric.values())))}

    def get_std(self):
        return {'Std.': float(np.std(list(self.metric.metric.values())))}

    def clear(self):
        self.metric.metric = {}


class AdvRobustEvaluator(object):
    parse_line = staticmethod(parse_line)

    def eval(self, clean_path, adv_path, num=None):
        lines_clean, lines_att = _lines(clean_path), _lines(adv_path)
        n = num if num is not None else len(lines_clean)
        before = after = 0
        for i in range(n):
            a, b = parse_line(lines_clean[i])
            if a == b:
                before += 1
                c, d = parse_line(lines_att[i])
                if c == d:
                    after += 1
        ar = after / before * 100
        print('Clean Acc: {}, Adversarial Robustness: {}'.format(before / n * 100, ar))
        return ar


class WorstCaseAdvRobustEvaluator(object):
    parse_line = staticmethod(parse_line)

    def eval(self, clean_path, multi_adv_result_paths, num=None):
        lines_clean = _lines(clean_path)
        atts = [_lines(p) for p in multi_adv_result_paths]
        n = num if num is not None else len(lines_clean)
        before = after = 0
        for i in range(n):
            a, b = parse_line(lines_clean[i])
            if a == b:
                before += 1
                ok = 1
                for la in atts:
                    c, d = parse_line(la[i])
                    if c != d:
                        ok = 0
                after += ok
        wcar = after / before * 100
        print('Worst-Case Adversarial Robustness: {}'.format(wcar))
        return wcar


def transfer_rate(src_clean_path, tgt_clean_path, transfer_path):
    """parse_transfer.py:33-42: among samples both the source and the target model classify correctly, the fraction
    the transferred adversarial examples make the TARGET model misclassify."""
    ls, lt, lx = _lines(src_clean_path), _lines(tgt_clean_path), _lines(transfer_path)
    if not (len(ls) == len(lt) == len(lx)):
        raise ValueError('transfer_rate: result files differ in length')
    before = after = 0
    for i in range(len(ls)):
        a, b = parse_line(ls[i])
        c, d = parse_line(lt[i])
        if a == b and c == d:
            before += 1
            e, f = parse_line(lx[i])
            if e != f:
                after += 1
    return after / before
