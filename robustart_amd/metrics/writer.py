"""JSON-lines result files, one process per GPU.

Layout (exprs/nips_benchmark/batch_eval_transfer/parse_transfer.py:27-31, batch_eval_adv/eval.sh): a run writes
`<root>/<model>/<noise>_<eps>/results.txt.all` (clean: `none_0`; transfer: `<src>_To_<tgt>/<attack>_<eps>/...`).
Every rank appends its shard to `results.txt.rank<r>` as {"prediction": p, "label": y, "score": [...], "index": i};
rank 0 merges the shards ordered by the global sample index, so the file is identical for every world size.
"""
import json
import os


def result_dir(root, model_name, noise='none', eps='0', tgt_name=None):
    top = model_name if tgt_name is None else '%s_To_%s' % (model_name, tgt_name)
    return os.path.join(root, top, '%s_%s' % (noise, eps))


class ResultWriter(object):
    def __init__(self, save_dir, rank=0, world=1, with_score=True):
        self.dir, self.rank, self.world, self.with_score = save_dir, rank, world, with_score
        os.makedirs(save_dir, exist_ok=True)
        self.path = os.path.join(save_dir, 'results.txt.rank%d' % rank)
        self.f = open(self.path, 'w')

    def write_batch(self, logits, labels, indices):
        """logits: (b, classes) tensor on any device; labels: (b,) tensor; indices: global sample indices."""
        pred = logits.argmax(dim=1).tolist()
        lab = labels.tolist()
        sc = logits.float().cpu().tolist() if self.with_score else None
        for j in range(len(lab)):
            rec = {'prediction': int(pred[j]), 'label': int(lab[j])}
            if sc is not None:
                rec['score'] = sc[j]
            rec['index'] = int(indices[j])
            self.f.write(json.dumps(rec) + '\n')

    def close(self, barrier=None):
        """Flush this rank's shard; after `barrier()` rank 0 merges all shards into results.txt.all."""
        self.f.close()
        if barrier is not None:
            barrier()
        out = os.path.join(self.dir, 'results.txt.all')
        if self.rank == 0:
            recs = []
            for r in range(self.world):
                with open(os.path.join(self.dir, 'results.txt.rank%d' % r)) as f:
                    recs.extend(json.loads(ln) for ln in f)
            recs.sort(key=lambda x: x['index'])
            with open(out, 'w') as f:
                for rec in recs:
                    f.write(json.dumps(rec) + '\n')
        return out
