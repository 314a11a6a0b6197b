"""Result writer and robustness metrics: the step immediately after the hot path.

Mirrors RobustART/metrics/ (names, arguments, printed quantities):
    ImageNetCEvaluator   RobustART/metrics/imagenetc_evaluator.py:26-75    top-k from `score` / `label` lines
    ImageNetSEvaluator   RobustART/metrics/imagenets_evaluator.py:9-66     per (decoder, resize) top-1, mean / std
    AdvRobustEvaluator   RobustART/metrics/AR_evaluator.py:9-39            AR = survived / clean-correct * 100
    WorstCaseAdvRobustEvaluator  RobustART/metrics/WCAR_evaluator.py:9-44  WCAR = survived ALL attacks / clean-correct
    transfer_rate        exprs/nips_benchmark/batch_eval_transfer/parse_transfer.py:10-46
and writes the JSON-lines `results.txt.all` files they read (one line per sample, `prediction` first, `label`
second: AR/WCAR/transfer parse the first two values of a line).

Departures from the reference as written (each one is a defect there):
  * AR / WCAR define `parse_line(line)` without `self` and call it as a method, so `eval` raises TypeError; here it
    is a staticmethod with the same parsing;
  * the sample count is hard-coded to 50000 there; here it is the length of the clean file (`num=` overrides);
  * ImageNetSEvaluator keys its result dict with a list (unhashable) and iterates `for key, item in dict`; here the
    key is the tuple (decoder_type, resize_type) and mean / std run over the values.
"""
from .evaluators import (AdvRobustEvaluator, ImageNetCEvaluator, ImageNetSEvaluator,   # noqa: F401
                         WorstCaseAdvRobustEvaluator, parse_line, transfer_rate)
from .writer import ResultWriter, result_dir                                          # noqa: F401
