"""Run statistics with the report format of the reference's plnlp/logger.py (its text is the
contract, fixture G9): per run the test score at the best-validation evaluation point; over runs
the mean and (unbiased) standard deviation of those picks."""
import sys
from typing import List, Sequence, Tuple

import torch

_RUN_LINES = ("Run {run:02d}:", "Highest Valid: {valid:.2f}", "Highest Eval Point: {point}", "   Final Test: {test:.2f}")
_ALL_LINES = ("All runs:", "Highest Valid: {vm:.2f}  {vs:.2f}", "   Final Test: {tm:.2f}  {ts:.2f}")


def _pick(history: Sequence[Tuple[float, float]], last_best: bool):
    """(best validation score, its 0-based evaluation point, the test score there), in percent"""
    scores = torch.tensor(history, dtype=torch.get_default_dtype()) * 100
    valid = scores[:, 0]
    at = int(valid.argmax())
    if last_best:                       # the LAST evaluation point that reaches the maximum
        at = valid.numel() - 1 - int(valid.flip(0).argmax())
    return valid.max(), at, scores[at, 1]


class Logger(object):
    def __init__(self, runs, info=None):
        self.info = info
        self.results: List[list] = [[] for _ in range(runs)]

    def add_result(self, run, result):
        if len(result) != 2:
            raise ValueError("result must be (valid, test)")
        if not 0 <= run < len(self.results):
            raise IndexError(run)
        self.results[run].append(result)

    def print_statistics(self, run=None, f=sys.stdout, last_best=False):
        if run is None:
            picks = torch.stack([torch.stack([p[0], p[2]]) for p in (_pick(h, last_best) for h in self.results)])
            fields = dict(vm=picks[:, 0].mean(), vs=picks[:, 0].std(), tm=picks[:, 1].mean(), ts=picks[:, 1].std())
            text = [line.format(**fields) for line in _ALL_LINES]
        else:
            valid, at, test = _pick(self.results[run], last_best)
            text = [line.format(run=run + 1, valid=valid, point=at + 1, test=test) for line in _RUN_LINES]
        for line in text:
            print(line, file=f)
