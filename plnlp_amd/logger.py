"""Run statistics, same report as the reference's plnlp/logger.py: per run the
test score at the best-validation evaluation point, over runs mean and std."""
import sys

import torch


class Logger(object):
    def __init__(self, runs, info=None):
        self.info = info
        self.results = [[] for _ in range(runs)]

    def add_result(self, run, result):
        if len(result) != 2:
            raise ValueError("result must be (valid, test)")
        if not 0 <= run < len(self.results):
            raise IndexError(run)
        self.results[run].append(result)

    @staticmethod
    def _best_index(valid: torch.Tensor, last_best: bool) -> int:
        if last_best:  # last occurrence of the maximum
            return valid.numel() - 1 - int(valid.flip(dims=[0]).argmax())
        return int(valid.argmax())

    def print_statistics(self, run=None, f=sys.stdout, last_best=False):
        if run is not None:
            table = 100 * torch.tensor(self.results[run])
            best = self._best_index(table[:, 0], last_best)
            print(f'Run {run + 1:02d}:', file=f)
            print(f'Highest Valid: {table[:, 0].max():.2f}', file=f)
            print(f'Highest Eval Point: {best + 1}', file=f)
            print(f'   Final Test: {table[best, 1]:.2f}', file=f)
            return
        picks = []
        for table in 100 * torch.tensor(self.results):
            best = self._best_index(table[:, 0], last_best)
            picks.append((table[:, 0].max().item(), table[best, 1].item()))
        picks = torch.tensor(picks)
        print('All runs:', file=f)
        print(f'Highest Valid: {picks[:, 0].mean():.2f}  {picks[:, 0].std():.2f}', file=f)
        print(f'   Final Test: {picks[:, 1].mean():.2f}  {picks[:, 1].std():.2f}', file=f)
