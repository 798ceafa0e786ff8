"""gcc/utils/misc.py of the reference, verbatim semantics (host-side, trivial)."""
import numpy as np


def warmup_linear(x, warmup=0.002):
    """misc.py:5-10: triangular schedule, peak at ``warmup``, zero at 1."""
    if x < warmup:
        return x / warmup
    return max((x - 1.0) / (warmup - 1.0), 0)


def adjust_learning_rate(epoch, opt, optimizer):
    """misc.py:13-19."""
    steps = np.sum(epoch > np.asarray(opt.lr_decay_epochs))
    if steps > 0:
        new_lr = opt.learning_rate * (opt.lr_decay_rate ** steps)
        for param_group in optimizer.param_groups:
            param_group["lr"] = new_lr


class AverageMeter(object):
    """misc.py:22-42."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
