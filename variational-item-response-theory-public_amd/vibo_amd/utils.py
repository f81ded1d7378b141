"""Meters and checkpoint writer (reference: src/utils.py:7-31)."""
import os
import shutil

import torch


class AverageMeter:
    """Running weighted mean; .val last value, .avg mean, .sum, .count."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def save_checkpoint(state, is_best, folder='./', filename='checkpoint.pth.tar'):
    """folder/checkpoint.pth.tar, copied to folder/model_best.pth.tar when is_best."""
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, filename)
    torch.save(state, path)
    if is_best:
        shutil.copyfile(path, os.path.join(folder, 'model_best.pth.tar'))
