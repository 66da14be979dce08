"""Counterpart of the reference's `util_modules.py`: positional encoding module on the HIP
encoders, the two losses and the learning-rate schedule (same names and constructor
arguments, util_modules.py:7-93)."""
import numpy as np
import torch
import torch.nn as nn

from . import ops


class MipLRDecay(torch.optim.lr_scheduler._LRScheduler):
    """Log-linear decay from lr_init to lr_final with an optional sine warm-up
    (util_modules.py:7-26)."""

    def __init__(self, optimizer, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1):
        self.lr_init = lr_init
        self.lr_final = lr_final
        self.max_steps = max_steps
        self.lr_delay_steps = lr_delay_steps
        self.lr_delay_mult = lr_delay_mult
        super().__init__(optimizer)

    def get_lr(self):
        step = self.last_epoch
        delay = 1.
        if self.lr_delay_steps > 0:
            delay = self.lr_delay_mult + (1 - self.lr_delay_mult) * np.sin(
                0.5 * np.pi * np.clip(step / self.lr_delay_steps, 0, 1))
        t = np.clip(step / self.max_steps, 0, 1)
        return [delay * np.exp(np.log(self.lr_init) * (1 - t) + np.log(self.lr_final) * t)]


class PositionalEncoding(nn.Module):
    """`forward(x, y=None)`: integrated positional encoding of (mean, diagonal covariance) when
    `y` is given, plain sinusoidal encoding otherwise (util_modules.py:29-50).

    `scales` is kept as a non-trainable int64 parameter so `state_dict()` carries the same key.
    The IPE call returns `(encoding, variance)` like the reference (util_modules.py:44-46); the models'
    own forward passes take the encoding alone (model.py:113 reads `[0]`) through the fused encoders."""

    def __init__(self, min_deg, max_deg):
        super().__init__()
        self.min_deg = min_deg
        self.max_deg = max_deg
        self.scales = nn.Parameter(torch.tensor([2 ** i for i in range(min_deg, max_deg)]), requires_grad=False)

    def forward(self, x, y=None):
        if y is not None:
            return ops.ipe_encode(x, y, self.min_deg, self.max_deg, with_variance=True)
        return ops.pe_encode(x, self.min_deg, self.max_deg)


def mse_to_psnr(mse):
    return -10.0 * torch.log10(mse)


class NeRFLoss(torch.nn.modules.loss._Loss):
    """Per-level MSE, coarse levels down-weighted (util_modules.py:53-71).  With a mask the sum of
    squared errors is divided by `mask.sum()` (not by 3 * mask.sum()), as upstream."""

    def __init__(self, coarse_weight_decay=0.1):
        super().__init__()
        self.coarse_weight_decay = coarse_weight_decay

    def forward(self, input, target, mask=None):
        losses = []
        for rgb in input:
            sq = (rgb - target[..., :3]) ** 2
            losses.append(sq.mean() if mask is None else (mask * sq).sum() / mask.sum())
        losses = torch.stack(losses)
        loss = self.coarse_weight_decay * torch.sum(losses[:-1]) + losses[-1]
        return loss, mse_to_psnr(losses.detach()).cpu()


class SARNeRFLoss(torch.nn.modules.loss._Loss):
    """util_modules.py:74-89."""

    def __init__(self, coarse_weight_decay=0.1):
        super().__init__()
        self.coarse_weight_decay = coarse_weight_decay

    def level_losses(self, input, target):
        """-> total loss, per-level MSEs; both stay on the device (no host copy: usable under graph capture)."""
        losses = torch.stack([((x - target) ** 2).mean() for x in input])
        return self.coarse_weight_decay * torch.sum(losses[:-1]) + losses[-1], losses

    def forward(self, input, target):
        loss, losses = self.level_losses(input, target)
        return loss, mse_to_psnr(losses.detach()).cpu()
