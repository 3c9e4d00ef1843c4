"""ORACLE (test infrastructure only): RAdam restated in plain torch.

The reference optimises with `blox.torch.radam.RAdam(params, lr, betas=(adam_beta, 0.999))`
(/root/reference/gcp/prediction/training/gcp_builder.py:27,88-89,178-179).  blox is an absent submodule, so this follows
the published algorithm (Liu et al., "On the Variance of the Adaptive Learning Rate and Beyond", 2019, Alg. 2) in the
form of the authors' reference implementation: rectified Adam step when rho_t >= 5, bias-corrected momentum SGD
otherwise; eps added to sqrt(v) before the division; no weight decay.  PARITY UNPINNED against blox.
"""
import math

import torch


class RAdamOracle:
    def __init__(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.betas, self.eps = lr, betas, eps
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, params, grads):
        """in-place update of the tensors in `params` (dict) with `grads` (dict)"""
        b1, b2 = self.betas
        self.t += 1
        t = self.t
        b2t = b2 ** t
        sma_max = 2.0 / (1.0 - b2) - 1.0
        sma = sma_max - 2.0 * t * b2t / (1.0 - b2t)
        for k, g in grads.items():
            p = params[k]
            m = self.m.setdefault(k, torch.zeros_like(p))
            v = self.v.setdefault(k, torch.zeros_like(p))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            if sma >= 5:
                step = math.sqrt((1 - b2t) * (sma - 4) / (sma_max - 4) * (sma - 2) / sma * sma_max / (sma_max - 2)) / (1 - b1 ** t)
                p.addcdiv_(m, v.sqrt().add_(self.eps), value=-step * self.lr)
            else:
                step = 1.0 / (1 - b1 ** t)
                p.add_(m, alpha=-step * self.lr)
