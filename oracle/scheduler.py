"""Oracle (CPU fp32) for DPMSolverMultistepScheduler(sde-dpmsolver++, karras, order 2, midpoint).
TEST INFRASTRUCTURE.  PARITY UNPINNED (diffusers==0.32.1 absent; restates the published update)."""
import numpy as np
import torch


class Scheduler:
    def __init__(self, n, beta_start=0.00085, beta_end=0.012, train=1000):
        betas = torch.linspace(beta_start, beta_end, train, dtype=torch.float32)
        ac = torch.cumprod(1 - betas.double(), 0).numpy()
        sig = ((1 - ac) / ac) ** 0.5
        log_sig = np.log(sig)
        smax, smin = sig[-1], sig[0]
        ramp = np.linspace(0, 1, n)
        k = (smax ** (1 / 7) + ramp * (smin ** (1 / 7) - smax ** (1 / 7))) ** 7
        ts = []
        for s in k:
            ls = np.log(np.maximum(s, 1e-10))
            d = ls - log_sig[:, None]
            low = np.cumsum(d >= 0, axis=0).argmax(axis=0).clip(max=train - 2)
            lo, hi = log_sig[low], log_sig[low + 1]
            w = np.clip((lo - ls) / (lo - hi), 0, 1)
            ts.append(((1 - w) * low + w * (low + 1)).item())
        self.timesteps = np.array(ts).round().astype(np.int64)
        self.sigmas = torch.tensor(np.concatenate([k, [0.0]]), dtype=torch.float32)
        self.outs = []
        self.i = 0

    def step(self, eps, x, noise):
        s = self.sigmas
        conv = lambda sg: (1 / (sg ** 2 + 1) ** 0.5, sg / (sg ** 2 + 1) ** 0.5)
        a_s, st_s = conv(s[self.i])
        x = x.float()
        x0 = (x - st_s * eps.float()) / a_s
        self.outs.append(x0)
        a_t, st_t = conv(s[self.i + 1])
        lam_t, lam_s = torch.log(a_t) - torch.log(st_t), torch.log(a_s) - torch.log(st_s)
        h = lam_t - lam_s
        last = self.i == len(self.timesteps) - 1
        D = x0
        if len(self.outs) >= 2 and not last:
            a_p, st_p = conv(s[self.i - 1])
            r0 = (lam_s - (torch.log(a_p) - torch.log(st_p))) / h
            D = x0 + 0.5 * (1 / r0) * (x0 - self.outs[-2])
        x = (st_t / st_s * torch.exp(-h)) * x + a_t * (1 - torch.exp(-2 * h)) * D + st_t * torch.sqrt(1 - torch.exp(-2 * h)) * noise.float()
        self.i += 1
        return x
