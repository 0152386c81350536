"""SDE-DPM-Solver++(2M) with Karras sigmas -- the scheduler `init_iclight` builds (utils/model_utils.py:71-78):
DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012 [beta_schedule default "linear"],
algorithm_type="sde-dpmsolver++", use_karras_sigmas=True, steps_offset=1), solver_order 2, midpoint, lower_order_final,
final_sigmas_type "zero".  diffusers==0.32.1 is not available here: this restates the published algorithm (parity unpinned).

Surface kept from the reference's use (generate.py:561,211,235): set_timesteps(n), .timesteps, .init_noise_sigma,
step(model_output, timestep, sample, noise=...) -- the noise tensor is explicit instead of a generator list.
The update itself is one HIP kernel (tcl_dpm_sde_step_f16); this class only derives its scalar coefficients.
"""
import math

import numpy as np
import torch

from .lib import lib, stream


class DPMSolverSDEScheduler:
    init_noise_sigma = 1.0
    order = 2

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float64).astype(np.float32)
        self.alphas_cumprod = np.cumprod(1.0 - betas.astype(np.float64))
        self.num_train_timesteps = num_train_timesteps
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        ac = self.alphas_cumprod
        sig = ((1 - ac) / ac) ** 0.5
        log_sig = np.log(sig)
        s = np.flip(sig).copy()
        rho = 7.0
        ramp = np.linspace(0, 1, n)
        mn, mx = s[-1] ** (1 / rho), s[0] ** (1 / rho)
        karras = (mx + ramp * (mn - mx)) ** rho
        ts = []
        for sg in karras:                       # _sigma_to_t: log-sigma interpolation to a (fractional) train timestep
            ls = math.log(max(sg, 1e-10))
            d = ls - log_sig
            low = min(int(np.cumsum(d >= 0).argmax()), len(log_sig) - 2)
            lo, hi = log_sig[low], log_sig[low + 1]
            w = min(max((lo - ls) / (lo - hi), 0.0), 1.0)
            ts.append((1 - w) * low + w * (low + 1))
        self.sigmas = np.concatenate([karras, [0.0]]).astype(np.float32)
        self.timesteps = torch.from_numpy(np.array(ts).round().astype(np.int64))
        self.num_inference_steps = n
        self._i = 0
        self._lower = 0
        self._m = [None, None]

    @staticmethod
    def _alpha_sigma(sigma):
        a = 1.0 / math.sqrt(sigma * sigma + 1.0)
        return a, sigma * a

    def coefficients(self, i, second_order):
        """-> (sigma_t, alpha_t of the CURRENT sigma for x0 conversion, ca, cb0, cb1, cc) for step i -> i+1."""
        s_next, s_cur = float(self.sigmas[i + 1]), float(self.sigmas[i])
        a_cur, st_cur = self._alpha_sigma(s_cur)
        a_t, st_t = self._alpha_sigma(s_next)
        lam_s = math.log(a_cur) - math.log(st_cur)
        if s_next == 0.0:                       # final step: h = +inf
            e_h, e_2h = 0.0, 0.0
            ca = 0.0
        else:
            h = (math.log(a_t) - math.log(st_t)) - lam_s
            e_h, e_2h = math.exp(-h), math.exp(-2 * h)
            ca = st_t / st_cur * e_h
        cB = a_t * (1 - e_2h)
        cc = st_t * math.sqrt(1 - e_2h)
        if second_order:
            a_p, st_p = self._alpha_sigma(float(self.sigmas[i - 1]))
            h0 = lam_s - (math.log(a_p) - math.log(st_p))
            r0 = h0 / h
            return st_cur, a_cur, ca, cB * (1 + 0.5 / r0), -cB * 0.5 / r0, cc
        return st_cur, a_cur, ca, cB, 0.0, cc

    def step(self, model_output, timestep, sample, noise=None, return_dict=False):
        """In place on `sample` ([N,4,h,w] f16 device); model_output f16; noise f16 or None."""
        i = self._i
        last = i == len(self.timesteps) - 1
        second = (self._lower >= 1) and not last         # lower_order_final with final_sigmas_type == "zero"
        sig_t, al_t, ca, cb0, cb1, cc = self.coefficients(i, second)
        n = sample.numel()
        if self._m[0] is None:
            self._m = [torch.empty(n, dtype=torch.float32, device=sample.device) for _ in range(2)]
        m0, m1 = self._m[i % 2], self._m[(i + 1) % 2]
        lib().tcl_dpm_sde_step_f16(sample, model_output, m0, m1 if second else 0, noise if noise is not None else 0, n,
                                   sig_t, al_t, ca, cb0, cb1, cc, stream())
        self._i += 1
        if self._lower < self.order:
            self._lower += 1
        return (sample,)
