"""Oracle (TEST INFRASTRUCTURE): UniPC (bh2, order 2, x0-prediction) and DDIM (eta=0) restated.

Restates D/schedulers/scheduling_unipc_multistep.py:179-279 (tables), :282-360 (set_timesteps),
:453-521 (convert_model_output), :523-650 (UniP predictor), :652-787 (UniC corrector), :822-901 (step) and
D/schedulers/scheduling_ddim.py:297-340 (set_timesteps), :342-468 (step), for the SD-1.5 scheduler config
(beta 0.00085 -> 0.012 scaled_linear, 1000 train steps, steps_offset 1, epsilon prediction).
Scalar maths is done with fp32 torch CPU scalars in the reference's operation order.
"""
import numpy as np
import torch


def _alphas_cumprod(num_train=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class UniPCOracle:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train=1000, beta_start=0.00085, beta_end=0.012, solver_order=2):
        self.num_train = num_train
        self.alphas_cumprod = _alphas_cumprod(num_train, beta_start, beta_end)
        self.solver_order = solver_order

    def set_timesteps(self, n):
        ts = np.linspace(0, self.num_train - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)   # :294-300
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)                                             # :334
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0]]).astype(np.float32))                # :343 zero final
        self.timesteps = torch.from_numpy(ts)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = 0
        self.this_order = None

    @staticmethod
    def _alpha_sigma(sigma):                               # :423-427
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def _bh_terms(self, h, rks, order):
        """Shared R / b construction (:590-611, :728-749), predict_x0 => hh = -h, bh2 => B_h = expm1(hh)."""
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)
        R, b = [], []
        fact = 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return h_phi_1, B_h, torch.stack(R), torch.tensor(b)

    def _lambda(self, idx):
        a, s = self._alpha_sigma(self.sigmas[idx])
        return torch.log(a) - torch.log(s)

    def _predict(self, sample, order):
        m0 = self.model_outputs[-1]
        i = self.step_index
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i + 1])
        alpha_s0, sigma_s0 = self._alpha_sigma(self.sigmas[i])
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks, D1s = [], []
        for k in range(1, order):
            rk = (self._lambda(i - k) - lambda_s0) / h
            rks.append(rk)
            D1s.append((self.model_outputs[-(k + 1)] - m0) / rk)
        rks.append(1.0)
        rks = torch.tensor(rks)
        h_phi_1, B_h, R, b = self._bh_terms(h, rks, order)
        x_t_ = sigma_t / sigma_s0 * sample - alpha_t * h_phi_1 * m0
        if D1s:
            assert order == 2                                   # :619-620 rhos_p = 0.5
            pred_res = 0.5 * D1s[0]
        else:
            pred_res = 0
        return x_t_ - alpha_t * B_h * pred_res

    def _correct(self, model_t, last_sample, order):
        m0 = self.model_outputs[-1]
        i = self.step_index
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i])
        alpha_s0, sigma_s0 = self._alpha_sigma(self.sigmas[i - 1])
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks, D1s = [], []
        for k in range(1, order):
            rk = (self._lambda(i - (k + 1)) - lambda_s0) / h
            rks.append(rk)
            D1s.append((self.model_outputs[-(k + 1)] - m0) / rk)
        rks.append(1.0)
        rks = torch.tensor(rks)
        h_phi_1, B_h, R, b = self._bh_terms(h, rks, order)
        if order == 1:
            rhos_c = torch.tensor([0.5])                        # :757-758
        else:
            rhos_c = torch.linalg.solve(R, b)                   # :760
        x_t_ = sigma_t / sigma_s0 * last_sample - alpha_t * h_phi_1 * m0
        corr = 0
        for k, d in enumerate(D1s):
            corr = corr + rhos_c[k] * d
        return x_t_ - alpha_t * B_h * (corr + rhos_c[-1] * (model_t - m0))

    def step(self, model_output, sample):
        i = self.step_index
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i])
        x0 = (sample - sigma_t * model_output) / alpha_t        # :499
        if i > 0 and self.last_sample is not None:              # :854-868
            sample = self._correct(x0, self.last_sample, self.this_order)
        for k in range(self.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = x0
        this_order = min(self.solver_order, len(self.timesteps) - i)     # lower_order_final :877-880
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self._predict(sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev


class DDIMOracle:
    """DDIM with SD-1.5 config: timestep_spacing='leading', steps_offset=1, clip_sample=False,
    set_alpha_to_one=False, eta=0 (scheduling_ddim.py:297-340, 342-468)."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train=1000, beta_start=0.00085, beta_end=0.012):
        self.num_train = num_train
        self.alphas_cumprod = _alphas_cumprod(num_train, beta_start, beta_end)
        self.final_alpha_cumprod = self.alphas_cumprod[0]      # set_alpha_to_one=False  (:206)

    def set_timesteps(self, n):
        self.n = n
        ratio = self.num_train // n                            # :326-331
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + 1
        self.timesteps = torch.from_numpy(ts)
        self.step_index = 0

    def step(self, model_output, sample):
        t = int(self.timesteps[self.step_index])
        prev_t = t - self.num_train // self.n                  # :399
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        beta_t = 1 - a_t
        x0 = (sample - beta_t ** 0.5 * model_output) / a_t ** 0.5           # :410
        dir_xt = (1 - a_prev) ** 0.5 * model_output                          # :444 (std_dev_t = 0)
        self.step_index += 1
        return a_prev ** 0.5 * x0 + dir_xt                                   # :447
