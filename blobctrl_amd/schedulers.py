"""Scheduler coefficient tables for the hipGraph-captured denoise loop.

The reference schedulers mutate Python state and build small tensors every step
(D/schedulers/scheduling_unipc_multistep.py:822-901, scheduling_ddim.py:342-468).  Every quantity they compute depends
only on the sigma / alpha tables, so each step is a fixed LINEAR combination of
    x (current latents), eps (guided noise prediction), last_sample, x0_{i-1}, x0_{i-2}
whose scalar coefficients are precomputed here per step index, uploaded once, and applied by one fused kernel
(`bc_cfg_scheduler_step`).  Row layout (16 floats per step):
    [0] 1/alpha_t   [1] sigma_t/alpha_t                    x0   = x*c0 - eps*c1
    [2] use_corrector
    [3] cc_x [4] cc_m0 [5] cc_m1 [6] cc_mt                 x_c  = c3*last + c4*x0_{i-1} + c5*x0_{i-2} + c6*x0      (UniC)
    [7] cp_x [8] cp_m0 [9] cp_m1 [10] cp_eps               x'   = c7*x_c + c8*x0 + c9*x0_{i-1} + c10*eps           (UniP / DDIM)
Scalar maths follows the reference in fp32 torch CPU ops (same operation order) so the tables match it to rounding.

SD-1.5 scheduler config: betas 0.00085 -> 0.012 scaled_linear, 1000 train steps, steps_offset 1,
epsilon prediction; UniPC: solver_order 2, bh2, predict_x0, lower_order_final, linspace spacing, final sigma 0;
DDIM: leading spacing, clip_sample False, set_alpha_to_one False, eta 0.
"""
import numpy as np
import torch


def _alphas_cumprod(num_train, beta_start, beta_end):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class _Base:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        self.num_train = num_train_timesteps
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end)
        self.timesteps = None
        self.coef = None

    def scale_model_input(self, sample, timestep=None):        # identity for both (pipe:1032)
        return sample

    def table(self) -> torch.Tensor:
        """[num_steps, 16] float32 coefficient rows."""
        return self.coef


class UniPCTable(_Base):
    """UniPCMultistepScheduler restated as a coefficient table (scheduling_unipc_multistep.py:282-360, 453-901)."""

    def __init__(self, solver_order=2, **kw):
        super().__init__(**kw)
        assert solver_order == 2, "the coefficient form is written for the reference's solver_order=2 / bh2"

    def set_timesteps(self, n, device=None):
        ts = np.linspace(0, self.num_train - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = n
        self.coef = self._build(n)
        return self

    @staticmethod
    def _as(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def _lam(self, i):
        a, s = self._as(self.sigmas[i])
        return torch.log(a) - torch.log(s)

    @staticmethod
    def _bh(h, rks, order):
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

    def _build(self, n):
        coef = torch.zeros(n, 16, dtype=torch.float32)
        lower_order_nums = 0
        prev_order = None
        for i in range(n):
            alpha_t, sigma_t = self._as(self.sigmas[i])
            coef[i, 0] = 1.0 / alpha_t
            coef[i, 1] = sigma_t / alpha_t
            # ---- UniC corrector with the PREVIOUS step's order (:854-868, :652-787)
            if i > 0:
                order = prev_order
                alpha_s0, sigma_s0 = self._as(self.sigmas[i - 1])
                lam_t = torch.log(alpha_t) - torch.log(sigma_t)
                lam_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
                h = lam_t - lam_s0
                rks = []
                for k in range(1, order):
                    rks.append((self._lam(i - (k + 1)) - lam_s0) / h)
                rks.append(1.0)
                rks_t = torch.tensor(rks)
                h_phi_1, B_h, R, b = self._bh(h, rks_t, order)
                rhos_c = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
                cc_x = sigma_t / sigma_s0
                cc_m0 = -alpha_t * h_phi_1
                cc_m1 = torch.tensor(0.0)
                cc_mt = -alpha_t * B_h * rhos_c[-1]
                cc_m0 = cc_m0 + alpha_t * B_h * rhos_c[-1]                 # -(alpha B_h rho_last) * (-m0)
                if order == 2:
                    w = alpha_t * B_h * rhos_c[0] / rks_t[0]               # D1 = (m1 - m0) / rk
                    cc_m1 = -w
                    cc_m0 = cc_m0 + w
                coef[i, 2] = 1.0
                coef[i, 3], coef[i, 4], coef[i, 5], coef[i, 6] = cc_x, cc_m0, cc_m1, cc_mt
            # ---- UniP predictor (:877-888, :523-650)
            this_order = min(2, n - i)
            this_order = min(this_order, lower_order_nums + 1)
            alpha_n, sigma_n = self._as(self.sigmas[i + 1])
            lam_n = torch.log(alpha_n) - torch.log(sigma_n)
            lam_t = torch.log(alpha_t) - torch.log(sigma_t)
            h = lam_n - lam_t
            rks = []
            for k in range(1, this_order):
                rks.append((self._lam(i - k) - lam_t) / h)
            rks.append(1.0)
            rks_t = torch.tensor(rks)
            h_phi_1, B_h, R, b = self._bh(h, rks_t, this_order)
            cp_x = sigma_n / sigma_t
            cp_m0 = -alpha_n * h_phi_1
            cp_m1 = torch.tensor(0.0)
            if this_order == 2:
                w = alpha_n * B_h * 0.5 / rks_t[0]                          # rhos_p = 0.5 (:619-620)
                cp_m1 = -w
                cp_m0 = cp_m0 + w
            coef[i, 7], coef[i, 8], coef[i, 9] = cp_x, cp_m0, cp_m1
            prev_order = this_order
            if lower_order_nums < 2:
                lower_order_nums += 1
        assert torch.isfinite(coef).all(), "non-finite UniPC coefficient"
        return coef


class DDIMTable(_Base):
    """DDIMScheduler (eta = 0) as a coefficient table (scheduling_ddim.py:297-340, 342-468)."""

    def set_timesteps(self, n, device=None):
        ratio = self.num_train // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + 1
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = n
        coef = torch.zeros(n, 16, dtype=torch.float32)
        final_alpha = self.alphas_cumprod[0]
        for i, t in enumerate(ts.tolist()):
            prev_t = t - ratio
            a_t = self.alphas_cumprod[t]
            a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else final_alpha
            coef[i, 0] = 1.0 / a_t ** 0.5
            coef[i, 1] = (1 - a_t) ** 0.5 / a_t ** 0.5
            coef[i, 8] = a_prev ** 0.5
            coef[i, 10] = (1 - a_prev) ** 0.5
        self.coef = coef
        return self


def apply_table_step(coef_row, eps, x, hist):
    """Host (torch) evaluation of one table row - the same arithmetic `bc_cfg_scheduler_step` performs after CFG.
    Used by the drop-in scheduler objects; `hist` = dict(m0, m1, last)."""
    c = coef_row
    x0 = x * c[0] - eps * c[1]
    xc = x
    if c[2] != 0:
        xc = c[3] * hist["last"] + c[4] * hist["m0"] + c[5] * hist["m1"] + c[6] * x0
    xn = c[7] * xc + c[8] * x0 + c[9] * hist["m0"] + c[10] * eps
    hist["m1"], hist["m0"], hist["last"] = hist["m0"], x0, xc
    return xn


class TableScheduler:
    """Drop-in for `pipeline.scheduler`: set_timesteps / timesteps / init_noise_sigma /
    scale_model_input / step(noise_pred, t, latents, return_dict=False)[0] / order."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, kind="unipc", **kw):
        self.table_impl = UniPCTable(**kw) if kind == "unipc" else DDIMTable(**kw)
        self.kind = kind

    def set_timesteps(self, num_inference_steps, device=None):
        self.table_impl.set_timesteps(num_inference_steps)
        self.timesteps = self.table_impl.timesteps.to(device) if device is not None else self.table_impl.timesteps
        self._i = 0
        self._hist = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict=False, **kw):
        if self._hist is None:
            z = torch.zeros_like(sample)
            self._hist = dict(m0=z, m1=z.clone(), last=z.clone())
        row = self.table_impl.coef[self._i].tolist()
        out = apply_table_step(row, model_output, sample, self._hist)
        self._i += 1
        return (out,)


class SchedulerConfig(dict):
    """`scheduler.config` with attribute and mapping access (the scripts do `X.from_config(pipeline.scheduler.config)`, inf:276-277)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


_SD15 = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
             prediction_type="epsilon")


class _ConfiguredScheduler(TableScheduler):
    _kind = None
    _defaults = {}

    def __init__(self, **kw):
        cfg = dict(_SD15)
        cfg.update(self._defaults)
        cfg.update({k: v for k, v in kw.items() if v is not None})
        if cfg["beta_schedule"] != "scaled_linear" or cfg["prediction_type"] != "epsilon":
            raise NotImplementedError("only the SD-1.5 configuration (scaled_linear betas, epsilon prediction) is tabulated")
        self._check(cfg)
        super().__init__(self._kind, num_train_timesteps=cfg["num_train_timesteps"], beta_start=cfg["beta_start"],
                         beta_end=cfg["beta_end"])
        self.config = SchedulerConfig(cfg)
        # like diffusers' `_use_default_values`: keys the caller did not give are NOT inherited by another class's from_config
        self.config["_use_default_values"] = sorted(k for k in cfg if k not in kw or kw[k] is None)
        self.timesteps = None

    def _check(self, cfg):
        pass

    @classmethod
    def from_config(cls, config, **kw):
        """Like diffusers' ConfigMixin.from_config: keys this class knows are taken from `config` (another scheduler's config is
        fine - unknown keys such as PNDM's `skip_prk_steps` are dropped), everything else keeps this class's defaults."""
        known = set(_SD15) | set(cls._defaults)
        defaulted = set(dict(config).get("_use_default_values", ()))
        src = {k: v for k, v in dict(config).items() if k not in defaulted}
        src.update(kw)
        return cls(**{k: v for k, v in src.items() if k in known})

    def table_params(self):
        """(num_train_timesteps, beta_start, beta_end): what the engine builds its coefficient table from."""
        return (int(self.config["num_train_timesteps"]), float(self.config["beta_start"]), float(self.config["beta_end"]))

    @property
    def kind(self):
        return self._kind

    @kind.setter
    def kind(self, v):
        pass


class PNDMScheduler(_ConfiguredScheduler):
    """What SD-1.5's `scheduler/scheduler_config.json` names.  The reference scripts only read its `.config` and build UniPC (or
    DDIM) from it (inf:276-277), so this class carries the configuration and refuses to step."""
    _kind = None
    _defaults = dict(skip_prk_steps=True, set_alpha_to_one=False, timestep_spacing="leading", clip_sample=False)

    def __init__(self, **kw):
        cfg = dict(_SD15)
        cfg.update(self._defaults)
        cfg.update({k: v for k, v in kw.items() if v is not None})
        self.config = SchedulerConfig(cfg)
        self.config["_use_default_values"] = sorted(k for k in cfg if k not in kw or kw[k] is None)
        self.timesteps = None
        self.table_impl = None

    def set_timesteps(self, *a, **k):
        raise NotImplementedError("PNDM is not tabulated: replace it as the scripts do, "
                                  "pipeline.scheduler = UniPCMultistepScheduler.from_config(pipeline.scheduler.config)")

    step = set_timesteps


def scheduler_from_config_dir(path):
    """`<model>/scheduler/scheduler_config.json` -> the scheduler object its `_class_name` names (PNDM / UniPC / DDIM)."""
    import json
    import os
    with open(os.path.join(path, "scheduler_config.json")) as f:
        cfg = json.load(f)
    name = cfg.get("_class_name", "PNDMScheduler")
    classes = {"PNDMScheduler": PNDMScheduler, "UniPCMultistepScheduler": UniPCMultistepScheduler, "DDIMScheduler": DDIMScheduler}
    if name not in classes:
        raise NotImplementedError(f"scheduler class {name} is not available (PNDM config holder, UniPC and DDIM are)")
    cls = classes[name]
    known = set(_SD15) | set(cls._defaults)
    return cls(**{k: v for k, v in cfg.items() if k in known})


class UniPCMultistepScheduler(_ConfiguredScheduler):
    """Drop-in for diffusers' UniPCMultistepScheduler as the scripts configure it (`from_config(PNDM config)`)."""
    _kind = "unipc"
    _defaults = dict(solver_order=2, solver_type="bh2", predict_x0=True, lower_order_final=True, timestep_spacing="linspace",
                     final_sigmas_type="zero", thresholding=False)

    def _check(self, cfg):
        if (cfg["solver_order"], cfg["solver_type"], cfg["predict_x0"], cfg["lower_order_final"], cfg["timestep_spacing"],
                cfg["final_sigmas_type"], cfg["thresholding"]) != (2, "bh2", True, True, "linspace", "zero", False):
            raise NotImplementedError("UniPC is tabulated for solver_order=2 / bh2 / predict_x0 / lower_order_final / linspace / "
                                      "final sigma zero (what the reference scripts run)")


class DDIMScheduler(_ConfiguredScheduler):
    """Drop-in for diffusers' DDIMScheduler with the SD-1.5 scheduler_config.json values (eta = 0)."""
    _kind = "ddim"
    _defaults = dict(clip_sample=False, set_alpha_to_one=False, timestep_spacing="leading", thresholding=False)

    def _check(self, cfg):
        if cfg["clip_sample"] or cfg["set_alpha_to_one"] or cfg["timestep_spacing"] != "leading" or cfg["steps_offset"] != 1 or \
                cfg["thresholding"]:
            raise NotImplementedError("DDIM is tabulated for clip_sample=False, set_alpha_to_one=False, leading spacing, steps_offset=1")
