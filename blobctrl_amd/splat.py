"""Gaussian-blob splat rasteriser on MI355X: drop-in for `blobctrl.utils.utils.splat_features(...)`.

Keeps the reference call signature for the branch the pipeline uses (blobctrl/utils/utils.py:80-96, 145-194 and the call
sites scripts/blobctrl_inference.py:112-117, scripts/blobctrl_app.py:653-658):
    splat_features(xs, ys, covs, sizes, score_size=(h, w), return_d_score=True) -> Tensor[N, M+1, h, w]  (float64)
with N = M = 1 as hard-coded by that branch (ut:157-158).  The rasterisation is ONE HIP kernel (bc_splat_scores), fp64.
Also hosts the ellipse -> normalised Gaussian helpers of scripts/blobctrl_inference.py:23-109.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def ellipse_to_gaussian(ellipse):
    """((xc,yc),(d1,d2),angle_deg) -> (mean[2], cov[2,2]); scripts/blobctrl_inference.py:23-86."""
    (xc, yc), (d1, d2), angle = ellipse
    theta = np.radians((((180 - angle) % 180) + 90) % 180)
    a, b = d1 / 2.0, d2 / 2.0
    cov = np.array([[b ** 2, 0.0], [0.0, a ** 2]])
    R = np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
    cov = R @ cov @ R.T
    cov[0, 1] *= -1
    cov[1, 0] *= -1
    return np.array([xc, yc], dtype=np.float64), cov


def normalize_gs(mean, cov, width, height):
    """scripts/blobctrl_inference.py:88-98."""
    return mean / np.array([width, height]), cov / (np.sqrt(width ** 2 + height ** 2) ** 2)


def blob_dict_from_ellipse(ellipse, width, height):
    """scripts/blobctrl_inference.py:78-109 chained: returns the kwargs `splat_features(**blob, ...)` expects."""
    mean, cov = ellipse_to_gaussian(ellipse)
    nm = mean / np.array([width, height])
    nc = cov / (np.sqrt(width ** 2 + height ** 2) ** 2)
    return {"xs": torch.tensor(nm[0]).unsqueeze(0), "ys": torch.tensor(nm[1]).unsqueeze(0),
            "covs": torch.tensor(nc).unsqueeze(0).unsqueeze(0), "sizes": torch.tensor([1.0]).unsqueeze(0)}


def splat_features(xs, ys, covs, sizes, score_size=None, return_d_score=False, device="cuda:0", **kwargs):
    """HIP replacement of utils.splat_features for the tuple-`score_size`, `return_d_score=True` branch."""
    if not return_d_score or not isinstance(score_size, (tuple, list)):
        raise NotImplementedError("only the pipeline's branch is provided: score_size=(h, w), return_d_score=True")
    h, w = int(score_size[0]), int(score_size[1])
    xs = torch.as_tensor(xs, dtype=torch.float64).reshape(-1)
    ys = torch.as_tensor(ys, dtype=torch.float64).reshape(-1)
    covs = torch.as_tensor(covs, dtype=torch.float64).reshape(-1, 2, 2)
    sizes = torch.as_tensor(sizes, dtype=torch.float64).reshape(-1)
    n = xs.numel()
    if not (ys.numel() == n and covs.shape[0] == n and sizes.numel() == n):
        raise ValueError("xs, ys, covs, sizes must describe the same number of blobs")
    if n != 1:
        raise ValueError("the reference branch hard-codes batch = 1, n_gaussians = 1 (utils.py:157-158)")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.BlobCtrlHipError("splat_features runs on MI355X only; there is no CPU fallback")
    prm = torch.zeros(n, 8, dtype=torch.float64)
    prm[:, 0], prm[:, 1] = xs, ys
    prm[:, 2:6] = covs.reshape(n, 4)
    prm[:, 6] = sizes
    from . import ops  # noqa: F401  (registers torch.ops.blobctrl.*)
    return torch.ops.blobctrl.splat_scores(prm, h, w, dev.index if dev.index is not None else 0)


def _splat_scores_impl(params: torch.Tensor, h: int, w: int, dev: torch.device) -> torch.Tensor:
    """Body of torch.ops.blobctrl.splat_scores: one bc_splat_scores launch (fp64) on `dev`."""
    lib = _lib.load()
    p = params.detach().to("cpu", torch.float64).contiguous()
    n = p.shape[0]
    prm = (C.c_double * (8 * n))(*p.reshape(-1).tolist())
    out = torch.empty(n, 2, h, w, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.bc_splat_scores(prm, n, h, w, out.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "bc_splat_scores")
    return out
