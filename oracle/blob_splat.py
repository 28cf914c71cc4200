"""Oracle (TEST INFRASTRUCTURE): Gaussian-blob maths restated in numpy float64.

Restates, for the single-blob / tuple-`score_size` branch the pipeline actually uses:
  * scripts/blobctrl_inference.py:23-75   ellipse -> Gaussian (mean, covariance with negated off-diagonals)
  * scripts/blobctrl_inference.py:78-98   normalisation by (W, H) and by the image diagonal
  * blobctrl/utils/utils.py:145-194       `splat_features(..., score_size=(h, w), return_d_score=True)`
  * blobctrl/pipelines/pipeline_blobnet.py:706-721  rank-1 feature splat (einsum 'nmhw,nmc->nchw')
"""
import numpy as np


def ellipse_to_gaussian(ellipse):
    """OpenCV ellipse ((xc,yc),(d1,d2),angle_deg) -> (mean[2], cov[2,2]).  inf:23-75,78-86."""
    (xc, yc), (d1, d2), angle_clockwise_short_axis = ellipse
    anti_short = (180 - angle_clockwise_short_axis) % 180          # inf:71-75
    anti_long = (anti_short + 90) % 180
    theta = np.radians(anti_long)
    a = d1 / 2.0
    b = d2 / 2.0
    cov = np.array([[b ** 2, 0.0], [0.0, a ** 2]])                    # inf:50-54 (sigma_x = b, sigma_y = a)
    R = np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
    cov = R @ cov @ R.T
    cov[0, 1] *= -1                                                   # inf:62-63
    cov[1, 0] *= -1
    return np.array([xc, yc], dtype=np.float64), cov


def normalize_gaussian(mean, cov, width, height):
    """inf:88-98."""
    nmean = mean / np.array([width, height])
    max_length = np.sqrt(width ** 2 + height ** 2)
    return nmean, cov / (max_length ** 2)


def splat_scores(xs, ys, cov, size, h, w):
    """`splat_features(xs, ys, covs, sizes, score_size=(h, w), return_d_score=True)` for one blob.

    ut:145-160 (delta, Mahalanobis via solve), ut:162-163 (2*sigmoid(-m) clamped at 1),
    ut:165-172 (sizes < 0.5 -> 1e-6), ut:175-183 (bg=1 + alpha composite), ut:193-194 (-> [n, m, h, w]).
    Returns float64 [1, 2, h, w]: channel 0 = background score, channel 1 = foreground score.
    """
    gx = np.tile(np.arange(w, dtype=np.float64), h)                  # ut:148
    gy = np.repeat(np.arange(h, dtype=np.float64), w)                # ut:149
    dx = (gx - xs * w) / w                                           # ut:151-153
    dy = (gy - ys * h) / h
    delta = np.stack([dx, dy], 0)                                    # [2, h*w]
    sol = np.linalg.solve(np.asarray(cov, dtype=np.float64), delta)  # ut:156
    m = (delta * sol).sum(0).reshape(h, w)
    with np.errstate(over="ignore"):
        s = 1.0 / (1.0 + np.exp(m))                                  # sigmoid(-m)  ut:162
    s = np.minimum(2.0 * s, 1.0)                                     # ut:163
    if size < 0.5:                                                   # ut:167-172
        s = np.full_like(s, np.float32(1e-6))                      # torch.tensor(1e-6) is float32 in the reference
    scores = np.stack([np.ones_like(s), s], -1)                      # ut:175-176  [h, w, 2]
    # alpha composite ut:179-181: d[..., i] = prod_{j>i}(1 - s_j) * s_i ; d[..., -1] = s[..., -1]
    d_bg = (1.0 - scores[..., 1]) * scores[..., 0]
    d_fg = scores[..., 1]
    return np.stack([d_bg, d_fg], 0)[None]


def splat_scores_from_ellipse(ellipse, img_w, img_h, h, w, size=1.0):
    """inf:78-117 chained: ellipse -> normalised Gaussian -> [1,2,h,w] scores."""
    mean, cov = ellipse_to_gaussian(ellipse)
    nmean, ncov = normalize_gaussian(mean, cov, img_w, img_h)
    return splat_scores(nmean[0], nmean[1], ncov, size, h, w)


def splat_feature_map(scores, feats):
    """pipe:718-721 einsum 'nmhw,nmc->nchw'.  scores [N,M,h,w], feats [N,M,C] -> [N,C,h,w]."""
    return np.einsum("nmhw,nmc->nchw", scores, feats)
