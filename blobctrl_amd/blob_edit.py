"""Blob-editing geometry of the reference app (host side): how add / remove / move / resize / rotate / replace turn
an ellipse into the three pipeline inputs (fg image, bg image, gs_score) - scripts/blobctrl_app.py:452-620, 661-688, 1095-1126,
762-792 (`app:` below).  Ellipses use the OpenCV `fitEllipse` convention ((xc, yc), (d1, d2), angle_deg).

Pinned: every function marked [pinned] is checked against vectors produced by EXECUTING the reference's own definitions
(tools/make_golden.py::golden_blob_edit extracts them from scripts/blobctrl_app.py with `ast`; fixture tests/golden/blob_edit.json).
Not pinned (OpenCV, a third-party dependency that is absent here): mask -> ellipse fitting (`cv2.fitEllipse` on the convex hull,
app:382-389) is not provided; `ellipse_mask` replaces `cv2.ellipse(..., -1, lineType=LINE_AA)` + the app's `> 0` threshold by an exact
"pixel square touches the ellipse" test on a 4 x 4 sub-pixel grid, which can differ from OpenCV's rasteriser on boundary pixels.
"""
import math
from typing import Sequence, Tuple

import numpy as np

from .splat import blob_dict_from_ellipse, splat_features

Ellipse = Tuple[Tuple[float, float], Tuple[float, float], float]


def normalize_ellipse(ellipse: Ellipse, width: int, height: int):
    """[pinned] app:452-459."""
    (xc, yc), (d1, d2), angle = ellipse
    max_length = np.sqrt(width ** 2 + height ** 2)
    return xc / width, yc / height, d1 / max_length, d2 / max_length, angle


def is_point_in_ellipse(point: Sequence[float], ellipse: Ellipse) -> bool:
    """[pinned] app:479-499."""
    (xc, yc), (d1, d2), angle = ellipse
    theta = np.radians(angle)
    xp, yp = point[0] - xc, point[1] - yc
    xr = xp * np.cos(theta) - yp * np.sin(theta)
    yr = xp * np.sin(theta) + yp * np.cos(theta)
    return bool((xr ** 2) / ((d1 / 2) ** 2) + (yr ** 2) / ((d2 / 2) ** 2) <= 1)


def calculate_ellipse_vertices(ellipse: Ellipse) -> np.ndarray:
    """[pinned] app:502-532: the four axis end points, rotated and translated."""
    (xc, yc), (d1, d2), angle = ellipse
    a = np.deg2rad(angle)
    rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    v = np.array([[d1 / 2, 0], [-d1 / 2, 0], [0, d2 / 2], [0, -d2 / 2]])
    return np.dot(v, rot.T) + np.array([xc, yc])


def move_ellipse(ellipse: Ellipse, tracking_points: Sequence[Sequence[float]]) -> Ellipse:
    """[pinned] app:535-543: translate by the last drag segment."""
    (xc, yc), (d1, d2), angle = ellipse
    (lx, ly), (px, py) = tracking_points[-1], tracking_points[-2]
    return (xc + (lx - px), yc + (ly - py)), (d1, d2), angle


def resize_blob(ellipse: Ellipse, resizing_factor: float, height: int, width: int, resize_type: int = 0,
                min_blob_area: float = 1600, exceed_threshold: float = 0.4):
    """[pinned] app:546-592: scale both axes (type 0), d2 only (1) or d1 only (2); the factor is walked in 0.1 steps until the blob
    stays within [-0.4, 1.4] x the image and above 1600 px^2.  Returns (ellipse, factor, too_big, too_small) - the reference reports
    the last two through gr.Warning."""
    (xc, yc), (d1, d2), angle = ellipse
    too_big = too_small = False
    while True:
        if resize_type == 0:
            r1, r2 = d1 * resizing_factor, d2 * resizing_factor
        elif resize_type == 1:
            r1, r2 = d1, d2 * resizing_factor
        elif resize_type == 2:
            r1, r2 = d1 * resizing_factor, d2
        else:
            raise ValueError("resize_type must be 0, 1 or 2")
        resized = ((xc, yc), (r1, r2), angle)
        v = calculate_ellipse_vertices(resized) / np.array([width, height])
        if resizing_factor != 1:
            if np.all(v >= -exceed_threshold) and np.all(v <= 1 + exceed_threshold):
                area = np.pi * (r1 / 2) * (r2 / 2)
                if area >= min_blob_area:
                    break
                too_small = True
                resizing_factor += 0.1
                if area < 1e-6:
                    break
            else:
                too_big = True
                resizing_factor -= 0.1
        else:
            break
    return resized, resizing_factor, too_big, too_small


def rotate_blob(ellipse: Ellipse, rotation_degree: float):
    """[pinned] app:595-601."""
    (xc, yc), (d1, d2), angle = ellipse
    return ((xc, yc), (d1, d2), (angle + rotation_degree) % 180), rotation_degree


def composite_mask_and_image(mask: np.ndarray, image: np.ndarray, masked_color=(0, 0, 0)) -> np.ndarray:
    """[pinned] app:461-476 on arrays: pixels where the mask is set (> 0, or channel sum > 255 for RGB masks) take `masked_color`."""
    mask, image = np.asarray(mask), np.asarray(image)
    ind = (mask > 0).astype(np.uint8) if mask.ndim == 2 else (mask.sum(-1) > 255).astype(np.uint8)
    out = image * (1 - ind[:, :, None]) + np.asarray(masked_color) * ind[:, :, None]
    return out.astype(np.uint8)


def object_region_from_mask(mask: np.ndarray, image: np.ndarray) -> np.ndarray:
    """[pinned, with cv2.boundingRect restated as the min / max of the set pixels] app:661-688: cut the masked object out of `image`,
    paste it centred on a white canvas of the same size (the pipeline's fg_image)."""
    mask, image = np.asarray(mask), np.asarray(image)
    ind = (mask > 0).astype(np.uint8) if mask.ndim == 2 else (mask.sum(-1) > 255).astype(np.uint8)
    ys, xs = np.nonzero(ind)
    if len(ys) == 0:
        raise ValueError("empty mask")
    x, y, w, h = xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1
    rect = np.where(ind[y:y + h, x:x + w, None] > 0, image[y:y + h, x:x + w], 255)
    H, W = image.shape[:2]
    out = np.ones((H, W, 3), dtype=np.uint8) * 255
    sy, sx = (H - h) // 2, (W - w) // 2
    out[sy:sy + h, sx:sx + w] = rect
    return out


def ellipse_mask(ellipse: Ellipse, height: int, width: int, sub: int = 4) -> np.ndarray:
    """[NOT pinned - replaces cv2.ellipse(..., thickness=-1, LINE_AA) followed by the app's `> 0`, app:1113-1121] uint8 mask, 255
    where the pixel square [x-0.5, x+0.5] x [y-0.5, y+0.5] contains a point of the filled ellipse (sampled on a sub x sub grid)."""
    (xc, yc), (d1, d2), angle = ellipse
    t = math.radians(angle)
    a, b = max(d1 / 2.0, 1e-9), max(d2 / 2.0, 1e-9)
    off = (np.arange(sub) + 0.5) / sub - 0.5
    ys = (np.arange(height)[:, None] + off[None, :]).reshape(-1)                  # [H*sub]
    xs = (np.arange(width)[:, None] + off[None, :]).reshape(-1)                   # [W*sub]
    X, Y = xs[None, :] - xc, ys[:, None] - yc
    u = X * math.cos(t) + Y * math.sin(t)                                         # coordinates along the d1 / d2 axes
    v = -X * math.sin(t) + Y * math.cos(t)
    inside = (u / a) ** 2 + (v / b) ** 2 <= 1.0
    cover = inside.reshape(height, sub, width, sub).any(axis=(1, 3))
    return (cover * 255).astype(np.uint8)


def build_edit_inputs(op: str, image: np.ndarray, start_ellipse: Ellipse, target_ellipse: Ellipse = None, strength: float = 1.0,
                      device: str = "cuda:0"):
    """The three things an edit hands to the pipeline: (bg_image uint8 [H,W,3], gs_score [1,2,h,w], strength).
      move / resize / rotate / add / replace: start ellipse painted white, then the target ellipse painted black (app:1113-1126);
                                              score = splat of the TARGET ellipse (app:774-778);
      remove: start ellipse painted white; score overwritten bg = 1, fg = 0; strength 0.0 (app:779-792, inf:175-188).
    The fg image (object on white) comes from `object_region_from_mask` or from an uploaded object image."""
    image = np.asarray(image)
    H, W = image.shape[:2]
    white = composite_mask_and_image(ellipse_mask(start_ellipse, H, W), image, (255, 255, 255))
    if op == "remove":
        blob = blob_dict_from_ellipse([list(start_ellipse[0]), list(start_ellipse[1]), start_ellipse[2]], W, H)
        score = splat_features(**blob, score_size=(H // 8, W // 8), return_d_score=True, device=device)
        score[:, 0] = 1.0
        score[:, 1] = 0.0
        return white, score, 0.0
    if op not in ("move", "resize", "rotate", "add", "replace"):
        raise ValueError(f"unknown edit operation {op!r}")
    if target_ellipse is None:
        raise ValueError(f"{op} needs a target ellipse")
    bg = composite_mask_and_image(ellipse_mask(target_ellipse, H, W), white, (0, 0, 0))
    blob = blob_dict_from_ellipse([list(target_ellipse[0]), list(target_ellipse[1]), target_ellipse[2]], W, H)
    score = splat_features(**blob, score_size=(H // 8, W // 8), return_d_score=True, device=device)
    return bg, score, float(strength)
