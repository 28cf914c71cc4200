"""Blob-editing geometry of the reference app (host side): how add / remove / move / resize / rotate / replace turn
an ellipse into the three pipeline inputs (fg image, bg image, gs_score) - scripts/blobctrl_app.py:452-620, 661-688, 1095-1126,
762-792 (`app:` below).  Ellipses use the OpenCV `fitEllipse` convention ((xc, yc), (d1, d2), angle_deg).

Pinned: every function marked [pinned] is checked against vectors produced by EXECUTING the reference's own definitions
(tools/make_golden.py::golden_blob_edit extracts them from scripts/blobctrl_app.py with `ast`; fixture tests/golden/blob_edit.json).
Not pinned to OpenCV (a third-party dependency that is absent here), but defined in closed form and tested against hand-checkable
vectors: `ellipse_from_mask` / `fit_ellipse` / `convex_hull` stand in for `cv2.fitEllipse(cv2.convexHull(contours))` (app:382-389) with a
direct least-squares fit that returns exact ellipses exactly; `ellipse_mask` replaces `cv2.ellipse(..., -1, lineType=LINE_AA)` + the
app's `> 0` threshold by the exact "pixel square intersects the filled ellipse" predicate, which can differ from OpenCV's polygon
rasteriser on the outermost ring of pixels.
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


def ellipse_mask(ellipse: Ellipse, height: int, width: int) -> np.ndarray:
    """[replaces cv2.ellipse(..., thickness=-1, lineType=LINE_AA) followed by the app's `> 0` (app:1113-1121, composite_mask_and_image
    app:461-476): every pixel the anti-aliased fill gives ANY coverage is masked.  Measured against the filled masks the reference
    ships (assets/results/demo/*/ori_result_gallery_3.png = cv2.ellipse(.., 255, -1), app:717; tests/golden/blob_edit_cv.npz): the
    masks disagree on 0.26-1.43 % of the ellipse's pixels (the outermost ring: OpenCV fills a polygon approximation)]
    uint8 mask, 255 exactly where the closed pixel square [x - 0.5, x + 0.5] x [y - 0.5, y + 0.5] intersects the filled ellipse.
    Exact, not sampled: the ellipse is mapped to the unit disc, the pixel square to a parallelogram, and the distance from the
    origin to that parallelogram is compared with 1 (closed forms for circles / axis-aligned ellipses: tests/test_blob_edit_cpu.py).
    OpenCV's fill is a polygon approximation of the ellipse with a ~1-pixel anti-aliasing footprint; the two masks can differ on
    the outermost ring of pixels."""
    (xc, yc), (d1, d2), angle = ellipse
    t = math.radians(angle)
    a, b = max(d1 / 2.0, 1e-9), max(d2 / 2.0, 1e-9)
    ct, st = math.cos(t), math.sin(t)
    X, Y = np.meshgrid(np.arange(width, dtype=np.float64) - xc, np.arange(height, dtype=np.float64) - yc)

    def to_disc(x, y):                                  # image offset -> unit-disc coordinates (u along the d1 axis, v along d2)
        return (x * ct + y * st) / a, (-x * st + y * ct) / b
    corners = [to_disc(X + dx, Y + dy) for dx, dy in ((-0.5, -0.5), (0.5, -0.5), (0.5, 0.5), (-0.5, 0.5))]
    cu, cv = to_disc(X, Y)
    inside = cu * cu + cv * cv <= 1.0                   # (a pixel whose centre is inside intersects trivially)
    best = np.full(X.shape, np.inf)
    # origin inside the parallelogram <=> the ellipse centre lies in the pixel square
    origin_in = (np.abs(X) <= 0.5) & (np.abs(Y) <= 0.5)
    for k in range(4):
        (pu, pv), (qu, qv) = corners[k], corners[(k + 1) % 4]
        eu, ev = qu - pu, qv - pv
        tt = np.clip(-(pu * eu + pv * ev) / np.maximum(eu * eu + ev * ev, 1e-300), 0.0, 1.0)
        du, dv = pu + tt * eu, pv + tt * ev
        best = np.minimum(best, du * du + dv * dv)
    cover = inside | origin_in | (best <= 1.0)
    return (cover * 255).astype(np.uint8)


def convex_hull(points: np.ndarray) -> np.ndarray:
    """cv2.convexHull restated (Andrew's monotone chain): hull vertices [K, 2] without collinear points."""
    pts = np.unique(np.asarray(points, dtype=np.float64).reshape(-1, 2), axis=0)
    if len(pts) <= 2:
        return pts

    def half(seq):
        out = []
        for p_ in seq:
            while len(out) >= 2 and ((out[-1][0] - out[-2][0]) * (p_[1] - out[-2][1]) - (out[-1][1] - out[-2][1]) * (p_[0] - out[-2][0])) <= 0:
                out.pop()
            out.append(tuple(p_))
        return out
    lower, upper = half(pts), half(pts[::-1])
    return np.array(lower[:-1] + upper[:-1], dtype=np.float64)


def fit_ellipse(points: np.ndarray) -> Ellipse:
    """cv2.fitEllipse (app:387) restated.  OpenCV (opencv-python, pinned in the reference's requirements.txt; not under /root/reference
    and absent from this image) implements it as `fitEllipseNoDirect` (modules/imgproc/src/shapedescr.cpp, "New fitellipse algorithm,
    contributed by Dr. Daniel Weiss"), three linear least-squares solves on the points centred at their mean:
      1. general conic  -A x^2 - B y^2 - C xy + D x + E y = const  (5 unknowns),
      2. its centre from the gradient equations  [2A C; C 2B] (cx, cy) = (D, E),
      3. re-fit  A (x-cx)^2 + B (y-cy)^2 + C (x-cx)(y-cy) = 1  (3 unknowns),
    then  angle = -atan2(C, B - A) / 2,  t = C / sin(-2 angle) (or B - A when C = 0),  radii = sqrt(2 / |A + B -+ t|),
    RotatedRect convention ((xc, yc), (d1, d2), angle): full axis lengths with d1 <= d2 (swap adds 90 degrees), angle in degrees.
    PINNED to OpenCV's own outputs that the reference ships: tests/golden/blob_edit_cv.npz holds the SAM masks and the fitted
    ellipses of the demo states (assets/results/demo/*/state/state.json, ellipse_lists[0] = fitEllipse(hull) enlarged by 1.05,
    app:382-389,902); centres and axes agree to < 0.01 px and angles to < 0.01 degree on every demo whose first ellipse is the
    unedited fit (tests/test_blob_edit_cpu.py)."""
    P = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    if len(P) < 5:
        raise ValueError("fit_ellipse needs at least 5 points")
    c = P.mean(0)
    p = P - c
    scale = 100.0 / max(float(np.abs(p).sum()), 1e-30)               # conditioning only: the solves are scale-invariant
    p = p * scale
    x, y = p[:, 0], p[:, 1]
    gfp = np.linalg.lstsq(np.stack([-x * x, -y * y, -x * y, x, y], 1), np.full(len(p), 10000.0), rcond=None)[0]
    rp = np.linalg.lstsq(np.array([[2.0 * gfp[0], gfp[2]], [gfp[2], 2.0 * gfp[1]]]), np.array([gfp[3], gfp[4]]), rcond=None)[0]
    if not np.all(np.isfinite(rp)):
        raise ValueError("fit_ellipse: the points do not determine an ellipse")
    qx, qy = x - rp[0], y - rp[1]
    g = np.linalg.lstsq(np.stack([qx * qx, qy * qy, qx * qy], 1), np.ones(len(p)), rcond=None)[0]
    ang = -0.5 * math.atan2(g[2], g[1] - g[0])
    t = g[2] / math.sin(-2.0 * ang) if abs(g[2]) > 1e-8 else g[1] - g[0]
    r = [abs(g[0] + g[1] - t), abs(g[0] + g[1] + t)]
    if min(r) <= 1e-8 or 4.0 * g[0] * g[1] - g[2] * g[2] <= 0:
        raise ValueError("fit_ellipse: the points do not determine an ellipse")
    w, h = (2.0 * math.sqrt(2.0 / v) / scale for v in r)
    angle = math.degrees(ang)
    if w > h:
        w, h = h, w
        angle += 90.0
    if angle < -180.0:
        angle += 360.0
    if angle > 360.0:
        angle -= 360.0
    return ((float(rp[0] / scale + c[0]), float(rp[1] / scale + c[1])), (float(w), float(h)), float(angle))


def ellipse_from_mask(mask: np.ndarray) -> Ellipse:
    """app:382-389 `_get_ellipse`: ellipse fitted to the convex hull of the mask's external contours.  The hull of the contour points
    equals the hull of the set pixels, so no contour tracing is needed."""
    mask = np.asarray(mask)
    ind = mask > 0 if mask.ndim == 2 else mask.sum(-1) > 0
    ys, xs = np.nonzero(ind)
    if len(xs) == 0:
        raise ValueError("empty mask")
    return fit_ellipse(convex_hull(np.stack([xs, ys], 1)))


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
