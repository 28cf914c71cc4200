"""Blob-editing geometry (host side) against vectors produced by executing the reference app's own function
definitions (tests/golden/blob_edit.json, tools/make_golden.py::golden_blob_edit).  The ellipse rasteriser replaces OpenCV and is
checked through properties instead (it is documented as unpinned)."""
import json
import os

import numpy as np
import pytest

from blobctrl_amd import blob_edit as be


@pytest.fixture(scope="module")
def z(golden_dir):
    return json.load(open(os.path.join(golden_dir, "blob_edit.json")))


def _ell(e):
    return (tuple(e[0]), tuple(e[1]), e[2])


def test_pure_geometry_matches_reference(z):
    ells = [_ell(e) for e in z["ellipses"]]
    for e, nrm, vert, ins, mv in zip(ells, z["normalize"], z["vertices"], z["inside"], z["move"]):
        np.testing.assert_allclose(be.normalize_ellipse(e, 512, 384), nrm, rtol=1e-12)
        np.testing.assert_allclose(be.calculate_ellipse_vertices(e), vert, rtol=1e-12, atol=1e-12)
        for x, y, inside in ins:
            assert be.is_point_in_ellipse((x, y), e) == inside
        m = be.move_ellipse(e, mv["points"])
        np.testing.assert_allclose([*m[0], *m[1], m[2]], [*mv["out"][0], *mv["out"][1], mv["out"][2]], rtol=1e-12)
    for r in z["resize"]:
        out, f, big, small = be.resize_blob(ells[r["ellipse"]], r["factor"], 512, 512, r["type"])
        np.testing.assert_allclose([*out[0], *out[1], out[2]], [*r["out"][0], *r["out"][1], r["out"][2]], rtol=1e-12)
        assert f == pytest.approx(r["factor_out"], rel=1e-12) and big == r["too_big"] and small == r["too_small"]
    for r in z["rotate"]:
        out, _ = be.rotate_blob(ells[r["ellipse"]], r["deg"])
        assert out[2] == pytest.approx(r["out"][2], abs=1e-12) and out[:2] == (tuple(r["out"][0]), tuple(r["out"][1]))
    with pytest.raises(ValueError):
        be.resize_blob(ells[0], 1.5, 512, 512, 3)


def test_compositing_matches_reference(z):
    img, mask = np.array(z["image"], np.uint8), np.array(z["mask"], np.uint8)
    assert np.array_equal(be.composite_mask_and_image(mask, img, (255, 255, 255)), np.array(z["composite_white"], np.uint8))
    mask3 = np.stack([mask, mask, np.zeros_like(mask)], -1)
    assert np.array_equal(be.composite_mask_and_image(mask3, img, (0, 0, 0)), np.array(z["composite_rgbmask_black"], np.uint8))
    assert np.array_equal(be.object_region_from_mask(mask, img), np.array(z["object_region"], np.uint8))
    with pytest.raises(ValueError):
        be.object_region_from_mask(np.zeros_like(mask), img)


def test_ellipse_mask_properties():
    """The OpenCV-free rasteriser: centre and axis end points inside, area close to pi a b, consistent with is_point_in_ellipse."""
    e = ((40.0, 30.0), (36.0, 20.0), 25.0)
    m = be.ellipse_mask(e, 64, 80)
    assert m.dtype == np.uint8 and m.shape == (64, 80) and set(np.unique(m)) <= {0, 255}
    assert m[30, 40] == 255
    area = (m > 0).sum()
    exact = np.pi * 18.0 * 10.0
    assert exact <= area <= exact * 1.25                     # "touches the pixel" over-covers by about one boundary ring
    # pixels whose centre is inside are always set
    ys, xs = np.mgrid[0:64, 0:80]
    t = np.radians(25.0)
    u = (xs - 40.0) * np.cos(t) + (ys - 30.0) * np.sin(t)
    v = -(xs - 40.0) * np.sin(t) + (ys - 30.0) * np.cos(t)
    centre_in = (u / 18.0) ** 2 + (v / 10.0) ** 2 <= 1.0
    assert np.all(m[centre_in] == 255)
    assert be.ellipse_mask(((40.0, 30.0), (1e-5, 1e-5), 0.0), 64, 80).sum() <= 255 * 1          # the degenerate `add` start ellipse


def test_ellipse_mask_matches_closed_forms_for_circles_and_axis_aligned_ellipses():
    """Hand-checkable closed form: for an axis-aligned ellipse the pixel square nearest point to the centre is (clamp(|dx| - 0.5, 0),
    clamp(|dy| - 0.5, 0)) in each axis independently (scaling x by 1/a and y by 1/b keeps the square a box), so the pixel is masked
    iff (ex / a)^2 + (ey / b)^2 <= 1.  OpenCV (cv2.ellipse LINE_AA + `> 0`, app:1113-1121) is absent in this image: these vectors pin
    the definition 'every pixel the filled ellipse touches', not OpenCV's polygon approximation."""
    H, W = 48, 64
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    for (xc, yc, d1, d2, ang) in [(30.0, 20.0, 18.0, 18.0, 0.0), (31.5, 22.25, 25.0, 9.0, 0.0), (12.0, 40.0, 7.0, 30.0, 90.0),
                                  (63.0, 0.0, 10.0, 14.0, 180.0), (20.3, 20.7, 1e-5, 1e-5, 0.0)]:
        a, b = (d1 / 2, d2 / 2) if ang % 180.0 == 0.0 else (d2 / 2, d1 / 2)          # 90 degrees swaps the axes
        ex = np.maximum(np.abs(xx - xc) - 0.5, 0.0) / max(a, 1e-9)
        ey = np.maximum(np.abs(yy - yc) - 0.5, 0.0) / max(b, 1e-9)
        ref = ((ex * ex + ey * ey <= 1.0) * 255).astype(np.uint8)
        got = be.ellipse_mask(((xc, yc), (d1, d2), ang), H, W)
        assert np.array_equal(got, ref), (xc, yc, d1, d2, ang, int((got != ref).sum()))
    # rotated ellipse: contains every pixel whose CENTRE is inside, stays within a one-pixel dilation of that set, 180-degree symmetric
    e = ((31.5, 23.5), (40.0, 12.0), 33.0)
    m = be.ellipse_mask(e, H, W) > 0
    t = np.radians(33.0)
    u = (xx - 31.5) * np.cos(t) + (yy - 23.5) * np.sin(t)
    v = -(xx - 31.5) * np.sin(t) + (yy - 23.5) * np.cos(t)
    centre_in = (u / 20.0) ** 2 + (v / 6.0) ** 2 <= 1.0
    assert (m | ~centre_in).all()
    dil = np.zeros_like(centre_in)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            dil |= np.roll(np.roll(centre_in, dy, 0), dx, 1)
    assert (dil | ~m).all() and m.sum() > centre_in.sum()
    assert np.array_equal(m, m[::-1, ::-1])                                           # (31.5, 23.5) is the centre of symmetry of the 48 x 64 pixel grid


def test_fit_ellipse_recovers_exact_ellipses_and_mask_hulls():
    """cv2.fitEllipse stand-in (app:382-389): points ON an ellipse give that ellipse back, in OpenCV's ((xc, yc), (d1 <= d2), angle of
    the d1 axis in [0, 180)) convention; the ellipse fitted to the hull of a rasterised ellipse mask reproduces it to a pixel."""
    for (xc, yc, d1, d2, ang) in [(361.1067, 367.8526, 85.4812, 103.6543, 87.3739), (100.0, 90.0, 60.0, 200.0, 10.0),
                                  (256.0, 256.0, 120.0, 121.0, 135.0), (40.5, 470.25, 40.0, 300.0, 179.0)]:
        th = np.linspace(0, 2 * np.pi, 23, endpoint=False) + 0.1
        t = np.radians(ang)
        u, v = d1 / 2 * np.cos(th), d2 / 2 * np.sin(th)
        pts = np.stack([xc + u * np.cos(t) - v * np.sin(t), yc + u * np.sin(t) + v * np.cos(t)], 1)
        (fx, fy), (f1, f2), fa = be.fit_ellipse(pts)
        np.testing.assert_allclose([fx, fy, f1, f2], [xc, yc, d1, d2], rtol=1e-9, atol=1e-7)
        assert f1 <= f2 and 0.0 <= fa < 180.0 and min(abs(fa - ang), 180.0 - abs(fa - ang)) < 1e-6
        # same orientation convention as the app's calculate_ellipse_vertices (app:502-532) and cv2.ellipse: the four axis end
        # points of the FITTED ellipse lie on the ORIGINAL one
        for vx, vy in be.calculate_ellipse_vertices(((fx, fy), (f1, f2), fa)):
            uu = (vx - xc) * np.cos(t) + (vy - yc) * np.sin(t)
            vv = -(vx - xc) * np.sin(t) + (vy - yc) * np.cos(t)
            assert abs((uu / (d1 / 2)) ** 2 + (vv / (d2 / 2)) ** 2 - 1.0) < 1e-6
    e = ((120.0, 96.0), (70.0, 110.0), 25.0)
    mask = be.ellipse_mask(e, 200, 240)
    (fx, fy), (f1, f2), fa = be.ellipse_from_mask(mask)
    assert abs(fx - 120.0) < 0.6 and abs(fy - 96.0) < 0.6 and abs(f1 - 70.0) < 2.0 and abs(f2 - 110.0) < 2.0 and abs(fa - 25.0) < 1.5
    hull = be.convex_hull(np.array([[0, 0], [4, 0], [2, 0], [4, 4], [0, 4], [2, 2], [1, 3]]))
    assert sorted(map(tuple, hull)) == [(0.0, 0.0), (0.0, 4.0), (4.0, 0.0), (4.0, 4.0)]
    with pytest.raises(ValueError):
        be.fit_ellipse(np.zeros((4, 2)))


def _cv_fixture(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "blob_edit_cv.npz"))
    out = {}
    for name in [str(n) for n in z["names"]]:
        H, W = [int(v) for v in z[f"{name}_shape"]]
        unpack = lambda k: np.unpackbits(z[f"{name}_{k}"])[: H * W].reshape(H, W).astype(bool)
        e = z[f"{name}_ellipse0"]
        out[name] = dict(mask=unpack("mask"), filled=unpack("filled"), ellipse=((e[0], e[1]), (e[2], e[3]), e[4]))
    return out, float(z["enlarge_factor"])


# demos whose first state ellipse is the unedited fit x 1.05 (the others were enlarged further by the app's blob-resize walk or
# edited before the state was saved: their centre and angle still match, their axes carry another factor)
FIT_PINNED = ["move_hat", "remove_cow", "replace_knife", "resize_teddy_bear"]


def test_ellipse_from_mask_matches_the_opencv_fits_the_reference_ships(golden_dir):
    """f4 pinned to reference-held OpenCV outputs: SAM mask -> convex hull -> fitEllipse against the ellipses in
    assets/results/demo/*/state/state.json (cv2.fitEllipse(cv2.convexHull(contours)) x 1.05, app:382-389,902)."""
    cases, factor = _cv_fixture(golden_dir)
    for name in FIT_PINNED:
        (xc, yc), (d1, d2), ang = be.ellipse_from_mask(cases[name]["mask"])
        (rx, ry), (r1, r2), ra = cases[name]["ellipse"]
        assert abs(xc - rx) < 0.01 and abs(yc - ry) < 0.01, (name, xc, yc, rx, ry)
        assert abs(d1 * factor - r1) < 0.01 and abs(d2 * factor - r2) < 0.01, (name, d1 * factor, d2 * factor, r1, r2)
        assert abs(ang - ra) < 0.01, (name, ang, ra)
    # enlarged / edited states: same centre and orientation (the fit), other axis factors (measured 1.09 - 1.26, per axis for shrink_dragon)
    for name in ("enlarge_deer", "remove_shit", "shrink_dragon"):
        (xc, yc), (d1, d2), ang = be.ellipse_from_mask(cases[name]["mask"])
        (rx, ry), (r1, r2), ra = cases[name]["ellipse"]
        assert abs(xc - rx) < 0.01 and abs(yc - ry) < 0.01 and abs(ang - ra) < 0.01, name
        assert 1.04 < r1 / d1 < 1.3 and 1.04 < r2 / d2 < 1.3, (name, r1 / d1, r2 / d2)
    # move_cup: the shipped mask image is not the one the state was fitted on (centre 0.25 / 0.41 px off, angle 1.03 degrees across
    # the 0 / 180 wrap: OpenCV reports 0.83 where the d1 <= d2 normalisation of this mask gives 179.80) - documented, not pinned
    (xc, yc), (d1, d2), ang = be.ellipse_from_mask(cases["move_cup"]["mask"])
    (rx, ry), (r1, r2), ra = cases["move_cup"]["ellipse"]
    assert abs(xc - rx) < 0.5 and abs(yc - ry) < 0.5 and abs(d1 * factor - r1) < 1.0 and abs(d2 * factor - r2) < 1.0
    assert min(abs(ang - ra), 180.0 - abs(ang - ra)) < 1.5


def test_ellipse_mask_against_the_reference_held_cv2_ellipse_fills(golden_dir):
    """`ellipse_mask` (any-coverage rule) against cv2.ellipse(mask, ellipse, 255, -1) (app:717) as shipped in
    ori_result_gallery_3.png: disagreement in % of the ellipse's pixels, per case (measured 0.26 - 1.43; bound 1.5)."""
    cases, _ = _cv_fixture(golden_dir)
    worst = 0.0
    for name in FIT_PINNED + ["remove_shit", "move_cup"]:
        c = cases[name]
        mine = be.ellipse_mask(c["ellipse"], *c["filled"].shape) > 0
        pct = 100.0 * float((mine != c["filled"]).sum()) / float(c["filled"].sum())
        worst = max(worst, pct)
        assert pct < 1.5, (name, pct)
    assert worst > 0.0                      # (not the same rasteriser: documented deviation, not bit equality)
