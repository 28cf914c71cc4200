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
