"""
The data edge of the hot path (dataset.py:52-111, scripts/render_nerf.py:85-101) — oracle/dataset.py pinned
by closed-form cases and by the invariants of the reference's own test (learn_nerf/test_dataset.py:49-81), and the
package's host path (learn_nerf.dataset, scripts/render_nerf.py) checked against that oracle.  CPU only.
"""
import math

import numpy as np
import torch

from oracle import dataset as OD


def _camera():
    z = np.array([0.3, -0.5, 0.81])
    z /= np.linalg.norm(z)
    x = np.cross(z, [0.0, 0.0, 1.0])
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return dict(camera_direction=tuple(z), camera_origin=(1.5, -2.0, 0.7), x_axis=tuple(x), y_axis=tuple(y), x_fov=0.69,
                y_fov=0.43)


def test_oracle_bare_rays_closed_form():
    # axis-aligned camera looking down +z, 90 degree fields of view: corner pixels point at (+-1, +-1, 1) / sqrt(3),
    # row 0 is -y_axis, column 0 is -x_axis, raster order is row-major (y outer)
    rays = OD.bare_rays((0, 0, 1), (0.5, 0.25, -1.0), (1, 0, 0), (0, 1, 0), math.pi / 2, math.pi / 2, 3, 5)
    assert rays.shape == (15, 2, 3) and rays.dtype == np.float32
    assert np.array_equal(rays[:, 0], np.tile(np.float32([0.5, 0.25, -1.0]), (15, 1)))
    s3 = 1 / math.sqrt(3)
    assert np.allclose(rays[0, 1], [-s3, -s3, s3], atol=1e-6) and np.allclose(rays[2, 1], [s3, -s3, s3], atol=1e-6)
    assert np.allclose(rays[14, 1], [s3, s3, s3], atol=1e-6) and np.allclose(rays[7, 1], [0, 0, 1], atol=1e-6)
    assert np.allclose(np.linalg.norm(rays[:, 1], axis=-1), 1, atol=1e-6)
    # W = 1 / H = 1: linspace(-1, 1, 1) = [-1] (the reference's end-point-inclusive grid degenerates to the left / top edge)
    one = OD.bare_rays((0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0), math.pi / 2, math.pi / 2, 1, 1)
    assert np.allclose(one[0, 1], [-s3, -s3, s3], atol=1e-6)


def test_oracle_colour_and_quantisation_known_answers():
    img = np.array([[[0, 127, 255], [128, 1, 254]]], dtype=np.uint8)
    bare = OD.bare_rays((0, 0, 1), (0, 0, 0), (1, 0, 0), (0, 1, 0), 1.0, 1.0, 2, 1)
    rows = OD.rays_with_colors(bare, img)
    assert rows.shape == (2, 3, 3)
    assert np.allclose(rows[:, 2], [[-1, 127 / 127.5 - 1, 1], [128 / 127.5 - 1, 1 / 127.5 - 1, 254 / 127.5 - 1]], atol=1e-7)
    # alpha premultiply: round half to even, as jnp.round
    rgba = np.array([[[255, 100, 1, 255], [255, 100, 1, 128], [51, 85, 3, 5], [200, 200, 200, 0]]], dtype=np.uint8)
    pm = OD.premultiply_alpha(rgba)
    assert pm.tolist() == [[[255, 100, 1], [128, 50, 1], [1, 2, 0], [0, 0, 0]]]  # 100 * 128 / 255 = 50.196; 85 * 5 / 255 = 1.667
    # truncating uint8 (render_nerf.py:93-96): -1 -> 0, 1 -> 255, 0 -> 127 (127.5 truncated), just below 1 -> 254
    q = OD.quantise_pixels(np.array([[-1.0, 0.0, 1.0], [0.999, -0.999, 0.00784]]), 1, 2)
    assert q.tolist() == [[[0, 127, 255], [254, 0, 128]]]


def test_package_host_path_matches_oracle(tmp_path):
    from PIL import Image

    from learn_nerf.dataset import CameraView, FileNeRFView

    cam = _camera()
    for w, h in ((1, 1), (1, 7), (9, 1), (33, 17)):
        want = OD.bare_rays(width=w, height=h, **cam)
        got = CameraView(**cam).bare_rays(w, h)
        assert isinstance(got, torch.Tensor) and got.shape == want.shape
        assert np.abs(got.numpy() - want).max() < 1e-6
    gen = np.random.default_rng(0)
    rgba = gen.integers(0, 256, size=(6, 5, 4), dtype=np.uint8)
    path = str(tmp_path / "v.png")
    Image.fromarray(rgba, "RGBA").save(path)
    view = FileNeRFView(image_path=path, **cam)
    assert np.array_equal(view.image(), OD.premultiply_alpha(rgba))
    want_rows = OD.rays_with_colors(OD.bare_rays(width=5, height=6, **cam), OD.premultiply_alpha(rgba))
    got_rows = view.rays().numpy()
    assert got_rows.shape == want_rows.shape == (30, 3, 3)
    assert np.abs(got_rows - want_rows).max() < 1e-6 and np.array_equal(got_rows[:, 2], want_rows[:, 2])


def test_render_session_quantisation_matches_oracle():
    """scripts/render_nerf.py's image assembly (RenderSession.render_view's last step) against the oracle's truncation."""
    import inspect

    from learn_nerf.scripts import render_nerf

    gen = np.random.default_rng(1)
    colors = gen.uniform(-1, 1, size=(12, 3)).astype(np.float32)
    colors[0] = [-1, 0, 1]
    want = OD.quantise_pixels(colors, 3, 4)
    got = render_nerf.quantise(colors.reshape(3, 4, 3))
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    assert "quantise(" in inspect.getsource(render_nerf.RenderSession.render_view)
