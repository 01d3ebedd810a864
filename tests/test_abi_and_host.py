"""
CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/lnrf.h declares,
the ctypes prototypes cover the header, host-side logic (parameter trees, keys, dataset format,
camera rays) behaves like the reference, and the product path refuses to run without a GPU.
"""
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lnrf.h")


def header_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lnrf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    from learn_nerf import _lib

    lib = _lib.lib()  # raises if the .so is missing: build() must have run
    syms = header_symbols()
    assert len(syms) >= 25
    for name in syms:
        assert hasattr(lib, name), f"liblnrf.so does not export {name}"
        assert name in _lib.PROTOTYPES, f"no ctypes prototype for {name}"
    assert set(_lib.PROTOTYPES) == set(syms), "prototype table and header disagree"
    assert lib.lnrf_version() == 100


def test_size_queries_without_gpu():
    import ctypes

    from learn_nerf import _lib

    lib = _lib.lib()
    shape = _lib.NerfShape(5, 4, 256, 128, 10, 4)
    assert lib.lnrf_nerf_param_count(ctypes.byref(shape)) == 593_924
    assert lib.lnrf_nerf_packed_bytes(ctypes.byref(shape)) == (1200 + 1120) * 1024 + 10240
    assert lib.lnrf_nerf_save_bytes(ctypes.byref(shape), 4096 * 192) == 167 * 24576 * 1024
    # gradient dump (tiles padded to 8-wave workgroups) + 512 partial-sum slabs of 8 waves x (9 tiles x 4 KiB + 4 bias rows)
    assert lib.lnrf_nerf_bwd_scratch_bytes(ctypes.byref(shape), 33) == 156 * 8 * 1024 + 512 * 8 * (9 * 4096 + 4 * 256)
    other = _lib.NerfShape(5, 4, 128, 128, 10, 4)
    assert lib.lnrf_nerf_param_count(ctypes.byref(other)) > 0
    assert lib.lnrf_nerf_packed_bytes(ctypes.byref(other)) == -1  # fused path: default shape only


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from learn_nerf import ops
    from learn_nerf.model import NeRFModel

    with pytest.raises(RuntimeError, match="GPU"):
        NeRFModel().init(dict(params=0))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.ray_aabb_stratified(torch.zeros(4, 2, 3), (-1, -1, -1), (1, 1, 1), 4)


def test_param_tree_views_and_spec():
    from learn_nerf.model import NeRFModel
    from learn_nerf.params import build_tree, flat_of

    m = NeRFModel()
    spec = m.param_spec()
    assert [s[0] for s in spec[:2]] == ["Dense_0", "Dense_0"] and spec[0][2] == (60, 256)
    assert m.num_params() == 593_924
    flat = torch.arange(m.num_params(), dtype=torch.float32)
    tree = build_tree(flat, spec)
    assert tree["Dense_5"]["kernel"].shape == (316, 256) and tree["Dense_11"]["bias"].shape == (3,)
    tree["Dense_0"]["bias"][0] = -5.0  # a view: writes go to the flat buffer
    assert flat[60 * 256] == -5.0
    assert flat_of(tree, spec) is flat
    plain = {k: dict(v) for k, v in tree.items()}  # e.g. a tree restored from a checkpoint
    assert torch.equal(flat_of(plain, spec), flat)
    gen = torch.Generator().manual_seed(0)
    f2 = torch.zeros(m.num_params())
    m.init_flat_(f2, gen)
    t2 = build_tree(f2, spec)
    k0 = t2["Dense_0"]["kernel"]
    assert abs(k0.std().item() - (1 / 60) ** 0.5) < 0.01  # lecun_normal: var = 1/fan_in
    assert k0.abs().max().item() <= 2.0 / 0.87962566 * (1 / 60) ** 0.5 + 1e-6  # truncated at 2 sigma
    assert (t2["Dense_3"]["bias"] == 0).all()


def test_key_split_is_deterministic():
    from learn_nerf.rng import Key, Uniforms, sampler_args, split

    a, b = split(Key(5), 2)
    a2, b2 = split(5, 2)
    assert (a, b) == (a2, b2) and a.seed != b.seed
    assert Key(5, ray_offset=7).split(2)[0].ray_offset == 7
    u = Uniforms(torch.zeros(2, 3))
    assert sampler_args(u, 0)["u"] is not None and sampler_args(a, 1)["seed"] == a.seed
    with pytest.raises(TypeError):
        split(u, 2)


def _views():
    from learn_nerf.dataset import NeRFView

    class DummyView(NeRFView):
        def __init__(self, img, **kw):
            super().__init__(**kw)
            self._img = img

        def image(self):
            return self._img

    rng = np.random.default_rng(0)
    v1 = DummyView((rng.random((10, 10, 3)) * 255).astype(np.uint8), camera_direction=(0.0, 1.0, 0.0),
                   camera_origin=(2.0, 2.0, 2.0), x_axis=(-1.0, 0.0, 0.0), y_axis=(0.0, 0.0, 1.0), x_fov=1.0,
                   y_fov=1.0)
    v2 = DummyView((rng.random((10, 10, 3)) * 255).astype(np.uint8), camera_direction=(1.0, 0.0, 0.0),
                   camera_origin=(-2.0, 2.0, 2.0), x_axis=(0.0, 0.0, -1.0), y_axis=(0.0, 1.0, 0.0), x_fov=1.0,
                   y_fov=1.0)
    return [v1, v2]


def test_dataset_iterate_batches_invariants(tmp_path):
    """The invariants of the reference's only test (learn_nerf/test_dataset.py:49-81)."""
    from learn_nerf.dataset import ModelMetadata, NeRFDataset

    views = _views()
    ds = NeRFDataset(metadata=ModelMetadata((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), views=views)
    batches = list(ds.iterate_batches(str(tmp_path / "sh"), 1234, batch_size=51, repeat=False))
    assert len(batches) == 4 and batches[-1].shape[0] == 200 - 51 * 3
    combined = torch.cat(batches, 0)
    assert combined.shape == (200, 3, 3)
    for v in views:
        origin = torch.tensor(v.camera_origin)
        sel = (combined[:, 0] - origin).abs().sum(-1) < 1e-5
        assert int(sel.sum()) == 100
        mean_dir = combined[sel][:, 1].mean(0)
        mean_dir = mean_dir / mean_dir.norm()
        assert abs(float((mean_dir * torch.tensor(v.camera_direction)).sum()) - 1) < 1e-5
        mean_col = combined[sel][:, 2].mean(0)
        actual = torch.from_numpy(v.image().astype(np.float32) / 127.5 - 1).mean((0, 1))
        assert float((mean_col - actual).abs().mean()) < 1e-5
    # shards are raw little-endian fp32 [k,3,3] files "0".."31" + "done" (dataset.py:255-263)
    names = sorted(os.listdir(tmp_path / "sh"))
    assert "done" in names and len(names) == 33
    total = sum(os.path.getsize(tmp_path / "sh" / str(i)) for i in range(32))
    assert total == 200 * 9 * 4
    # a second iterator re-uses the shards and repeats forever
    it = ds.iterate_batches(str(tmp_path / "sh"), 1234, batch_size=64, repeat=True)
    assert all(next(it).shape == (64, 3, 3) for _ in range(7))


def test_camera_rays_and_json_roundtrip(tmp_path):
    from learn_nerf.dataset import CameraView, ModelMetadata

    v = CameraView(camera_direction=(0.0, 0.0, -1.0), camera_origin=(0.0, 0.0, 4.0), x_axis=(1.0, 0.0, 0.0),
                   y_axis=(0.0, -1.0, 0.0), x_fov=0.6, y_fov=0.6)
    rays = v.bare_rays(5, 3)
    assert rays.shape == (15, 2, 3) and rays.dtype == torch.float32
    assert torch.allclose(rays[:, 1].norm(dim=-1), torch.ones(15), atol=1e-6)
    centre = rays[7, 1]  # middle pixel looks along the camera direction (dataset.py:59-70)
    assert torch.allclose(centre, torch.tensor([0.0, 0.0, -1.0]), atol=1e-6)
    import math

    corner = rays[0, 1]  # row 0, col 0: -x_axis, -y_axis
    expect = torch.tensor([-math.tan(0.3), math.tan(0.3), -1.0])
    assert torch.allclose(corner, expect / expect.norm(), atol=1e-6)
    p = tmp_path / "0000.json"
    p.write_text(v.to_json())
    v2 = CameraView.from_json(str(p))
    assert v2 == v
    (tmp_path / "metadata.json").write_text(json.dumps({"min": [-1, -1, -1], "max": [1, 1, 1]}))
    md = ModelMetadata.from_json(str(tmp_path / "metadata.json"))
    assert md.bbox_min == (-1, -1, -1) and md.bbox_max == (1, 1, 1)


def test_blender_converter_and_loader_roundtrip(tmp_path):
    """convert_dataset/blender.py semantics (axes :39-41, y_fov :42, bbox :59-60) + optional downscale."""
    import math

    from PIL import Image

    sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd", "learn_nerf", "scripts"))
    import convert_blender
    from learn_nerf.dataset import load_dataset

    src = tmp_path / "blender"
    (src / "train").mkdir(parents=True)
    frames = []
    rng = np.random.default_rng(0)
    for i in range(2):
        img = (rng.random((8, 16, 4)) * 255).astype(np.uint8)
        Image.fromarray(img, "RGBA").save(src / "train" / f"r_{i}.png")
        m = np.eye(4)
        m[:3, 3] = [1.0 + i, 2.0, 3.0]
        frames.append(dict(file_path=f"./train/r_{i}", transform_matrix=m.tolist()))
    (src / "transforms_train.json").write_text(json.dumps(dict(camera_angle_x=0.8, frames=frames)))
    out = tmp_path / "out"
    assert convert_blender.convert(str(src), str(out), resize=8) == 2
    ds = load_dataset(str(out))
    assert ds.metadata.bbox_min == (-1.0, -1.0, -1.0) and len(ds.views) == 2
    v = ds.views[0]
    assert v.camera_origin == (1.0, 2.0, 3.0) and v.x_axis == (1.0, 0.0, 0.0)
    assert v.y_axis == (0.0, -1.0, 0.0) and v.camera_direction == (0.0, 0.0, -1.0)
    assert abs(v.y_fov - 2 * math.atan(math.tan(0.4) * 8 / 16)) < 1e-12
    assert v.image().shape == (4, 8, 3)  # downscaled 16x8 -> 8x4
    assert v.rays().shape == (32, 3, 3)
    with pytest.raises(FileExistsError):
        convert_blender.convert(str(src), str(out))


def test_z_depth_and_random_view_match_reference_formulas():
    """scripts/render_new_dataset.py:78-116 restated with NumPy: z-depth normalisation and the pose sampler."""
    import numpy as np
    import torch

    from learn_nerf.render import z_depth
    from learn_nerf.scripts.render_new_dataset import random_view

    gen = np.random.RandomState(0)
    n, max_depth = 200, 7.0
    coords = gen.normal(size=(n, 3)).astype(np.float32) * 3
    alphas = gen.uniform(0.5, 1.0, size=(n, 1)).astype(np.float32)
    alphas[:5] = 0.9  # boundary: not a hit (strict >)
    origin, direction = (0.5, -1.0, 2.0), (0.0, 0.6, -0.8)
    got = z_depth(torch.from_numpy(coords), torch.from_numpy(alphas), origin, direction, max_depth).numpy()
    along = ((coords - np.array(origin, np.float32)) @ np.array(direction, np.float32))[:, None] / (alphas + 1e-8)
    want = np.clip(np.where(alphas > 0.9, along, max_depth), 0.0, max_depth) / max_depth
    assert got.shape == (n, 1) and np.abs(got - want).max() < 1e-6
    assert (got[:5] == 1.0).all() and got.min() >= 0.0 and got.max() <= 1.0
    # 16-bit quantisation used by the script: truncation of depth * 0xFFFF
    q = (got.reshape(-1) * 0xFFFF).astype(np.uint32)
    assert q.max() == 0xFFFF and q.dtype == np.uint32

    center, radius = np.array([0.1, 0.2, -0.3]), 3.5
    view = random_view(np.random.RandomState(4), center, radius)
    z, x, y = (np.array(v) for v in (view.camera_direction, view.x_axis, view.y_axis))
    assert abs(np.linalg.norm(z) - 1) < 1e-12 and abs(np.linalg.norm(x) - 1) < 1e-12 and x[2] == 0.0
    assert abs(z @ x) < 1e-12 and np.abs(np.cross(z, x) - y).max() < 1e-12
    assert np.abs(np.array(view.camera_origin) - (center - z * radius)).max() < 1e-12  # looks at the centre
    assert view.x_fov == view.y_fov == 60.0 * np.pi / 180


def test_pan_and_spin_camera_paths():
    """scripts/render_nerf_pan.py:22-52 and render_nerf_spin.py:22-33 restated with explicit formulas."""
    import math

    import numpy as np

    from learn_nerf.dataset import CameraView
    from learn_nerf.scripts.camera_paths import pan_views, spin_views

    lo, hi = (-1.0, -2.0, 0.0), (1.0, 2.0, 1.0)
    views = list(pan_views(lo, hi, 8, 2.0))
    center, diag = (np.array(lo) + np.array(hi)) / 2, np.linalg.norm(np.array(lo) - np.array(hi))
    assert len(views) == 8
    for i, v in enumerate(views):
        th = 2 * math.pi * i / 8
        z = np.array([math.cos(th), -math.sin(th), 0.0])  # e1 = x, e2 = (0,0,-1) x (1,0,0) = (0,-1,0)
        assert np.abs(np.array(v.camera_direction) - z).max() < 1e-12
        assert np.abs(np.array(v.camera_origin) - (center - z * diag * 2.0)).max() < 1e-12
        assert v.y_axis == (0.0, 0.0, -1.0) and abs(np.dot(v.x_axis, v.camera_direction)) < 1e-12
        assert v.x_fov == v.y_fov == math.pi / 3
    custom = list(pan_views(lo, hi, 4, 1.0, axis=(0.0, 3.0, 4.0)))
    assert np.abs(np.array(custom[0].y_axis) - np.array([0.0, 0.6, 0.8])).max() < 1e-12
    assert np.abs(np.array(custom[0].camera_direction) - np.array([-1.0, 0.0, 0.0])).max() < 1e-12
    base = CameraView(camera_direction=(0.0, 0.0, 1.0), camera_origin=(1.0, 2.0, 3.0), x_axis=(1.0, 0.0, 0.0),
                      y_axis=(0.0, 1.0, 0.0), x_fov=0.5, y_fov=0.4)
    spun = list(spin_views(base, 4))
    assert spun[0].x_axis == (1.0, 0.0, 0.0) and spun[0].camera_direction == (0.0, 0.0, 1.0)
    assert np.abs(np.array(spun[1].x_axis) - np.array([0.0, 0.0, 1.0])).max() < 1e-12      # quarter turn: x -> z
    assert np.abs(np.array(spun[1].camera_direction) - np.array([-1.0, 0.0, 0.0])).max() < 1e-12
    assert all(v.camera_origin == base.camera_origin and v.y_axis == base.y_axis and v.x_fov == 0.5 for v in spun)
    assert base.x_axis == (1.0, 0.0, 0.0)  # the input view is not modified
