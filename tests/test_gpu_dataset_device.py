"""
SURVEY.md section 8 row f1 on the GPU: ray generation (CameraView.bare_rays, dataset.py:52-78) by the HIP kernel
lnrf_camera_rays and the shuffled batch iterator (ShuffledDataset.iterate_batches, dataset.py:222-240) with the
shards resident in HBM and batches assembled by lnrf_gather_rows.  Ray generation is checked against oracle/dataset.py
(the CPU restatement of dataset.py:52-78, pinned in tests/test_dataset_oracle.py); the iterator against the host iterator
(bit-identical) and against the invariants of the reference's own test (learn_nerf/test_dataset.py:49-81).
"""
import numpy as np
import pytest
import torch

from test_abi_and_host import _views

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("width,height", [(1, 1), (1, 7), (9, 1), (33, 17), (400, 400)])
def test_camera_rays_kernel_matches_oracle(width, height):
    from learn_nerf.dataset import CameraView
    from oracle import dataset as OD

    z = np.array([0.3, -0.5, 0.81])
    z /= np.linalg.norm(z)
    x = np.cross(z, [0.0, 0.0, 1.0])
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    view = CameraView(camera_direction=tuple(z), camera_origin=(1.5, -2.0, 0.7), x_axis=tuple(x), y_axis=tuple(y),
                      x_fov=0.69, y_fov=0.43)
    host = torch.from_numpy(OD.bare_rays(tuple(z), (1.5, -2.0, 0.7), tuple(x), tuple(y), 0.69, 0.43, width, height))
    dev = view.bare_rays(width, height, device="cuda")
    assert dev.is_cuda and dev.shape == host.shape == (width * height, 2, 3) and dev.dtype == torch.float32
    assert torch.equal(dev[:, 0].cpu(), host[:, 0])  # origins are copied
    err = (dev[:, 1].cpu() - host[:, 1]).abs().max().item()
    print(f"{width}x{height}: max |dir - oracle| = {err:.2e}")
    assert err < 1e-6  # raster order, end-point-inclusive linspace, W = 1 / H = 1 -> -1
    assert (dev[:, 1].norm(dim=-1) - 1).abs().max().item() < 1e-6


@pytest.mark.parametrize("batch_size,repeat", [(51, False), (64, True), (200, False), (7, False)])
def test_device_iterator_yields_the_host_batches(tmp_path, batch_size, repeat):
    from learn_nerf.dataset import ModelMetadata, NeRFDataset

    ds = NeRFDataset(metadata=ModelMetadata((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), views=_views())
    shards = str(tmp_path / "sh")
    host_it = ds.iterate_batches(shards, 1234, batch_size=batch_size, repeat=repeat)
    dev_it = ds.iterate_batches(shards, 1234, batch_size=batch_size, repeat=repeat, device="cuda")
    count = 0
    for _ in range(12 if repeat else 10 ** 6):  # repeat: three epochs' worth, crossing epoch boundaries
        h = next(host_it, None)
        d = next(dev_it, None)
        if h is None or d is None:
            assert h is None and d is None
            break
        assert d.is_cuda and d.shape == h.shape and d.dtype == torch.float32
        assert torch.equal(d.cpu(), h), "device gather must reproduce the host iterator bit for bit"
        count += 1
    assert count == (12 if repeat else -(-200 // batch_size))


def test_device_iterator_invariants(tmp_path):
    """The invariants of the reference's own test (learn_nerf/test_dataset.py:49-81) on the device path."""
    from learn_nerf.dataset import ModelMetadata, NeRFDataset

    views = _views()
    ds = NeRFDataset(metadata=ModelMetadata((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)), views=views)
    batches = list(ds.iterate_batches(str(tmp_path / "sh"), 99, batch_size=51, repeat=False, device="cuda"))
    assert len(batches) == 4 and batches[-1].shape[0] == 200 - 51 * 3 and all(b.is_cuda for b in batches)
    combined = torch.cat(batches, 0).cpu()
    for v in views:
        sel = (combined[:, 0] - torch.tensor(v.camera_origin)).abs().sum(-1) < 1e-5
        assert int(sel.sum()) == 100
        mean_col = combined[sel][:, 2].mean(0)
        actual = torch.from_numpy(v.image().astype(np.float32) / 127.5 - 1).mean((0, 1))
        assert float((mean_col - actual).abs().mean()) < 1e-5
    # every ray exactly once
    rows = {tuple(r.reshape(-1).tolist()) for r in combined}
    assert len(rows) == 200


def test_gather_rows_out_of_range_index_writes_zeros():
    from learn_nerf import ops

    src = torch.arange(45, dtype=torch.float32, device="cuda").view(5, 9)
    idx = torch.tensor([4, -1, 0, 5, 2], dtype=torch.int32, device="cuda")
    out = torch.full((5, 9), 7.0, device="cuda")
    ops.gather_rows_into(src, idx, out)
    assert torch.equal(out[0], src[4]) and torch.equal(out[2], src[0]) and torch.equal(out[4], src[2])
    assert (out[1] == 0).all() and (out[3] == 0).all()
