"""
learn_nerf.dataset — the reference's data layer (learn_nerf/dataset.py) for the HIP train step:
CameraView, NeRFView, FileNeRFView, ModelMetadata, NeRFDataset, ShuffledDataset, load_dataset.

File formats are the reference's: one PNG + one JSON {origin, x, y, z, x_fov, y_fov} per view, a global
metadata.json {min, max}, and the on-disk two-stage shuffle — `num_shards` files named "0".."N-1" holding raw
little-endian fp32 rows of 9 floats (origin, direction, colour) plus a "done" marker — so dataset and shard
directories can be shared with the JAX implementation.  Keys are int seeds instead of jax PRNG keys and
batches are torch tensors.  Ray generation also exists as a HIP kernel (bare_rays(..., device=cuda)).
"""
import json
import math
import os
from dataclasses import dataclass, fields
from typing import Iterator, List, Tuple

import numpy as np
import torch

from .params import split_seed

Vec3 = Tuple[float, float, float]

# JSON key -> CameraView attribute (dataset.py:26-50)
_CAMERA_JSON = (("z", "camera_direction"), ("origin", "camera_origin"), ("x", "x_axis"), ("y", "y_axis"))


def _seed_of(key) -> int:
    return int(getattr(key, "seed", key)) & 0xFFFFFFFFFFFFFFFF


@dataclass
class CameraView:
    camera_direction: Vec3
    camera_origin: Vec3
    x_axis: Vec3
    y_axis: Vec3
    x_fov: float
    y_fov: float

    @classmethod
    def from_json(cls, path: str, **kwargs) -> "CameraView":
        with open(path, "rb") as handle:
            doc = json.load(handle)
        values = {attr: tuple(doc[name]) for name, attr in _CAMERA_JSON}
        values.update(x_fov=float(doc["x_fov"]), y_fov=float(doc["y_fov"]))
        return cls(**values, **kwargs)

    def to_json(self) -> str:
        doc = {name: getattr(self, attr) for name, attr in _CAMERA_JSON}
        doc.update(x_fov=self.x_fov, y_fov=self.y_fov)
        return json.dumps(doc)

    def bare_rays(self, width: int, height: int, device=None) -> torch.Tensor:
        """
        Every ray of the view in raster-scan order as [width*height, 2, 3] (origin, unit direction).
        Pixel (col, row) looks along z + tan(x_fov/2)*u*x_axis + tan(y_fov/2)*v*y_axis with u, v sampled by
        linspace(-1, 1, ·) including both ends (dataset.py:52-78).  On a CUDA/ROCm `device` the HIP kernel
        lnrf_camera_rays produces the tensor in place on the GPU.
        """
        if device is not None and torch.device(device).type == "cuda":
            from . import ops

            return ops.camera_rays(self.camera_origin, self.x_axis, self.y_axis, self.camera_direction, self.x_fov,
                                   self.y_fov, width, height, torch.device(device))
        f32 = np.float32
        u = np.linspace(-1, 1, num=width, dtype=f32) * f32(math.tan(self.x_fov / 2))
        v = np.linspace(-1, 1, num=height, dtype=f32) * f32(math.tan(self.y_fov / 2))
        grid = (u[None, :, None] * np.asarray(self.x_axis, dtype=f32)
                + v[:, None, None] * np.asarray(self.y_axis, dtype=f32)
                + np.asarray(self.camera_direction, dtype=f32)).reshape(-1, 3)
        grid /= np.linalg.norm(grid, axis=-1, keepdims=True)
        out = np.empty((grid.shape[0], 2, 3), dtype=f32)
        out[:, 0] = np.asarray(self.camera_origin, dtype=f32)
        out[:, 1] = grid
        return torch.from_numpy(out)


@dataclass
class NeRFView(CameraView):
    """A camera with pixels: subclasses provide image() -> uint8 [H, W, 3]."""

    def image(self) -> np.ndarray:
        raise NotImplementedError

    def rays(self) -> torch.Tensor:
        """[H*W, 3, 3] rows of (origin, direction, colour), colours mapped to [-1, 1] (dataset.py:89-101)."""
        pixels = np.asarray(self.image())
        height, width = pixels.shape[:2]
        geometry = self.bare_rays(width, height).numpy()
        colours = pixels.reshape(-1, 1, 3).astype(np.float32) / 127.5 - 1
        return torch.from_numpy(np.concatenate([geometry, colours], axis=1).astype(np.float32))


@dataclass
class FileNeRFView(NeRFView):
    image_path: str = None

    def image(self) -> np.ndarray:
        from PIL import Image

        rgba = np.array(Image.open(self.image_path).convert("RGBA"))
        # alpha is pre-multiplied so that transparent borders become black, not garbage (dataset.py:108-111)
        return np.round(rgba[:, :, :3] * (rgba[:, :, 3:] / 255)).astype(np.uint8)


@dataclass
class ModelMetadata:
    bbox_min: Vec3  # scene bounding box
    bbox_max: Vec3

    @classmethod
    def from_json(cls, path: str) -> "ModelMetadata":
        with open(path, "rb") as handle:
            doc = json.load(handle)
        return cls(bbox_min=tuple(doc["min"]), bbox_max=tuple(doc["max"]))


@dataclass
class NeRFDataset:
    metadata: ModelMetadata
    views: List[NeRFView]

    def iterate_batches(self, dir_path: str, key, batch_size: int, repeat: bool = True,
                        num_shards: int = 32) -> Iterator[torch.Tensor]:
        """
        Shuffled [batch_size, 3, 3] batches of coloured rays (dataset.py:134-159).  `dir_path` holds the
        shard files (created on first use); with repeat=False the last batch may be short.
        """
        with ShuffledDataset(dir_path, self, key, num_shards=num_shards) as shuffled:
            yield from shuffled.iterate_batches(batch_size, repeat=repeat)


class ShuffledDataset:
    """
    Two-stage on-disk shuffle (dataset.py:162-263): stage 1 deals every ray to a random shard file, stage 2
    (per epoch) visits the shards in random order and permutes each in memory.
    """

    ROW_FLOATS = 9

    def __init__(self, dir_path: str, dataset: NeRFDataset, key, num_shards: int = 32):
        self.num_shards = num_shards
        deal_seed, self._epoch_seed = split_seed(_seed_of(key), 2)
        os.makedirs(dir_path, exist_ok=True)
        marker = os.path.join(dir_path, "done")
        paths = [os.path.join(dir_path, str(i)) for i in range(num_shards)]
        if os.path.exists(marker):
            self.fds = [open(p, "rb") for p in paths]
        else:
            self.fds = [open(p, "wb+") for p in paths]
            self._deal(dataset, np.random.default_rng(deal_seed))
            with open(marker, "wb+") as handle:
                handle.write(b"done\n")

    def _deal(self, dataset: NeRFDataset, rng: np.random.Generator):
        for view in dataset.views:
            rows = view.rays().numpy().reshape(-1, self.ROW_FLOATS).astype("<f4")
            owner = rng.integers(0, self.num_shards, size=rows.shape[0])
            for shard in np.unique(owner):
                self.fds[shard].write(rows[owner == shard].tobytes())
        for fd in self.fds:
            fd.flush()

    def _shard(self, index: int) -> np.ndarray:
        fd = self.fds[index]
        fd.seek(0)
        return np.frombuffer(fd.read(), dtype="<f4").reshape(-1, 3, 3)

    def iterate_batches(self, batch_size: int, repeat: bool = False) -> Iterator[torch.Tensor]:
        rng = np.random.default_rng(self._epoch_seed)
        pending = np.empty((0, 3, 3), dtype=np.float32)
        while True:
            for index in rng.permutation(self.num_shards):
                rows = self._shard(int(index))
                pending = np.concatenate([pending, rows[rng.permutation(rows.shape[0])]], axis=0)
                while pending.shape[0] >= batch_size:
                    yield torch.from_numpy(pending[:batch_size].copy())
                    pending = pending[batch_size:]
            if not repeat:
                break
        if pending.shape[0]:
            yield torch.from_numpy(pending.copy())

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for fd in self.fds:
            fd.close()


def load_dataset(directory: str) -> NeRFDataset:
    """Read `directory`: every X.png with its X.json camera, plus metadata.json (dataset.py:266-286)."""
    metadata = ModelMetadata.from_json(os.path.join(directory, "metadata.json"))
    views = []
    for name in sorted(os.listdir(directory)):
        if name.endswith(".png") and not name.startswith("."):
            stem = os.path.join(directory, name[:-4])
            views.append(FileNeRFView.from_json(stem + ".json", image_path=stem + ".png"))
    return NeRFDataset(metadata=metadata, views=views)
