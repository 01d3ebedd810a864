"""
learn_nerf.dataset — CameraView, NeRFView, FileNeRFView, ModelMetadata, NeRFDataset,
ShuffledDataset, load_dataset (reference: learn_nerf/dataset.py).

Host-side data plumbing (NumPy + PIL) kept API- and file-format-compatible with the reference:
per-view PNG + JSON {origin,x,y,z,x_fov,y_fov}, metadata.json {min,max}, and the on-disk
two-stage shuffle (raw little-endian fp32 [k,3,3] shards "0".."31" + "done", dataset.py:140-263).
Batches are returned as torch tensors (CPU, pinned when possible); callers move them to the GPU.
"""
import json
import math
import os
from abc import abstractmethod
from dataclasses import dataclass
from typing import Iterator, List, Tuple

import numpy as np
import torch

from .params import split_seed

Vec3 = Tuple[float, float, float]


def _rng(key) -> np.random.Generator:
    """jax.random key stand-in: an int seed (or anything int()-able)."""
    seed = int(getattr(key, "seed", key)) & 0xFFFFFFFFFFFFFFFF
    return np.random.default_rng(seed)


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
        with open(path, "rb") as f:
            camera_info = json.load(f)
        return cls(
            camera_direction=tuple(camera_info["z"]),
            camera_origin=tuple(camera_info["origin"]),
            x_axis=tuple(camera_info["x"]),
            y_axis=tuple(camera_info["y"]),
            x_fov=float(camera_info["x_fov"]),
            y_fov=float(camera_info["y_fov"]),
            **kwargs,
        )

    def to_json(self) -> str:
        return json.dumps(
            dict(z=self.camera_direction, origin=self.camera_origin, x=self.x_axis, y=self.y_axis,
                 x_fov=self.x_fov, y_fov=self.y_fov)
        )

    def bare_rays(self, width: int, height: int) -> torch.Tensor:
        """
        All rays of the view in raster scan order: [N x 2 x 3] (origin, direction) (dataset.py:52-78).
        """
        z = np.array(self.camera_direction, dtype=np.float32)
        ys = (np.float32(math.tan(self.y_fov / 2)) * np.linspace(-1, 1, num=height, dtype=np.float32)[:, None, None]
              * np.array(self.y_axis, dtype=np.float32))
        xs = (np.float32(math.tan(self.x_fov / 2)) * np.linspace(-1, 1, num=width, dtype=np.float32)[None, :, None]
              * np.array(self.x_axis, dtype=np.float32))
        directions = np.reshape(xs + ys + z, [-1, 3])
        directions = directions / np.linalg.norm(directions, axis=-1, keepdims=True)
        origins = np.broadcast_to(np.array(self.camera_origin, dtype=np.float32), directions.shape)
        return torch.from_numpy(np.stack([origins, directions], axis=1).astype(np.float32))


@dataclass
class NeRFView(CameraView):
    @abstractmethod
    def image(self) -> np.ndarray:
        """Load the image as a [Height x Width x 3] array of uint8 RGB values."""

    def rays(self) -> torch.Tensor:
        """[N x 3 x 3] (origin, direction, color in [-1, 1]) (dataset.py:89-101)."""
        img = np.asarray(self.image())
        bare = self.bare_rays(img.shape[1], img.shape[0]).numpy()
        colors = np.reshape(img, [-1, 3]).astype(np.float32) / 127.5 - 1
        return torch.from_numpy(np.concatenate([bare, colors[:, None]], axis=1).astype(np.float32))


@dataclass
class FileNeRFView(NeRFView):
    image_path: str = None

    def image(self) -> np.ndarray:
        # Premultiply alpha to prevent egregious errors at the border (dataset.py:108-111).
        from PIL import Image

        rgba = np.array(Image.open(self.image_path).convert("RGBA"))
        return np.round(rgba[:, :, :3] * (rgba[:, :, 3:] / 255)).astype(np.uint8)


@dataclass
class ModelMetadata:
    bbox_min: Vec3
    bbox_max: Vec3

    @classmethod
    def from_json(cls, path: str) -> "ModelMetadata":
        with open(path, "rb") as f:
            metadata = json.load(f)
        return ModelMetadata(bbox_min=tuple(metadata["min"]), bbox_max=tuple(metadata["max"]))


@dataclass
class NeRFDataset:
    metadata: ModelMetadata
    views: List[NeRFView]

    def iterate_batches(self, dir_path: str, key, batch_size: int, repeat: bool = True,
                        num_shards: int = 32) -> Iterator[torch.Tensor]:
        """
        Shuffled [N x 3 x 3] batches of (origin, direction, color) rays (dataset.py:134-159).
        """
        with ShuffledDataset(dir_path, self, key, num_shards=num_shards) as sd:
            yield from sd.iterate_batches(batch_size, repeat=repeat)


class ShuffledDataset:
    """
    A pre-shuffled version of the rays in a NeRFDataset: the two-stage on-disk shuffle of
    dataset.py:162-263 (same file names and raw fp32 layout, so shard directories interchange).
    """

    def __init__(self, dir_path: str, dataset: NeRFDataset, key, num_shards: int = 32):
        self.num_shards = num_shards
        shard_seed, shuffle_seed = split_seed(int(getattr(key, "seed", key)), 2)
        self.shard_rng = np.random.default_rng(shard_seed)
        self.shuffle_seed = shuffle_seed
        if not os.path.exists(dir_path):
            os.mkdir(dir_path)
        done_path = os.path.join(dir_path, "done")
        if os.path.exists(done_path):
            self.fds = [open(os.path.join(dir_path, f"{i}"), "rb") for i in range(num_shards)]
        else:
            self.fds = [open(os.path.join(dir_path, f"{i}"), "wb+") for i in range(num_shards)]
            self._create_shards(dataset)
            with open(done_path, "wb+") as f:
                f.write(b"done\n")

    def iterate_batches(self, batch_size: int, repeat: bool = False) -> Iterator[torch.Tensor]:
        rng = np.random.default_rng(self.shuffle_seed)
        cur_batch = None
        while True:
            for shard in rng.permutation(self.num_shards).tolist():
                shard_rays = self._read_shard(shard)
                shard_rays = shard_rays[rng.permutation(shard_rays.shape[0])]
                cur_batch = shard_rays if cur_batch is None else np.concatenate([cur_batch, shard_rays], axis=0)
                while cur_batch.shape[0] >= batch_size:
                    yield torch.from_numpy(np.ascontiguousarray(cur_batch[:batch_size]))
                    cur_batch = cur_batch[batch_size:]
            if not repeat:
                break
        if cur_batch is not None and cur_batch.shape[0]:
            yield torch.from_numpy(np.ascontiguousarray(cur_batch))

    def __enter__(self):
        return self

    def __exit__(self, *args):
        for fd in self.fds:
            fd.close()

    def _create_shards(self, dataset: NeRFDataset):
        for view in dataset.views:
            rays = view.rays().numpy()
            assignments = self.shard_rng.integers(0, self.num_shards, size=rays.shape[0])
            for shard in range(self.num_shards):
                sub_batch = rays[assignments == shard]
                if sub_batch.shape[0]:
                    self.fds[shard].write(sub_batch.astype("<f4").tobytes())
        for fd in self.fds:
            fd.flush()

    def _read_shard(self, shard: int) -> np.ndarray:
        f = self.fds[shard]
        f.seek(0)
        return np.frombuffer(f.read(), dtype="<f4").reshape([-1, 3, 3])


def load_dataset(directory: str) -> NeRFDataset:
    """
    Load a dataset from a directory on disk (dataset.py:266-286): X.png + X.json per view and a
    global metadata.json with the scene bounding box.
    """
    dataset = NeRFDataset(metadata=ModelMetadata.from_json(os.path.join(directory, "metadata.json")), views=[])
    for img_name in sorted(os.listdir(directory)):
        if img_name.startswith(".") or not img_name.endswith(".png"):
            continue
        img_path = os.path.join(directory, img_name)
        json_path = img_path[: -len(".png")] + ".json"
        dataset.views.append(FileNeRFView.from_json(json_path, image_path=img_path))
    return dataset
