"""
Create a new NeRF dataset from a trained model by rendering random viewing angles, with the command line
and file formats of the reference's scripts/render_new_dataset.py: `NNNNN.json` camera (CameraView.to_json),
`NNNNN.png` colour ((c + 1) * 127.5 truncated to uint8, :120-123) and `NNNNN_depth.png` 16-bit z-depth
(:96-116, 124-130), plus a copy of the metadata file.  Everything per ray (rendering, depth) runs on the GPU.
"""
import argparse
import math
import os
import shutil

import numpy as np
import torch

from learn_nerf.dataset import CameraView
from learn_nerf.render import z_depth
from learn_nerf.scripts.render_nerf import RENDER_FLAGS, RenderSession
from learn_nerf.scripts.train_nerf import add_model_args

FIELD_OF_VIEW = 60.0 * math.pi / 180  # both axes (:89-90)


def argparser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description=__doc__)
    for flag, typ, default, text in RENDER_FLAGS:
        if flag not in ("--width", "--height"):
            parser.add_argument(flag, type=typ, default=default, help=text)
    parser.add_argument("--num_images", type=int, default=100)
    parser.add_argument("--size", type=int, default=512)
    parser.add_argument("--distance", type=float, default=1.0)
    parser.add_argument("--max_depth", type=float, default=10.0)
    add_model_args(parser)
    parser.add_argument("metadata_json", type=str)
    parser.add_argument("output_dir", type=str)
    return parser


def random_view(rng: np.random.RandomState, center: np.ndarray, radius: float) -> CameraView:
    """Camera on a sphere of `radius` about `center`, looking at the centre; x axis horizontal (:78-91)."""
    z = rng.normal(size=(3,))
    z = z / np.linalg.norm(z)
    x = np.array([z[1], -z[0], 0.0])
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    return CameraView(camera_direction=tuple(float(v) for v in z),
                      camera_origin=tuple(float(v) for v in (center - z * radius)),
                      x_axis=tuple(float(v) for v in x), y_axis=tuple(float(v) for v in y),
                      x_fov=FIELD_OF_VIEW, y_fov=FIELD_OF_VIEW)


class DatasetRenderSession(RenderSession):
    """RenderSession that keeps the whole fine result (colour, hit probability, collision point)."""

    def render_fn(self, key, rays: torch.Tensor):
        return self.renderer.render_rays(key, rays)["fine"]

    def render_view_with_depth(self, view: CameraView, size: int, max_depth: float):
        rays = view.bare_rays(size, size, device=self.device)  # lnrf_camera_rays: generated on the GPU
        colors, depths = [], []
        for start in range(0, rays.shape[0], self.args.batch_size):
            self.key, slice_key = self.key.split(2)
            fine = self.render_fn(slice_key, rays[start:start + self.args.batch_size].contiguous())
            colors.append(fine["outputs"])
            depths.append(z_depth(fine["coords"], fine["alphas"], view.camera_origin, view.camera_direction,
                                  max_depth))
        image = ((torch.cat(colors).cpu().numpy().reshape(size, size, 3) + 1) * 127.5).astype(np.uint8)
        depth = (torch.cat(depths).cpu().numpy().reshape(size, size) * 0xFFFF).astype(np.uint32)
        return image, depth


def main():
    from PIL import Image

    args = argparser().parse_args()
    if os.path.exists(args.output_dir):
        raise FileExistsError(f"output directory exists: {args.output_dir}")
    args.width = args.height = args.size
    session = DatasetRenderSession(args)
    os.makedirs(args.output_dir)
    shutil.copy(args.metadata_json, os.path.join(args.output_dir, "metadata.json"))
    lo = np.array(session.metadata.bbox_min, dtype=np.float64)
    hi = np.array(session.metadata.bbox_max, dtype=np.float64)
    radius = float(np.linalg.norm(lo - hi)) * args.distance  # bbox diagonal x --distance (:74, 86)
    rng = np.random.RandomState(args.seed)  # the reference draws the poses from NumPy's global, unseeded state
    for frame in range(args.num_images):
        print(f"sampling frame {frame}...")
        view = random_view(rng, (lo + hi) / 2, radius)
        stem = os.path.join(args.output_dir, f"{frame:05}")
        with open(stem + ".json", "w") as handle:
            handle.write(view.to_json())
        image, depth = session.render_view_with_depth(view, args.size, args.max_depth)
        Image.fromarray(image).save(stem + ".png")
        Image.fromarray(depth).save(stem + "_depth.png")


if __name__ == "__main__":
    main()
