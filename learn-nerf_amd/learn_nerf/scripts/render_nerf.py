"""
Rendering CLI with the command line of the reference's scripts/render_nerf.py: flags, positional
`metadata_json view_json... output_png`, views laid side by side, colours quantised by truncation
((c + 1) * 127.5 -> uint8, scripts/render_nerf.py:93-96), fresh sampling noise per slice (:90).
`argparser()` and `RenderSession` keep their names because the reference's pan / spin scripts build on them.
"""
import argparse
import random

import numpy as np
import torch

from learn_nerf.dataset import CameraView, ModelMetadata
from learn_nerf.render import NeRFRenderer
from learn_nerf.rng import Key
from learn_nerf.scripts.train_nerf import add_model_args, create_model
from learn_nerf.train import load_params

RENDER_FLAGS = (
    ("--seed", int, None, None),
    ("--batch_size", int, 1024, "rays per batch"),
    ("--coarse_samples", int, 64, "samples per coarse ray"),
    ("--fine_samples", int, 128, "samples per fine ray (not including coarse samples)"),
    ("--width", int, 512, None),
    ("--height", int, 512, None),
    ("--model_path", str, "nerf.pkl", None),
)


def argparser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description=__doc__)
    for flag, typ, default, text in RENDER_FLAGS:
        parser.add_argument(flag, type=typ, default=default, help=text)
    add_model_args(parser)
    parser.add_argument("metadata_json", type=str)
    return parser


def quantise(colors: np.ndarray) -> np.ndarray:
    """Colours in [-1, 1] -> uint8 by TRUNCATION, no rounding and no clip (reference scripts/render_nerf.py:93-96)."""
    return ((colors + 1) * 127.5).astype(np.uint8)


class RenderSession:
    """Loads a checkpoint once, renders any number of camera views, then writes them into one PNG."""

    def __init__(self, args: argparse.Namespace):
        print("loading metadata...")
        self.metadata = ModelMetadata.from_json(args.metadata_json)
        print("loading model...")
        self.device = torch.device("cuda", torch.cuda.current_device())
        coarse, fine, _ = create_model(args, self.metadata)
        params = load_params(args.model_path, coarse, fine, self.device)
        self.renderer = NeRFRenderer(coarse=coarse, fine=fine, coarse_params=params["coarse"],
                                     fine_params=params["fine"], background=params["background"],
                                     bbox_min=self.metadata.bbox_min, bbox_max=self.metadata.bbox_max,
                                     coarse_ts=args.coarse_samples, fine_ts=args.fine_samples)
        self.key = Key(args.seed if args.seed is not None else random.randint(0, 2 ** 32 - 1))
        self.args = args
        self.images = []

    def render_fn(self, key, rays: torch.Tensor) -> torch.Tensor:
        return self.renderer.render_rays(key, rays)["fine"]["outputs"]

    def render_view(self, view: CameraView):
        width, height, step = self.args.width, self.args.height, self.args.batch_size
        rays = view.bare_rays(width, height, device=self.device)  # lnrf_camera_rays: generated on the GPU
        pieces = []
        for start in range(0, rays.shape[0], step):
            self.key, slice_key = self.key.split(2)
            pieces.append(self.render_fn(slice_key, rays[start:start + step].contiguous()))
        colors = torch.cat(pieces, dim=0).cpu().numpy().reshape(height, width, 3)
        self.images.append(quantise(colors))

    def save(self, output_path: str):
        from PIL import Image

        Image.fromarray(np.concatenate(self.images, axis=1)).save(output_path)


def main():
    parser = argparser()
    parser.add_argument("view_json", type=str, nargs="+")
    parser.add_argument("output_png", type=str)
    args = parser.parse_args()
    session = RenderSession(args)
    for path in args.view_json:
        print(f"rendering view {path}...")
        session.render_view(CameraView.from_json(path))
    session.save(args.output_png)


if __name__ == "__main__":
    main()
