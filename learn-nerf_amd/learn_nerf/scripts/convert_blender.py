"""
Convert a Blender dataset of the original NeRF release (transforms_<split>.json + PNGs) into the on-disk
format of this package (reference: convert_dataset/blender.py; axis convention :39-41, y_fov :42, bounding
box fixed to [-1, 1]^3 :59-60).  Additive option --resize W downsamples the images with a box filter
(the Lego scenes are 800x800; BASELINE config 2 trains on 400x400); the camera JSON does not depend on the
resolution, so only the PNGs change.
"""
import argparse
import json
import math
import os
import shutil

import numpy as np


def convert(input_dir: str, output_dir: str, split: str = "train", resize: int = None) -> int:
    from PIL import Image

    if os.path.exists(output_dir):
        raise FileExistsError(f"output path exists: {output_dir}")
    os.mkdir(output_dir)
    with open(os.path.join(input_dir, f"transforms_{split}.json"), "r") as f:
        info = json.load(f)
    x_fov = info["camera_angle_x"]
    for i, frame in enumerate(info["frames"]):
        img_path = os.path.join(input_dir, frame["file_path"] + ".png")
        img = Image.open(img_path)
        width, height = img.size
        matrix = np.array(frame["transform_matrix"], dtype=np.float64)
        origin, rot = matrix[:3, -1], matrix[:3, :3]
        x_axis = rot @ np.array([1.0, 0.0, 0.0])
        y_axis = rot @ np.array([0.0, -1.0, 0.0])  # image rows grow downwards
        z_axis = rot @ np.array([0.0, 0.0, -1.0])  # Blender cameras look along -z
        y_fov = 2 * math.atan(math.tan(x_fov / 2) * height / width)
        base = os.path.join(output_dir, f"{i:04}")
        with open(base + ".json", "w") as f:
            json.dump(dict(origin=origin.tolist(), x_fov=x_fov, y_fov=y_fov, x=x_axis.tolist(), y=y_axis.tolist(),
                           z=z_axis.tolist()), f)
        if resize is not None and resize != width:
            new_h = max(1, round(height * resize / width))
            img.convert("RGBA").resize((resize, new_h), Image.BOX).save(base + ".png")
        else:
            shutil.copyfile(img_path, base + ".png")
    with open(os.path.join(output_dir, "metadata.json"), "w") as f:
        json.dump(dict(min=[-1.0] * 3, max=[1.0] * 3), f)
    return len(info["frames"])


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--split", type=str, default="train")
    parser.add_argument("--resize", type=int, default=None, help="(additive) output image width, e.g. 400")
    parser.add_argument("input_dir", type=str)
    parser.add_argument("output_dir", type=str)
    args = parser.parse_args()
    convert(args.input_dir, args.output_dir, args.split, args.resize)


if __name__ == "__main__":
    main()
