"""
Synthetic dataset producer: an analytic ray-cast of a coloured, shaded unit cube, written in the
reference's on-disk format (NNNN.png + NNNN.json{origin,x,y,z,x_fov,y_fov} + metadata.json{min,max};
dataset.py:266-286, simple_dataset/main.go:89-156).  Stands in for the reference's Go tool
(simple_dataset/, needs the un-vendored model3d module; Go is absent here) for BASELINE config 1.
"""
import argparse
import json
import math
import os

import numpy as np

FACE_COLORS = np.array([[0.9, 0.2, 0.2], [0.2, 0.9, 0.2], [0.2, 0.3, 0.9],
                        [0.9, 0.9, 0.2], [0.9, 0.2, 0.9], [0.2, 0.9, 0.9]], dtype=np.float64)


def render_cube(origin, x_axis, y_axis, z_axis, fov, size, half=0.5):
    lin = np.linspace(-1, 1, size)
    t = math.tan(fov / 2)
    d = z_axis[None, None] + t * lin[None, :, None] * x_axis[None, None] + t * lin[:, None, None] * y_axis[None, None]
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (-half - origin) / d
        t1 = (half - origin) / d
    lo, hi = np.minimum(t0, t1), np.maximum(t0, t1)
    t_near, t_far = lo.max(-1), hi.min(-1)
    hit = (t_near < t_far) & (t_far > 0)
    axis = lo.argmax(-1)
    sign = np.take_along_axis(np.sign(-d), axis[..., None], -1)[..., 0]
    face = axis * 2 + (sign > 0)
    normal = np.zeros(d.shape)
    np.put_along_axis(normal, axis[..., None], sign[..., None], -1)
    light = np.array([0.4, 0.5, 0.76])
    shade = 0.35 + 0.65 * np.clip((normal * light).sum(-1), 0, 1)
    rgb = FACE_COLORS[face] * shade[..., None]
    rgba = np.zeros((size, size, 4), dtype=np.uint8)
    rgba[..., :3] = np.where(hit[..., None], np.round(rgb * 255), 0).astype(np.uint8)
    rgba[..., 3] = np.where(hit, 255, 0).astype(np.uint8)
    return rgba


def random_camera(rng, radius):
    v = rng.normal(size=3)
    origin = radius * v / np.linalg.norm(v)
    z = -origin / np.linalg.norm(origin)
    up = np.array([0.0, 0.0, 1.0]) if abs(z[2]) < 0.95 else np.array([0.0, 1.0, 0.0])
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)  # points down the image, as the reference's convention (blender.py:39-41)
    return origin, x, y, z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=30)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--fov_degrees", type=float, default=40.0)
    ap.add_argument("--radius", type=float, default=2.5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("out_dir")
    args = ap.parse_args()
    from PIL import Image

    os.makedirs(args.out_dir, exist_ok=True)
    rng = np.random.default_rng(args.seed)
    fov = math.radians(args.fov_degrees)
    for i in range(args.views):
        origin, x, y, z = random_camera(rng, args.radius)
        img = render_cube(origin, x, y, z, fov, args.size)
        Image.fromarray(img, "RGBA").save(os.path.join(args.out_dir, f"{i:04d}.png"))
        with open(os.path.join(args.out_dir, f"{i:04d}.json"), "w") as f:
            json.dump(dict(origin=origin.tolist(), x=x.tolist(), y=y.tolist(), z=z.tolist(), x_fov=fov, y_fov=fov), f)
    with open(os.path.join(args.out_dir, "metadata.json"), "w") as f:
        json.dump({"min": [-1.0, -1.0, -1.0], "max": [1.0, 1.0, 1.0]}, f)


if __name__ == "__main__":
    main()
