"""
Camera paths used by the pan / spin renderers (reference: scripts/render_nerf_pan.py:22-52,
scripts/render_nerf_spin.py:22-33), as pure functions so they can be checked without a GPU.
"""
import math
from dataclasses import replace
from typing import Iterator, Optional, Sequence

import numpy as np

from learn_nerf.dataset import CameraView

FIELD_OF_VIEW = 60.0 * math.pi / 180


def _t(v) -> tuple:
    return tuple(float(c) for c in v)


def pan_views(bbox_min: Sequence[float], bbox_max: Sequence[float], frames: int, distance: float,
              axis: Optional[Sequence[float]] = None) -> Iterator[CameraView]:
    """
    Orbit around the centre of the bounding box at `distance` bbox diagonals, always looking at the centre.
    Default axis (0, 0, -1) with first basis vector (1, 0, 0); a custom axis a gets the basis (-a_z, 0, a_x).
    """
    lo, hi = np.asarray(bbox_min, dtype=np.float64), np.asarray(bbox_max, dtype=np.float64)
    radius = float(np.linalg.norm(lo - hi)) * distance
    center = (lo + hi) / 2
    if axis is None:
        up, e1 = np.array([0.0, 0.0, -1.0]), np.array([1.0, 0.0, 0.0])
    else:
        up = np.asarray(axis, dtype=np.float64)
        up = up / np.linalg.norm(up)
        e1 = np.array([-up[2], 0.0, up[0]])
        e1 = e1 / np.linalg.norm(e1)
    e2 = np.cross(up, e1)
    for frame in range(frames):
        theta = 2 * math.pi * frame / frames
        look = math.cos(theta) * e1 + math.sin(theta) * e2
        right = math.cos(theta + math.pi / 2) * e1 + math.sin(theta + math.pi / 2) * e2
        yield CameraView(camera_direction=_t(look), camera_origin=_t(center - look * radius), x_axis=_t(right),
                         y_axis=_t(up), x_fov=FIELD_OF_VIEW, y_fov=FIELD_OF_VIEW)


def spin_views(view: CameraView, frames: int) -> Iterator[CameraView]:
    """Rotate the camera about its own y axis in `frames` steps of a full turn; origin and y axis stay fixed."""
    x, z = np.asarray(view.x_axis, dtype=np.float64), np.asarray(view.camera_direction, dtype=np.float64)
    for i in range(frames):
        theta = 2 * math.pi * i / frames
        s, c = math.sin(theta), math.cos(theta)
        yield replace(view, x_axis=_t(c * x + s * z), camera_direction=_t(-s * x + c * z))
