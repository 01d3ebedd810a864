"""
Render a panning view of a NeRF model: command line of the reference's scripts/render_nerf_pan.py
(render_nerf.py flags + --frames, --distance, --random_axis, output_png); frames side by side in one PNG.
"""
import numpy as np

from learn_nerf.scripts.camera_paths import pan_views
from learn_nerf.scripts.render_nerf import RenderSession, argparser


def main():
    parser = argparser()
    parser.add_argument("--frames", type=int, default=10)
    parser.add_argument("--distance", type=float, default=2.0)
    parser.add_argument("--random_axis", action="store_true")
    parser.add_argument("output_png", type=str)
    args = parser.parse_args()
    session = RenderSession(args)
    axis = np.random.normal(size=(3,)) if args.random_axis else None
    meta = session.metadata
    for frame, view in enumerate(pan_views(meta.bbox_min, meta.bbox_max, args.frames, args.distance, axis)):
        print(f"sampling frame {frame}...")
        session.render_view(view)
    session.save(args.output_png)


if __name__ == "__main__":
    main()
