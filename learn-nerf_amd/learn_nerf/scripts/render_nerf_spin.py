"""
Spin around the y axis from a fixed camera view: command line of the reference's scripts/render_nerf_spin.py
(render_nerf.py flags + --frames, view_json, output_png); frames side by side in one PNG.
"""
from learn_nerf.dataset import CameraView
from learn_nerf.scripts.camera_paths import spin_views
from learn_nerf.scripts.render_nerf import RenderSession, argparser


def main():
    parser = argparser()
    parser.add_argument("--frames", type=int, default=10)
    parser.add_argument("view_json", type=str)
    parser.add_argument("output_png", type=str)
    args = parser.parse_args()
    session = RenderSession(args)
    for view in spin_views(CameraView.from_json(args.view_json), args.frames):
        session.render_view(view)
    session.save(args.output_png)


if __name__ == "__main__":
    main()
