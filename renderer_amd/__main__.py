"""Command line counterpart of the reference's ``potato <config> <envmap> <scene>`` for this
path: render one frame of a ``scene.bin`` (or a built-in synthetic scene) on the GPU and write
the image and, optionally, the raw buffers a comparison with a reference-side dump needs.

  python -m renderer_amd scene.bin --size 3840 2160 --out frame.png
  python -m renderer_amd synth:atrium --skybox --dump-vis vis.u32x2 --dump-color color.rgba8
"""
import argparse
import sys

import numpy as np

from . import Renderer, RendererConfig, synth
from .imageio import write_png
from .scene import Camera
from .scene_bin import load_scene_bin, save_scene_bin


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m renderer_amd", description=__doc__.split("\n\n")[0])
    ap.add_argument("scene", help="path of a scene.bin / .gltf / .glb / .obj (several .obj separated by commas), or synth:field / synth:atrium")
    ap.add_argument("--size", type=int, nargs=2, default=(1280, 720), metavar=("W", "H"))
    ap.add_argument("--camera", type=float, nargs=6, metavar=("PX", "PY", "PZ", "DX", "DY", "DZ"))
    ap.add_argument("--sun", type=float, nargs=3)
    ap.add_argument("--env", help="environment JSON naming five KTX2 files (skybox, brdf_lut, lambertian, ggx, charlie), "
                                  "as the reference's second command-line argument; default: a synthetic sky")
    ap.add_argument("--lod", action="store_true", help="RendererConfig.lod_enabled")
    ap.add_argument("--skybox", action="store_true", help="run the skybox pass on uncovered pixels")
    ap.add_argument("--frames", type=int, default=1, help="render this many frames and report the mean time")
    ap.add_argument("--out", help="PNG of the shaded frame")
    ap.add_argument("--dump-vis", help="raw W*H*2 u32 (reference texel format: meshlet<<8|triangle, draw)")
    ap.add_argument("--dump-color", help="raw W*H*4 RGBA8 (sRGB encoded, row 0 = top)")
    ap.add_argument("--save-scene", help="write the scene as scene.bin (for synth: scenes)")
    args = ap.parse_args(argv)

    if args.scene == "synth:field":
        scene = synth.make_field(1250, 5, lods=3, with_env=True)
    elif args.scene == "synth:atrium":
        scene = synth.make_atrium(columns=10, spheres=60, sphere_subdiv=5)
    elif args.scene.lower().endswith((".gltf", ".glb")):
        from .gltf import load_gltf
        scene = load_gltf(args.scene, env=synth.default_env())  # IBL inputs: the synthetic sky (see renderer_amd.bake for real ones)
    elif args.scene.lower().endswith(".obj") or ".obj," in args.scene.lower():
        from .obj import load_obj
        scene = load_obj(args.scene, env=synth.default_env())
    else:
        scene = load_scene_bin(args.scene)
    if args.env:
        from .ktx2 import load_env_json
        scene.env = load_env_json(args.env)
    if args.camera:
        scene.camera = Camera(tuple(args.camera[:3]), tuple(args.camera[3:]))
    if args.sun:
        scene.sun_direction = tuple(args.sun)
    if args.save_scene:
        save_scene_bin(scene, args.save_scene)
    w, h = args.size
    r = Renderer(w, h, RendererConfig(lod_enabled=args.lod, skybox=args.skybox, collect_stats=True, time_kernels=True))
    r.set_scene(scene)
    for _ in range(max(1, args.frames)):
        r.render()
    r.sync()
    st = r.stats()
    frame_ms = sum(st["kernel_ms"].values())
    print(f"{w}x{h}: {frame_ms:.3f} ms/frame on the GPU ({1e3 / frame_ms:.0f} frames/s), "
          f"{st['triangles_submitted']} triangles submitted, {st['triangles_rasterised']} rasterised, "
          f"{st['pixels_covered']} pixels covered")
    print("kernel ms:", {k: round(v, 4) for k, v in st["kernel_ms"].items() if v > 0})
    if args.out:
        write_png(args.out, r.read_color())
    if args.dump_vis:
        r.read_visbuffer().astype("<u4").tofile(args.dump_vis)
    if args.dump_color:
        np.ascontiguousarray(r.read_color()).tofile(args.dump_color)
    r.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
