"""Parity evidence: HIP path vs the CPU oracle on a few scenes; prints visibility-key mismatches
and the error histogram / PSNR of the shaded image.  (Run on a GPU box; the tests assert the same
bars.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import psnr  # noqa: E402
from oracle import orc  # noqa: E402
from renderer_amd import Renderer, RendererConfig, synth  # noqa: E402

cases = [
    ("atrium textured 640x360", synth.make_atrium(), 640, 360, RendererConfig()),
    ("atrium textured 1280x720", synth.make_atrium(), 1280, 720, RendererConfig()),
    ("field 300 x icosphere-3, lod on, 10% alpha-masked, 960x540", synth.make_field(300, 3, lods=3, alpha_fraction=0.1, with_env=True),
     960, 540, RendererConfig(lod_enabled=True)),
    ("field 150 x icosphere-4, 1280x720", synth.make_field(150, 4, lods=1, with_env=True), 1280, 720, RendererConfig(lod_enabled=False)),
]
for name, scene, w, h, conf in cases:
    o = orc.OracleRenderer(w, h, conf)
    o.set_scene(scene)
    ovis = o.render_visbuffer()
    ref = o.shade()
    g = Renderer(w, h, conf)
    g.set_scene(scene)
    g.render()
    gvis = g.read_visbuffer_u64()
    img = g.read_color()
    err = np.abs(img.astype(np.int32) - ref.astype(np.int32)).max(axis=-1)
    hist = np.bincount(err.ravel(), minlength=4)
    print(f"{name}: keys differing {int((gvis != ovis).sum())} of {ovis.size}; "
          f"pixels by max channel error 0/1/2/3+: {hist[0]}/{hist[1]}/{hist[2]}/{int(hist[3:].sum())}; "
          f"PSNR {psnr(img, ref):.1f} dB", flush=True)
    g.close()
