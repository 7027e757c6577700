"""Regenerate the golden fixtures: visibility buffers and shaded images of small deterministic
scenes, produced by the CPU oracle (oracle/liborc.so).

The reference ships no golden images or vectors and cannot be built or run here (Rust, Vulkan
and its assets are absent), so these fixtures pin the oracle against *itself over time* (any
change to the oracle that alters an output shows up as a fixture diff) and give the GPU tests a
second, data-only anchor.  Inputs are regenerated from code (renderer_amd.synth); only outputs
are stored.

    python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import orc  # noqa: E402
from renderer_amd import RendererConfig, synth  # noqa: E402
from renderer_amd.scene import Camera  # noqa: E402

CASES = {
    # name: (scene factory, width, height, config kwargs)
    "atrium_160x90": (lambda: synth.make_atrium(), 160, 90, {}),
    "field_lod_192x108": (lambda: _field_close(), 192, 108, {"lod_enabled": True}),
    "field_mask_160x90": (lambda: _field_mask(), 160, 90, {}),
}


def _field_close():
    sc = synth.make_field(40, 3, lods=3)
    sc.camera = Camera((0.0, 3.0, -11.0), (0.0, 0.22, -1.0))
    return sc


def _field_mask():
    sc = synth.make_field(24, 3, alpha_fraction=0.5)
    sc.camera = Camera((0.0, 3.0, -9.0), (0.0, 0.25, -1.0))
    return sc


def render_case(name):
    factory, w, h, cfg = CASES[name]
    sc = factory()
    o = orc.OracleRenderer(w, h, RendererConfig(**cfg))
    o.set_scene(sc)
    vis = o.render_visbuffer().copy()
    color = o.shade()
    return sc, o, vis, color


def main():
    orc.build()
    for name in CASES:
        sc, o, vis, color = render_case(name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, vis=vis, color=color, draw_lod=o.draw_lod[: len(sc.mesh_draws)],
                            stats=np.array([o.stats.triangles_rasterised, o.stats.fragments, o.stats.meshlets_visible], np.uint64))
        print(name, vis.shape, "covered", int((vis != 0x7FFFFFFFFFFFFFFF).sum()), os.path.getsize(path), "bytes",
              "crc", zlib.crc32(vis.tobytes()))


if __name__ == "__main__":
    main()
