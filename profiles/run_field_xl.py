"""Field-XL: 50 000 instances (1.02 G triangles submitted) at 7680x4320 on one GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import Renderer, RendererConfig, synth  # noqa: E402

scene = synth.make_field(50000, 5, lods=1, with_env=True, prepared=True)
r = Renderer(7680, 4320, RendererConfig(lod_enabled=False, cull_enabled=False, collect_stats=True, time_kernels=True))
r.set_scene(scene)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    r.render()
r.sync()
st = r.stats()
print(st, "frame_ms", sum(st["kernel_ms"].values()))
