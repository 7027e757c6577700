"""Profiling driver: renders a few frames of the benchmark workload (for rocprofv3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import Renderer, RendererConfig, synth  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scene_name = sys.argv[2] if len(sys.argv) > 2 else "field"
w, h = 3840, 2160
scene = synth.make_field(1250, 5) if scene_name == "field" else synth.make_atrium(columns=10, spheres=60, sphere_subdiv=5)
r = Renderer(w, h, RendererConfig(collect_stats=True, time_kernels=True))
r.set_scene(scene)
for _ in range(frames):
    r.render()
r.sync()
print(r.stats())
