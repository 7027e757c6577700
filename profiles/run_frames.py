"""Profiling driver: renders a few frames of the benchmark workload (for rocprofv3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import Renderer, RendererConfig, synth  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scene_name = sys.argv[2] if len(sys.argv) > 2 else "field"
w, h = 3840, 2160
if scene_name == "field":  # exactly bench.py's single-GPU workload
    scene = synth.make_field(1250, 5, lods=1, with_env=True, prepared=True)
    conf = RendererConfig(lod_enabled=False, cull_enabled=False, collect_stats=True, time_kernels=True)
elif scene_name == "occfield":  # the headline field with two-phase occlusion culling on
    scene = synth.make_field(1250, 5, lods=1, with_env=True, prepared=True)
    conf = RendererConfig(lod_enabled=False, cull_enabled=True, collect_stats=True, time_kernels=True)
elif scene_name == "lodfield":  # the field with LOD selection on: coarse levels, mid-size triangles dominate
    scene = synth.make_field(1250, 5, lods=8, with_env=True, prepared=True)
    conf = RendererConfig(lod_enabled=True, cull_enabled=False, collect_stats=True, time_kernels=True)
elif scene_name == "alphafield":  # half of the draws alpha-masked with a checkerboard albedo
    scene = synth.make_field(1250, 5, lods=1, alpha_fraction=0.5, with_env=True, prepared=True)
    conf = RendererConfig(lod_enabled=False, cull_enabled=False, collect_stats=True, time_kernels=True)
elif scene_name == "fieldxl":  # 50 000 instances = 1.02 G triangles at 7680 x 4320
    w, h = 7680, 4320
    scene = synth.make_field(50000, 5, lods=1, with_env=True, prepared=True)
    conf = RendererConfig(lod_enabled=False, cull_enabled=False, collect_stats=True, time_kernels=True)
else:
    scene = synth.make_atrium(columns=10, spheres=60, sphere_subdiv=5)
    conf = RendererConfig(cull_enabled=os.environ.get("RDR_OCCLUSION", "0") != "0", collect_stats=True, time_kernels=True)
r = Renderer(w, h, conf)
r.set_scene(scene)
if os.environ.get("RUN_WORLD"):  # one rank's share of a sort-last split, alone: RUN_WORLD=8 RUN_RANK=0 [RUN_REPLICATE=2048]
    world, rank = int(os.environ["RUN_WORLD"]), int(os.environ.get("RUN_RANK", "0"))
    r.set_partition_interleaved(64, rank, world)
    r.set_shade_bands(32, rank, world)
    r.set_partition_replication(float(os.environ.get("RUN_REPLICATE", "0")))
for _ in range(frames):
    r.render()
r.sync()
st = r.stats()
print(st)
print("frame_ms (sum of kernel times)", sum(st["kernel_ms"].values()), "frame status", r.frame_status())
