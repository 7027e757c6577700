R=$GRAFT_REPO_ROOT
cd $R
RDR_LIB_PATH=$R/renderer_amd/variants/libstamps.so timeout -k 10 150 python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from renderer_amd import Renderer, RendererConfig, synth
scene = synth.make_field(1250, 5, lods=1, with_env=True, prepared=True)
r = Renderer(3840, 2160, RendererConfig(lod_enabled=False, cull_enabled=False, collect_stats=True, time_kernels=True))
r.set_scene(scene)
for _ in range(3): r.render()
r.sync(); r.stats()
r.render(); r.sync()
st = r.stats()
n = st["meshlets_visible"]
print("meshlets", n, "raster ms", st["kernel_ms"]["raster"])
for name, key in (("A vertex phase (barrier1..barrier2)", "triangles_tested"), ("B cull phase", "triangles_rasterised"), ("C barrier3 + settle", "triangles_big"), ("D scan", "fragments"), ("E loop end + barrier1", "vertices_transformed")):
    print(f"{name:40s} {st[key] * 16 / n:9.0f} cycles per meshlet")
PY
