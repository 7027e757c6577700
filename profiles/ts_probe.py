"""Scratch: reads the wall-clock stamps an instrumented k_tile_setup left (instrumented build only)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from renderer_amd import Renderer, RendererConfig, synth
scene = synth.make_atrium(columns=10, spheres=60, sphere_subdiv=5)
r = Renderer(3840, 2160, RendererConfig(cull_enabled=False, collect_stats=True))
r.set_scene(scene)
for _ in range(30):
    r.render()
r.sync()
out = np.zeros(1024 * 8, dtype=np.uint64)
L = r._L
L.rdr_debug_ts.argtypes = [ctypes.c_void_p]
print("rc", L.rdr_debug_ts(out.ctypes.data))
np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "ts.npy"), out.reshape(1024, 8))
