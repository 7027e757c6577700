import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import Renderer, RendererConfig, synth
scene = synth.make_field(1250, 5, lods=1, with_env=True, prepared=True)
for tk in (False, True, False, True):
    r = Renderer(3840, 2160, RendererConfig(lod_enabled=False, cull_enabled=False, collect_stats=True, time_kernels=tk))
    r.set_scene(scene)
    for _ in range(20): r.render()
    r.sync()
    t0 = time.perf_counter()
    for _ in range(300): r.render()
    r.sync()
    print("time_kernels", tk, (time.perf_counter() - t0) / 300 * 1e3, "ms")
    r.close()
