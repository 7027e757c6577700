"""Times the two bake entry points at the reference's sizes (256 x 256 table with 10 000 samples; three 256 x 256
prefiltered maps with full mip chains and 1024 samples), copies to and from the host included."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderer_amd import bake, synth  # noqa: E402

sky = synth.default_env().skybox
bake.bake_brdf_lut(16, 64)  # first call: library load, context
for name, fn in (("brdf_lut 256^2 x 10000", lambda: bake.bake_brdf_lut(256, 10000)),
                 ("lambertian 256^2 x 1024", lambda: bake.bake_env_map(sky, bake.LAMBERTIAN, 256, None, 1024)),
                 ("ggx 256^2 x 1024", lambda: bake.bake_env_map(sky, bake.GGX, 256, None, 1024)),
                 ("charlie 256^2 x 1024", lambda: bake.bake_env_map(sky, bake.CHARLIE, 256, None, 1024))):
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t)
    print(f"{name}: {best * 1e3:.1f} ms", flush=True)
