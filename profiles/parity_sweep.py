"""Extended sweep of the triangle-soup parity test (tests/test_gpu_random.py): 60 more seeds with random
resolutions, triangle counts and alpha-masking, with and without occlusion culling.  Prints every frame whose
visibility keys differ from the CPU checker's or whose colour differs by more than 2/255."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import orc as oracle
from renderer_amd import Renderer, RendererConfig
from renderer_amd.scene import Camera
from test_gpu_random import soup

bad = 0
for seed in range(100, 160):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    masked = bool(rng.integers(0, 2))
    sc = soup(rng, int(rng.integers(50, 1500)), masked)
    for occlusion in (False, True):
        conf = RendererConfig(cull_enabled=occlusion)
        o = oracle.OracleRenderer(w, h, conf); o.set_scene(sc)
        g = Renderer(w, h, conf); g.set_scene(sc)
        for k in range(2):
            a = 0.5 * k
            cam = Camera((0.3 * k, 0.1 * k, -0.5 * k), (np.sin(a), 0.05 * k, -np.cos(a)))
            ovis = o.render_visbuffer(cam, (0.3, 0.8, -0.4))
            g.render(cam, (0.3, 0.8, -0.4))
            gvis = g.read_visbuffer_u64()
            d = int((gvis != ovis).sum())
            ocol = o.shade()
            gcol = g.read_color()
            e = int(np.abs(ocol.astype(int) - gcol.astype(int)).max()) if ocol.size else 0
            if d or e > 2:
                bad += 1
                print("MISMATCH seed", seed, w, h, masked, occlusion, k, "keys", d, "colour", e, flush=True)
        g.close()
    if seed % 10 == 0:
        print("seed", seed, "ok so far, bad =", bad, flush=True)
print("done, mismatching frames:", bad)
