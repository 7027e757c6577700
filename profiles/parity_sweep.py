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

SCALE = int(os.environ.get("SWEEP_SCALE", "1"))  # more seeds of every part
bad = 0
for seed in (range(100, 100 + 60 * SCALE) if os.environ.get("SWEEP_PART", "all") in ("all", "soups") else ()):
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

# second part: instanced fields (prepared meshlets, simplified LOD chains, alpha-masked draws) from random cameras
from renderer_amd import synth
bad = 0
for seed in ((range(20000, 20000 + 30 * SCALE) if SCALE > 1 else range(200, 230)) if os.environ.get("SWEEP_PART", "all") in ("all", "fields") else ()):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(64, 900)), int(rng.integers(64, 600))
    subdiv = int(rng.integers(1, 4))
    prepared = bool(rng.integers(0, 2))
    lods = int(rng.integers(1, 5)) if prepared else int(rng.integers(1, subdiv + 2))  # hand-made chains stop at the icosahedron
    sc = synth.make_field(int(rng.integers(20, 300)), subdiv, lods=lods, alpha_fraction=float(rng.choice([0.0, 0.3, 1.0])),
                          with_env=True, seed=seed, prepared=prepared)
    for occlusion in (False, True):
        conf = RendererConfig(cull_enabled=occlusion, lod_enabled=lods > 1)
        o = oracle.OracleRenderer(w, h, conf); o.set_scene(sc)
        g = Renderer(w, h, conf); g.set_scene(sc)
        base = sc.camera
        for k in range(2):
            p = np.asarray(base.position, np.float64) + rng.normal(size=3) * 2.0 * k
            d = np.asarray(base.direction, np.float64) + rng.normal(size=3) * 0.2 * k
            cam = Camera(tuple(p), tuple(d / np.linalg.norm(d)))
            ovis = o.render_visbuffer(cam, (0.3, 0.8, -0.4))
            g.render(cam, (0.3, 0.8, -0.4))
            dk = int((g.read_visbuffer_u64() != ovis).sum())
            e = int(np.abs(o.shade().astype(int) - g.read_color().astype(int)).max())
            if dk or e > 2:
                bad += 1
                print("MISMATCH field seed", seed, w, h, "lods", lods, occlusion, k, "keys", dk, "colour", e, flush=True)
        g.close()
    if seed % 10 == 0:
        print("field seed", seed, "ok so far, bad =", bad, flush=True)
print("fields done, mismatching frames:", bad)

# third part: sort-last partitions of random scenes, composited with a 64-bit minimum on the host, against the
# unpartitioned frame of the same renderer (bit for bit)
bad = 0
for seed in ((range(30000, 30000 + 30 * SCALE) if SCALE > 1 else range(300, 330)) if os.environ.get("SWEEP_PART", "all") in ("all", "partitions") else ()):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(32, 700)), int(rng.integers(32, 500))
    if seed % 2:
        sc = soup(rng, int(rng.integers(100, 1200)), bool(rng.integers(0, 2)))
    else:
        sc = synth.make_field(int(rng.integers(20, 200)), int(rng.integers(1, 4)), lods=1, alpha_fraction=float(rng.choice([0.0, 0.5])),
                              with_env=False, seed=seed, prepared=True)
    conf = RendererConfig(cull_enabled=bool(rng.integers(0, 2)))
    g = Renderer(w, h, conf); g.set_scene(sc)
    for _ in range(2):
        g.render_visbuffer()
    full = g.read_visbuffer_u64().copy()
    parts = int(rng.integers(2, 8))
    total = int(sc.meshlet_visibility_count)
    cuts = sorted(set([0, total] + [int(c) for c in rng.integers(0, total + 1, parts - 1)]))
    acc = np.full_like(full, np.uint64(0x7FFFFFFFFFFFFFFF))
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        p = Renderer(w, h, conf); p.set_scene(sc); p.set_partition(lo, hi)
        for _ in range(2):
            p.render_visbuffer()
        acc = np.minimum(acc, p.read_visbuffer_u64())
        p.close()
    d = int((acc != full).sum())
    if d:
        bad += 1
        print("MISMATCH partitions seed", seed, w, h, cuts, "keys", d, flush=True)
    g.close()
print("partitions done, mismatching scenes:", bad)

# fourth part: block-compressed textures of random shapes, GPU decode against the CPU decode
if os.environ.get("SWEEP_PART", "all") in ("all", "bc"):
    import ctypes as C
    from renderer_amd import dds
    from renderer_amd.scene import BlockImage
    bad = 0
    for seed in range(400, 600):
        rng = np.random.default_rng(seed)
        fmt = int(rng.integers(3, 9))  # RDR_FORMAT_BC1 .. BC7_SRGB
        w, h = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        mips = int(rng.integers(1, int(np.log2(max(w, h))) + 2))
        img = BlockImage(np.zeros(0, np.uint8), w, h, mips, fmt)
        img.data = rng.integers(0, 256, sum(img.level_bytes(l) for l in range(mips)), dtype=np.uint8)
        got = dds.decode(img).mips
        dims = [(max(1, h >> l), max(1, w >> l)) for l in range(mips)]
        want = np.zeros(sum(a * b for a, b in dims) * 4, np.uint8)
        assert oracle.lib().orc_decode_bc(fmt, w, h, mips, img.data.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p)) == 0
        flat = np.concatenate([m.reshape(-1) for m in got])
        if not np.array_equal(flat, want):
            bad += 1
            print("MISMATCH bc seed", seed, fmt, w, h, mips, flush=True)
    print("bc done, mismatching images:", bad)
