"""The frames of the extended random sweep (profiles/parity_sweep.py) that once exceeded the colour
tolerance, as collected tests: two triangle soups whose huge triangles make the barycentrics cancel
catastrophically, and one instanced field with a simplified LOD chain.  Keys bit-exact, colour within 2/255."""
import numpy as np
import pytest

from renderer_amd import Renderer, RendererConfig, synth
from renderer_amd.scene import Camera
from test_gpu_random import soup

pytestmark = pytest.mark.gpu
SUN = (0.3, 0.8, -0.4)


def compare(oracle, sc, w, h, conf, cam):
    o = oracle.OracleRenderer(w, h, conf)
    o.set_scene(sc)
    g = Renderer(w, h, conf)
    g.set_scene(sc)
    ovis = o.render_visbuffer(cam, SUN)
    g.render(cam, SUN)
    assert int((g.read_visbuffer_u64() != ovis).sum()) == 0
    err = np.abs(o.shade().astype(int) - g.read_color().astype(int))
    g.close()
    assert err.max() <= 2, f"colour off by {err.max()}/255 at {np.argwhere(err == err.max())[:4].tolist()}"


@pytest.mark.parametrize("seed,k", [(118, 1), (157, 0)])
def test_soups_with_triangles_many_screens_large(oracle, seed, k):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    masked = bool(rng.integers(0, 2))
    sc = soup(rng, int(rng.integers(50, 1500)), masked)
    a = 0.5 * k
    cam = Camera((0.3 * k, 0.1 * k, -0.5 * k), (np.sin(a), 0.05 * k, -np.cos(a)))
    compare(oracle, sc, w, h, RendererConfig(cull_enabled=False), cam)


def test_field_with_lod_chain_seed_215(oracle):
    seed = 215
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(64, 900)), int(rng.integers(64, 600))
    subdiv = int(rng.integers(1, 4))
    prepared = bool(rng.integers(0, 2))
    lods = int(rng.integers(1, 5)) if prepared else int(rng.integers(1, subdiv + 2))
    sc = synth.make_field(int(rng.integers(20, 300)), subdiv, lods=lods, alpha_fraction=float(rng.choice([0.0, 0.3, 1.0])),
                          with_env=True, seed=seed, prepared=prepared)
    compare(oracle, sc, w, h, RendererConfig(cull_enabled=False, lod_enabled=lods > 1), sc.camera)
