"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the CPU
oracle): the oracle must keep reproducing them, and the HIP path must match them."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402

from renderer_amd import RendererConfig  # noqa: E402
from helpers import psnr  # noqa: E402

NAMES = sorted(make_golden.CASES)


def load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(oracle, name):
    g = load(name)
    sc, o, vis, color = make_golden.render_case(name)
    assert np.array_equal(vis, g["vis"])
    assert np.array_equal(o.draw_lod[: len(sc.mesh_draws)], g["draw_lod"])
    assert [o.stats.triangles_rasterised, o.stats.fragments, o.stats.meshlets_visible] == g["stats"].tolist()
    # transcendental functions come from the host libm: allow one 8-bit step
    assert np.abs(color.astype(int) - g["color"].astype(int)).max() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_golden(name):
    from renderer_amd import Renderer
    g = load(name)
    factory, w, h, cfg = make_golden.CASES[name]
    sc = factory()
    r = Renderer(w, h, RendererConfig(**cfg))
    r.set_scene(sc)
    r.render()
    assert np.array_equal(r.read_visbuffer_u64(), g["vis"])
    col = r.read_color()
    assert np.abs(col.astype(int) - g["color"].astype(int)).max() <= 2
    assert psnr(col, g["color"]) >= 40.0
