"""The C++ host (renderer_amd/host/harness.cpp over renderer.hpp over the C ABI): it builds, and on a
GPU it renders a scene dump to the same bytes as the Python host."""
import os
import subprocess

import numpy as np
import pytest

from renderer_amd import RendererConfig, synth
from renderer_amd.scene_bin import save_scene_bin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "renderer_amd", "host", "rdr_harness")


def test_harness_is_built_and_links_the_library():
    assert os.path.exists(HARNESS), "run `make -C renderer_amd/csrc` (builds renderer_amd/host/rdr_harness)"
    out = subprocess.run(["ldd", HARNESS], capture_output=True, text=True).stdout
    assert "librdr_hip.so" in out and "not found" not in out.split("librdr_hip.so")[1].splitlines()[0]


def test_harness_reports_errors_like_the_library(tmp_path):
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"not a scene")
    p = subprocess.run([HARNESS, str(bad), "64", "36"], capture_output=True, text=True)
    assert p.returncode == 1 and "scene.bin" in p.stderr


@pytest.mark.gpu
def test_harness_renders_the_same_bytes_as_the_python_host(tmp_path):
    from renderer_amd import Renderer
    sc = synth.make_atrium()
    save_scene_bin(sc, tmp_path / "atrium.bin")
    w, h = 256, 144
    p = subprocess.run([HARNESS, str(tmp_path / "atrium.bin"), str(w), str(h), "--skybox", "--frames", "2",
                        "--dump-vis", str(tmp_path / "vis.u32x2"), "--dump-color", str(tmp_path / "color.rgba8")],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "pixels covered" in p.stdout
    g = Renderer(w, h, RendererConfig(cull_enabled=False, skybox=True))
    g.set_scene(sc)
    g.render()
    g.render()
    vis = np.fromfile(tmp_path / "vis.u32x2", np.uint32).reshape(h, w, 2)
    color = np.fromfile(tmp_path / "color.rgba8", np.uint8).reshape(h, w, 4)
    assert np.array_equal(vis, g.read_visbuffer())
    assert np.array_equal(color, g.read_color())
