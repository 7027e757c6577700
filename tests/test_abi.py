"""CPU-side checks of the C-ABI library: it loads, exports exactly what include/rdr_abi.h
declares, its host-only arithmetic matches the oracle, and it refuses to run without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from renderer_amd import abi, renderer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rdr_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rdr_[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree():
    L = renderer.load_library()
    names = declared_symbols()
    assert set(names) == set(renderer.ABI_SYMBOLS)
    for n in names:
        assert hasattr(L, n), f"librdr_hip.so does not export {n}"


def test_struct_sizes_match_reference_layouts():
    # scenery/src/mesh.rs, vertex.rs, renderer.rs:56-76
    assert abi.VERTEX.itemsize == 64
    assert abi.MESHLET.itemsize == 32 and abi.MESHLET.fields["cone_axis"][1] == 16
    assert abi.MESHLET.fields["vertex_offset"][1] == 20 and abi.MESHLET.fields["triangle_count"][1] == 28
    assert abi.MESH.itemsize == 160 and abi.MESH.fields["lods"][1] == 32
    assert abi.MESH_DRAW.itemsize == 16 and abi.MATERIAL.itemsize == 272
    assert abi.MATERIAL.fields["albedo_texture"][1] == 192 and abi.MATERIAL.fields["ior"][1] == 256
    assert C.sizeof(abi.Ubo) == 288
    for name, off in (("view_proj", 0), ("view", 64), ("proj", 128), ("camera_position", 192), ("near_plane", 208),
                      ("far_plane", 212), ("depth_pyramid_width", 216), ("frustum", 224), ("draw_count", 240),
                      ("lod_enabled", 244), ("cull_enabled", 248), ("lod_target", 252), ("screen_width", 256),
                      ("screen_height", 260), ("sun_direction", 272)):
        assert getattr(abi.Ubo, name).offset == off


@pytest.mark.parametrize("w,h", [(3840, 2160), (1280, 720), (64, 36), (7680, 4320)])
def test_ubo_matches_oracle_bitwise(oracle, w, h):
    rng = np.random.default_rng(w)
    for _ in range(20):
        pos = rng.uniform(-100, 100, 3)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        if abs(d[1]) > 0.99:
            continue
        sun = rng.normal(size=3)
        a = renderer.build_ubo(w, h, pos, d, sun, 123, True, False)
        b = oracle.build_ubo(w, h, pos, d, sun, 123, True, False)
        assert bytes(a) == bytes(b)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(renderer.RendererError):
        renderer.Renderer(64, 36)


def test_bad_arguments_are_errors_not_crashes():
    L = renderer.load_library()
    out = C.c_void_p()
    dev = (C.c_int * 1)(0)
    assert L.rdr_create(None, 0, 10, dev, 1, C.byref(out)) != 0
    assert b"extent" in L.rdr_last_error()
    assert L.rdr_create(None, 64, 64, dev, 2, C.byref(out)) != 0
    assert b"one HIP device" in L.rdr_last_error()
    assert L.rdr_set_scene(None, None) != 0
    assert L.rdr_render(None, None, None, None) != 0
    assert L.rdr_read_color(None, None) != 0
    L.rdr_destroy(None)
