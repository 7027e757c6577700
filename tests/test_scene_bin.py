"""scene.bin round trip: the flat dump a reference-side writer would produce is read back into
byte-identical arrays, and renders to the identical visibility buffer and image."""
import numpy as np

from renderer_amd import RendererConfig, synth
from renderer_amd.scene_bin import load_scene_bin, save_scene_bin


def test_round_trip_is_byte_identical_and_renders_the_same(oracle, tmp_path):
    sc = synth.make_atrium()
    path = tmp_path / "scene.bin"
    save_scene_bin(sc, path)
    back = load_scene_bin(path)
    for name in ("vertices", "indices", "meshlet_vertices", "meshlet_triangles", "meshlets", "meshes", "mesh_draws",
                 "materials"):
        assert getattr(back, name).tobytes() == getattr(sc, name).tobytes(), name
    assert np.asarray(back.global_transforms, np.float32).tobytes() == np.asarray(sc.global_transforms, np.float32).tobytes()
    assert back.meshlet_visibility_count == sc.meshlet_visibility_count
    assert len(back.images) == len(sc.images)
    for a, b in zip(back.images, sc.images):
        assert a.srgb == b.srgb and all(np.array_equal(x, y) for x, y in zip(a.mips, b.mips))
    for name in ("skybox", "lambertian", "ggx", "charlie"):
        a, b = getattr(back.env, name), getattr(sc.env, name)
        assert all(np.array_equal(x, y) for x, y in zip(a.mips, b.mips)), name
    w, h = 160, 90
    imgs = []
    for s in (sc, back):
        o = oracle.OracleRenderer(w, h, RendererConfig(skybox=True))
        o.set_scene(s)
        imgs.append((o.render_visbuffer().copy(), o.shade()))
    assert np.array_equal(imgs[0][0], imgs[1][0]) and np.array_equal(imgs[0][1], imgs[1][1])


def test_rejects_foreign_files(tmp_path):
    p = tmp_path / "x.bin"
    p.write_bytes(b"not a scene" * 10)
    try:
        load_scene_bin(p)
    except ValueError as e:
        assert "magic" in str(e)
    else:
        raise AssertionError("accepted a foreign file")
