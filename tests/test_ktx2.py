"""KTX 2.0 reader / writer for the environment inputs: header fields, level order (smallest first),
round trips, the reference's EnvMap JSON, files of BC7 blocks (what the reference's bake writes for its
skybox and converted textures, bake/src/image_utils.rs:57-65), and the formats that are refused."""
import ctypes as C
import struct

import numpy as np
import pytest

from renderer_amd import abi, synth
from renderer_amd.ktx2 import (IDENTIFIER, Ktx2Error, load_cubemap_ktx2, load_env_json, load_image_ktx2, read_ktx2,
                               read_ktx2_blocks, save_cubemap_ktx2, save_env_json, save_image_ktx2, write_ktx2,
                               write_ktx2_blocks)
from renderer_amd.scene import BlockImage, Cubemap, Image


def test_container_layout(tmp_path):
    rng = np.random.default_rng(1)
    mips = [rng.random((6, n, n, 4), dtype=np.float32) for n in (8, 4, 2, 1)]
    path = tmp_path / "c.ktx2"
    save_cubemap_ktx2(path, Cubemap(mips))
    raw = path.read_bytes()
    assert raw[:12] == IDENTIFIER
    vk, type_size, w, h, d, layers, faces, levels, scheme = struct.unpack_from("<9I", raw, 12)
    assert (vk, type_size, w, h, d, layers, faces, levels, scheme) == (109, 4, 8, 8, 0, 0, 6, 4, 0)
    dfd_off, dfd_len, kvd_off, kvd_len = struct.unpack_from("<4I", raw, 48)
    assert dfd_off == 80 + 24 * 4 and struct.unpack_from("<I", raw, dfd_off)[0] == dfd_len and (kvd_off, kvd_len) == (0, 0)
    index = [struct.unpack_from("<3Q", raw, 80 + 24 * l) for l in range(4)]
    assert [i[1] for i in index] == [6 * n * n * 16 for n in (8, 4, 2, 1)] and all(i[1] == i[2] for i in index)
    assert index[3][0] < index[2][0] < index[1][0] < index[0][0]         # the smallest level comes first in the file
    assert all(i[0] % 16 == 0 for i in index)
    assert index[0][0] + index[0][1] == len(raw)
    back = load_cubemap_ktx2(path)
    assert len(back.mips) == 4 and all(np.array_equal(a, b) for a, b in zip(back.mips, mips))


def test_images_and_srgb_flag(tmp_path):
    rng = np.random.default_rng(2)
    img = Image([rng.integers(0, 256, (8 >> l, 16 >> l, 4), dtype=np.uint8) for l in range(3)], srgb=True)
    save_image_ktx2(tmp_path / "i.ktx2", img)
    back = load_image_ktx2(tmp_path / "i.ktx2")
    assert back.srgb and len(back.mips) == 3 and all(np.array_equal(a, b) for a, b in zip(back.mips, img.mips))
    assert struct.unpack_from("<I", (tmp_path / "i.ktx2").read_bytes(), 12)[0] == 43
    lut = Image([rng.random((16, 16, 4), dtype=np.float32)])
    save_image_ktx2(tmp_path / "l.ktx2", lut)
    assert np.array_equal(load_image_ktx2(tmp_path / "l.ktx2").mips[0], lut.mips[0])
    with pytest.raises(Ktx2Error, match="2D image was expected"):
        save_cubemap_ktx2(tmp_path / "c.ktx2", Cubemap([np.zeros((6, 2, 2, 4), np.uint8)]))
        load_image_ktx2(tmp_path / "c.ktx2")


def test_refused_files(tmp_path):
    (tmp_path / "x.ktx2").write_bytes(b"not a ktx file at all")
    with pytest.raises(Ktx2Error, match="not a KTX"):
        read_ktx2(tmp_path / "x.ktx2")
    write_ktx2(tmp_path / "ok.ktx2", [np.zeros((1, 4, 4, 4), np.uint8)])
    raw = bytearray((tmp_path / "ok.ktx2").read_bytes())
    bc7 = bytearray(raw)
    struct.pack_into("<I", bc7, 12, 145)               # VK_FORMAT_BC7_UNORM_BLOCK: what the reference's skybox holds
    (tmp_path / "bc7.ktx2").write_bytes(bc7)
    with pytest.raises(Ktx2Error, match="BC7"):               # read_ktx2 gives texels; blocks have their own reader
        read_ktx2(tmp_path / "bc7.ktx2")
    bc1 = bytearray(raw)
    struct.pack_into("<I", bc1, 12, 133)               # VK_FORMAT_BC1_RGBA_UNORM_BLOCK
    (tmp_path / "bc1.ktx2").write_bytes(bc1)
    with pytest.raises(Ktx2Error, match="VkFormat 133"):
        load_image_ktx2(tmp_path / "bc1.ktx2")
    zstd = bytearray(raw)
    struct.pack_into("<I", zstd, 44, 2)                # supercompressionScheme = Zstandard
    (tmp_path / "z.ktx2").write_bytes(zstd)
    with pytest.raises(Ktx2Error, match="supercompression"):
        read_ktx2(tmp_path / "z.ktx2")
    (tmp_path / "short.ktx2").write_bytes(raw[:-8])
    with pytest.raises(Ktx2Error, match="truncated"):
        read_ktx2(tmp_path / "short.ktx2")


def _bc7_levels(rng, faces, width, height, levels):
    out = []
    for l in range(levels):
        w, h = max(1, width >> l), max(1, height >> l)
        out.append(rng.integers(0, 256, (faces, ((w + 3) // 4) * ((h + 3) // 4) * 16), dtype=np.uint8))
    return out


@pytest.mark.parametrize("srgb", [False, True])
def test_bc7_image_file_becomes_a_block_image(tmp_path, srgb):
    rng = np.random.default_rng(3)
    levels = _bc7_levels(rng, 1, 20, 12, 4)  # 20x12, 10x6, 5x3, 2x1: 5x3, 3x2, 2x1 and 1 blocks
    path = tmp_path / "t.ktx2"
    write_ktx2_blocks(path, levels, 20, 12, srgb=srgb)
    raw = path.read_bytes()
    vk, type_size, w, h, d, layers, faces, n, scheme = struct.unpack_from("<9I", raw, 12)
    assert (vk, type_size, w, h, d, layers, faces, n, scheme) == (146 if srgb else 145, 1, 20, 12, 0, 0, 1, 4, 0)
    index = [struct.unpack_from("<3Q", raw, 80 + 24 * l) for l in range(4)]
    assert [i[1] for i in index] == [240, 96, 32, 16] and all(i[0] % 16 == 0 for i in index)
    assert index[3][0] < index[2][0] < index[1][0] < index[0][0]
    dfd_off = struct.unpack_from("<I", raw, 48)[0]
    assert raw[dfd_off + 12] == 134 and raw[dfd_off + 14] == (2 if srgb else 1)   # colour model BC7, transfer function
    img = load_image_ktx2(path)
    assert isinstance(img, BlockImage) and (img.width, img.height, img.mip_count) == (20, 12, 4)
    assert img.format == (abi.FORMAT_BC7_SRGB if srgb else abi.FORMAT_BC7)
    assert np.array_equal(img.data, np.concatenate([lv[0] for lv in levels]))     # level 0 first, as the library takes them
    back, faces, w, h, fmt = read_ktx2_blocks(path)
    assert (faces, w, h, fmt) == (1, 20, 12, img.format) and all(np.array_equal(a, b) for a, b in zip(back, levels))
    path.write_bytes(raw[:-8])
    with pytest.raises(Ktx2Error, match="truncated"):
        load_image_ktx2(path)
    with pytest.raises(Ktx2Error, match="do not hold"):
        write_ktx2_blocks(path, [levels[1]], 20, 12)


def test_gltf_takes_ktx2_images(tmp_path):
    """image_loader.rs:264: the extension picks the loader; converted textures are .ktx2 files of BC7 blocks."""
    from renderer_amd.gltf import load_gltf, save_gltf
    import os
    sc = synth.make_triangle_scene([(0, 0, 5), (1, 0, 5), (0, 1, 5)])
    tex = sc.add_image(synth.checker_texture(16, 2))
    sc.materials = sc.materials[:0]
    sc.add_material(albedo_texture=tex)
    path = str(tmp_path / "s.gltf")
    save_gltf(sc, path)
    png = [f for f in os.listdir(tmp_path) if f.endswith(".png")][0]
    levels = _bc7_levels(np.random.default_rng(5), 1, 16, 16, 3)
    write_ktx2_blocks(tmp_path / (os.path.splitext(png)[0] + ".ktx2"), levels, 16, 16)
    doc = open(path).read().replace(png, os.path.splitext(png)[0] + ".ktx2")
    open(path, "w").write(doc)
    got = load_gltf(path).images[0]
    assert isinstance(got, BlockImage) and got.format == abi.FORMAT_BC7 and got.mip_count == 3
    assert np.array_equal(got.data, np.concatenate([lv[0] for lv in levels]))


@pytest.mark.gpu
def test_bc7_cubemap_file_is_expanded_with_the_library_decode(tmp_path, oracle):
    rng = np.random.default_rng(4)
    levels = _bc7_levels(rng, 6, 16, 16, 5)  # faces of 16, 8, 4, 2, 1
    path = tmp_path / "sky.ktx2"
    write_ktx2_blocks(path, levels, 16, 16)
    cm = load_cubemap_ktx2(path)
    assert [m.shape for m in cm.mips] == [(6, n, n, 4) for n in (16, 8, 4, 2, 1)] and cm.format == abi.FORMAT_RGBA8_UNORM
    for l, n in enumerate((16, 8, 4, 2, 1)):
        for f in range(6):
            want = np.zeros((n, n, 4), np.uint8)
            blocks = np.ascontiguousarray(levels[l][f])
            assert oracle.lib().orc_decode_bc(abi.FORMAT_BC7, n, n, 1, blocks.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(cm.mips[l][f], want), (l, f)
    with pytest.raises(Ktx2Error, match="2D image was expected"):
        load_image_ktx2(path)
    write_ktx2_blocks(path, levels, 16, 16, srgb=True)
    with pytest.raises(Ktx2Error, match="sRGB cubemap"):
        load_cubemap_ktx2(path)


@pytest.mark.gpu
def test_env_json_with_a_bc7_skybox_feeds_the_skybox_pass(tmp_path, oracle):
    """The reference's environment description names a skybox file of BC7 blocks (bake/src/cubemaps.rs:158-163): read
    through load_env_json it must give the skybox pass the texels the CPU decode gives."""
    from renderer_amd import Renderer, RendererConfig
    from renderer_amd.scene import Camera
    env = synth.default_env()
    save_env_json(tmp_path / "env.json", env)
    rng = np.random.default_rng(11)
    n = 32
    levels = _bc7_levels(rng, 6, n, n, 3)
    write_ktx2_blocks(tmp_path / "env_skybox.ktx2", levels, n, n)
    back = load_env_json(tmp_path / "env.json")
    want = []
    for l, lv in enumerate(levels):
        m = max(1, n >> l)
        faces = np.zeros((6, m, m, 4), np.uint8)
        for f in range(6):
            blocks = np.ascontiguousarray(lv[f])
            assert oracle.lib().orc_decode_bc(abi.FORMAT_BC7, m, m, 1, blocks.ctypes.data_as(C.c_void_p), faces[f].ctypes.data_as(C.c_void_p)) == 0
        want.append(faces)
    assert all(np.array_equal(a, b) for a, b in zip(back.skybox.mips, want))
    sc = synth.make_triangle_scene([(0, 0, 5), (1, 0, 5), (0, 1, 5)])
    sc.env = back
    W, H = 96, 64
    conf = RendererConfig(skybox=True)
    r = Renderer(W, H, conf)
    r.set_scene(sc)
    cam = Camera((0.0, 0.0, 0.0), (0.3, 0.2, 1.0))
    r.render(cam)
    got = r.read_color()
    r.close()
    o = oracle.OracleRenderer(W, H, conf)
    o.set_scene(sc)
    o.render_visbuffer(cam)
    ref = o.shade()
    assert int(np.abs(got.astype(int) - ref.astype(int)).max()) <= 2
    assert len(np.unique(got.reshape(-1, 4), axis=0)) > 50  # the random sky shows


def test_env_json_round_trip(tmp_path):
    env = synth.default_env()
    save_env_json(tmp_path / "env.json", env)
    back = load_env_json(tmp_path / "env.json")
    assert np.array_equal(back.brdf_lut.mips[0], env.brdf_lut.mips[0])
    for k in ("skybox", "lambertian", "ggx", "charlie"):
        a, b = getattr(back, k), getattr(env, k)
        assert len(a.mips) == len(b.mips) and all(np.array_equal(x, y) for x, y in zip(a.mips, b.mips)), k
