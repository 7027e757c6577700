"""KTX 2.0 reader / writer for the environment inputs: header fields, level order (smallest first),
round trips, the reference's EnvMap JSON, and the formats that are refused."""
import struct

import numpy as np
import pytest

from renderer_amd import synth
from renderer_amd.ktx2 import (IDENTIFIER, Ktx2Error, load_cubemap_ktx2, load_env_json, load_image_ktx2, read_ktx2,
                               save_cubemap_ktx2, save_env_json, save_image_ktx2, write_ktx2)
from renderer_amd.scene import Cubemap, Image


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
    with pytest.raises(Ktx2Error, match="BC7"):
        read_ktx2(tmp_path / "bc7.ktx2")
    zstd = bytearray(raw)
    struct.pack_into("<I", zstd, 44, 2)                # supercompressionScheme = Zstandard
    (tmp_path / "z.ktx2").write_bytes(zstd)
    with pytest.raises(Ktx2Error, match="supercompression"):
        read_ktx2(tmp_path / "z.ktx2")
    (tmp_path / "short.ktx2").write_bytes(raw[:-8])
    with pytest.raises(Ktx2Error, match="truncated"):
        read_ktx2(tmp_path / "short.ktx2")


def test_env_json_round_trip(tmp_path):
    env = synth.default_env()
    save_env_json(tmp_path / "env.json", env)
    back = load_env_json(tmp_path / "env.json")
    assert np.array_equal(back.brdf_lut.mips[0], env.brdf_lut.mips[0])
    for k in ("skybox", "lambertian", "ggx", "charlie"):
        a, b = getattr(back, k), getattr(env, k)
        assert len(a.mips) == len(b.mips) and all(np.array_equal(x, y) for x, y in zip(a.mips, b.mips)), k
