"""KTX 2.0 containers for the environment inputs (the files the reference's ``EnvMap`` JSON names,
scenery/src/scene.rs:26-33, read through libktx in renderer/src/renderer/image_loader.rs:180-232 and
written by bake/src/image_utils.rs:6-115): RGBA8 / RGBA8-sRGB / RGBA32F texels or BC7 blocks, images
and cubemaps with mip chains, no supercompression.  The reference's bake writes its BRDF table and
its three prefiltered maps as texels; its skybox (bake/src/cubemaps.rs:158-163) and the textures it
converts (bake/src/ptto_converter.rs:9-39) go through Basis to BC7 blocks (image_utils.rs:57-65), which
the file then holds as VK_FORMAT_BC7_UNORM_BLOCK.  A 2D image of blocks becomes a ``BlockImage`` (the
library expands it at upload); a cubemap of blocks is expanded here, face by face, with the library's
decode kernel, because the cubemap boundary takes texels.

File layout (Khronos KTX 2.0): 12-byte identifier, 13 u32 header words + 2 u64, one
(offset, length, uncompressed length) u64 triple per level, data format descriptor, then the level
data with the *smallest level first*, each level holding its faces in +X,-X,+Y,-Y,+Z,-Z order."""
import json
import os
import struct

import numpy as np

from . import abi
from .scene import BlockImage, Cubemap, EnvMap, Image

IDENTIFIER = b"\xabKTX 20\xbb\r\n\x1a\n"
VK_R8G8B8A8_UNORM, VK_R8G8B8A8_SRGB, VK_R32G32B32A32_SFLOAT = 37, 43, 109
VK_BC7_UNORM_BLOCK, VK_BC7_SRGB_BLOCK = 145, 146
_FORMATS = {VK_R8G8B8A8_UNORM: (np.uint8, 4, False), VK_R8G8B8A8_SRGB: (np.uint8, 4, True),
            VK_R32G32B32A32_SFLOAT: (np.float32, 16, False)}
_BLOCK_FORMATS = {VK_BC7_UNORM_BLOCK: abi.FORMAT_BC7, VK_BC7_SRGB_BLOCK: abi.FORMAT_BC7_SRGB}


class Ktx2Error(ValueError):
    pass


def _open(path):
    """-> (file bytes, VkFormat, width, height, faces, level count) of a file this module can read."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:12] != IDENTIFIER:
        raise Ktx2Error(f"{path}: not a KTX 2.0 file")
    if len(data) < 80:
        raise Ktx2Error(f"{path}: header is truncated")
    (vk_format, _type_size, width, height, depth, layers, faces, levels, scheme) = struct.unpack_from("<9I", data, 12)
    if scheme != 0:
        raise Ktx2Error(f"{path}: supercompression scheme {scheme} is not supported (store the levels uncompressed)")
    if vk_format not in _FORMATS and vk_format not in _BLOCK_FORMATS:
        raise Ktx2Error(f"{path}: VkFormat {vk_format} is not supported; only RGBA8 (37, 43), RGBA32F (109) and "
                        "BC7 blocks (145, 146)")
    if depth > 1 or layers > 1 or faces not in (1, 6):
        raise Ktx2Error(f"{path}: only 2D images and cubemaps without array layers are supported")
    levels = max(1, levels)
    if levels > 17 or width == 0 or 80 + 24 * levels > len(data):
        raise Ktx2Error(f"{path}: level index is truncated or the dimensions are invalid")
    return data, vk_format, width, max(1, height), faces, levels


def read_ktx2(path):
    """Returns (levels, faces, srgb): ``levels[l]`` is a (faces, h, w, 4) array, level 0 first.
    Files of texels only; for files of BC7 blocks see ``read_ktx2_blocks``."""
    data, vk_format, width, height, faces, levels = _open(path)
    if vk_format in _BLOCK_FORMATS:
        raise Ktx2Error(f"{path}: holds BC7 blocks (VkFormat {vk_format}), not texels: load it with load_image_ktx2 / "
                        "load_cubemap_ktx2 or read_ktx2_blocks")
    dtype, texel, srgb = _FORMATS[vk_format]
    out = []
    for l in range(levels):
        off, length, _ = struct.unpack_from("<3Q", data, 80 + 24 * l)
        w, h = max(1, width >> l), max(1, height >> l)
        need = faces * w * h * texel
        if length < need or off + need > len(data):
            raise Ktx2Error(f"{path}: level {l} is truncated")
        out.append(np.frombuffer(data, dtype=dtype, count=need // np.dtype(dtype).itemsize, offset=off)
                   .reshape(faces, h, w, 4).copy())
    return out, faces, srgb


def read_ktx2_blocks(path):
    """A file of BC7 blocks -> (levels, faces, width, height, format): ``levels[l]`` is a (faces, bytes)
    uint8 array holding each face's ceil(w/4) x ceil(h/4) blocks row-major, level 0 first; ``format``
    is ``abi.FORMAT_BC7`` or ``abi.FORMAT_BC7_SRGB``."""
    data, vk_format, width, height, faces, levels = _open(path)
    if vk_format not in _BLOCK_FORMATS:
        raise Ktx2Error(f"{path}: holds texels (VkFormat {vk_format}), not blocks")
    out = []
    for l in range(levels):
        off, length, _ = struct.unpack_from("<3Q", data, 80 + 24 * l)
        w, h = max(1, width >> l), max(1, height >> l)
        per_face = ((w + 3) // 4) * ((h + 3) // 4) * 16
        if length < faces * per_face or off + faces * per_face > len(data):
            raise Ktx2Error(f"{path}: level {l} is truncated")
        out.append(np.frombuffer(data, np.uint8, count=faces * per_face, offset=off).reshape(faces, per_face).copy())
    return out, faces, width, height, _BLOCK_FORMATS[vk_format]


def _dfd(vk_format):
    """Basic data format descriptor block for the three formats written here."""
    dtype, texel, srgb = _FORMATS[vk_format]
    bits = 32 if dtype == np.float32 else 8
    samples = b""
    for i, channel in enumerate((0, 1, 2, 15)):  # R, G, B, A
        qual = 0
        if dtype == np.float32:
            qual, lo, hi = 0x80 | 0x40, 0xBF800000, 0x3F800000  # float, signed; -1.0 .. 1.0
        else:
            lo, hi = 0, 255
            if srgb and channel == 15:
                qual = 0x10  # alpha stays linear
        samples += struct.pack("<HBB4BII", i * bits, bits - 1, channel | qual, 0, 0, 0, 0, lo, hi)
    block_size = 24 + len(samples)
    block = struct.pack("<IHH", 0, 2, block_size)                          # vendor 0 / basic, version 2, size
    block += struct.pack("<4B", 1, 1, 2 if srgb else 1, 0)                 # RGBSDA, BT.709, transfer, straight alpha
    block += struct.pack("<4B", 0, 0, 0, 0) + struct.pack("<8B", texel, 0, 0, 0, 0, 0, 0, 0)
    block += samples
    return struct.pack("<I", 4 + len(block)) + block


def write_ktx2(path, levels, srgb=False):
    """``levels[l]``: (faces, h, w, 4) uint8 or float32 arrays, level 0 first (faces 1 or 6)."""
    first = np.asarray(levels[0])
    faces, height, width = first.shape[0], first.shape[1], first.shape[2]
    if first.dtype == np.float32:
        vk_format = VK_R32G32B32A32_SFLOAT
    elif first.dtype == np.uint8:
        vk_format = VK_R8G8B8A8_SRGB if srgb else VK_R8G8B8A8_UNORM
    else:
        raise Ktx2Error("levels must be uint8 or float32")
    _, texel, _ = _FORMATS[vk_format]
    dfd = _dfd(vk_format)
    n = len(levels)
    dfd_off = 80 + 24 * n
    pos = dfd_off + len(dfd)
    align = 16 if texel == 16 else 4
    placed = {}
    for l in range(n - 1, -1, -1):  # smallest level first in the file
        pos = (pos + align - 1) // align * align
        blob = np.ascontiguousarray(levels[l]).tobytes()
        placed[l] = (pos, blob)
        pos += len(blob)
    out = bytearray(pos)
    out[:12] = IDENTIFIER
    struct.pack_into("<9I", out, 12, vk_format, 4 if texel == 16 else 1, width, height, 0, 0, faces, n, 0)
    struct.pack_into("<4I2Q", out, 48, dfd_off, len(dfd), 0, 0, 0, 0)
    for l in range(n):
        off, blob = placed[l]
        struct.pack_into("<3Q", out, 80 + 24 * l, off, len(blob), len(blob))
        out[off:off + len(blob)] = blob
    out[dfd_off:dfd_off + len(dfd)] = dfd
    with open(path, "wb") as f:
        f.write(out)


def _dfd_bc7(srgb):
    """Data format descriptor of a BC7 file: one 128-bit sample covering the 4 x 4 texel block."""
    sample = struct.pack("<HBB4BII", 0, 127, 0, 0, 0, 0, 0, 0, 0xFFFFFFFF)
    block = struct.pack("<IHH", 0, 2, 24 + len(sample))
    block += struct.pack("<4B", 134, 1, 2 if srgb else 1, 0)               # BC7, BT.709, transfer, straight alpha
    block += struct.pack("<4B", 3, 3, 0, 0) + struct.pack("<8B", 16, 0, 0, 0, 0, 0, 0, 0)  # 4 x 4 texels, 16 bytes
    block += sample
    return struct.pack("<I", 4 + len(block)) + block


def write_ktx2_blocks(path, levels, width, height, srgb=False):
    """``levels[l]``: (faces, bytes) uint8 arrays of BC7 blocks, level 0 first (test fixtures and tools)."""
    faces = np.asarray(levels[0]).shape[0]
    dfd = _dfd_bc7(srgb)
    n = len(levels)
    dfd_off = 80 + 24 * n
    pos = dfd_off + len(dfd)
    placed = {}
    for l in range(n - 1, -1, -1):  # smallest level first in the file
        pos = (pos + 15) // 16 * 16
        blob = np.ascontiguousarray(levels[l], np.uint8).tobytes()
        w, h = max(1, width >> l), max(1, height >> l)
        if len(blob) != faces * ((w + 3) // 4) * ((h + 3) // 4) * 16:
            raise Ktx2Error(f"level {l}: {len(blob)} bytes do not hold {faces} faces of {w}x{h} texels in BC7 blocks")
        placed[l] = (pos, blob)
        pos += len(blob)
    out = bytearray(pos)
    out[:12] = IDENTIFIER
    struct.pack_into("<9I", out, 12, VK_BC7_SRGB_BLOCK if srgb else VK_BC7_UNORM_BLOCK, 1, width, height, 0, 0, faces, n, 0)
    struct.pack_into("<4I2Q", out, 48, dfd_off, len(dfd), 0, 0, 0, 0)
    for l in range(n):
        off, blob = placed[l]
        struct.pack_into("<3Q", out, 80 + 24 * l, off, len(blob), len(blob))
        out[off:off + len(blob)] = blob
    out[dfd_off:dfd_off + len(dfd)] = dfd
    with open(path, "wb") as f:
        f.write(out)


def _holds_blocks(path):
    with open(path, "rb") as f:
        head = f.read(16)
    return len(head) == 16 and head[:12] == IDENTIFIER and struct.unpack_from("<I", head, 12)[0] in _BLOCK_FORMATS


def load_cubemap_ktx2(path, device=0):
    """-> ``Cubemap``.  A file of BC7 blocks (the reference's skybox) is expanded to RGBA8 on GPU ``device``."""
    if _holds_blocks(path):
        from . import dds
        levels, faces, width, height, fmt = read_ktx2_blocks(path)
        if faces != 6 or width != height:
            raise Ktx2Error(f"{path}: not a cubemap")
        if fmt != abi.FORMAT_BC7:
            raise Ktx2Error(f"{path}: an sRGB cubemap; cubemaps are RGBA8_UNORM or RGBA32F at the library's boundary")
        mips = []
        for l, level in enumerate(levels):
            n = max(1, width >> l)
            mips.append(np.stack([dds.decode(BlockImage(level[f], n, n, 1, fmt), device).mips[0] for f in range(6)]))
        return Cubemap(mips)
    levels, faces, _ = read_ktx2(path)
    if faces != 6:
        raise Ktx2Error(f"{path}: not a cubemap")
    return Cubemap(levels)


def load_image_ktx2(path):
    """-> ``Image``, or ``BlockImage`` for a file of BC7 blocks (image_loader.rs:180-241 passes the file's
    VkFormat on unchanged, so the sRGB code stays what the file says)."""
    if _holds_blocks(path):
        levels, faces, width, height, fmt = read_ktx2_blocks(path)
        if faces != 1:
            raise Ktx2Error(f"{path}: a cubemap where a 2D image was expected")
        return BlockImage(np.concatenate([lv[0] for lv in levels]), width, height, len(levels), fmt)
    levels, faces, srgb = read_ktx2(path)
    if faces != 1:
        raise Ktx2Error(f"{path}: a cubemap where a 2D image was expected")
    return Image([lv[0] for lv in levels], srgb=srgb)


def save_cubemap_ktx2(path, cubemap):
    write_ktx2(path, cubemap.mips)


def save_image_ktx2(path, image):
    write_ktx2(path, [np.asarray(m)[None] for m in image.mips], srgb=image.srgb)


_ENV_FIELDS = ("skybox", "brdf_lut", "lambertian", "ggx", "charlie")


def load_env_json(path):
    """The reference's environment description: a JSON object naming five KTX2 files
    (app/src/main.rs:32, scenery/src/scene.rs:26-33).  Relative paths are taken from the JSON's folder."""
    with open(path) as f:
        doc = json.load(f)
    base = os.path.dirname(os.path.abspath(path))
    files = {k: os.path.join(base, doc[k]) for k in _ENV_FIELDS}
    return EnvMap(brdf_lut=load_image_ktx2(files["brdf_lut"]), lambertian=load_cubemap_ktx2(files["lambertian"]),
                  ggx=load_cubemap_ktx2(files["ggx"]), charlie=load_cubemap_ktx2(files["charlie"]),
                  skybox=load_cubemap_ktx2(files["skybox"]))


def save_env_json(path, env, stem="env"):
    """Write ``env`` as five KTX2 files next to ``path`` and the JSON that names them."""
    base = os.path.dirname(os.path.abspath(path))
    names = {k: f"{stem}_{k}.ktx2" for k in _ENV_FIELDS}
    save_image_ktx2(os.path.join(base, names["brdf_lut"]), env.brdf_lut)
    for k in ("skybox", "lambertian", "ggx", "charlie"):
        save_cubemap_ktx2(os.path.join(base, names[k]), getattr(env, k))
    with open(path, "w") as f:
        json.dump(names, f, indent=1)
