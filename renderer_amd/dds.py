"""DirectDraw Surface (.dds) files with block-compressed data -> ``BlockImage``: the counterpart of
``PixelData::from_dds`` (renderer/src/renderer/image_loader.rs:63-135), which reads the file with
the ddsfile crate and hands the blocks to Vulkan unchanged.

Followed from the reference: the format comes from the DX10 header when there is one and from
the legacy four-character code otherwise; the sRGB variants of BC1-BC3 are bound as UNORM
(image_loader.rs:72-74) while BC7 keeps its sRGB variant (:81-82); the level count is the header's mip-map count (1 when absent); levels
are ceil(w/4) x ceil(h/4) blocks, tightly packed, largest first (:96-121).

Not supported here, with a clear error: BC6H (no decoder in this library), BC4 / BC5
signed, and BC5, which the reference binds as a BC4 image while keeping BC5's 16-byte block stride
(:77-78,88-91), i.e. it cannot render such a file correctly either.
"""
import ctypes as C
import struct

import numpy as np

from . import abi
from .scene import BlockImage, Image

_MAGIC = b"DDS "
_DDPF_FOURCC = 0x4
_DDSD_MIPMAPCOUNT = 0x20000
# DXGI_FORMAT codes of the DX10 header
_DXGI = {71: abi.FORMAT_BC1, 72: abi.FORMAT_BC1, 74: abi.FORMAT_BC2, 75: abi.FORMAT_BC2,
         77: abi.FORMAT_BC3, 78: abi.FORMAT_BC3, 80: abi.FORMAT_BC4, 98: abi.FORMAT_BC7, 99: abi.FORMAT_BC7_SRGB}
_DXGI_UNSUPPORTED = {81: "BC4_SNORM", 83: "BC5_UNORM", 84: "BC5_SNORM", 95: "BC6H_UF16", 96: "BC6H_SF16"}
# legacy four-character codes and the DXGI format ddsfile derives from them
_FOURCC = {b"DXT1": 71, b"DXT2": 74, b"DXT3": 74, b"DXT4": 77, b"DXT5": 77, b"ATI1": 80, b"BC4U": 80,
           b"BC4S": 81, b"ATI2": 83, b"BC5U": 83, b"BC5S": 84}
_DXGI_OF_FORMAT = {abi.FORMAT_BC1: 71, abi.FORMAT_BC2: 74, abi.FORMAT_BC3: 77, abi.FORMAT_BC4: 80,
                   abi.FORMAT_BC7: 98, abi.FORMAT_BC7_SRGB: 99}


class DdsError(ValueError):
    pass


def read_dds(path):
    """-> ``BlockImage``."""
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 128 or raw[:4] != _MAGIC:
        raise DdsError(f"{path}: not a DDS file")
    size, flags, height, width, _pitch, _depth, mips = struct.unpack_from("<7I", raw, 4)
    pf_size, pf_flags = struct.unpack_from("<2I", raw, 76)
    fourcc = raw[84:88]
    if size != 124 or pf_size != 32:
        raise DdsError(f"{path}: malformed DDS header")
    offset = 128
    if (pf_flags & _DDPF_FOURCC) and fourcc == b"DX10":
        if len(raw) < 148:
            raise DdsError(f"{path}: truncated DX10 header")
        dxgi = struct.unpack_from("<I", raw, 128)[0]
        offset = 148
    elif (pf_flags & _DDPF_FOURCC) and fourcc in _FOURCC:
        dxgi = _FOURCC[fourcc]
    else:
        raise DdsError(f"{path}: DDS file doesn't have a format")  # image_loader.rs:68
    if dxgi in _DXGI_UNSUPPORTED:
        raise DdsError(f"{path}: {_DXGI_UNSUPPORTED[dxgi]} is not supported")
    if dxgi not in _DXGI:
        raise DdsError(f"{path}: Format is not supported")  # image_loader.rs:84
    if width == 0 or height == 0:
        raise DdsError(f"{path}: empty image")
    mip_count = mips if (flags & _DDSD_MIPMAPCOUNT) and mips else 1
    if mip_count > 17:
        raise DdsError(f"{path}: {mip_count} mip levels")
    img = BlockImage(np.frombuffer(raw, np.uint8, offset=offset).copy(), width, height, mip_count, _DXGI[dxgi])
    need = sum(img.level_bytes(l) for l in range(mip_count))
    if img.data.size < need:
        raise DdsError(f"{path}: {img.data.size} bytes of blocks, {mip_count} levels of {width}x{height} need {need}")
    return img


def write_dds(path, image, dx10=False):
    """Writes a ``BlockImage`` (test fixtures and tools).  ``dx10``: with the extended header, which
    BC7 always gets (it has no legacy code)."""
    legacy = {abi.FORMAT_BC1: b"DXT1", abi.FORMAT_BC2: b"DXT3", abi.FORMAT_BC3: b"DXT5", abi.FORMAT_BC4: b"ATI1"}
    dx10 = dx10 or image.format not in legacy
    flags = 0x1 | 0x2 | 0x4 | 0x1000 | 0x80000 | (_DDSD_MIPMAPCOUNT if image.mip_count > 1 else 0)
    head = struct.pack("<7I", 124, flags, image.height, image.width, image.level_bytes(0), 0,
                       image.mip_count if image.mip_count > 1 else 0) + b"\0" * 44
    head += struct.pack("<2I", 32, _DDPF_FOURCC) + (b"DX10" if dx10 else legacy[image.format]) + b"\0" * 20
    head += struct.pack("<5I", 0x1000 | (0x400008 if image.mip_count > 1 else 0), 0, 0, 0, 0)
    assert len(head) == 124
    with open(path, "wb") as f:
        f.write(_MAGIC + head)
        if dx10:
            f.write(struct.pack("<5I", _DXGI_OF_FORMAT[image.format], 3, 0, 1, 0))
        need = sum(image.level_bytes(l) for l in range(image.mip_count))
        f.write(np.ascontiguousarray(image.data, np.uint8).reshape(-1)[:need].tobytes())


def decode(image, device=0):
    """``BlockImage`` -> ``Image`` of RGBA8 levels, decoded on the GPU by the kernel the scene upload
    uses (``rdr_decode_bc``)."""
    from .renderer import RendererError, load_library
    L = load_library()
    dims = [(max(1, image.height >> l), max(1, image.width >> l)) for l in range(image.mip_count)]
    out = np.empty(sum(h * w for h, w in dims) * 4, np.uint8)
    blocks = np.ascontiguousarray(image.data, np.uint8).reshape(-1)
    if L.rdr_decode_bc(int(device), image.format, image.width, image.height, image.mip_count,
                       blocks.ctypes.data_as(C.c_void_p), blocks.size, out.ctypes.data_as(C.c_void_p)):
        raise RendererError(L.rdr_last_error().decode())
    mips, at = [], 0
    for h, w in dims:
        mips.append(out[at:at + h * w * 4].reshape(h, w, 4).copy())
        at += h * w * 4
    return Image(mips, srgb=image.format == abi.FORMAT_BC7_SRGB)
