"""Minimal PNG writer (no third-party imaging library in the image)."""
import struct
import zlib

import numpy as np


def write_png(path, rgba):
    a = np.ascontiguousarray(rgba, np.uint8)
    if a.ndim == 2:
        a = np.stack([a, a, a, np.full_like(a, 255)], -1)
    if a.shape[2] == 3:
        a = np.concatenate([a, np.full(a.shape[:2] + (1,), 255, np.uint8)], -1)
    h, w = a.shape[:2]
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def id_colors(vis_xy):
    """False-colour a (H, W, 2) uint32 visibility buffer for inspection."""
    x = vis_xy[..., 0].astype(np.uint64) * 2654435761 + vis_xy[..., 1].astype(np.uint64) * 40503
    x = (x ^ (x >> 15)) & 0xFFFFFFFF
    rgb = np.stack([(x >> 0) & 255, (x >> 8) & 255, (x >> 16) & 255], -1).astype(np.uint8)
    rgb[vis_xy[..., 1] == 0xFFFFFFFF] = 0
    return rgb
