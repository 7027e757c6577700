"""Host mirror of the reference's ``bake`` tool for the two IBL inputs the shading pass samples
(bake/src/main.rs sub-commands ``brdf-lut`` and ``env-map``): runs the HIP kernels behind
``rdr_bake_brdf_lut`` / ``rdr_bake_env_map``.  No CPU path."""
import ctypes as C

import numpy as np

from . import abi
from .renderer import RendererError, load_library
from .scene import Cubemap, EnvMap, Image

LAMBERTIAN, GGX, CHARLIE = 0, 1, 2


def _last_error(L):
    return L.rdr_last_error().decode()


def cubemap_desc(cm, keep):
    """``abi.CubemapDesc`` over a ``Cubemap`` (mip-major, face-major contiguous copy kept in ``keep``)."""
    blob = np.concatenate([np.ascontiguousarray(m).reshape(-1) for m in cm.mips])
    keep.append(blob)
    d = abi.CubemapDesc()
    d.data = blob.ctypes.data
    d.size = cm.mips[0].shape[1]
    d.mip_count = len(cm.mips)
    d.format = cm.format
    return d


def bake_brdf_lut(size=256, samples=10000, device=0):
    """The 2D table of brdf.comp.glsl as an ``Image`` (RGBA32F, one level)."""
    L = load_library()
    out = np.zeros((size, size, 4), np.float32)
    if L.rdr_bake_brdf_lut(int(device), int(size), int(samples), out.ctypes.data):
        raise RendererError(_last_error(L))
    return Image([out], srgb=False)


def bake_env_map(skybox, distribution, size=256, mips=None, samples=1024, device=0):
    """``skybox`` prefiltered with filter_envmap.frag.glsl as a ``Cubemap`` of ``mips`` levels
    (default log2(size), as bake/src/brdf.rs:115)."""
    L = load_library()
    mips = int(np.log2(size)) if mips is None else int(mips)
    mips = max(1, mips)
    sizes = [max(1, size >> l) for l in range(mips)]
    out = np.zeros(sum(6 * n * n for n in sizes) * 4, np.float32)
    keep = []
    desc = cubemap_desc(skybox, keep)
    if L.rdr_bake_env_map(int(device), C.byref(desc), int(distribution), int(size), mips, int(samples), out.ctypes.data):
        raise RendererError(_last_error(L))
    levels, off = [], 0
    for n in sizes:
        levels.append(out[off:off + 6 * n * n * 4].reshape(6, n, n, 4).copy())
        off += 6 * n * n * 4
    return Cubemap(levels)


def bake_env(skybox, size=256, samples=1024, lut_size=256, lut_samples=10000, device=0):
    """Everything ``EnvMap`` needs from one sky cubemap, as the reference's bake produces it."""
    return EnvMap(brdf_lut=bake_brdf_lut(lut_size, lut_samples, device),
                  lambertian=bake_env_map(skybox, LAMBERTIAN, size, None, samples, device),
                  ggx=bake_env_map(skybox, GGX, size, None, samples, device),
                  charlie=bake_env_map(skybox, CHARLIE, size, None, samples, device),
                  skybox=skybox)


# ---------------------------------------------------------------------------------------------
# Sky conversion: bake/src/cubemaps.rs (the reference's `bake cubemap` sub-command)
# ---------------------------------------------------------------------------------------------
def read_radiance_hdr(path):
    """Radiance RGBE (.hdr) picture as (h, w, 4) float32 with alpha 1.  (The reference reads OpenEXR,
    bake/src/cubemaps.rs:121-127; no EXR decoder exists here, .hdr carries the same kind of data.)"""
    with open(path, "rb") as f:
        data = f.read()
    pos = data.index(b"\n\n") + 2
    end = data.index(b"\n", pos)
    dims = data[pos:end].split()
    if len(dims) != 4 or dims[0] != b"-Y" or dims[2] != b"+X":
        raise ValueError(f"{path}: only the standard -Y h +X w orientation is supported")
    h, w = int(dims[1]), int(dims[3])
    pos = end + 1
    rgbe = np.zeros((h, w, 4), np.uint8)
    for y in range(h):
        if 8 <= w < 32768 and data[pos] == 2 and data[pos + 1] == 2 and (data[pos + 2] << 8 | data[pos + 3]) == w:
            pos += 4  # new-style run-length encoding, one channel after the other
            for c in range(4):
                x = 0
                while x < w:
                    n = data[pos]
                    if n > 128:
                        rgbe[y, x:x + n - 128, c] = data[pos + 1]
                        x += n - 128
                        pos += 2
                    else:
                        rgbe[y, x:x + n, c] = np.frombuffer(data, np.uint8, n, pos + 1)
                        x += n
                        pos += 1 + n
        else:
            rgbe[y] = np.frombuffer(data, np.uint8, w * 4, pos).reshape(w, 4)
            pos += w * 4
    scale = np.where(rgbe[..., 3] > 0, np.ldexp(1.0, rgbe[..., 3].astype(np.int32) - 136), 0.0).astype(np.float32)
    out = np.ones((h, w, 4), np.float32)
    out[..., :3] = rgbe[..., :3].astype(np.float32) * scale[..., None]
    return out


def _face_coords_to_xyz(i, j, face, face_size):
    """bake/src/cubemaps.rs:9-22 (faces of the vertical cross, not yet in cubemap order)."""
    a = np.float32(2.0) * i.astype(np.float32) / np.float32(face_size)
    b = np.float32(2.0) * j.astype(np.float32) / np.float32(face_size)
    one = np.float32(1.0)
    return [(-one + 0 * a, a - one, b - one), (a - one, -one + 0 * a, one - b), (one + 0 * a, a - one, one - b),
            (one - a, one + 0 * a, one - b), (b - one, a - one, one + 0 * a), (one - b, a - one, -one + 0 * a)][face]


def equirect_to_cubemap(pixels, mips=None):
    """An equirectangular (h, w, 4) float picture as the RGBA8 cubemap the reference's
    ``convert_hdr_cubemap_to_ktx`` stores (bake/src/cubemaps.rs:24-185): face size w / 4, bilinear
    fetch with clamped neighbours, assembled as a vertical cross and cut into the six faces,
    ``(p * 255) as u8`` (truncating, saturating), ilog2(face size) levels.  The lower levels are
    resampled with Pillow's Lanczos filter (the reference uses the ``image`` crate's Lanczos3:
    same kernel family, not bit-identical)."""
    from PIL import Image as PilImage
    data = np.asarray(pixels, np.float32)
    height, width = data.shape[:2]
    fs = width // 4
    cross = np.zeros((fs * 4, fs * 3, 4), np.float32)  # [row, column]
    offsets = [(fs, fs * 3), (0, fs), (fs, fs), (fs * 2, fs), (fs, 0), (fs, fs * 2)]
    ii, jj = np.meshgrid(np.arange(fs), np.arange(fs), indexing="ij")
    pi = np.float32(np.pi)
    for face in range(6):
        x, y, z = _face_coords_to_xyz(ii, jj, face, fs)
        r = np.hypot(x, y).astype(np.float32)
        theta, phi = np.arctan2(y, x).astype(np.float32), np.arctan2(z, r).astype(np.float32)
        uf = np.float32(2.0) * np.float32(fs) * (theta + pi) / pi
        vf = np.float32(2.0) * np.float32(fs) * (pi / np.float32(2.0) - phi) / pi
        u1 = np.clip(np.floor(uf).astype(np.int64), 0, width - 1)
        v1 = np.clip(np.floor(vf).astype(np.int64), 0, height - 1)
        u2, v2 = np.clip(u1 + 1, 0, width - 1), np.clip(v1 + 1, 0, height - 1)
        s, t = (uf - u1.astype(np.float32))[..., None], (vf - v1.astype(np.float32))[..., None]
        a, b, c, d = data[v1, u1], data[v1, u2], data[v2, u1], data[v2, u2]
        color = a * (1 - s) * (1 - t) + b * s * (1 - t) + c * (1 - s) * t + d * s * t
        ox, oy = offsets[face]
        cross[jj + oy, ii + ox] = color
    w3, h4 = fs * 3, fs * 4
    faces = np.zeros((6, fs, fs, 4), np.float32)
    j, i = np.meshgrid(np.arange(fs), np.arange(fs), indexing="ij")
    src = [(2 * fs + i, fs + j), (i, fs + j), (fs + i, j), (fs + i, 2 * fs + j), (fs + i, fs + j),
           (2 * fs - (i + 1), h4 - (j + 1))]  # +X, -X, +Y, -Y, +Z, -Z out of the cross (cubemaps.rs:100-107)
    for face in range(6):
        x, y = src[face]
        faces[face] = cross[y, x]
    q = np.clip(np.nan_to_num(faces * np.float32(255.0), nan=0.0), 0.0, 255.0).astype(np.uint8)  # `as u8`: truncate, saturate
    levels = int(np.log2(fs)) if mips is None else int(mips)
    out, cur, size = [], q, fs
    for _ in range(max(1, levels)):
        out.append(cur)
        size = max(size >> 1, 1)
        # channel by channel: the four channels are filtered alike, alpha is not a weight
        cur = np.stack([np.stack([np.asarray(PilImage.fromarray(np.ascontiguousarray(f[..., c]), "L")
                                             .resize((size, size), PilImage.LANCZOS)) for c in range(4)], -1) for f in cur])
    return Cubemap(out)


def main(argv=None):
    """``python -m renderer_amd.bake sky.hdr out/env.json``: sky -> cubemap -> BRDF table and the three
    prefiltered maps on the GPU -> five KTX2 files and the JSON that names them (what the reference's
    ``bake`` tool produces in three runs)."""
    import argparse
    from .ktx2 import save_env_json
    ap = argparse.ArgumentParser(prog="python -m renderer_amd.bake")
    ap.add_argument("sky", help="equirectangular Radiance .hdr picture, or a .npy holding (h, w, 4) float32")
    ap.add_argument("out", help="environment JSON to write (the KTX2 files go next to it)")
    ap.add_argument("--size", type=int, default=256, help="face size of the prefiltered maps")
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--lut-size", type=int, default=256)
    ap.add_argument("--lut-samples", type=int, default=10000)
    args = ap.parse_args(argv)
    pixels = np.load(args.sky) if args.sky.endswith(".npy") else read_radiance_hdr(args.sky)
    env = bake_env(equirect_to_cubemap(pixels), args.size, args.samples, args.lut_size, args.lut_samples)
    save_env_json(args.out, env)
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
