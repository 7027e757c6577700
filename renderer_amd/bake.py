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
