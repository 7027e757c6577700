"""ctypes binding of the CPU oracle (``oracle/liborc.so``).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py; the ``renderer_amd`` package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from renderer_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        f3 = C.POINTER(C.c_float)
        L.orc_build_ubo.argtypes = [C.c_uint32, C.c_uint32, f3, f3, f3, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.POINTER(abi.Ubo)]
        L.orc_build_ubo.restype = None
        L.orc_render_visbuffer.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.Ubo), C.c_uint32, C.c_uint32,
                                           C.c_void_p, C.c_void_p, C.POINTER(abi.Stats)]
        L.orc_render_visbuffer_interleaved.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.Ubo), C.c_uint32, C.c_uint32,
                                                       C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(abi.Stats)]
        L.orc_resolve.argtypes = [C.POINTER(abi.SceneDesc), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_shade.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.Ubo), C.c_void_p, C.c_void_p, C.c_uint32,
                                C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_shade_quad_lod.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.Ubo), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_uint64)]
        L.orc_skybox.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.Ubo), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_bake_brdf_lut.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_bake_env_map.argtypes = [C.POINTER(abi.CubemapDesc), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_decode_bc.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def build_ubo(width, height, cam_pos, cam_dir, sun_dir, draw_count, lod_enabled=False, cull_enabled=True):
    u = abi.Ubo()
    lib().orc_build_ubo(width, height, _f3(cam_pos), _f3(cam_dir), _f3(sun_dir), draw_count,
                        int(lod_enabled), int(cull_enabled), C.byref(u))
    return u


class OracleRenderer:
    """Same call shape as ``renderer_amd.Renderer`` (new / set_scene / render), on the CPU."""

    def __init__(self, width, height, config=None):
        self.width, self.height = width, height
        self.lod_enabled = bool(config.lod_enabled) if config else False
        self.cull_enabled = bool(config.cull_enabled) if config else True
        self.skybox = bool(getattr(config, "skybox", False)) if config else False
        self.part = (0, 0xFFFFFFFF)
        self.interleave = (64, 0, 1)  # chunk, rank, world
        self.stats = abi.Stats()

    def set_scene(self, scene):
        self.scene = scene
        self.desc, self._keep = scene.to_desc()
        self.draw_lod = np.zeros(max(1, len(scene.mesh_draws)), np.uint8)

    def set_partition(self, lo, hi):
        self.part = (lo, hi)

    def set_partition_interleaved(self, chunk, rank, world):
        self.interleave = (chunk, rank, world)

    def render_visbuffer(self, camera=None, sun=None):
        cam = camera or self.scene.camera
        sun = sun or self.scene.sun_direction
        self.ubo = build_ubo(self.width, self.height, cam.position, cam.direction, sun,
                             len(self.scene.mesh_draws), self.lod_enabled, self.cull_enabled)
        self.vis = np.empty(self.width * self.height, np.uint64)
        rc = lib().orc_render_visbuffer_interleaved(C.byref(self.desc), C.byref(self.ubo), self.part[0], self.part[1],
                                                    *self.interleave, self.vis.ctypes.data, self.draw_lod.ctypes.data,
                                                    C.byref(self.stats))
        assert rc == 0
        return self.vis.reshape(self.height, self.width)

    def shade(self, vis=None, rows=None, want_linear=False, bands=None):
        """``rows`` = (lo, hi), or ``bands`` = (band, rank, world): the rows of rdr_set_shade_bands."""
        vis = self.vis if vis is None else np.ascontiguousarray(vis, np.uint64).reshape(-1)
        spans = [rows or (0, self.height)]
        if bands:
            band, rank, world = bands
            spans = [(lo, min(lo + band, self.height)) for lo in range(rank * band, self.height, band * world)]
        out = np.zeros((self.height, self.width, 4), np.uint8)
        lin = np.zeros((self.height, self.width, 4), np.float32) if want_linear else None
        for lo, hi in spans:
            rc = lib().orc_shade(C.byref(self.desc), C.byref(self.ubo), vis.ctypes.data, self.draw_lod.ctypes.data,
                                 lo, hi, out.ctypes.data, lin.ctypes.data if want_linear else None)
            assert rc == 0
            if self.skybox:
                rc = lib().orc_skybox(C.byref(self.desc), C.byref(self.ubo), vis.ctypes.data, lo, hi, out.ctypes.data)
                assert rc == 0
        return (out, lin) if want_linear else out

    def shade_quad_lod(self):
        """(image, level per pixel, undefined count): the frame with texture(lambertian, n) at the level of detail a
        2 x 2-quad-difference implementation would use instead of level 0."""
        out = np.zeros((self.height, self.width, 4), np.uint8)
        lod = np.zeros((self.height, self.width), np.float32)
        undefined = C.c_uint64(0)
        rc = lib().orc_shade_quad_lod(C.byref(self.desc), C.byref(self.ubo), self.vis.ctypes.data, self.draw_lod.ctypes.data,
                                      out.ctypes.data, lod.ctypes.data, C.byref(undefined))
        assert rc == 0
        return out, lod, int(undefined.value)

    def render(self, camera=None, sun=None):
        self.render_visbuffer(camera, sun)
        return self.shade()

    def read_visbuffer(self, vis=None):
        vis = self.vis if vis is None else np.ascontiguousarray(vis, np.uint64).reshape(-1)
        out = np.zeros((self.height, self.width, 2), np.uint32)
        lib().orc_resolve(C.byref(self.desc), self.width, self.height, vis.ctypes.data,
                          self.draw_lod.ctypes.data, out.ctypes.data)
        return out


def num_threads():
    return int(lib().orc_num_threads())


def bake_brdf_lut(size, samples):
    """(size, size, 4) float32: CPU restatement of brdf.comp.glsl."""
    out = np.zeros((size, size, 4), np.float32)
    assert lib().orc_bake_brdf_lut(size, samples, out.ctypes.data) == 0
    return out


def bake_env_map(skybox, distribution, size, mips, samples):
    """List of (6, n, n, 4) float32 levels: CPU restatement of filter_envmap.frag.glsl."""
    from renderer_amd.bake import cubemap_desc
    keep = []
    desc = cubemap_desc(skybox, keep)
    sizes = [max(1, size >> l) for l in range(mips)]
    out = np.zeros(sum(6 * n * n for n in sizes) * 4, np.float32)
    assert lib().orc_bake_env_map(C.byref(desc), distribution, size, mips, samples, out.ctypes.data) == 0
    levels, off = [], 0
    for n in sizes:
        levels.append(out[off:off + 6 * n * n * 4].reshape(6, n, n, 4).copy())
        off += 6 * n * n * 4
    return levels
