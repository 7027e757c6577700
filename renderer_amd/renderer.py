"""Host-side mirror of the reference's ``Renderer`` (renderer/src/renderer.rs:780-1060) over the
C ABI of ``include/rdr_abi.h``: ``Renderer(width, height, config)`` / ``set_scene`` / ``render``,
plus the readbacks the reference lacks.  There is no CPU path: construction fails without the
HIP library and a GPU.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# RDR_LIB_PATH: another build of the same library (kernel experiments); there is still no fallback
LIB_PATH = os.environ.get("RDR_LIB_PATH") or os.path.join(_HERE, "librdr_hip.so")

# every symbol include/rdr_abi.h declares
ABI_SYMBOLS = [
    "rdr_create", "rdr_set_scene", "rdr_update_transforms", "rdr_render", "rdr_render_visbuffer", "rdr_shade",
    "rdr_set_partition", "rdr_set_shade_rows", "rdr_set_partition_interleaved", "rdr_set_shade_bands", "rdr_set_partition_replication", "rdr_visbuffer_device_ptr", "rdr_color_device_ptr",
    "rdr_set_stream", "rdr_sync", "rdr_read_visbuffer", "rdr_read_visbuffer_u64", "rdr_read_color",
    "rdr_get_stats", "rdr_build_ubo", "rdr_get_ubo", "rdr_destroy", "rdr_last_error",
    "rdr_bake_brdf_lut", "rdr_bake_env_map", "rdr_mesh_build", "rdr_mesh_result", "rdr_mesh_free", "rdr_decode_bc",
    "rdr_decode_bc_noalloc", "rdr_frame_status",
    "rdr_composite_export", "rdr_composite_connect", "rdr_composite_exchange", "rdr_composite_gather", "rdr_composite_disconnect", "rdr_generate_tangents",
]

_LIB = None


class RendererError(RuntimeError):
    pass


def load_library():
    """Load ``librdr_hip.so`` (built in-tree by ``make -C renderer_amd/csrc``)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RendererError(f"{LIB_PATH} is missing: build it with `make -C renderer_amd/csrc` "
                            "(there is no fallback path)")
    L = C.CDLL(LIB_PATH)
    f3 = C.POINTER(C.c_float)
    vp = C.c_void_p
    L.rdr_create.argtypes = [C.POINTER(abi.Config), C.c_uint32, C.c_uint32, C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.rdr_set_scene.argtypes = [vp, C.POINTER(abi.SceneDesc)]
    L.rdr_update_transforms.argtypes = [vp, f3, C.c_uint32]
    L.rdr_render.argtypes = [vp, f3, f3, f3]
    L.rdr_render_visbuffer.argtypes = [vp, f3, f3, f3]
    L.rdr_shade.argtypes = [vp]
    L.rdr_bake_brdf_lut.argtypes = [C.c_int, C.c_uint32, C.c_uint32, vp]
    L.rdr_bake_env_map.argtypes = [C.c_int, C.POINTER(abi.CubemapDesc), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.rdr_set_partition.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.rdr_set_shade_rows.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.rdr_set_partition_interleaved.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.rdr_set_shade_bands.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.rdr_set_partition_replication.argtypes = [vp, C.c_float]
    L.rdr_visbuffer_device_ptr.argtypes = [vp]
    L.rdr_visbuffer_device_ptr.restype = vp
    L.rdr_color_device_ptr.argtypes = [vp]
    L.rdr_color_device_ptr.restype = vp
    L.rdr_set_stream.argtypes = [vp, vp]
    L.rdr_sync.argtypes = [vp]
    L.rdr_read_visbuffer.argtypes = [vp, vp]
    L.rdr_read_visbuffer_u64.argtypes = [vp, vp]
    L.rdr_read_color.argtypes = [vp, vp]
    L.rdr_get_stats.argtypes = [vp, C.POINTER(abi.Stats)]
    L.rdr_frame_status.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.rdr_composite_export.argtypes = [vp, vp]
    L.rdr_composite_connect.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.rdr_composite_exchange.argtypes = [vp]
    L.rdr_composite_gather.argtypes = [vp]
    L.rdr_composite_disconnect.argtypes = [vp]
    L.rdr_build_ubo.argtypes = [C.c_uint32, C.c_uint32, f3, f3, f3, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(abi.Ubo)]
    L.rdr_get_ubo.argtypes = [vp, C.POINTER(abi.Ubo)]
    L.rdr_destroy.argtypes = [vp]
    L.rdr_destroy.restype = None
    L.rdr_last_error.restype = C.c_char_p
    L.rdr_mesh_build.argtypes = [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.rdr_mesh_result.argtypes = [vp, C.POINTER(abi.BuiltMesh)]
    L.rdr_mesh_free.argtypes = [vp]
    L.rdr_mesh_free.restype = None
    L.rdr_generate_tangents.argtypes = [vp, C.c_uint32, vp, C.c_uint32]
    L.rdr_decode_bc.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp]
    L.rdr_decode_bc_noalloc.argtypes = L.rdr_decode_bc.argtypes
    _LIB = L
    return L


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def build_ubo(width, height, cam_pos, cam_dir, sun_dir, draw_count, lod_enabled=False, cull_enabled=True):
    """``Renderer::update_uniform_object`` (renderer.rs:1148-1197); host arithmetic only."""
    L = load_library()
    u = abi.Ubo()
    rc = L.rdr_build_ubo(width, height, _f3(cam_pos), _f3(cam_dir), _f3(sun_dir), draw_count,
                         int(lod_enabled), int(cull_enabled), C.byref(u))
    if rc:
        raise RendererError(L.rdr_last_error().decode())
    return u


class _DevArray:
    """Expose library-owned device memory through ``__cuda_array_interface__``."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}
        self._owner = owner


class Renderer:
    def __init__(self, width, height, config=None, device=0):
        self._L = load_library()
        self._h = C.c_void_p()
        self.width, self.height = int(width), int(height)
        cfg = abi.Config(int(config.lod_enabled) if config else 0, int(config.cull_enabled) if config else 1,
                         int(config.collect_stats) if config else 1, int(config.time_kernels) if config else 0,
                         int(getattr(config, "skybox", False)) if config else 0)
        dev = (C.c_int * 1)(int(device))
        self._check(self._L.rdr_create(C.byref(cfg), self.width, self.height, dev, 1, C.byref(self._h)))
        self.scene = None
        self._keep = None

    def _check(self, rc):
        if rc:
            raise RendererError(self._L.rdr_last_error().decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.rdr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- reference API -----------------------------------------------------------
    def set_scene(self, scene):
        desc, keep = scene.to_desc()
        self._check(self._L.rdr_set_scene(self._h, C.byref(desc)))
        self.scene = scene
        self._keep = (desc, keep)

    def update_transforms(self, transforms):
        t = np.ascontiguousarray(transforms, np.float32).reshape(-1, 16)
        self._check(self._L.rdr_update_transforms(self._h, t.ctypes.data_as(C.POINTER(C.c_float)), len(t)))

    def _cam(self, camera, sun):
        cam = camera or self.scene.camera
        sun = sun or self.scene.sun_direction
        return _f3(cam.position), _f3(cam.direction), _f3(sun)

    def render(self, camera=None, sun=None):
        self._check(self._L.rdr_render(self._h, *self._cam(camera, sun)))

    # --- sort-last halves ----------------------------------------------------------
    def render_visbuffer(self, camera=None, sun=None):
        self._check(self._L.rdr_render_visbuffer(self._h, *self._cam(camera, sun)))

    def shade(self):
        self._check(self._L.rdr_shade(self._h))

    def set_partition(self, lo, hi):
        self._check(self._L.rdr_set_partition(self._h, lo, hi))

    def set_shade_rows(self, lo, hi):
        self._check(self._L.rdr_set_shade_rows(self._h, lo, hi))

    def set_partition_interleaved(self, chunk, rank, world):
        """Chunk j (``chunk`` meshlets) of draw d belongs to rank (j + d) % world."""
        self._check(self._L.rdr_set_partition_interleaved(self._h, chunk, rank, world))

    def set_partition_replication(self, min_triangle_pixels):
        """Draws whose triangles average more than this many pixels are drawn by every rank, large triangles on the
        rank's own rows only (0 = off)."""
        self._check(self._L.rdr_set_partition_replication(self._h, float(min_triangle_pixels)))

    def set_shade_bands(self, band, rank, world):
        """Bands of ``band`` rows dealt round-robin: band b is shaded by rank b % world."""
        self._check(self._L.rdr_set_shade_bands(self._h, band, rank, world))

    # --- peer-to-peer sort-last exchange (one process per GPU) -------------------------
    COMPOSITE_HANDLE_BYTES = 208

    def composite_export(self):
        """This renderer's IPC handles and exchange counter (208 bytes) for the other ranks."""
        buf = C.create_string_buffer(self.COMPOSITE_HANDLE_BYTES)
        self._check(self._L.rdr_composite_export(self._h, buf))
        return buf.raw

    def composite_connect(self, rank, world, handles):
        """``handles``: every rank's ``composite_export()`` in rank order."""
        assert len(handles) == world and all(len(h) == self.COMPOSITE_HANDLE_BYTES for h in handles)
        blob = C.create_string_buffer(b"".join(handles), world * self.COMPOSITE_HANDLE_BYTES)
        self._check(self._L.rdr_composite_connect(self._h, rank, world, blob))

    def composite_exchange(self):
        self._check(self._L.rdr_composite_exchange(self._h))

    def composite_gather(self):
        self._check(self._L.rdr_composite_gather(self._h))

    def composite_disconnect(self):
        self._check(self._L.rdr_composite_disconnect(self._h))

    def set_stream(self, stream_handle):
        self._check(self._L.rdr_set_stream(self._h, C.c_void_p(stream_handle)))

    def sync(self):
        self._check(self._L.rdr_sync(self._h))

    def visbuffer_device_array(self):
        """The W*H visibility buffer in HBM as int64 (bit 63 of a key is never set, so signed and
        unsigned order agree) for ``torch.as_tensor(..., device='cuda')``."""
        return _DevArray(self._L.rdr_visbuffer_device_ptr(self._h), (self.height * self.width,), "<i8", self)

    def color_device_array(self):
        return _DevArray(self._L.rdr_color_device_ptr(self._h), (self.height, self.width, 4), "|u1", self)

    # --- readback ------------------------------------------------------------------
    def read_visbuffer(self):
        out = np.empty((self.height, self.width, 2), np.uint32)
        self._check(self._L.rdr_read_visbuffer(self._h, out.ctypes.data))
        return out

    def read_visbuffer_u64(self):
        out = np.empty((self.height, self.width), np.uint64)
        self._check(self._L.rdr_read_visbuffer_u64(self._h, out.ctypes.data))
        return out

    def read_color(self):
        out = np.empty((self.height, self.width, 4), np.uint8)
        self._check(self._L.rdr_read_color(self._h, out.ctypes.data))
        return out

    def frame_status(self):
        """Waits for the submitted frames; ``abi.FRAME_*`` flags of the last one (0 = nothing was dropped)."""
        flags = C.c_uint32(0)
        self._check(self._L.rdr_frame_status(self._h, C.byref(flags)))
        return int(flags.value)

    def stats(self):
        st = abi.Stats()
        self._check(self._L.rdr_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def ubo(self):
        u = abi.Ubo()
        self._check(self._L.rdr_get_ubo(self._h, C.byref(u)))
        return u
