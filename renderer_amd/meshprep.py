"""Mesh preparation through the library's host-side builder: the counterpart of
``Geometry::append_mesh`` (scenery/src/mesh.rs:197-374) — de-duplication, cache / fetch
ordering, meshlets with bounds and the simplified LOD chain.  Needs the library, not a GPU."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import abi
from .renderer import RendererError, load_library


@dataclass
class BuiltMesh:
    vertices: np.ndarray           # abi.VERTEX, de-duplicated, fetch order
    indices: np.ndarray            # level 0 triangle list
    meshlet_vertices: np.ndarray
    meshlet_triangles: np.ndarray
    meshlets: np.ndarray           # abi.MESHLET, offsets local to the two arrays above
    mesh: np.ndarray               # abi.MESH scalar, offsets local to these arrays

    def lod_triangles(self, lod):
        """(n, 3) vertex ids of one level, expanded from its meshlets."""
        L = self.mesh["lods"][lod]
        out = []
        for m in self.meshlets[int(L["meshlet_offset"]):int(L["meshlet_offset"]) + int(L["meshlet_count"])]:
            nt, vo, to = int(m["triangle_count"]), int(m["vertex_offset"]), int(m["triangle_offset"])
            local = self.meshlet_triangles[to:to + nt * 3].reshape(-1, 3)
            out.append(self.meshlet_vertices[vo + local.astype(np.int64)])
        return np.concatenate(out) if out else np.zeros((0, 3), np.uint32)


def build_mesh(vertices, indices, max_lods=8, fast_mode=False):
    L = load_library()
    v = np.ascontiguousarray(vertices, abi.VERTEX)
    i = np.ascontiguousarray(indices, np.uint32).reshape(-1)
    handle = C.c_void_p()
    if L.rdr_mesh_build(v.ctypes.data_as(C.c_void_p), len(v), i.ctypes.data_as(C.c_void_p), len(i), max_lods, 1 if fast_mode else 0, C.byref(handle)):
        raise RendererError(L.rdr_last_error().decode())
    try:
        r = abi.BuiltMesh()
        if L.rdr_mesh_result(handle, C.byref(r)):
            raise RendererError(L.rdr_last_error().decode())

        def view(ptr, count, dtype):
            if not count:
                return np.zeros(0, dtype)
            nbytes = count * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=dtype, count=count).copy()

        mesh = np.frombuffer(bytes(r.mesh), dtype=abi.MESH, count=1)[0].copy()
        return BuiltMesh(view(r.vertices, r.vertex_count, abi.VERTEX), view(r.indices, r.index_count, np.uint32),
                         view(r.meshlet_vertices, r.meshlet_vertex_count, np.uint32),
                         view(r.meshlet_triangles, r.meshlet_triangle_bytes, np.uint8),
                         view(r.meshlets, r.meshlet_count, abi.MESHLET), mesh)
    finally:
        L.rdr_mesh_free(handle)


def generate_tangents(vertices, indices):
    """``scenery::vertex::generate_tangents`` (scenery/src/vertex.rs:86-90, the mikktspace crate): fills the ``t``
    field of ``vertices`` (abi.VERTEX, modified in place) from positions, normals and uv0 over the triangle list."""
    L = load_library()
    assert vertices.dtype == abi.VERTEX and vertices.flags["C_CONTIGUOUS"]
    i = np.ascontiguousarray(indices, np.uint32).reshape(-1)
    if L.rdr_generate_tangents(vertices.ctypes.data_as(C.c_void_p), len(vertices), i.ctypes.data_as(C.c_void_p), len(i)):
        raise RendererError(L.rdr_last_error().decode())
    return vertices
