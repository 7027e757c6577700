"""Wavefront OBJ ingestion: ``Scene::load_obj_file`` / ``Geometry::load_obj_file``
(scenery/src/scene.rs:262-310, scenery/src/mesh.rs:164-195).

The reference reads position + normal vertices (the ``obj`` crate's ``obj::Vertex``), gives every
vertex uv (0, 0) and tangent (1, 0, 0, 1), runs each file through ``append_mesh`` and then draws
*only mesh 0* once, with the identity transform, ``Material::default()``, the camera at (0, 0, 2)
looking along +z and the default sun — quirks kept as they are.  Polygons with more than three
corners are fanned here (a superset of what the crate accepts)."""
import numpy as np

from . import abi
from .scene import DEFAULT_SUN, Camera, Scene, mat4_trs


def read_obj(path):
    """(vertices, indices): one vertex per distinct (position, normal) pair, in order of first use."""
    pos, nrm, verts, index, tris = [], [], [], {}, []
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split("#", 1)[0].split()
            if not t:
                continue
            if t[0] == "v":
                pos.append([float(x) for x in t[1:4]])
            elif t[0] == "vn":
                nrm.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                corner = []
                for c in t[1:]:
                    parts = c.split("/")
                    vi = int(parts[0])
                    ni = int(parts[2]) if len(parts) > 2 and parts[2] else 0
                    vi = vi - 1 if vi > 0 else len(pos) + vi
                    ni = (ni - 1 if ni > 0 else len(nrm) + ni) if ni else -1
                    key = (vi, ni)
                    if key not in index:
                        index[key] = len(verts)
                        verts.append(key)
                    corner.append(index[key])
                for k in range(1, len(corner) - 1):
                    tris.append((corner[0], corner[k], corner[k + 1]))
    if not tris:
        raise ValueError(f"{path}: no faces")
    v = np.zeros(len(verts), abi.VERTEX)
    p = np.asarray(pos, np.float32)
    n = np.asarray(nrm, np.float32).reshape(-1, 3)
    for i, (vi, ni) in enumerate(verts):
        v["p"][i, :3] = p[vi]
        if ni >= 0:
            v["n"][i, :3] = n[ni]
    v["t"][:] = (1.0, 0.0, 0.0, 1.0)
    return v, np.asarray(tris, np.uint32).reshape(-1)


def load_obj(model_files, env=None, max_lods=8, fast_mode=False):
    """``model_files``: one path or several separated by commas, as on the reference's command line."""
    sc = Scene()
    for name in [f for f in str(model_files).split(",") if f]:
        v, idx = read_obj(name)
        sc.append_mesh_auto(v, idx, max_lods, fast_mode)
    sc.add_material()
    sc.add_draw(0, mat4_trs(), 0, 0)
    sc.camera = Camera((0.0, 0.0, 2.0), (0.0, 0.0, 1.0))
    sc.sun_direction = DEFAULT_SUN
    sc.env = env
    return sc
