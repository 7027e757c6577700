"""glTF 2.0 (.gltf + buffers, or .glb) -> ``Scene``: the counterpart of ``Scene::load_gltf_file``
(scenery/src/scene.rs:62-232) with ``build_material`` (:346-577) and ``build_scene_mesh_draws``
(:579-664), plus a small exporter used to test it.

Differences from the reference, all upstream of the frame path: meshlets and the LOD chain come
from this library's own mesh preparation (``rdr_mesh_build``: same limits and rules as the
reference's use of meshoptimizer, different algorithms, so IDs differ from the reference's; use
``scene.bin`` for identical meshlets), and missing tangents are derived from the uv gradients
instead of mikktspace.  Everything else follows the reference:
meshes are appended per (node, primitive) in breadth-first node order, one transform per node in
depth-first order, images are indexed by *texture* index, loaded as a single mip level, sRGB iff
the file path contains "basecolor" (renderer/src/renderer/image_loader.rs:47-51).
"""
import base64
import json
import os
import struct

import numpy as np

from . import abi
from .dds import DdsError, read_dds
from .scene import Camera, Image, Scene

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_COUNT = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}
DEFAULT_SUN = (-0.23623444, 0.9302103, 0.2808959)  # scenery/src/scene.rs:115


# ---------------------------------------------------------------------------------------------
# container
# ---------------------------------------------------------------------------------------------
def _read_container(path):
    with open(path, "rb") as f:
        head = f.read(12)
        if head[:4] == b"glTF":
            _, length = struct.unpack("<II", head[4:12])
            doc, bin_chunk = None, None
            while f.tell() < length:
                clen, ctype = struct.unpack("<II", f.read(8))
                data = f.read(clen)
                if ctype == 0x4E4F534A:
                    doc = json.loads(data.decode("utf-8"))
                elif ctype == 0x004E4942:
                    bin_chunk = data
            if doc is None:
                raise ValueError(f"{path}: GLB without a JSON chunk")
            return doc, bin_chunk
    with open(path, "r", encoding="utf-8") as f:
        return json.load(f), None


def _load_buffers(doc, base_dir, bin_chunk):
    out = []
    for b in doc.get("buffers", []):
        uri = b.get("uri")
        if uri is None:
            if bin_chunk is None:
                raise ValueError("buffer without uri outside a GLB")
            out.append(bin_chunk)
        elif uri.startswith("data:"):
            out.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            with open(os.path.join(base_dir, uri), "rb") as f:
                out.append(f.read())
    return out


def _accessor(doc, buffers, index, as_f32=False):
    acc = doc["accessors"][index]
    if "sparse" in acc:
        raise ValueError("sparse accessors are not supported")
    dt = np.dtype(_COMPONENT[acc["componentType"]]).newbyteorder("<")
    n = _COUNT[acc["type"]]
    count = acc["count"]
    if "bufferView" not in acc:
        arr = np.zeros((count, n), dt)
    else:
        bv = doc["bufferViews"][acc["bufferView"]]
        start = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
        stride = bv.get("byteStride") or n * dt.itemsize
        raw = np.frombuffer(buffers[bv["buffer"]], np.uint8, offset=start,
                            count=min(stride * (count - 1) + n * dt.itemsize, len(buffers[bv["buffer"]]) - start) if count else 0)
        if stride == n * dt.itemsize:
            arr = raw[:count * stride].view(dt).reshape(count, n)
        else:
            rows = np.lib.stride_tricks.as_strided(raw, (count, n * dt.itemsize), (stride, 1))
            arr = np.ascontiguousarray(rows).view(dt).reshape(count, n)
    if as_f32 and arr.dtype.kind != "f":  # normalised integers (the gltf crate's into_f32)
        info = np.iinfo(arr.dtype)
        arr = np.maximum(arr.astype(np.float32) / np.float32(info.max), -1.0) if info.min < 0 else arr.astype(np.float32) / np.float32(info.max)
    return np.array(arr)


# ---------------------------------------------------------------------------------------------
# transforms (glam conventions, column-major 16 floats)
# ---------------------------------------------------------------------------------------------
def _quat_to_mat3(q):
    x, y, z, w = [np.float32(c) for c in q]
    x2, y2, z2 = x + x, y + y, z + z
    xx, xy, xz, yy, yz, zz = x * x2, x * y2, x * z2, y * y2, y * z2, z * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    one = np.float32(1.0)
    return np.array([[one - (yy + zz), xy + wz, xz - wy],      # column 0
                     [xy - wz, one - (xx + zz), yz + wx],      # column 1
                     [xz + wy, yz - wx, one - (xx + yy)]], np.float32)  # column 2


def _trs_to_mat4(t, r, s):
    """``Mat4::from_scale_rotation_translation`` (glam), column-major."""
    c = _quat_to_mat3(r)
    m = np.zeros((4, 4), np.float32)  # m[col][row]
    for k in range(3):
        m[k, :3] = c[k] * np.float32(s[k])
    m[3, :3] = np.asarray(t, np.float32)
    m[3, 3] = 1.0
    return m


def _mat3_to_quat(c):
    """Rotation matrix (columns c[0..2]) -> quaternion (x, y, z, w)."""
    m00, m01, m02 = c[0][0], c[1][0], c[2][0]
    m10, m11, m12 = c[0][1], c[1][1], c[2][1]
    m20, m21, m22 = c[0][2], c[1][2], c[2][2]
    tr = m00 + m11 + m22
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        return ((m21 - m12) / s, (m02 - m20) / s, (m10 - m01) / s, 0.25 * s)
    if m00 > m11 and m00 > m22:
        s = np.sqrt(1.0 + m00 - m11 - m22) * 2
        return (0.25 * s, (m01 + m10) / s, (m02 + m20) / s, (m21 - m12) / s)
    if m11 > m22:
        s = np.sqrt(1.0 + m11 - m00 - m22) * 2
        return ((m01 + m10) / s, 0.25 * s, (m12 + m21) / s, (m02 - m20) / s)
    s = np.sqrt(1.0 + m22 - m00 - m11) * 2
    return ((m02 + m20) / s, (m12 + m21) / s, 0.25 * s, (m10 - m01) / s)


def _node_trs(node):
    """(translation, rotation xyzw, scale) as the gltf crate's ``Transform::decomposed``."""
    if "matrix" in node:
        m = np.asarray(node["matrix"], np.float64).reshape(4, 4)  # m[col][row]
        t = m[3, :3]
        cols = m[:3, :3].copy()
        sx, sy = np.linalg.norm(cols[0]), np.linalg.norm(cols[1])
        sz = np.sign(np.linalg.det(cols)) * np.linalg.norm(cols[2])
        s = np.array([sx, sy, sz])
        rot = cols / np.where(s == 0, 1.0, s)[:, None]
        return t.astype(np.float32), np.asarray(_mat3_to_quat(rot), np.float32), s.astype(np.float32)
    return (np.asarray(node.get("translation", (0, 0, 0)), np.float32), np.asarray(node.get("rotation", (0, 0, 0, 1)), np.float32),
            np.asarray(node.get("scale", (1, 1, 1)), np.float32))


def _local_matrix(node):
    if "matrix" in node:
        return np.asarray(node["matrix"], np.float32).reshape(4, 4)
    return _trs_to_mat4(*_node_trs(node))


def _mul(a, b):
    """Column-major product a * b with m[col][row] storage."""
    return (a.T @ b.T).T.astype(np.float32)


# ---------------------------------------------------------------------------------------------
# geometry
# ---------------------------------------------------------------------------------------------
def _primitive_vertices(doc, buffers, prim):
    if prim.get("mode", 4) != 4:
        raise ValueError("only Triangles are supported")
    at = prim["attributes"]
    pos = _accessor(doc, buffers, at["POSITION"]).astype(np.float32)
    n = len(pos)
    v = np.zeros(n, abi.VERTEX)
    v["p"][:, :3] = pos
    if "NORMAL" in at:
        v["n"][:, :3] = _accessor(doc, buffers, at["NORMAL"]).astype(np.float32)[:n]
    if "TEXCOORD_0" in at:
        v["uv"][:, 0:2] = _accessor(doc, buffers, at["TEXCOORD_0"], as_f32=True)[:n]
    if "TEXCOORD_1" in at:
        v["uv"][:, 2:4] = _accessor(doc, buffers, at["TEXCOORD_1"], as_f32=True)[:n]
    if "indices" in prim:
        idx = _accessor(doc, buffers, prim["indices"]).astype(np.uint32).reshape(-1)
    else:
        idx = np.zeros(0, np.uint32)  # the reference reads no indices for non-indexed primitives either
    if "TANGENT" in at:
        v["t"] = _accessor(doc, buffers, at["TANGENT"]).astype(np.float32)[:n]
    else:
        v["t"] = (1.0, 0.0, 0.0, 1.0)
        if len(idx):  # generate_tangents (scenery/src/scene.rs:207)
            from .meshprep import generate_tangents
            generate_tangents(v, idx)
    return v, idx


# ---------------------------------------------------------------------------------------------
# materials
# ---------------------------------------------------------------------------------------------
def _uv_transform(info):
    ext = (info or {}).get("extensions", {}).get("KHR_texture_transform")
    if not ext:
        return (1.0, 1.0, 0.0, 0.0)
    off, sc = ext.get("offset", (0.0, 0.0)), ext.get("scale", (1.0, 1.0))
    return (sc[0], sc[1], off[0], off[1])


def _tex(info):
    """(texture index, uv set, uv transform) of a textureInfo, or the reference's "none"."""
    if info is None:
        return abi.NO_TEXTURE, 0, (1.0, 1.0, 0.0, 0.0)
    return info["index"], info.get("texCoord", 0), _uv_transform(info)


def build_material(gm):
    """``build_material`` (scenery/src/scene.rs:346-577), field for field."""
    m = np.zeros((), abi.MATERIAL)
    pbr = gm.get("pbrMetallicRoughness", {})
    ext = gm.get("extensions", {})
    sg = ext.get("KHR_materials_pbrSpecularGlossiness")
    base = list(pbr.get("baseColorFactor", (1.0, 1.0, 1.0, 1.0)))
    if "baseColorTexture" in pbr:
        alb = _tex(pbr["baseColorTexture"])
    elif sg is not None:
        alb = _tex(sg.get("diffuseTexture"))
    else:
        alb = _tex(None)
    nrm, emi, occ = _tex(gm.get("normalTexture")), _tex(gm.get("emissiveTexture")), _tex(gm.get("occlusionTexture"))
    mr = _tex(pbr.get("metallicRoughnessTexture"))
    mtype = abi.MAT_METAL_ROUGHNESS
    spec_gloss = (0.0, 0.0, 0.0, 0.0)
    if mr[0] == abi.NO_TEXTURE:
        if sg is not None:
            mtype = abi.MAT_SPEC_GLOSS
            sf = sg.get("specularFactor", (1.0, 1.0, 1.0))
            spec_gloss = (sf[0], sf[1], sf[2], sg.get("glossinessFactor", 1.0))
            base = list(sg.get("diffuseFactor", (1.0, 1.0, 1.0, 1.0)))
            mr = _tex(sg.get("specularGlossinessTexture"))
        else:
            mr = _tex(None)
    spec_factor = [1.0, 1.0, 1.0, 1.0]
    spec_color_tex, spec_color_uv, spec_tex, spec_uv = abi.NO_TEXTURE, 0, abi.NO_TEXTURE, 0
    sp = ext.get("KHR_materials_specular")
    if sp is not None:
        c = sp.get("specularColorFactor", (1.0, 1.0, 1.0))
        spec_factor = [c[0], c[1], c[2], sp.get("specularFactor", 1.0)]
        if "specularColorTexture" in sp:
            spec_color_tex, spec_color_uv = sp["specularColorTexture"]["index"], sp["specularColorTexture"].get("texCoord", 0)
        if "specularTexture" in sp:
            spec_tex, spec_uv = sp["specularTexture"]["index"], sp["specularTexture"].get("texCoord", 0)
        else:
            spec_uv = abi.NO_TEXTURE  # the reference's unwrap_or(!0) (scene.rs:460)
        mtype |= abi.MAT_SPEC
    clearcoat = [1.0, 0.0, 0.0, 0.0]
    cc_tex = abi.NO_TEXTURE
    cc = ext.get("KHR_materials_clearcoat")
    if cc is not None:
        clearcoat = [cc.get("clearcoatFactor", 0.0), cc.get("clearcoatRoughnessFactor", 0.0), 0.0, 0.0]
        cc_tex = cc.get("clearcoatNormalTexture", {}).get("index", abi.NO_TEXTURE)
        mtype |= abi.MAT_CLEARCOAT
    sheen = [1.0, 1.0, 1.0, 1.0]
    sh = ext.get("KHR_materials_sheen")
    if sh is not None:
        c = sh.get("sheenColorFactor", (0.0, 0.0, 0.0))
        sheen = [c[0], c[1], c[2], sh.get("sheenRoughnessFactor", 0.0)]
        mtype |= abi.MAT_SHEEN
    m["albedo_texture"], m["albedo_texture_uv"], m["albedo_texture_transform"] = alb
    m["normal_texture"], m["normal_texture_uv"], m["normal_texture_transform"] = nrm
    m["metallic_roughness_texture"], m["metallic_roughness_texture_uv"], m["metallic_roughness_texture_transform"] = mr
    m["emissive_texture"], m["emissive_texture_uv"], m["emissive_texture_transform"] = emi
    m["occlusion_texture"], m["occlusion_texture_uv"], m["occlusion_texture_transform"] = occ
    m["clearcoat_texture"], m["clearcoat_texture_uv"] = cc_tex, 0
    m["specular_color_texture"], m["specular_color_texture_uv"] = spec_color_tex, spec_color_uv
    m["specular_texture"], m["specular_texture_uv"] = spec_tex, spec_uv
    m["sheen_color_roughness_factor"] = sheen
    m["base_color_factor"] = base
    m["metallic_roughness_normal_occlusion"] = (pbr.get("metallicFactor", 1.0), pbr.get("roughnessFactor", 1.0),
                                                (gm.get("normalTexture") or {}).get("scale", 1.0),
                                                (gm.get("occlusionTexture") or {}).get("strength", 1.0))
    e = gm.get("emissiveFactor", (0.0, 0.0, 0.0))
    m["emissive_factor_alpha_cutoff"] = (e[0], e[1], e[2], gm.get("alphaCutoff", 0.5))
    m["alpha_mode"] = 0 if gm.get("alphaMode", "OPAQUE") == "OPAQUE" else 1
    m["material_type"] = mtype
    m["specular_glossiness_factor"] = spec_gloss
    m["clearcoat_factor"] = clearcoat
    m["ior"] = ext.get("KHR_materials_ior", {}).get("ior", 1.5)
    m["specular_factor"] = spec_factor
    return m


def _load_texture_images(doc, base_dir):
    """One ``Image`` per glTF *texture* (the reference indexes its image list by texture index,
    scenery/src/scene.rs:74-95): RGBA8, one level, sRGB iff the path contains "basecolor"."""
    from PIL import Image as PilImage
    out, cache = [], {}
    for tex in doc.get("textures", []):
        src = tex.get("source")
        uri = doc["images"][src].get("uri") if src is not None else None
        if uri is None or uri.startswith("data:"):
            out.append(None)  # buffer-view / embedded images: the reference skips them with a warning
            continue
        path = os.path.join(base_dir, uri)
        if path not in cache:
            cache[path] = None
            ext = os.path.splitext(path)[1].lower()
            if ext in (".jpg", ".jpeg", ".png", ".dds"):
                # a .dds file of the same name wins over a .jpg / .png (image_loader.rs:254-263);
                # any failure to load it falls back to the named file
                try:
                    cache[path] = read_dds(os.path.splitext(path)[0] + ".dds")
                except (OSError, DdsError):
                    if ext == ".dds":
                        raise
            if ext == ".ktx2":  # image_loader.rs:264
                from .ktx2 import load_image_ktx2
                cache[path] = load_image_ktx2(path)
            if cache[path] is None:
                with PilImage.open(path) as im:
                    cache[path] = Image([np.array(im.convert("RGBA"), np.uint8)], srgb="basecolor" in path)
        out.append(cache[path])
    return out


# ---------------------------------------------------------------------------------------------
# loader
# ---------------------------------------------------------------------------------------------
def load_gltf(path, env=None, max_lods=8, fast_mode=False):
    doc, bin_chunk = _read_container(path)
    base_dir = os.path.dirname(os.path.abspath(path))
    buffers = _load_buffers(doc, base_dir, bin_chunk)
    nodes_json = doc.get("nodes", [])
    scenes = doc.get("scenes", [])
    if not scenes:
        raise ValueError("GLTF file doesn't have a default scene")
    roots = scenes[doc.get("scene", 0)].get("nodes", [])

    sc = Scene()
    sc.env = env
    # breadth-first node list (scene.rs:103-112)
    order = list(roots)
    k = 0
    while k < len(order):
        order.extend(nodes_json[order[k]].get("children", []))
        k += 1
    camera = None
    sun = DEFAULT_SUN
    lights = doc.get("extensions", {}).get("KHR_lights_punctual", {}).get("lights", [])
    prim_mesh = {}
    for ni in order:
        node = nodes_json[ni]
        if "camera" in node:
            cam = doc["cameras"][node["camera"]]
            if cam.get("type") != "perspective":
                raise ValueError("only Perspective is supported")
            t, r, _ = _node_trs(node)
            camera = Camera(tuple(float(c) for c in t), tuple(float(c) for c in _quat_to_mat3(r)[2]))  # rotation * Z
        li = node.get("extensions", {}).get("KHR_lights_punctual", {}).get("light")
        if li is not None and lights[li].get("type") == "directional":
            sun = tuple(float(c) for c in _local_matrix(node)[2, :3])
        if "mesh" in node:
            for pi, prim in enumerate(doc["meshes"][node["mesh"]]["primitives"]):
                v, idx = _primitive_vertices(doc, buffers, prim)
                prim_mesh[node["mesh"] * 16 + pi] = sc.append_mesh_auto(v, idx, max_lods, fast_mode)  # Geometry::append_mesh (scenery/src/scene.rs:210)
    sc.materials = (np.stack([build_material(m) for m in doc.get("materials", [])]) if doc.get("materials")
                    else np.zeros(0, abi.MATERIAL))
    sc.images = _load_texture_images(doc, base_dir)
    sc.camera = camera or Camera((0.0, 0.0, 0.0), (0.0, 0.0, 1.0))
    sc.sun_direction = sun

    transforms, draws = [], []

    def visit(ni, parent):
        node = nodes_json[ni]
        world = _mul(parent, _trs_to_mat4(*_node_trs(node)))
        model_index = len(transforms)
        transforms.append(world.reshape(16))
        if "mesh" in node:
            for pi, prim in enumerate(doc["meshes"][node["mesh"]]["primitives"]):
                mesh_index = prim_mesh[node["mesh"] * 16 + pi]
                mat = prim.get("material")
                masked = mat is not None and doc["materials"][mat].get("alphaMode", "OPAQUE") != "OPAQUE"
                d = np.zeros((), abi.MESH_DRAW)
                d["alpha_mode"] = 1 if masked else 0
                d["meshlet_visibility_offset"] = sc.meshlet_visibility_count
                d["model_index"] = model_index
                d["mesh_index"] = mesh_index
                d["material_index"] = 0xFFFF if mat is None else mat
                draws.append(d)
                sc.meshlet_visibility_count += int(sc.meshes[mesh_index]["lods"][0]["meshlet_count"])
        for c in node.get("children", []):
            visit(c, world)

    ident = np.eye(4, dtype=np.float32)
    for ni in roots:
        visit(ni, ident)
    sc.global_transforms = np.asarray(transforms, np.float32).reshape(-1, 16)
    sc.mesh_draws = np.stack(draws) if draws else np.zeros(0, abi.MESH_DRAW)
    return sc


# ---------------------------------------------------------------------------------------------
# exporter (tests, and handing synthetic scenes to other tools)
# ---------------------------------------------------------------------------------------------
def save_gltf(scene, path):
    """Write ``scene`` as ``path`` (.gltf) + one .bin + one PNG per image.  One node per draw
    (matrix transform), one glTF mesh per scene mesh, metallic-roughness materials with the
    KHR_texture_transform / ior extensions this package's scenes use."""
    from .imageio import write_png
    base_dir = os.path.dirname(os.path.abspath(path))
    stem = os.path.splitext(os.path.basename(path))[0]
    blob = bytearray()
    views, accessors = [], []

    def add(arr, comp, typ, target=None, minmax=False):
        while len(blob) % 4:
            blob.append(0)
        a = np.ascontiguousarray(arr)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": a.nbytes, **({"target": target} if target else {})})
        blob.extend(a.tobytes())
        acc = {"bufferView": len(views) - 1, "componentType": comp, "count": len(a), "type": typ}
        if minmax:
            acc["min"], acc["max"] = a.min(0).tolist(), a.max(0).tolist()
        accessors.append(acc)
        return len(accessors) - 1

    # glTF binds the material to the primitive: one glTF mesh per (scene mesh, material) pair in use
    meshes, mesh_of_pair, geometry = [], {}, {}
    index_ends = [int(m["index_offset"]) for m in scene.meshes[1:]] + [len(scene.indices)]
    for d in scene.mesh_draws:
        pair = (int(d["mesh_index"]), int(d["material_index"]))
        if pair in mesh_of_pair:
            continue
        mi, mat = pair
        if mi not in geometry:
            m = scene.meshes[mi]
            v = scene.vertices[int(m["vertex_offset"]):int(m["vertex_offset"]) + int(m["vertex_count"])]
            idx = scene.indices[int(m["index_offset"]):index_ends[mi]]
            geometry[mi] = ({"POSITION": add(v["p"][:, :3].copy(), 5126, "VEC3", 34962, True),
                             "NORMAL": add(v["n"][:, :3].copy(), 5126, "VEC3", 34962),
                             "TEXCOORD_0": add(v["uv"][:, 0:2].copy(), 5126, "VEC2", 34962),
                             "TEXCOORD_1": add(v["uv"][:, 2:4].copy(), 5126, "VEC2", 34962),
                             "TANGENT": add(v["t"].copy(), 5126, "VEC4", 34962)},
                            add(idx.astype(np.uint32), 5125, "SCALAR", 34963))
        prim = {"attributes": dict(geometry[mi][0]), "indices": geometry[mi][1], "mode": 4}
        if mat != 0xFFFF:
            prim["material"] = mat
        mesh_of_pair[pair] = len(meshes)
        meshes.append({"primitives": [prim]})

    images, textures = [], []
    for k, img in enumerate(scene.images):
        name = f"{stem}_{'basecolor' if img.srgb else 'data'}_{k}.png"
        write_png(os.path.join(base_dir, name), img.mips[0])
        images.append({"uri": name})
        textures.append({"source": k})

    def tex_info(index, uv, tx):
        info = {"index": int(index), "texCoord": int(uv)}
        if tuple(float(c) for c in tx) != (1.0, 1.0, 0.0, 0.0):
            info["extensions"] = {"KHR_texture_transform": {"scale": [float(tx[0]), float(tx[1])], "offset": [float(tx[2]), float(tx[3])]}}
        return info

    materials = []
    for m in scene.materials:
        f = m["metallic_roughness_normal_occlusion"]
        pbr = {"baseColorFactor": [float(c) for c in m["base_color_factor"]], "metallicFactor": float(f[0]), "roughnessFactor": float(f[1])}
        gm = {"pbrMetallicRoughness": pbr, "emissiveFactor": [float(c) for c in m["emissive_factor_alpha_cutoff"][:3]],
              "alphaMode": "MASK" if int(m["alpha_mode"]) == 1 else "OPAQUE", "alphaCutoff": float(m["emissive_factor_alpha_cutoff"][3]),
              "extensions": {"KHR_materials_ior": {"ior": float(m["ior"])}}}
        if int(m["albedo_texture"]) != abi.NO_TEXTURE:
            pbr["baseColorTexture"] = tex_info(m["albedo_texture"], m["albedo_texture_uv"], m["albedo_texture_transform"])
        if int(m["metallic_roughness_texture"]) != abi.NO_TEXTURE:
            pbr["metallicRoughnessTexture"] = tex_info(m["metallic_roughness_texture"], m["metallic_roughness_texture_uv"],
                                                       m["metallic_roughness_texture_transform"])
        if int(m["normal_texture"]) != abi.NO_TEXTURE:
            gm["normalTexture"] = {**tex_info(m["normal_texture"], m["normal_texture_uv"], m["normal_texture_transform"]), "scale": float(f[2])}
        if int(m["occlusion_texture"]) != abi.NO_TEXTURE:
            gm["occlusionTexture"] = {**tex_info(m["occlusion_texture"], m["occlusion_texture_uv"], m["occlusion_texture_transform"]),
                                      "strength": float(f[3])}
        if int(m["emissive_texture"]) != abi.NO_TEXTURE:
            gm["emissiveTexture"] = tex_info(m["emissive_texture"], m["emissive_texture_uv"], m["emissive_texture_transform"])
        mt = int(m["material_type"])
        if mt & abi.MAT_CLEARCOAT:
            cc = {"clearcoatFactor": float(m["clearcoat_factor"][0]), "clearcoatRoughnessFactor": float(m["clearcoat_factor"][1])}
            if int(m["clearcoat_texture"]) != abi.NO_TEXTURE:
                cc["clearcoatNormalTexture"] = {"index": int(m["clearcoat_texture"])}
            gm["extensions"]["KHR_materials_clearcoat"] = cc
        if mt & abi.MAT_SHEEN:
            s = m["sheen_color_roughness_factor"]
            gm["extensions"]["KHR_materials_sheen"] = {"sheenColorFactor": [float(s[0]), float(s[1]), float(s[2])], "sheenRoughnessFactor": float(s[3])}
        if mt & abi.MAT_SPEC:
            s = m["specular_factor"]
            sp = {"specularColorFactor": [float(s[0]), float(s[1]), float(s[2])], "specularFactor": float(s[3])}
            if int(m["specular_color_texture"]) != abi.NO_TEXTURE:
                sp["specularColorTexture"] = {"index": int(m["specular_color_texture"]), "texCoord": int(m["specular_color_texture_uv"])}
            if int(m["specular_texture"]) != abi.NO_TEXTURE:
                sp["specularTexture"] = {"index": int(m["specular_texture"]), "texCoord": int(m["specular_texture_uv"])}
            gm["extensions"]["KHR_materials_specular"] = sp
        if mt & abi.MAT_SPEC_GLOSS:  # the loader takes this branch only without a metallic-roughness texture
            g = m["specular_glossiness_factor"]
            sg = {"diffuseFactor": [float(c) for c in m["base_color_factor"]], "specularFactor": [float(g[0]), float(g[1]), float(g[2])],
                  "glossinessFactor": float(g[3])}
            if "baseColorTexture" in pbr:
                sg["diffuseTexture"] = pbr.pop("baseColorTexture")
            if "metallicRoughnessTexture" in pbr:
                sg["specularGlossinessTexture"] = pbr.pop("metallicRoughnessTexture")
            gm["extensions"]["KHR_materials_pbrSpecularGlossiness"] = sg
        materials.append(gm)

    nodes = []
    for d in scene.mesh_draws:
        nodes.append({"mesh": mesh_of_pair[(int(d["mesh_index"]), int(d["material_index"]))],
                      "matrix": [float(c) for c in scene.global_transforms[int(d["model_index"])]]})
    doc = {"asset": {"version": "2.0", "generator": "renderer_amd"}, "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}],
           "nodes": nodes, "meshes": meshes, "materials": materials, "textures": textures, "images": images,
           "buffers": [{"uri": stem + ".bin", "byteLength": len(blob)}], "bufferViews": views, "accessors": accessors,
           "extensionsUsed": ["KHR_texture_transform", "KHR_materials_ior", "KHR_materials_clearcoat", "KHR_materials_sheen",
                              "KHR_materials_specular", "KHR_materials_pbrSpecularGlossiness"]}
    if not materials:
        del doc["materials"]
    with open(os.path.join(base_dir, stem + ".bin"), "wb") as f:
        f.write(bytes(blob))
    with open(path, "w", encoding="utf-8") as f:
        json.dump(doc, f)
