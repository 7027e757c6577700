"""``scene.bin``: a flat little-endian dump of exactly the arrays the renderer uploads
(``SceneResources::load_scene``, renderer/src/renderer.rs:426-586), so that a scene loaded by
the reference (glTF / .ptto through ``scenery``, meshlets built by meshoptimizer) can be handed
to this path with identical meshlets and IDs.  A reference-side writer is about 30 lines of Rust after
``Scene::load_*`` (one ``write_all`` per array below); this module reads and writes the same bytes.

Layout (all integers little-endian u32 unless noted, no padding between sections):

    magic     8 bytes  b"RDRSCN01"
    header    16 x u32: vertex_count, index_count, meshlet_vertex_count, meshlet_triangle_bytes,
                        meshlet_count, mesh_count, draw_count, material_count, transform_count,
                        meshlet_visibility_count, image_count, has_env, reserved x4
    camera    9 x f32: position xyz, direction xyz, sun direction xyz
    vertices              vertex_count   x 64 B  (scenery/src/vertex.rs:4-11)
    indices               index_count    x u32
    meshlet_vertices      ...            x u32
    meshlet_triangles     ...            x u8
    meshlets              meshlet_count  x 32 B  (scenery/src/mesh.rs:17-28)
    meshes                mesh_count     x 160 B (mesh.rs:30-49)
    mesh_draws            draw_count     x 16 B  (mesh.rs:61-70)
    materials             material_count x 272 B (mesh.rs:83-123)
    global_transforms     transform_count x 64 B, column-major
    images    image_count x { width, height, mip_count, format (RDR_FORMAT_*), texels mip-major }
    env       if has_env: brdf_lut as an image, then skybox, lambertian, ggx, charlie as
              { size, mip_count, format, texels mip-major / face-major }; size 0 = absent
"""
import struct

import numpy as np

from . import abi
from .scene import BlockImage, Camera, Cubemap, EnvMap, Image, Scene

MAGIC = b"RDRSCN01"


def _texel_dtype(fmt):
    return np.float32 if fmt == abi.FORMAT_RGBA32F else np.uint8


def _image_format(img):
    if img.mips[0].dtype == np.float32:
        return abi.FORMAT_RGBA32F
    return abi.FORMAT_RGBA8_SRGB if img.srgb else abi.FORMAT_RGBA8_UNORM


def _write_image(f, img):
    if img is None:
        f.write(struct.pack("<4I", 0, 0, 0, 0))
        return
    if isinstance(img, BlockImage):  # blocks verbatim
        f.write(struct.pack("<4I", img.width, img.height, img.mip_count, img.format))
        f.write(np.ascontiguousarray(img.data, np.uint8).reshape(-1)[:sum(img.level_bytes(l) for l in range(img.mip_count))].tobytes())
        return
    h, w = img.mips[0].shape[:2]
    f.write(struct.pack("<4I", w, h, len(img.mips), _image_format(img)))
    for m in img.mips:
        f.write(np.ascontiguousarray(m).tobytes())


def _read_image(f):
    w, h, mips, fmt = struct.unpack("<4I", f.read(16))
    if w == 0:
        return None
    if fmt >= abi.FORMAT_BC1:
        img = BlockImage(np.zeros(0, np.uint8), w, h, mips, fmt)
        img.data = np.frombuffer(f.read(sum(img.level_bytes(l) for l in range(mips))), np.uint8).copy()
        return img
    dt = _texel_dtype(fmt)
    out = []
    for l in range(mips):
        mw, mh = max(1, w >> l), max(1, h >> l)
        out.append(np.frombuffer(f.read(mw * mh * 4 * np.dtype(dt).itemsize), dt).reshape(mh, mw, 4).copy())
    return Image(out, srgb=(fmt == abi.FORMAT_RGBA8_SRGB))


def _write_cube(f, cm):
    if cm is None:
        f.write(struct.pack("<3I", 0, 0, 0))
        return
    f.write(struct.pack("<3I", cm.mips[0].shape[1], len(cm.mips), cm.format))
    for m in cm.mips:
        f.write(np.ascontiguousarray(m).tobytes())


def _read_cube(f):
    n, mips, fmt = struct.unpack("<3I", f.read(12))
    if n == 0:
        return None
    if fmt >= abi.FORMAT_BC1:
        raise ValueError(f"cubemap format {fmt}: cubemaps hold RGBA8 or RGBA32F texels, not blocks")
    dt = _texel_dtype(fmt)
    out = []
    for l in range(mips):
        s = max(1, n >> l)
        out.append(np.frombuffer(f.read(6 * s * s * 4 * np.dtype(dt).itemsize), dt).reshape(6, s, s, 4).copy())
    return Cubemap(out)


def save_scene_bin(scene, path):
    arrays = [(scene.vertices, abi.VERTEX), (scene.indices, np.uint32), (scene.meshlet_vertices, np.uint32),
              (scene.meshlet_triangles, np.uint8), (scene.meshlets, abi.MESHLET), (scene.meshes, abi.MESH),
              (scene.mesh_draws, abi.MESH_DRAW), (scene.materials, abi.MATERIAL),
              (np.asarray(scene.global_transforms, np.float32).reshape(-1, 16), np.float32)]
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<16I", len(scene.vertices), len(scene.indices), len(scene.meshlet_vertices),
                            len(scene.meshlet_triangles), len(scene.meshlets), len(scene.meshes), len(scene.mesh_draws),
                            len(scene.materials), len(arrays[8][0]), scene.meshlet_visibility_count,
                            len(scene.images), 1 if scene.env is not None else 0, 0, 0, 0, 0))
        f.write(struct.pack("<9f", *scene.camera.position, *scene.camera.direction, *scene.sun_direction))
        for a, dt in arrays:
            f.write(np.ascontiguousarray(a, dtype=dt).tobytes())
        for img in scene.images:
            _write_image(f, img)
        if scene.env is not None:
            _write_image(f, scene.env.brdf_lut)
            for cm in (scene.env.skybox, scene.env.lambertian, scene.env.ggx, scene.env.charlie):
                _write_cube(f, cm)


def load_scene_bin(path):
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not a scene.bin (bad magic)")
        hdr = struct.unpack("<16I", f.read(64))
        cam = struct.unpack("<9f", f.read(36))

        def arr(count, dt):
            n = count * np.dtype(dt).itemsize
            buf = f.read(n)
            if len(buf) != n:
                raise ValueError(f"{path}: truncated")
            return np.frombuffer(buf, dt).copy()

        sc = Scene()
        sc.vertices = arr(hdr[0], abi.VERTEX)
        sc.indices = arr(hdr[1], np.uint32)
        sc.meshlet_vertices = arr(hdr[2], np.uint32)
        sc.meshlet_triangles = arr(hdr[3], np.uint8)
        sc.meshlets = arr(hdr[4], abi.MESHLET)
        sc.meshes = arr(hdr[5], abi.MESH)
        sc.mesh_draws = arr(hdr[6], abi.MESH_DRAW)
        sc.materials = arr(hdr[7], abi.MATERIAL)
        sc.global_transforms = arr(hdr[8] * 16, np.float32).reshape(-1, 16)
        sc.meshlet_visibility_count = hdr[9]
        sc.images = [_read_image(f) for _ in range(hdr[10])]
        if hdr[11]:
            lut = _read_image(f)
            sky, lam, ggx, cha = (_read_cube(f) for _ in range(4))
            sc.env = EnvMap(brdf_lut=lut, lambertian=lam, ggx=ggx, charlie=cha, skybox=sky)
        sc.camera = Camera(tuple(cam[0:3]), tuple(cam[3:6]))
        sc.sun_direction = tuple(cam[6:9])
        return sc
