"""glTF ingestion (scenery/src/scene.rs:62-232, 346-664): a scene exported and loaded again renders
to the same image; hand-written documents check the node hierarchy, camera / sun extraction,
interleaved and normalised accessors, the GLB container and the material mapping."""
import json
import struct

import numpy as np
import pytest

from renderer_amd import RendererConfig, abi, synth
from renderer_amd.gltf import build_material, load_gltf, save_gltf
from renderer_amd.scene import Image
from helpers import psnr


def test_round_trip_renders_the_same_image(oracle, tmp_path):
    sc = synth.make_atrium()
    save_gltf(sc, tmp_path / "atrium.gltf")
    back = load_gltf(tmp_path / "atrium.gltf", env=sc.env)
    back.camera, back.sun_direction = sc.camera, sc.sun_direction
    assert len(back.mesh_draws) == len(sc.mesh_draws) == len(back.meshes)  # one mesh per (node, primitive), as the reference
    assert np.array_equal(back.mesh_draws["material_index"], sc.mesh_draws["material_index"])
    assert np.array_equal(back.mesh_draws["alpha_mode"], sc.mesh_draws["alpha_mode"])
    assert np.abs(back.global_transforms[back.mesh_draws["model_index"]] - sc.global_transforms[sc.mesh_draws["model_index"]]).max() < 1e-5
    carried = [n for n in sc.materials.dtype.names if n not in ("specular_glossiness_factor", "clearcoat_factor", "_pad")]
    for n in carried:
        assert np.array_equal(back.materials[n], sc.materials[n]), n
    assert all(a.srgb == b.srgb and np.array_equal(a.mips[0], b.mips[0]) for a, b in zip(back.images, sc.images))
    sc.images = [Image([im.mips[0]], srgb=im.srgb) for im in sc.images]  # the reference loads a single level per image
    w, h = 240, 135
    out = []
    for s in (sc, back):
        o = oracle.OracleRenderer(w, h, RendererConfig(skybox=True))
        o.set_scene(s)
        o.render_visbuffer()
        out.append((o.read_visbuffer()[..., 1].copy(), o.shade()))
    assert np.array_equal(out[0][0], out[1][0])           # the same draw in every pixel
    assert np.abs(out[0][1].astype(int) - out[1][1].astype(int)).max() <= 2 and psnr(out[0][1], out[1][1]) > 60


def _write_glb(path, doc, blob):
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    blob = bytes(blob) + b"\0" * (-len(blob) % 4)
    with open(path, "wb") as f:
        f.write(b"glTF" + struct.pack("<II", 2, 12 + 8 + len(js) + 8 + len(blob)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(blob), 0x004E4942) + blob)


def test_hierarchy_camera_light_and_accessor_forms(tmp_path):
    # one quad: interleaved position + normal (stride 24), uv as normalised u16, u16 indices, no tangents
    pos_nrm = np.array([[0, 0, 0, 0, 0, 1], [1, 0, 0, 0, 0, 1], [1, 1, 0, 0, 0, 1], [0, 1, 0, 0, 0, 1]], np.float32)
    uv = np.array([[0, 0], [65535, 0], [65535, 65535], [0, 65535]], np.uint16)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    blob = pos_nrm.tobytes() + uv.tobytes() + idx.tobytes()
    doc = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 3]}],
        "nodes": [
            {"translation": [10, 0, 0], "scale": [2, 2, 2], "children": [1, 2]},
            {"mesh": 0, "translation": [0, 5, 0]},
            {"camera": 0, "translation": [1, 2, 3], "rotation": [0, 0.70710678, 0, 0.70710678]},   # 90 degrees about +y
            {"extensions": {"KHR_lights_punctual": {"light": 0}}, "rotation": [0.70710678, 0, 0, 0.70710678]},  # 90 about +x
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 1.0, "znear": 0.1}}],
        "extensions": {"KHR_lights_punctual": {"lights": [{"type": "directional"}]}},
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0}]}],
        "materials": [{"alphaMode": "MASK"}],
        "buffers": [{"byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 96, "byteStride": 24},
                        {"buffer": 0, "byteOffset": 96, "byteLength": 16}, {"buffer": 0, "byteOffset": 112, "byteLength": 12}],
        "accessors": [{"bufferView": 0, "byteOffset": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 0, "byteOffset": 12, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5123, "normalized": True, "count": 4, "type": "VEC2"},
                      {"bufferView": 2, "componentType": 5123, "count": 6, "type": "SCALAR"}],
    }
    _write_glb(tmp_path / "q.glb", doc, blob)
    sc = load_gltf(tmp_path / "q.glb")
    assert np.array_equal(sc.vertices["p"][:, :3], pos_nrm[:, :3]) and np.array_equal(sc.vertices["n"][:, :3], pos_nrm[:, 3:])
    assert np.allclose(sc.vertices["uv"][:, :2], [[0, 0], [1, 0], [1, 1], [0, 1]])
    assert np.array_equal(sc.indices, idx.astype(np.uint32))
    # missing tangents are derived from the uv gradients: +x here, right-handed
    assert np.allclose(sc.vertices["t"], [[1, 0, 0, 1]] * 4, atol=1e-6)
    # one transform per node, depth first: root, its two children, the light node
    assert sc.global_transforms.shape == (4, 16)
    child = sc.global_transforms[1].reshape(4, 4)  # [col][row]
    assert np.allclose(child[3, :3], [10, 10, 0]) and np.allclose(np.diag(child)[:3], [2, 2, 2])  # parent scale applies to the child's offset
    d = sc.mesh_draws[0]
    assert (int(d["model_index"]), int(d["alpha_mode"]), int(d["material_index"]), int(d["mesh_index"])) == (1, 1, 0, 0)
    assert sc.meshlet_visibility_count == int(sc.meshes[0]["lods"][0]["meshlet_count"]) == 1
    # camera: position = the node's translation, direction = rotation * Z (scene.rs:121-129): +z turned about +y -> +x
    assert np.allclose(sc.camera.position, [1, 2, 3]) and np.allclose(sc.camera.direction, [1, 0, 0], atol=1e-6)
    # sun: third column of the light node's matrix (scene.rs:135-140): +z turned about +x -> -y
    assert np.allclose(sc.sun_direction, [0, -1, 0], atol=1e-6)


def test_build_material_follows_the_reference_mapping():
    m = build_material({
        "pbrMetallicRoughness": {"baseColorFactor": [0.1, 0.2, 0.3, 0.4], "metallicFactor": 0.5, "roughnessFactor": 0.6,
                                 "baseColorTexture": {"index": 3, "texCoord": 1,
                                                      "extensions": {"KHR_texture_transform": {"scale": [2, 3], "offset": [0.5, 0.25]}}}},
        "normalTexture": {"index": 4, "scale": 0.7}, "occlusionTexture": {"index": 5, "strength": 0.8},
        "emissiveFactor": [1, 2, 3], "alphaMode": "BLEND", "alphaCutoff": 0.3,
        "extensions": {"KHR_materials_ior": {"ior": 1.33},
                       "KHR_materials_sheen": {"sheenColorFactor": [0.1, 0.2, 0.3], "sheenRoughnessFactor": 0.4},
                       "KHR_materials_clearcoat": {"clearcoatFactor": 0.9, "clearcoatRoughnessFactor": 0.1, "clearcoatNormalTexture": {"index": 6}},
                       "KHR_materials_specular": {"specularFactor": 0.5, "specularColorFactor": [0.9, 0.8, 0.7], "specularColorTexture": {"index": 7}}},
    })
    assert tuple(m["albedo_texture_transform"]) == (2, 3, 0.5, 0.25) and (int(m["albedo_texture"]), int(m["albedo_texture_uv"])) == (3, 1)
    assert np.allclose(m["metallic_roughness_normal_occlusion"], [0.5, 0.6, 0.7, 0.8])
    assert np.allclose(m["emissive_factor_alpha_cutoff"], [1, 2, 3, 0.3]) and int(m["alpha_mode"]) == 1  # anything but OPAQUE is Mask
    assert int(m["material_type"]) == abi.MAT_METAL_ROUGHNESS | abi.MAT_SHEEN | abi.MAT_CLEARCOAT | abi.MAT_SPEC
    assert np.allclose(m["sheen_color_roughness_factor"], [0.1, 0.2, 0.3, 0.4]) and np.allclose(m["clearcoat_factor"], [0.9, 0.1, 0, 0])
    assert int(m["clearcoat_texture"]) == 6 and int(m["specular_color_texture"]) == 7 and int(m["specular_texture"]) == abi.NO_TEXTURE
    assert np.allclose(m["specular_factor"], [0.9, 0.8, 0.7, 0.5]) and np.isclose(m["ior"], 1.33)
    assert int(m["metallic_roughness_texture"]) == abi.NO_TEXTURE and tuple(m["specular_glossiness_factor"]) == (0, 0, 0, 0)
    # spec-gloss: taken only when there is no metallic-roughness texture; the diffuse factor replaces the base colour
    g = build_material({"extensions": {"KHR_materials_pbrSpecularGlossiness": {
        "diffuseFactor": [0.5, 0.5, 0.5, 1], "specularFactor": [0.2, 0.3, 0.4], "glossinessFactor": 0.9,
        "diffuseTexture": {"index": 1}, "specularGlossinessTexture": {"index": 2}}}})
    assert int(g["material_type"]) == abi.MAT_SPEC_GLOSS and np.allclose(g["specular_glossiness_factor"], [0.2, 0.3, 0.4, 0.9])
    assert int(g["albedo_texture"]) == 1 and int(g["metallic_roughness_texture"]) == 2 and np.allclose(g["base_color_factor"], [0.5, 0.5, 0.5, 1])
    # defaults (scene.rs:347-354)
    dflt = build_material({})
    assert np.allclose(dflt["base_color_factor"], 1) and np.allclose(dflt["metallic_roughness_normal_occlusion"], 1)
    assert np.isclose(dflt["emissive_factor_alpha_cutoff"][3], 0.5) and np.isclose(dflt["ior"], 1.5) and int(dflt["alpha_mode"]) == 0


def test_embedded_image_is_skipped_and_the_scene_still_renders(oracle, tmp_path):
    """The reference warns and skips images that live in a buffer view or a data URI
    (scenery/src/scene.rs:170-196 resolves file URIs only): the texture slot stays, without data, and
    samples as zero.  Such a scene must still go through the C ABI description and render."""
    pos = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float32)
    nrm = np.tile(np.array([[0, 0, 1]], np.float32), (4, 1))
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    blob = pos.tobytes() + nrm.tobytes() + uv.tobytes() + idx.tobytes() + b"\x89PNG not really an image"
    doc = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0}]}],
        "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}],
        "textures": [{"source": 0}], "images": [{"bufferView": 4, "mimeType": "image/png"}],
        "buffers": [{"byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 48},
                        {"buffer": 0, "byteOffset": 96, "byteLength": 32}, {"buffer": 0, "byteOffset": 128, "byteLength": 12},
                        {"buffer": 0, "byteOffset": 140, "byteLength": len(blob) - 140}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3", "min": [-1, -1, 0], "max": [1, 1, 0]},
                      {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 2, "componentType": 5126, "count": 4, "type": "VEC2"},
                      {"bufferView": 3, "componentType": 5123, "count": 6, "type": "SCALAR"}],
    }
    _write_glb(tmp_path / "embedded.glb", doc, blob)
    sc = load_gltf(tmp_path / "embedded.glb")
    assert sc.images == [None]
    desc, keep = sc.to_desc()
    assert desc.image_count == 1 and not desc.images[0].data
    from renderer_amd.scene import Camera
    sc.camera = Camera((0.0, 0.0, 4.0), (0.0, 0.0, 1.0))
    o = oracle.OracleRenderer(64, 36, RendererConfig())
    o.set_scene(sc)
    vis = o.render_visbuffer()
    assert (vis != np.uint64(abi.VIS_CLEAR)).any()
    o.shade()  # albedo samples as zero: no crash, a dark quad
    # the flat scene file tolerates the hole too
    from renderer_amd import scene_bin
    scene_bin.save_scene_bin(sc, tmp_path / "embedded.bin")
    assert scene_bin.load_scene_bin(tmp_path / "embedded.bin").images == [None]


@pytest.mark.gpu
def test_gltf_without_tangents_renders_on_the_gpu_like_the_oracle(oracle, tmp_path):
    """A glTF whose primitives carry no TANGENT attribute gets its tangents from the library
    (rdr_generate_tangents, the reference's generate_tangents at scenery/src/scene.rs:207); the loaded scene then
    renders through the HIP path to the oracle's keys and, within tolerance, its colours (normal-mapped materials)."""
    from renderer_amd import Renderer
    sc = synth.make_atrium()
    save_gltf(sc, tmp_path / "atrium.gltf")
    doc = json.load(open(tmp_path / "atrium.gltf"))
    stripped = 0
    for m in doc["meshes"]:
        for p in m["primitives"]:
            stripped += p["attributes"].pop("TANGENT", None) is not None
    assert stripped > 0
    json.dump(doc, open(tmp_path / "atrium.gltf", "w"))
    back = load_gltf(tmp_path / "atrium.gltf", env=sc.env)
    back.camera, back.sun_direction = sc.camera, sc.sun_direction
    t = back.vertices["t"]
    assert np.abs(np.linalg.norm(t[:, :3], axis=1) - 1.0).max() < 1e-3 and set(np.unique(t[:, 3])) <= {-1.0, 1.0}
    w, h = 240, 135
    conf = RendererConfig(cull_enabled=False, skybox=True)
    o = oracle.OracleRenderer(w, h, conf)
    o.set_scene(back)
    ovis = o.render_visbuffer()
    g = Renderer(w, h, conf)
    g.set_scene(back)
    g.render()
    assert np.array_equal(g.read_visbuffer_u64(), ovis)
    err = np.abs(g.read_color().astype(int) - o.shade().astype(int))
    assert err.max() <= 2 and psnr(g.read_color(), o.shade()) >= 40
    g.close()
