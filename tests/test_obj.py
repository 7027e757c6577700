"""OBJ ingestion (scenery/src/scene.rs:262-310, scenery/src/mesh.rs:164-195): positions and normals
only, fixed uv / tangent, one draw of mesh 0 with the reference's fixed camera."""
import numpy as np

from renderer_amd import RendererConfig, abi, synth
from renderer_amd.obj import load_obj, read_obj


def _write_obj(path, v, idx, quads=False):
    p, n = v["p"][:, :3], v["n"][:, :3]
    with open(path, "w") as f:
        f.write("# test mesh\no ball\n")
        for a in p:
            f.write("v %r %r %r\n" % tuple(float(x) for x in a))
        for a in n:
            f.write("vn %r %r %r\n" % tuple(float(x) for x in a))
        for t in np.asarray(idx).reshape(-1, 3) + 1:
            f.write("f %d//%d %d//%d %d//%d\n" % (t[0], t[0], t[1], t[1], t[2], t[2]))


def test_reader_forms(tmp_path):
    (tmp_path / "q.obj").write_text(
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\n"
        "f 1/1/1 2/1/1 3/1/1 4/1/1\n"      # a quad: fanned into two triangles
        "f -4//-1 -2//-1 -1//-1\n")         # negative (relative) indices
    v, idx = read_obj(tmp_path / "q.obj")
    assert len(v) == 4 and idx.tolist() == [0, 1, 2, 0, 2, 3, 0, 2, 3]
    assert np.array_equal(v["n"][:, :3], [[0, 0, 1]] * 4) and np.array_equal(v["uv"], np.zeros((4, 4)))
    assert np.array_equal(v["t"], [[1, 0, 0, 1]] * 4)


def test_scene_follows_the_reference_quirks(oracle, tmp_path):
    v, lods, _ = synth.icosphere(3, lods=1)
    _write_obj(tmp_path / "a.obj", v, lods[0])
    _write_obj(tmp_path / "b.obj", v, lods[0])
    sc = load_obj(f"{tmp_path / 'a.obj'},{tmp_path / 'b.obj'}", env=synth.default_env())
    assert len(sc.meshes) == 2 and len(sc.mesh_draws) == 1          # every file is a mesh, only mesh 0 is drawn
    d = sc.mesh_draws[0]
    assert (int(d["mesh_index"]), int(d["material_index"]), int(d["alpha_mode"]), int(d["model_index"])) == (0, 0, 0, 0)
    assert sc.meshlet_visibility_count == int(sc.meshes[0]["lods"][0]["meshlet_count"])
    assert len(sc.materials) == 1 and sc.materials[0] == abi.default_material()
    assert sc.camera.position == (0.0, 0.0, 2.0) and sc.camera.direction == (0.0, 0.0, 1.0)
    assert int(sc.meshes[0]["lod_count"]) >= 2                       # append_mesh built a LOD chain
    # the unit sphere two units in front of the camera (which looks back along its direction) is on screen
    o = oracle.OracleRenderer(96, 54, RendererConfig(cull_enabled=False))
    o.set_scene(sc)
    vis = o.render_visbuffer()
    covered = vis != np.uint64(abi.VIS_CLEAR)
    assert 0.05 < covered.mean() < 0.9 and covered.reshape(54, 96)[27, 48]
