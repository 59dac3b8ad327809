"""Scene ingestion (YAML + OBJ, the reference's live format, src/scene/loader.rs:23-232). The reference has no loader
tests and its parser crates cannot run here, so the expectations below are hand-derived from the format rules."""
import numpy as np
import pytest

from ray_tracer_amd import api, loaders
from ray_tracer_amd.scene import look_at_camera

OBJ = """# a unit quad with normals and uvs, then a bare triangle in a second group
o quad
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 0 1
f 1/1/1 2/2/1 3/3/1 4/4/1
g bare
v 2 0 0
v 3 0 0
v 2 1 0.5
f -3 -2 -1
"""

YAML = """
camera:
  resolution: [32, 24]
  focal_length: 2.2
  location: [0.5, 0.5, 4.0]
  look_at: [0.5, 0.5, 0.0]
skybox: [px.png, nx.png, py.png, ny.png, pz.png, nz.png]
materials:
  Floor:
    base_colour: [51, 102, 255]
    roughness: 0.75
    relative_ior: 1.5      # not a Material field: ignored
    thin: false
  Lamp:
    emission: [150, 150, 150]
  Glass: {spec_trans: 1.0, ior: 1.5, roughness: 0, shade_smooth: 0}
surfaces:
  - file: quad.obj
    material: Floor
    smooth: true
  - type: mesh
    material: Lamp
    vertices: [[0, 2, 0], [1, 2, 0], [0, 2, 1], [1, 2, 1]]
    triangles: [[0, 1, 2], [1, 3, 2]]
  - file: missing.obj
    material: Glass
"""


@pytest.fixture()
def scene_dir(tmp_path):
    from PIL import Image
    (tmp_path / "quad.obj").write_text(OBJ)
    (tmp_path / "scene.yaml").write_text(YAML)
    for i, name in enumerate(["px", "nx", "py", "ny", "pz", "nz"]):
        Image.new("RGB", (4, 4), (40 * i, 10, 200)).save(tmp_path / f"{name}.png")
    return tmp_path


def test_obj_reader_fans_polygons_and_resolves_negative_indices(scene_dir):
    models = loaders.load_obj(scene_dir / "quad.obj")
    assert [m["name"] for m in models] == ["quad", "bare"]
    q, b = models
    assert q["v"].shape == (2, 3, 3)                                  # quad -> fan (0,1,2), (0,2,3)
    assert np.array_equal(q["v"][0], [[0, 0, 0], [1, 0, 0], [1, 1, 0]]) and np.array_equal(q["v"][1], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])
    assert np.array_equal(q["uv"][1], [[0, 0], [1, 1], [0, 1]]) and (q["n"] == [0, 0, 1]).all()
    assert b["n"] is None and b["uv"] is None                         # no vn / vt on its faces: zeros later
    assert np.array_equal(b["v"][0], [[2, 0, 0], [3, 0, 0], [2, 1, 0.5]])


def test_yaml_scene_components(scene_dir):
    msgs = []
    sc = loaders.load_scene_yaml(scene_dir / "scene.yaml", log=msgs.append)
    assert any("missing.obj" in m for m in msgs)                      # a missing OBJ is reported, not fatal
    assert len(sc.triangles) == 3 + 2 and len(sc.materials) == 3
    floor, lamp, glass = sc.materials
    assert np.array_equal(floor["base_colour"][:3], np.float32([51, 102, 255]) / np.float32(255))
    assert floor["roughness"] == np.float32(0.75) and floor["ior"] == np.float32(1.45) and floor["shade_smooth"] == 1
    assert np.array_equal(lamp["emission"][:3], np.float32([150, 150, 150]) / np.float32(255))
    assert np.array_equal(lamp["base_colour"][:3], np.float32([0.8, 0.8, 0.8]))
    assert glass["spec_trans"] == 1 and glass["shade_smooth"] == 0 and glass["base_color_tex"] == api.NO_TEXTURE
    t = sc.triangles
    assert list(t["material_idx"]) == [0, 0, 0, 1, 1]
    assert (t["n"][2, :, :3] == 0).all() and (t["uv"][2] == 0).all()  # the bare OBJ triangle
    assert np.allclose(t["n"][3, :, :3], [0, -1, 0]) and np.allclose(t["n"][4, :, :3], [0, -1, 0])   # flat normals of the inline mesh
    cam = look_at_camera(32, 24, (0.5, 0.5, 4.0), (0.5, 0.5, 0.0), 2.2, 1.0, 0.0)
    assert sc.camera.tobytes() == cam.tobytes()
    assert sc.skybox.shape == (6, 4, 4, 4) and tuple(sc.skybox[2, 0, 0]) == (80, 10, 200, 255)


def test_from_new_builds_bvh_and_lights(scene_dir, oracle):
    sc = loaders.from_new(scene_dir / "scene.yaml", log=None)
    assert len(sc.bvh) >= 1 and len(sc.emissive_tris) == 2 and sc.total_power > 0
    img, _ = oracle.OracleScene(sc).render_radiance(0, 3)
    assert img.shape[:2] == (24, 32) and np.isfinite(img).all() and img.max() > 0


@pytest.mark.parametrize("bad,msg", [
    ("materials:\n  A: {base_colour: [0.5, 0.5, 0.5]}\n", "three non-negative integers"),
    ("materials:\n  A: {roughness: rough}\n", "must be a number"),
    ("materials: {A: {}}\nsurfaces:\n  - {file: quad.obj, material: B}\n", "unknown material"),
    ("materials: {A: {}}\nsurfaces:\n  - {file: quad.obj}\n", "Material not specified"),
    ("camera: {resolution: [4, 4], location: [0, 0, 1]}\n", "missing field `look_at`"),
    ("camera: {resolution: [4, 4], location: [0, 0, 1], look_at: [0, 0, 0], fov: 60}\n", "unknown field `fov`"),
    ("skybox: [a.png]\n", "six faces"),
])
def test_malformed_scenes_are_rejected(scene_dir, bad, msg):
    (scene_dir / "bad.yaml").write_text(bad)
    with pytest.raises(loaders.SceneFormatError, match=msg):
        loaders.load_scene_yaml(scene_dir / "bad.yaml")


def test_empty_document_sections_give_an_empty_scene(tmp_path):
    (tmp_path / "e.yaml").write_text("materials: {}\nsurfaces: []\n")
    sc = loaders.load_scene_yaml(tmp_path / "e.yaml")
    assert len(sc.triangles) == 0 and len(sc.materials) == 0 and sc.skybox is None


@pytest.mark.gpu
def test_loaded_scene_renders_like_the_oracle(scene_dir, rt, oracle, tmp_path):
    """End to end: YAML + OBJ + cube map -> spatial-split BVH -> HIP path vs the oracle, then the frame saver."""
    from util import rel_close
    sc = loaders.from_new(scene_dir / "scene.yaml", log=None)
    r = rt.HipRenderer.create(sc, max_bounces=4)
    o = oracle.OracleScene(sc)
    for frame in (0, 2):
        g = r.render_radiance(frame)
        h, _ = o.render_radiance(frame, 4)
        assert rel_close(g, h, 1e-3, 1e-4).all(axis=-1).mean() > 0.985
    r.render(4)
    out = loaders.save_frame(r, tmp_path / "frame.png")
    from PIL import Image
    with Image.open(out) as im:
        assert im.size == (32, 24) and im.mode == "RGBA"
        assert np.array_equal(np.asarray(im), r.read_framebuffer())
    r.destroy()
