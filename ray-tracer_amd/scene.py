"""Host-side `Scene`, mirroring the reference's `Scene` (src/scene/mod.rs:67-270): it owns the triangle /
material / light arrays in wire layout, builds the BVH and the light table through the library, and
produces the packed blob the renderer uploads. All heavy lifting happens in librt_hip.so."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import api
from .api import BVH_NODE, CAMERA, EMISSIVE_TRI, LIGHT, MATERIAL, TRIANGLE, check, load_library, ptr


def default_material(n: int = 1) -> np.ndarray:
    """`Material::default()` (src/types/material.rs:32-57)."""
    m = np.zeros(n, dtype=MATERIAL)
    m["base_colour"][:, :3] = 0.8
    m["roughness"] = 0.5
    m["sheen_tint"] = 0.5
    m["clearcoat_roughness"] = 0.03
    m["ior"] = 1.45
    m["shade_smooth"] = 1
    for k in ("base_color_tex", "metallic_roughness_tex", "normal_tex", "emission_tex"):
        m[k] = api.NO_TEXTURE
    return m


def make_triangles(verts: np.ndarray, normals: np.ndarray | None, uvs: np.ndarray | None, material_idx) -> np.ndarray:
    """verts: (n,3,3). normals default to the face normal, uvs to zero."""
    verts = np.asarray(verts, dtype=np.float32)
    n = verts.shape[0]
    t = np.zeros(n, dtype=TRIANGLE)
    t["v"][:, :, :3] = verts
    if normals is None:
        fn = np.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0])
        ln = np.linalg.norm(fn, axis=1, keepdims=True)
        fn = np.where(ln > 0, fn / np.maximum(ln, 1e-30), 0).astype(np.float32)
        normals = np.repeat(fn[:, None, :], 3, axis=1)
    t["n"][:, :, :3] = np.asarray(normals, dtype=np.float32)
    if uvs is not None:
        t["uv"] = np.asarray(uvs, dtype=np.float32)
    t["material_idx"] = np.asarray(material_idx, dtype=np.uint32)
    return t


def look_at_camera(width, height, location, look_at, focal_length=1.0, focus_distance=1.0, aperture_radius=0.0) -> np.ndarray:
    """Camera block from a look-at pose (src/types/camera_buffer_obj.rs:150-170), computed by the library."""
    lib = load_library()
    cam = np.zeros(1, dtype=CAMERA)
    loc = np.asarray(location, dtype=np.float32)
    tgt = np.asarray(look_at, dtype=np.float32)
    check(lib.rt_camera_look_at(ptr(cam), width, height, ptr(loc), ptr(tgt), focal_length, focus_distance, aperture_radius))
    return cam


@dataclass
class Scene:
    """Field names follow `SceneComponents` upstream."""
    triangles: np.ndarray
    materials: np.ndarray
    camera: np.ndarray                      # one CAMERA record
    lights: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=LIGHT))
    textures: list = field(default_factory=list)   # list of (h, w, 4) uint8 arrays, sRGB-encoded
    skybox: np.ndarray | None = None               # (6, h, w, 4) uint8 or None (white 1x1 cube)
    skybox_hdr: np.ndarray | None = None           # (6, h, w, 4) float32 linear radiance; used instead of `skybox` when set
    name: str = "scene"
    bvh: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=BVH_NODE))
    emissive_tris: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=EMISSIVE_TRI))
    cdf: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.float32))
    total_power: float = 0.0
    bvh_stats: dict = field(default_factory=dict)

    # Scene::build_bvh (src/scene/mod.rs:84-115): reorders the triangles in place. `spatial=True` selects the
    # reference's own policy, BinSah::spatial() (bin_sah.rs:94-100,127-134): triangles may then be duplicated.
    def build_bvh(self, threads: int = 0, min_leaf: int = 2, max_leaf: int = 8, spatial: bool = False,
                  alpha: float = 1e-5, budget: float = 0.3, sweep: bool = False):
        lib = load_library()
        n = len(self.triangles)
        self.triangles = np.ascontiguousarray(self.triangles)
        params = api.RtBvhParams(1.0, 1.0, min_leaf, max_leaf, threads)
        stats = api.RtBvhStats()
        n_nodes = C.c_uint32(0)
        if sweep:      # the reference's first builder, `Bvh` (bvh.rs:183-331)
            nodes = np.zeros(max(1, 2 * n), dtype=BVH_NODE)
            check(lib.rt_bvh_build_sweep(ptr(self.triangles), n, ptr(nodes), C.byref(n_nodes), C.byref(stats)))
        elif spatial:
            cap = n + int(n * max(budget, 0.0)) + 1
            out = np.zeros(cap, dtype=TRIANGLE)
            nodes = np.zeros(2 * cap, dtype=BVH_NODE)
            n_out = C.c_uint32(0)
            check(lib.rt_bvh_build_spatial(ptr(self.triangles), n, C.byref(params), alpha, budget, ptr(out), cap,
                                           C.byref(n_out), ptr(nodes), 2 * cap, C.byref(n_nodes), C.byref(stats)))
            self.triangles = out[: n_out.value].copy()
        else:
            nodes = np.zeros(max(1, 2 * n), dtype=BVH_NODE)
            check(lib.rt_bvh_build(ptr(self.triangles), n, C.byref(params), ptr(nodes), C.byref(n_nodes), C.byref(stats)))
        self.bvh = nodes[: n_nodes.value].copy()
        self.bvh_stats = stats.as_dict()
        self.bvh_stats["duplicated_refs"] = len(self.triangles) - n
        return self

    # Scene::build_lights (src/scene/mod.rs:207-270); must follow build_bvh.
    def build_lights(self):
        lib = load_library()
        n = len(self.triangles)
        em = np.zeros(max(1, n), dtype=EMISSIVE_TRI)
        cdf = np.zeros(max(1, n), dtype=np.float32)
        n_em = C.c_uint32(0)
        total = C.c_float(0)
        check(lib.rt_build_lights(ptr(self.triangles), n, ptr(self.materials), len(self.materials), ptr(em), ptr(cdf),
                                  C.byref(n_em), C.byref(total)))
        self.emissive_tris = em[: n_em.value].copy()
        self.cdf = cdf[: n_em.value].copy()
        self.total_power = float(total.value)
        return self

    def counts(self):
        return (len(self.bvh), len(self.materials), len(self.triangles), len(self.lights), len(self.emissive_tris))

    # Scene::get_buffer_sizes (src/scene/mod.rs:117-157)
    def get_buffer_sizes(self) -> np.ndarray:
        lib = load_library()
        sizes = np.zeros(6, dtype=np.uint64)
        check(lib.rt_scene_buffer_sizes(*self.counts(), ptr(sizes)))
        return sizes

    # Scene::write_buffers (src/scene/mod.rs:166-195)
    def write_buffers(self) -> np.ndarray:
        lib = load_library()
        sizes = self.get_buffer_sizes()
        blob = np.zeros(int(sizes.sum()), dtype=np.uint8)
        nb, nm, nt, nl, ne = self.counts()
        self.materials = np.ascontiguousarray(self.materials)
        check(lib.rt_scene_write_buffers(ptr(self.bvh), nb, ptr(self.materials), nm, ptr(self.triangles), nt,
                                         ptr(np.ascontiguousarray(self.lights)), nl, ptr(self.emissive_tris), ptr(self.cdf), ne,
                                         ptr(blob), blob.size))
        return blob

    def finalize(self, threads: int = 0, spatial: bool = False):
        """build_bvh + build_lights, the order `Scene::from_new` uses (src/scene/loader.rs:17-18)."""
        return self.build_bvh(threads=threads, spatial=spatial).build_lights()
