"""Seeded synthetic scenes standing in for the reference's assets (every shipped scene lacks its geometry blobs).

Each generator returns a `Scene` in wire layout with the exact triangle / material / texture counts of the
asset it replaces:
    cornell            38 tris,  8 materials   (matrices of scenes/cornell-box/scene_v3.xml; light split into 4 tris)
    atrium             262,267 tris, 25 materials  (Sponza-sized colonnade hall, open roof, camera_sponza of gltf.rs:59-80)
    glass_blob         134,995 tris,  3 materials  (smooth displaced sphere, spec_trans 1, ior 1.5)
    car                412,845 tris, 65 materials, 1 texture + 6x512^2 cube map
    interior           1,567,673 tris, 43 materials, 37 textures, emissive ceiling strips
Generators are pure numpy, seeded with 0x5EED0000 + config id, and accept `scale` < 1 to shrink triangle
counts for CPU-sized tests (the structure stays the same).
"""
from __future__ import annotations

import math

import numpy as np

from . import api
from .scene import Scene, default_material, look_at_camera, make_triangles


# ------------------------------------------------------------------------------------------- mesh helpers
class Mesh:
    def __init__(self):
        self.v, self.n, self.uv, self.m = [], [], [], []

    def count(self):
        return sum(len(x) for x in self.v)

    def add(self, verts, normals, uvs, mat):
        verts = np.asarray(verts, dtype=np.float32).reshape(-1, 3, 3)
        k = len(verts)
        if k == 0:
            return
        if normals is None:
            fn = np.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0])
            ln = np.linalg.norm(fn, axis=1, keepdims=True)
            fn = fn / np.maximum(ln, 1e-30)
            normals = np.repeat(fn[:, None, :], 3, axis=1)
        if uvs is None:
            uvs = np.zeros((k, 3, 2), dtype=np.float32)
        self.v.append(verts)
        self.n.append(np.asarray(normals, dtype=np.float32).reshape(-1, 3, 3))
        self.uv.append(np.asarray(uvs, dtype=np.float32).reshape(-1, 3, 2))
        self.m.append(np.broadcast_to(np.asarray(mat, dtype=np.uint32), (k,)).copy())

    def grid(self, fn, nu, nv, mat, smooth=True, uv_scale=(1.0, 1.0), wrap_u=False):
        """Parametric surface fn(u, v) -> (..., 3) on [0,1]^2, nu x nv quads, two triangles each."""
        u = np.linspace(0.0, 1.0, nu + 1)
        v = np.linspace(0.0, 1.0, nv + 1)
        U, V = np.meshgrid(u, v, indexing="ij")
        P = np.asarray(fn(U, V), dtype=np.float64)
        if smooth:
            e = 1e-4
            du = np.asarray(fn(np.clip(U + e, 0, 1), V)) - np.asarray(fn(np.clip(U - e, 0, 1), V))
            dv = np.asarray(fn(U, np.clip(V + e, 0, 1))) - np.asarray(fn(U, np.clip(V - e, 0, 1)))
            N = np.cross(du, dv)
            ln = np.linalg.norm(N, axis=-1, keepdims=True)
            N = np.where(ln > 1e-20, N / np.maximum(ln, 1e-20), np.array([0.0, 1.0, 0.0]))
        UV = np.stack([U * uv_scale[0], V * uv_scale[1]], axis=-1)
        i0, i1 = slice(0, nu), slice(1, nu + 1)
        j0, j1 = slice(0, nv), slice(1, nv + 1)

        def tri(a, b, c):
            return np.stack([a, b, c], axis=2).reshape(-1, 3, a.shape[-1])

        p00, p10, p11, p01 = P[i0, j0], P[i1, j0], P[i1, j1], P[i0, j1]
        verts = np.concatenate([tri(p00, p10, p11), tri(p00, p11, p01)])
        uvs = np.concatenate([tri(UV[i0, j0], UV[i1, j0], UV[i1, j1]), tri(UV[i0, j0], UV[i1, j1], UV[i0, j1])])
        normals = None
        if smooth:
            normals = np.concatenate([tri(N[i0, j0], N[i1, j0], N[i1, j1]), tri(N[i0, j0], N[i1, j1], N[i0, j1])])
        # drop degenerate triangles (poles) so no zero-area primitives are emitted
        a = np.linalg.norm(np.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0]), axis=1)
        keep = a > 1e-12
        self.add(verts[keep], None if normals is None else normals[keep], uvs[keep], mat)

    def box(self, lo, hi, mat, xform=None):
        lo, hi = np.asarray(lo, float), np.asarray(hi, float)
        c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                      [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]])
        if xform is not None:
            c = c @ xform[:3, :3].T + xform[:3, 3]
        quads = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 4, 7, 3)]  # outward winding
        verts, uvs = [], []
        quv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], float)
        for q in quads:
            verts += [c[[q[0], q[1], q[2]]], c[[q[0], q[2], q[3]]]]
            uvs += [quv[[0, 1, 2]], quv[[0, 2, 3]]]
        v = np.array(verts)
        if xform is not None and np.linalg.det(xform[:3, :3]) < 0:
            v = v[:, ::-1]
        self.add(v, None, np.array(uvs), mat)

    def quad(self, p, mat, split4=False):
        """p: 4 corners in order; two triangles, or a 4-triangle fan about the centre."""
        p = np.asarray(p, float)
        quv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], float)
        if split4:
            c, cuv = p.mean(axis=0), np.array([0.5, 0.5])
            v = [[p[i], p[(i + 1) % 4], c] for i in range(4)]
            uv = [[quv[i], quv[(i + 1) % 4], cuv] for i in range(4)]
        else:
            v = [p[[0, 1, 2]], p[[0, 2, 3]]]
            uv = [quv[[0, 1, 2]], quv[[0, 2, 3]]]
        self.add(np.array(v), None, np.array(uv), mat)

    def pebbles(self, rng, n, lo, hi, size, mat_choices):
        """n small random triangles inside the box [lo, hi]: pads a scene to an exact triangle count."""
        if n <= 0:
            return
        c = rng.uniform(lo, hi, size=(n, 1, 3))
        v = c + rng.uniform(-size, size, size=(n, 3, 3))
        self.add(v, None, rng.uniform(0, 1, size=(n, 3, 2)), rng.choice(mat_choices, size=n))

    def triangles(self):
        return make_triangles(np.concatenate(self.v), np.concatenate(self.n), np.concatenate(self.uv), np.concatenate(self.m))


def _rng(config_id):
    return np.random.Generator(np.random.PCG64(0x5EED0000 + config_id))


def _noise_texture(rng, size, base, contrast=0.35, channels=None):
    """Smooth seeded RGBA8 texture: low-resolution noise upsampled bilinearly."""
    g = max(4, size // 32)
    low = rng.uniform(-1, 1, size=(g + 1, g + 1, 3))
    xs = np.linspace(0, g, size, endpoint=False)
    x0 = np.floor(xs).astype(int)
    fx = (xs - x0)[:, None]
    rows = low[x0] * (1 - fx[:, :, None]) + low[x0 + 1] * fx[:, :, None]           # (size, g+1, 3)
    img = rows[:, x0] * (1 - fx[None, :, :]) + rows[:, x0 + 1] * fx[None, :, :]     # (size, size, 3)
    out = np.clip(np.asarray(base)[None, None, :] + contrast * img, 0, 1)
    if channels is not None:
        out = out * np.asarray(channels)[None, None, :]
    rgba = np.empty((size, size, 4), dtype=np.uint8)
    rgba[..., :3] = np.round(out * 255).astype(np.uint8)
    rgba[..., 3] = 255
    return rgba


def _sky_cube(rng, size):
    """Seeded 6-face sRGB cube map: vertical gradient plus a bright 'sun' blob on +Y and soft noise."""
    faces = np.zeros((6, size, size, 4), dtype=np.uint8)
    t = (np.arange(size) + 0.5) / size
    for f in range(6):
        n = _noise_texture(rng, size, (0.55, 0.65, 0.85), 0.15)[..., :3].astype(np.float32) / 255
        if f == 2:
            yy, xx = np.meshgrid(t, t, indexing="ij")
            n = np.clip(n + 0.9 * np.exp(-((xx - 0.6) ** 2 + (yy - 0.4) ** 2) / 0.01)[..., None], 0, 1)
        elif f == 3:
            n = n * 0.35
        else:
            n = n * (1.0 - 0.5 * t)[:, None, None]
        faces[f, ..., :3] = np.round(n * 255).astype(np.uint8)
        faces[f, ..., 3] = 255
    return faces


def _sky_cube_hdr(rng, size, sun=60.0):
    """The same sky as linear fp32 radiance, with a sun on +Y an 8-bit image cannot hold (peak `sun`)."""
    ldr = _sky_cube(rng, size)[..., :3].astype(np.float64) / 255.0
    lin = np.where(ldr <= 0.04045, ldr / 12.92, ((ldr + 0.055) / 1.055) ** 2.4)
    faces = np.ones((6, size, size, 4), dtype=np.float32)
    faces[..., :3] = lin
    t = (np.arange(size) + 0.5) / size
    yy, xx = np.meshgrid(t, t, indexing="ij")
    faces[2, ..., :3] += (sun * np.exp(-((xx - 0.6) ** 2 + (yy - 0.4) ** 2) / 0.002))[..., None].astype(np.float32)
    return faces


# ------------------------------------------------------------------------------------------- 1. cornell
_CORNELL = {  # name: (shape, row-major 3x4 to_world, material) — scenes/cornell-box/scene_v3.xml:66-116
    "Floor": ("rect", [-4.37114e-08, 1, 4.37114e-08, 0, 0, -8.74228e-08, 2, 0, 1, 4.37114e-08, 1.91069e-15, 0], 2),
    "Ceiling": ("rect", [-1, 7.64274e-15, -1.74846e-07, 0, 8.74228e-08, 8.74228e-08, -2, 2, 0, -1, -4.37114e-08, 0], 3),
    "BackWall": ("rect", [1.91069e-15, 1, 1.31134e-07, 0, 1, 3.82137e-15, -8.74228e-08, 1, -4.37114e-08, 1.31134e-07, -2, -1], 4),
    "RightWall": ("rect", [4.37114e-08, -1.74846e-07, 2, 1, 1, 3.82137e-15, -8.74228e-08, 1, 3.82137e-15, 1, 2.18557e-07, 0], 1),
    "LeftWall": ("rect", [-4.37114e-08, 8.74228e-08, -2, -1, 1, 3.82137e-15, -8.74228e-08, 1, 0, -1, -4.37114e-08, 0], 0),
    "ShortBox": ("cube", [0.0851643, 0.289542, 1.31134e-08, 0.328631, 3.72265e-09, 1.26563e-08, -0.3, 0.3, -0.284951, 0.0865363, 5.73206e-16, 0.374592], 5),
    "TallBox": ("cube", [0.286776, 0.098229, -2.29282e-15, -0.335439, -4.36233e-09, 1.23382e-08, -0.6, 0.6, -0.0997984, 0.282266, 2.62268e-08, -0.291415], 6),
    "Light": ("rect", [0.235, -1.66103e-08, -7.80685e-09, -0.005, -2.05444e-08, 3.90343e-09, -0.0893, 1.98, 2.05444e-08, 0.19, 8.30516e-09, -0.03], 7),
}
_CORNELL_ALBEDO = [(0.63, 0.065, 0.05), (0.14, 0.45, 0.091), (0.725, 0.71, 0.68), (0.725, 0.71, 0.68), (0.725, 0.71, 0.68),
                   (0.725, 0.71, 0.68), (0.725, 0.71, 0.68), (0.0, 0.0, 0.0)]


def cornell(width=256, height=256) -> Scene:
    mesh = Mesh()
    for name, (shape, m34, mat) in _CORNELL.items():
        M = np.eye(4)
        M[:3, :] = np.array(m34, dtype=np.float64).reshape(3, 4)
        if shape == "rect":
            p = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], float) @ M[:3, :3].T + M[:3, 3]
            mesh.quad(p, mat, split4=(name == "Light"))
        else:
            mesh.box([-1, -1, -1], [1, 1, 1], mat, xform=M)
    mats = default_material(8)
    for i, c in enumerate(_CORNELL_ALBEDO):
        mats["base_colour"][i, :3] = c
        mats["roughness"][i] = 1.0
    mats["emission"][7, :3] = (17.0, 12.0, 4.0)
    focal = 0.5 / math.tan(math.radians(19.5 / 2))
    cam = look_at_camera(width, height, (0.0, 1.0, 6.8), (0.0, 1.0, 0.0), focal_length=focal, focus_distance=1.0)
    return Scene(triangles=mesh.triangles(), materials=mats, camera=cam, name="cornell")


# ------------------------------------------------------------------------------------------- 2. atrium
def atrium(width=1920, height=1080, scale=1.0, lights=True) -> Scene:
    """Sponza-sized hall: 30 x 12 x 13, two storeys of colonnades with arches, hanging cloth, open roof."""
    rng = _rng(2)
    target = 262_267 if scale >= 1.0 else max(2_000, int(262_267 * scale))
    s = math.sqrt(min(1.0, scale))
    r = lambda k: max(2, int(round(k * s)))
    mesh = Mesh()
    X, Y, Z = 15.0, 12.0, 6.5
    n_mat = 25
    mats = default_material(n_mat)
    mats["shade_smooth"] = 0                      # the reference's glTF path shades flat (gltf.rs:292-413)
    mats["ior"] = 1.5
    mats["base_colour"][:, :3] = rng.uniform(0.25, 0.85, size=(n_mat, 3))
    mats["roughness"] = rng.uniform(0.3, 1.0, size=n_mat)
    LAMP = 24
    mats["base_colour"][LAMP, :3] = 0.0
    if lights:
        mats["emission"][LAMP, :3] = (40.0, 36.0, 28.0)
    stone = np.arange(0, 8)
    cloth = np.arange(8, 16)
    trim = np.arange(16, 24)

    bump = lambda U, V, a, f: a * np.sin(f * U * 2 * np.pi) * np.sin(f * 0.7 * V * 2 * np.pi)
    # floor and the four walls (gently displaced so the BVH sees real geometry, not 2 triangles)
    mesh.grid(lambda U, V: np.stack([-X + 2 * X * U, bump(U, V, 0.01, 40), Z - 2 * Z * V], -1), r(120), r(52), stone[0], False, (30, 13))
    mesh.grid(lambda U, V: np.stack([-X + 2 * X * U, Y * V, -Z + bump(U, V, 0.03, 25)], -1), r(150), r(60), stone[1], False, (15, 6))
    mesh.grid(lambda U, V: np.stack([X - 2 * X * U, Y * V, Z + bump(U, V, 0.03, 25)], -1), r(150), r(60), stone[2], False, (15, 6))
    mesh.grid(lambda U, V: np.stack([-X + bump(U, V, 0.03, 12), Y * V, Z - 2 * Z * U], -1), r(65), r(60), stone[3], False, (6, 6))
    mesh.grid(lambda U, V: np.stack([X + bump(U, V, 0.03, 12), Y * V, -Z + 2 * Z * U], -1), r(65), r(60), stone[4], False, (6, 6))
    # gallery slabs along both long walls
    for sz in (-1, 1):
        mesh.box([-X, 5.0, sz * Z - (2.5 if sz > 0 else 0)], [X, 5.3, sz * Z + (2.5 if sz < 0 else 0)], stone[5])
    # columns + arches, two storeys, two rows
    n_col = 10
    xs = np.linspace(-X + 2.0, X - 2.0, n_col)
    for level, (y0, y1) in enumerate(((0.0, 4.2), (5.3, 9.5))):
        for sz in (-1, 1):
            zc = sz * (Z - 2.5)
            for k, xc in enumerate(xs):
                rad = 0.32 if level == 0 else 0.25
                flute = lambda U, V, xc=xc, zc=zc, rad=rad, y0=y0, y1=y1: np.stack(
                    [xc + (rad + 0.02 * np.cos(16 * 2 * np.pi * U)) * np.cos(2 * np.pi * U) * (1 - 0.15 * V),
                     y0 + (y1 - y0) * V,
                     zc + (rad + 0.02 * np.cos(16 * 2 * np.pi * U)) * np.sin(2 * np.pi * U) * (1 - 0.15 * V)], -1)
                mesh.grid(flute, r(32), r(24), trim[(k + level) % 8], False, (4, 8))
                mesh.box([xc - 0.45, y1, zc - 0.45], [xc + 0.45, y1 + 0.25, zc + 0.45], trim[(k + 3) % 8])
            for k in range(n_col - 1):
                xa, xb = xs[k], xs[k + 1]
                R = 0.5 * (xb - xa)
                arch = lambda U, V, xa=xa, R=R, zc=zc, y1=y1: np.stack(
                    [xa + R - (R - 0.18 * np.cos(2 * np.pi * V)) * np.cos(np.pi * U),
                     y1 + 0.25 + (R * 0.55 - 0.18 * np.cos(2 * np.pi * V)) * np.sin(np.pi * U),
                     zc + 0.3 * np.sin(2 * np.pi * V)], -1)
                mesh.grid(arch, r(24), r(12), stone[6 + (k % 2)], False, (6, 2))
    # hanging cloth in the nave
    n_cloth = 6
    for k in range(n_cloth):
        xc = -X + 3.0 + k * (2 * X - 6.0) / (n_cloth - 1)
        ph = rng.uniform(0, 2 * np.pi)
        cl = lambda U, V, xc=xc, ph=ph: np.stack(
            [xc + 0.25 * np.sin(6 * np.pi * U + ph) * (0.3 + V) + 0.1 * np.sin(14 * np.pi * V),
             10.5 - 5.5 * V - 0.4 * np.sin(np.pi * U),
             -2.2 + 4.4 * U], -1)
        mesh.grid(cl, r(96), r(96), cloth[k % 8], False, (4, 4))
    # lamps: emissive quads facing down under the gallery slabs (2 triangles each, count kept a multiple of 4)
    if lights:
        for k in range(8):
            xc = -X + 3.5 + k * (2 * X - 7.0) / 7
            zc = (Z - 1.25) * (1 if k % 2 else -1)
            mesh.quad([[xc - 0.4, 4.95, zc - 0.4], [xc + 0.4, 4.95, zc - 0.4], [xc + 0.4, 4.95, zc + 0.4], [xc - 0.4, 4.95, zc + 0.4]], LAMP)
    left = target - mesh.count()
    if left < 0 and scale >= 1.0:
        raise RuntimeError(f"atrium generator overshot its triangle budget by {-left}")
    left = max(left, 0)   # reduced-scale variants keep their structural minimum
    mesh.pebbles(rng, left, [-X + 1, 0.02, -Z + 1], [X - 1, 0.12, Z - 1], 0.05, stone)
    cam = look_at_camera(width, height, (3.0, 2.0, 0.0), (-4.0, 2.0, 0.0), focal_length=1.0, focus_distance=55.0)
    return Scene(triangles=mesh.triangles(), materials=mats, camera=cam, name="atrium-262k" if scale >= 1.0 else f"atrium-{mesh.count()}")


# ------------------------------------------------------------------------------------------- 3. glass blob
def glass_blob(width=1920, height=1080, scale=1.0, roughness=0.0) -> Scene:
    rng = _rng(3)
    target = 134_995 if scale >= 1.0 else max(500, int(134_995 * scale))
    s = math.sqrt(min(1.0, scale))
    mesh = Mesh()
    mats = default_material(3)
    mats["base_colour"][0, :3] = (0.9, 0.95, 0.92)
    mats["spec_trans"][0] = 1.0
    mats["ior"][0] = 1.5
    mats["roughness"][0] = roughness
    mats["base_colour"][1, :3] = (0.6, 0.6, 0.6)
    mats["roughness"][1] = 0.9
    mats["base_colour"][2, :3] = 0.0
    mats["emission"][2, :3] = (30.0, 30.0, 28.0)
    ph = rng.uniform(0, 2 * np.pi, size=4)

    def blob(U, V):
        th, phi = np.pi * V, 2 * np.pi * U
        rr = 1.0 + np.sin(th) * (0.18 * np.sin(5 * phi + ph[0]) * np.sin(4 * th + ph[1]) + 0.07 * np.sin(11 * phi + ph[2]) * np.sin(9 * th + ph[3]))
        return np.stack([rr * np.sin(th) * np.cos(phi), 1.3 + rr * np.cos(th), rr * np.sin(th) * np.sin(phi)], -1)

    nu, nv = max(8, int(260 * s)), max(6, int(260 * s))
    mesh.grid(blob, nu, nv, 0, True)
    mesh.quad([[-6, 0, 6], [6, 0, 6], [6, 0, -6], [-6, 0, -6]], 1)
    mesh.quad([[-1.5, 5.0, -1.5], [1.5, 5.0, -1.5], [1.5, 5.0, 1.5], [-1.5, 5.0, 1.5]], 2, split4=True)
    left = target - mesh.count()
    if left < 0 and scale >= 1.0:
        raise RuntimeError("glass_blob overshot its triangle budget")
    left = max(left, 0)   # reduced-scale variants keep their structural minimum
    mesh.pebbles(rng, left, [-4, 0.01, -4], [4, 0.06, 4], 0.04, [1])
    cam = look_at_camera(width, height, (0.0, 2.2, 5.5), (0.0, 1.2, 0.0), focal_length=1.2, focus_distance=1.0)
    return Scene(triangles=mesh.triangles(), materials=mats, camera=cam, name="glass-blob-135k" if scale >= 1.0 else f"glass-blob-{mesh.count()}")


# ------------------------------------------------------------------------------------------- 4/5. object soups
def _soup(rng, mesh, n_objects, budget, region_lo, region_hi, mat_ids, res_hint):
    """Scatter smooth closed objects (ellipsoids / tori) with textured uvs until ~budget triangles are used."""
    per = max(40, budget // max(1, n_objects))
    for k in range(n_objects):
        if mesh.count() + per > budget + mesh._soup_base:
            break
        c = rng.uniform(region_lo, region_hi)
        ax = rng.uniform(0.3, 1.0, size=3) * res_hint
        nu = max(6, int(math.sqrt(per / 2) * 1.15))
        nv = max(4, per // (2 * nu))
        mat = int(rng.choice(mat_ids))
        if k % 3 == 2:
            R, rr = ax[0], 0.35 * ax[1]
            fn = lambda U, V, c=c, R=R, rr=rr: np.stack([c[0] + (R + rr * np.cos(2 * np.pi * V)) * np.cos(2 * np.pi * U),
                                                        c[1] + rr * np.sin(2 * np.pi * V),
                                                        c[2] + (R + rr * np.cos(2 * np.pi * V)) * np.sin(2 * np.pi * U)], -1)
        else:
            fn = lambda U, V, c=c, ax=ax: np.stack([c[0] + ax[0] * np.sin(np.pi * V) * np.cos(2 * np.pi * U),
                                                    c[1] + ax[1] * np.cos(np.pi * V),
                                                    c[2] + ax[2] * np.sin(np.pi * V) * np.sin(2 * np.pi * U)], -1)
        mesh.grid(fn, nu, nv, mat, True, (2, 2))


def car(width=1920, height=1080, scale=1.0, tex_size=2048, sky_size=512, hdr_sky=False) -> Scene:
    """412,845 triangles, 65 materials (clearcoat paint, metal, glass, rubber), one base-colour atlas, sRGB cube map
    (`hdr_sky`: additionally the same sky as an fp32 cube map with a sun, which then takes the 8-bit one's place)."""
    rng = _rng(4)
    target = 412_845 if scale >= 1.0 else max(2_000, int(412_845 * scale))
    if scale < 1.0:
        tex_size, sky_size = max(64, int(tex_size * scale)), max(16, int(sky_size * scale))
    n_mat = 65
    mats = default_material(n_mat)
    mats["base_colour"][:, :3] = rng.uniform(0.05, 0.9, size=(n_mat, 3))
    mats["roughness"] = rng.uniform(0.05, 0.8, size=n_mat)
    mats["clearcoat"][:16] = 1.0                              # paint
    mats["metallic"][16:32] = 1.0                             # metal trim
    mats["spec_trans"][32:40] = 1.0                           # glass
    mats["roughness"][32:40] = rng.uniform(0.0, 0.05, size=8)
    mats["ior"][32:40] = 1.5
    mats["base_color_tex"][:16] = 0                           # painted panels use the atlas
    mats["base_color_tex"][40:48] = 0
    mesh = Mesh()
    mesh._soup_base = 0
    mesh.grid(lambda U, V: np.stack([-12 + 24 * U, 0 * U, 12 - 24 * V], -1), 64, 64, 48, False, (8, 8))   # ground
    body = lambda U, V: np.stack([2.4 * np.sin(np.pi * V) * np.cos(2 * np.pi * U) * (1 + 0.15 * np.sin(6 * np.pi * V)),
                                  0.75 + 0.55 * np.cos(np.pi * V) * (1 + 0.3 * np.cos(2 * np.pi * U) ** 2),
                                  1.0 * np.sin(np.pi * V) * np.sin(2 * np.pi * U)], -1)
    nb = max(16, int(300 * math.sqrt(min(1.0, scale))))
    mesh.grid(body, nb, nb, 0, True, (4, 4))
    mesh._soup_base = mesh.count()
    _soup(rng, mesh, 260, int((target - mesh.count()) * 0.98), [-5, 0.3, -4], [5, 2.5, 4], np.arange(1, n_mat), 0.5)
    left = target - mesh.count()
    if left < 0 and scale >= 1.0:
        raise RuntimeError("car overshot its triangle budget")
    left = max(left, 0)   # reduced-scale variants keep their structural minimum
    mesh.pebbles(rng, left, [-8, 0.01, -8], [8, 0.05, 8], 0.04, [48, 49, 50])
    cam = look_at_camera(width, height, (6.5, 2.4, 6.0), (0.0, 0.8, 0.0), focal_length=1.4, focus_distance=1.0)
    sc = Scene(triangles=mesh.triangles(), materials=mats, camera=cam, textures=[_noise_texture(rng, tex_size, (0.6, 0.55, 0.5))],
               skybox=_sky_cube(rng, sky_size), name="car-413k" if scale >= 1.0 else f"car-{mesh.count()}")
    if hdr_sky:
        sc.skybox_hdr = _sky_cube_hdr(_rng(44), sky_size)
    return sc


def interior(width=3840, height=2160, scale=1.0, tex_size=1024) -> Scene:
    """1,567,673 triangles, 43 materials, 37 textures (base colour / metallic-roughness / normal / emission), closed room lit
    by emissive ceiling strips (strip triangle count is a multiple of 4)."""
    rng = _rng(5)
    target = 1_567_673 if scale >= 1.0 else max(4_000, int(1_567_673 * scale))
    if scale < 1.0:
        tex_size = max(32, int(tex_size * scale))
    n_mat, n_tex = 43, 37
    mats = default_material(n_mat)
    mats["base_colour"][:, :3] = rng.uniform(0.2, 0.9, size=(n_mat, 3))
    mats["roughness"] = rng.uniform(0.1, 1.0, size=n_mat)
    mats["metallic"][30:36] = 1.0
    mats["clearcoat"][20:24] = 0.8
    mats["spec_trans"][36:39] = 1.0
    mats["roughness"][36:39] = 0.02
    textures = []
    for t in range(n_tex):
        if t < 24:
            textures.append(_noise_texture(rng, tex_size, rng.uniform(0.3, 0.8, size=3)))
        elif t < 30:   # metallic (b) / roughness (g) maps
            textures.append(_noise_texture(rng, tex_size, (0.0, 0.7, 0.8), 0.3))
        elif t < 35:   # tangent-space normal maps around (0.5, 0.5, 1)
            textures.append(_noise_texture(rng, tex_size, (0.72, 0.72, 0.98), 0.08))
        else:          # emission masks
            textures.append(_noise_texture(rng, tex_size, (0.8, 0.8, 0.8), 0.4))
    for i in range(24):
        mats["base_color_tex"][i] = i
    for i in range(6):
        mats["metallic_roughness_tex"][30 + i] = 24 + i
    for i in range(5):
        mats["normal_tex"][i] = 30 + i
    STRIP = 42
    mats["base_colour"][STRIP, :3] = 0.0
    mats["emission"][STRIP, :3] = (22.0, 21.0, 19.0)
    mats["emission_tex"][STRIP] = 35
    mesh = Mesh()
    mesh._soup_base = 0
    X, Y, Z = 10.0, 4.0, 7.0
    sq = math.sqrt(min(1.0, scale))
    g = lambda k: max(2, int(k * sq))
    mesh.grid(lambda U, V: np.stack([-X + 2 * X * U, 0 * U, Z - 2 * Z * V], -1), g(200), g(140), 0, False, (10, 7))
    mesh.grid(lambda U, V: np.stack([-X + 2 * X * U, Y + 0 * U, -Z + 2 * Z * V], -1), g(100), g(70), 1, False, (10, 7))
    mesh.grid(lambda U, V: np.stack([-X + 2 * X * U, Y * V, -Z + 0 * U], -1), g(160), g(40), 2, False, (10, 2))
    mesh.grid(lambda U, V: np.stack([X - 2 * X * U, Y * V, Z + 0 * U], -1), g(160), g(40), 3, False, (10, 2))
    mesh.grid(lambda U, V: np.stack([-X + 0 * U, Y * V, Z - 2 * Z * U], -1), g(110), g(40), 4, False, (7, 2))
    mesh.grid(lambda U, V: np.stack([X + 0 * U, Y * V, -Z + 2 * Z * U], -1), g(110), g(40), 5, False, (7, 2))
    for k in range(6):   # ceiling strips, 2 quads each -> 24 emissive triangles in total
        xc = -X + 2.0 + k * (2 * X - 4.0) / 5
        for zc in (-3.0, 3.0):
            mesh.quad([[xc - 0.15, Y - 0.02, zc - 1.5], [xc + 0.15, Y - 0.02, zc - 1.5], [xc + 0.15, Y - 0.02, zc + 1.5], [xc - 0.15, Y - 0.02, zc + 1.5]], STRIP)
    mesh._soup_base = mesh.count()
    _soup(rng, mesh, 900, int((target - mesh.count()) * 0.985), [-X + 1, 0.3, -Z + 1], [X - 1, Y - 1.0, Z - 1], np.arange(6, 42), 0.45)
    left = target - mesh.count()
    if left < 0 and scale >= 1.0:
        raise RuntimeError("interior overshot its triangle budget")
    left = max(left, 0)   # reduced-scale variants keep their structural minimum
    mesh.pebbles(rng, left, [-X + 0.5, 0.01, -Z + 0.5], [X - 0.5, 0.05, Z - 0.5], 0.03, np.arange(6, 20))
    cam = look_at_camera(width, height, (8.5, 1.7, 5.5), (-2.0, 1.2, -1.0), focal_length=1.0, focus_distance=1.0)
    return Scene(triangles=mesh.triangles(), materials=mats, camera=cam, textures=textures,
                 name="interior-1.57M" if scale >= 1.0 else f"interior-{mesh.count()}")


# ------------------------------------------------------------------------------------------- analytic lights
def lit_room(width=96, height=64, n_point=3) -> Scene:
    """A small closed room lit by `n_point` analytic point lights (src/types/light.rs:25-32, type 0) AND two emissive ceiling
    panels (8 triangles), so the light selector (disney.comp:669-695) crosses from the analytic buckets into the mesh bucket.
    No shipped scene of the reference fills `lights`; this is the scene that exercises that branch of DirectLight."""
    rng = _rng(6)
    mesh = Mesh()
    mats = default_material(8)
    mats["base_colour"][:, :3] = rng.uniform(0.3, 0.85, size=(8, 3))
    mats["roughness"] = rng.uniform(0.15, 1.0, size=8)
    mats["metallic"][3] = 1.0
    mats["clearcoat"][4] = 1.0
    mats["spec_trans"][5] = 1.0                     # one glass box: shadow rays take the closest-hit (glass rule) variant
    mats["roughness"][5] = 0.05
    mats["base_colour"][7, :3] = 0.0
    mats["emission"][7, :3] = (9.0, 8.0, 6.0)
    X, Y, Z = 3.0, 2.5, 3.0
    mesh.box([-X, 0.0, -Z], [X, Y, Z], 0)            # the room itself (outward winding; paths see its inside, two-sided BSDF)
    for k in range(5):
        c = np.array([-2.0 + k * 1.0, 0.35 + 0.1 * k, -0.8 + 0.45 * k])
        mesh.box(c - 0.3, c + 0.3, 1 + k)
    for xc in (-1.2, 1.2):                          # ceiling panels facing down, 4 triangles each
        mesh.quad([[xc - 0.4, Y - 0.02, -0.4], [xc + 0.4, Y - 0.02, -0.4], [xc + 0.4, Y - 0.02, 0.4], [xc - 0.4, Y - 0.02, 0.4]], 7, split4=True)
    lights = np.zeros(n_point, dtype=api.LIGHT)
    for k in range(n_point):
        lights["position"][k, :3] = (-1.8 + 1.8 * k, 1.4 + 0.3 * k, 1.6 - 0.9 * k)
        lights["emission"][k, :3] = rng.uniform(6.0, 18.0, size=3)
        lights["type"][k] = 0
    cam = look_at_camera(width, height, (0.0, 1.3, 2.8), (0.0, 0.9, 0.0), focal_length=1.0, focus_distance=1.0)
    return Scene(triangles=mesh.triangles(), materials=mats, camera=cam, lights=lights, name=f"lit-room-{n_point}pt")


GENERATORS = {"cornell": cornell, "atrium": atrium, "glass_blob": glass_blob, "car": car, "interior": interior, "lit_room": lit_room}
