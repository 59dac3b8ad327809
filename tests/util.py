"""Helpers shared by the tests."""
import numpy as np

from ray_tracer_amd import api
from ray_tracer_amd.scene import Scene, default_material, look_at_camera, make_triangles


def pcg_reference(seed, n):
    """Independent pure-Python statement of random.comp:6-12 (words, states, floats)."""
    words, states, floats = [], [], []
    s = seed & 0xFFFFFFFF
    for _ in range(n):
        s = (s * 747796405 + 2891336453) & 0xFFFFFFFF
        w = ((((s >> ((s >> 28) + 4)) ^ s) & 0xFFFFFFFF) * 277803737) & 0xFFFFFFFF
        words.append(w)
        states.append(s)
        floats.append(np.float32(np.float32(w) / np.float32(4294967296.0)))
    return words, states, floats


def random_rays(rng, n, lo, hi):
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


def chain_scene(depth=60):
    """A deliberately deep, thin BVH: triangle k sits at z = k+1; every interior node has the remaining NEARER
    triangles in its left (interior) child and the farthest one as its right leaf. A ray along +z therefore stacks
    one far leaf per level: stack depth = `depth`, far beyond what the kernels keep in LDS."""
    n = depth + 1
    verts = np.zeros((n, 3, 3), np.float32)
    for k in range(n):
        z = float(k + 1)
        verts[k] = [[-1, -1, z], [1, -1, z], [0, 1.5, z]]
    tris = make_triangles(verts, None, None, np.zeros(n, np.uint32))
    nodes = np.zeros(2 * n - 1, dtype=api.BVH_NODE)

    def box(lo_k, hi_k):  # triangles lo_k..hi_k inclusive
        return (-1, -1, lo_k + 1), (1, 1.5, hi_k + 1)

    # node 0 covers all; interior i covers tris [0, n-1-level]
    idx = 0
    nxt = 1
    level = 0
    while True:
        hi_k = n - 1 - level
        mn, mx = box(0, hi_k)
        nodes[idx]["min"], nodes[idx]["max"] = mn, mx
        if hi_k == 0:
            nodes[idx]["idx"], nodes[idx]["amt"] = 0, 1
            break
        nodes[idx]["idx"], nodes[idx]["amt"] = nxt, 0
        left, right = nxt, nxt + 1
        nxt += 2
        mn, mx = box(hi_k, hi_k)
        nodes[right]["min"], nodes[right]["max"] = mn, mx
        nodes[right]["idx"], nodes[right]["amt"] = hi_k, 1
        idx = left
        level += 1
    mats = default_material(1)
    cam = look_at_camera(16, 16, (0.0, 0.0, -3.0), (0.0, 0.0, 0.0))
    sc = Scene(triangles=tris, materials=mats, camera=cam, name=f"chain-{depth}")
    sc.bvh = nodes[:nxt].copy()
    sc.build_lights()
    return sc


def rel_close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    return both_nan | (np.abs(a - b) <= atol + rtol * np.abs(b))


def two_leaf_scene(n_left=40, n_right=64, seed=3):
    """Root with two leaf children holding `n_left` and `n_right` small triangles (x < 0 and x > 0), every third triangle
    repeated right after itself (an exact tie in distance: the first copy must win). All rays reach a leaf on their first
    node visit, so the leaf phase sees anything from one to sixty-four lanes parked at once, on leaves far larger than
    the number of triangle columns a wavefront forms."""
    import numpy as np
    from ray_tracer_amd import api
    from ray_tracer_amd.scene import Scene, default_material, look_at_camera, make_triangles
    rng = np.random.default_rng(seed)

    def soup(n, x0):
        out = []
        while len(out) < n:
            c = np.array([x0 + rng.uniform(-0.9, 0.9), rng.uniform(-0.9, 0.9), rng.uniform(-0.5, 0.5)])
            tri = (c + rng.uniform(-0.35, 0.35, size=(3, 3))).astype(np.float32)
            out.append(tri)
            if len(out) % 3 == 0 and len(out) < n:
                out.append(tri.copy())
        return np.stack(out[:n])

    left, right = soup(n_left, -1.0), soup(n_right, 1.0)
    verts = np.concatenate([left, right]).astype(np.float32)
    tris = make_triangles(verts, None, None, np.zeros(len(verts), np.uint32))
    nodes = np.zeros(3, dtype=api.BVH_NODE)
    nodes[0]["min"], nodes[0]["max"] = verts.min(axis=(0, 1)), verts.max(axis=(0, 1))
    nodes[0]["idx"], nodes[0]["amt"] = 1, 0
    nodes[1]["min"], nodes[1]["max"] = left.min(axis=(0, 1)), left.max(axis=(0, 1))
    nodes[1]["idx"], nodes[1]["amt"] = 0, n_left
    nodes[2]["min"], nodes[2]["max"] = right.min(axis=(0, 1)), right.max(axis=(0, 1))
    nodes[2]["idx"], nodes[2]["amt"] = n_left, n_right
    sc = Scene(triangles=tris, materials=default_material(1), camera=look_at_camera(8, 8, (0, 0, 4), (0, 0, 0)))
    sc.bvh = nodes
    sc.build_lights()
    return sc


def big_tables_scene(width=80, height=48):
    """400 materials (one glass) and 640 emissive triangles: more than the shade kernel stages in LDS, so it reads its
    material / light tables from global memory, and the glass material forces closest-hit (not first-hit) shadow rays."""
    from ray_tracer_amd.scenes import Mesh
    rng = np.random.default_rng(31)
    n_mat = 400
    mats = default_material(n_mat)
    mats["base_colour"][:, :3] = rng.uniform(0.2, 0.9, size=(n_mat, 3))
    mats["roughness"] = rng.uniform(0.2, 1.0, size=n_mat)
    mats["spec_trans"][5] = 1.0
    mats["emission"][200:, :3] = rng.uniform(2.0, 6.0, size=(200, 1))
    mesh = Mesh()
    mesh.quad([[-6, 0, 6], [6, 0, 6], [6, 0, -6], [-6, 0, -6]], 0)
    for k in range(320):   # 320 small emissive quads facing down = 640 emissive triangles
        x, z = rng.uniform(-4, 4, size=2)
        mesh.quad([[x - 0.1, 3.0, z - 0.1], [x + 0.1, 3.0, z - 0.1], [x + 0.1, 3.0, z + 0.1], [x - 0.1, 3.0, z + 0.1]], 200 + (k % 200))
    for k in range(150):
        c = rng.uniform([-4, 0.3, -4], [4, 2.0, 4])
        mesh.box(c - 0.25, c + 0.25, 1 + (k % 199))
    cam = look_at_camera(width, height, (0.0, 2.0, 7.5), (0.0, 1.0, 0.0))
    sc = Scene(triangles=mesh.triangles(), materials=mats, camera=cam, name="big-tables").finalize()
    assert len(sc.emissive_tris) == 640 and len(sc.materials) == 400
    return sc


def pcg_float_of_state(state):
    """The random number the generator returned when it moved INTO `state` (random.comp:6-12): the output is a function
    of the updated state only, so the last draw of a path can be recovered from the state it left behind."""
    s = np.asarray(state, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    w = ((s >> ((s >> np.uint64(28)) + np.uint64(4))) ^ s) & np.uint64(0xFFFFFFFF)
    w = (w * np.uint64(277803737)) & np.uint64(0xFFFFFFFF)
    return (w.astype(np.float32) / np.float32(4294967296.0)).astype(np.float32)


TIE_EPS = 1e-6


def check_hits_up_to_ties(g, h, o, ro, rd, what=""):
    """Closest hits of the default (compressed 4-wide) traversal `g` against the oracle's `h` on the same rays. The rule:
      * same triangle id  -> t, u, v bit-identical (the triangle test is the shader's);
      * different id      -> either an eps-tie: both hit, |t_g - t_h| <= 1e-6 * t_h, and g's (t, u, v) are exactly what the
                             shader's hit_triangle gives for the triangle g names (visiting order decided between two
                             triangles a rounding error apart);
                             or a hit the reference's own box test loses to rounding: g is then strictly closer than h and
                             equals the brute-force closest hit over ALL triangles.
    Returns (number of ties, number of hits recovered from the reference's box test)."""
    g, h = np.asarray(g), np.asarray(h)
    same = g["prim"] == h["prim"]
    assert np.array_equal(g["t"][same], h["t"][same]) and np.array_equal(g["u"][same], h["u"][same]) and np.array_equal(g["v"][same], h["v"][same]), what
    diff = np.flatnonzero(~same)
    if len(diff) == 0:
        return 0, 0
    ro, rd = np.asarray(ro, np.float32).reshape(-1, 3), np.asarray(rd, np.float32).reshape(-1, 3)
    gd, hd = g[diff], h[diff]
    assert (gd["prim"] != 0xFFFFFFFF).all(), f"{what}: the default traversal missed a hit the reference finds"
    own = o.test_triangles(ro[diff], rd[diff], gd["prim"])          # the triangle g names really is hit, with exactly g's numbers
    assert np.array_equal(own["t"], gd["t"]) and np.array_equal(own["u"], gd["u"]) and np.array_equal(own["v"], gd["v"]), what
    both = hd["prim"] != 0xFFFFFFFF
    tie = both & (np.abs(gd["t"].astype(np.float64) - hd["t"]) <= TIE_EPS * np.abs(hd["t"].astype(np.float64)))
    rest = np.flatnonzero(~tie)
    if len(rest):
        assert (gd["t"][rest] < hd["t"][rest]).all(), f"{what}: a non-tie mismatch where the default traversal is farther"
        brute = o.trace_brute(ro[diff][rest], rd[diff][rest])
        assert np.array_equal(brute["t"], gd["t"][rest]), f"{what}: a closer hit that brute force does not confirm"
    return int(tie.sum()), int(len(rest))
