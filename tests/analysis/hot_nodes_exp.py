"""Analysis aid: how much of the interior-node traffic of a render falls on the K hottest BVH nodes?
Renders the scene with the CPU oracle (test infrastructure) at a small resolution with a per-node visit histogram
installed, then reports the share of interior-node pops covered by K nodes chosen (a) by measured frequency,
(b) by surface area (what a builder can know without tracing), (c) breadth-first from the root.

    python tests/analysis/hot_nodes_exp.py [atrium|glass|car|interior] [width height bounces]
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import scenes  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 240
H = int(sys.argv[3]) if len(sys.argv) > 3 else 135
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
maker = {"atrium": scenes.atrium, "glass": scenes.glass_blob, "car": scenes.car, "interior": scenes.interior}[name]
sc = maker(W, H).finalize()
nodes = sc.bvh
n = len(nodes)
o = po.OracleScene(sc)
hist = np.zeros(n, np.uint32)
po.lib().orc_set_visit_histogram(hist.ctypes.data_as(C.c_void_p))
_, cnt = o.render_radiance(0, B)
po.lib().orc_set_visit_histogram(None)
interior = nodes["amt"] == 0
vis = hist.astype(np.float64) * interior
total = vis.sum()
print(f"{name}: {n} nodes ({int(interior.sum())} interior), {cnt['rays'] + cnt['shadow_rays']} rays, {int(total)} interior pops")
ext = nodes["max"].astype(np.float64) - nodes["min"].astype(np.float64)
area = 2 * (ext[:, 0] * ext[:, 1] + ext[:, 1] * ext[:, 2] + ext[:, 2] * ext[:, 0])
area[~interior] = -1
# breadth-first rank
depth = np.full(n, 1 << 30, np.int64)
depth[0] = 0
order = [0]
i = 0
while i < len(order):
    k = order[i]; i += 1
    if nodes["amt"][k] == 0:
        for c in (int(nodes["idx"][k]), int(nodes["idx"][k]) + 1):
            depth[c] = depth[k] + 1
            order.append(c)
bfs_rank = np.empty(n, np.int64)
bfs_rank[np.array(order)] = np.arange(len(order))
bfs_rank[~interior] = 1 << 40
by_freq = np.argsort(-vis)
by_area = np.argsort(-area)
by_bfs = np.argsort(bfs_rank)
print(f"{'K':>6} {'by frequency':>14} {'by area':>10} {'breadth-first':>14}")
for K in (64, 128, 256, 512, 768, 1024, 1536, 2048, 4096, 8192):
    print(f"{K:6d} {vis[by_freq[:K]].sum() / total:14.3f} {vis[by_area[:K]].sum() / total:10.3f} {vis[by_bfs[:K]].sum() / total:14.3f}")
# per-depth share
md = int(depth[interior].max())
shares = [vis[(depth == d)].sum() / total for d in range(md + 1)]
print("share of interior pops by depth:", " ".join(f"{d}:{s:.3f}" for d, s in enumerate(shares) if s > 0.002))
