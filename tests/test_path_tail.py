"""The per-path kernel that finishes the last bounces of a short pass (k_path_tail) against the wavefront kernels it stands
in for: the same paths, the same hits, the same arithmetic in the same order, so images and ray counts must be IDENTICAL,
not merely close. The kernel is switched off and tuned through the environment, which a context reads once at creation."""
import os
from contextlib import contextmanager

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@contextmanager
def env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    try:
        for k, v in kw.items():
            os.environ[k] = str(v)
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def render(rt, sc, *, tail_max, tail_from=None, tail_slots=1 << 30, spp=3, bounces=8, flags=0, fpp=1, **kw):
    e = {"RT_HIP_TAIL_MAX": tail_max, "RT_HIP_TAIL_SLOTS": tail_slots}
    if tail_from is not None:
        e["RT_HIP_TAIL_FROM"] = tail_from
    with env(**e):
        r = rt.HipRenderer.create(sc, max_bounces=bounces, flags=flags, frames_per_pass=fpp, **kw)
    r.render(spp)
    img, cnt = r.read_sum(), r.counters()
    r.destroy()
    return img, cnt


CASES = [
    # scene, generator arguments, resolution, bounces: any-hit shadow rays / the glass rule / textures + sky without lights /
    # analytic point lights / more materials and lights than the LDS tables hold
    ("atrium", {"scale": 0.08}, (320, 180), 8),
    ("glass_blob", {"scale": 0.05}, (320, 180), 8),
    ("car", {"scale": 0.05}, (320, 180), 8),
    ("lit_room", {}, (320, 180), 8),
    ("interior", {"scale": 0.02}, (320, 180), 12),
]


@pytest.mark.parametrize("mode", ["default", "exact_arith"])
@pytest.mark.parametrize("name,kw,res,bounces", CASES)
def test_tail_kernel_changes_nothing(rt, scene_cache, name, kw, res, bounces, mode):
    sc = scene_cache(name, spatial=True, width=res[0], height=res[1], **kw)
    flags = rt.api.RT_EXACT_ARITH if mode == "exact_arith" else 0
    off, c_off = render(rt, sc, tail_max=0, bounces=bounces, flags=flags)
    assert c_off["tail_paths"] == 0
    for tail_from in (1, 3):
        on, c_on = render(rt, sc, tail_max=1 << 20, tail_from=tail_from, bounces=bounces, flags=flags)
        assert c_on["tail_paths"] > 0, "the kernel never took a queue over"
        print(f"{name} ({mode}), per-path kernel from bounce {tail_from}: {c_on['tail_paths']} paths taken over in 3 passes, "
              f"{c_on['extend_rays']} + {c_on['shadow_rays']} rays")
        assert np.array_equal(on, off)
        for k in ("camera_rays", "extend_rays", "shadow_rays"):
            assert c_on[k] == c_off[k], k


def test_tail_leaves_long_queues_alone_and_takes_short_ones(rt, scene_cache):
    sc = scene_cache("atrium", scale=0.08, width=320, height=180)
    npix = 320 * 180
    # a limit below any queue length of the first bounces, then one that only the late bounces meet
    _, c = render(rt, sc, tail_max=1, tail_from=1)
    assert c["tail_paths"] <= 3                       # at most a lone survivor per pass
    img_mid, c_mid = render(rt, sc, tail_max=npix // 4, tail_from=1)
    img_off, c_off = render(rt, sc, tail_max=0)
    assert 0 < c_mid["tail_paths"] <= 3 * (npix // 4)
    assert np.array_equal(img_mid, img_off) and c_mid["extend_rays"] == c_off["extend_rays"]


def test_tail_in_multi_frame_passes_and_both_lanes(rt, scene_cache):
    """Four frames per pass on two alternating lanes, with the shadow rays of a bounce on the side stream: the kernel's vertex
    adds must come after the previous bounce's shadow rays."""
    sc = scene_cache("lit_room", width=256, height=144)
    off, c_off = render(rt, sc, tail_max=0, spp=16, fpp=4)
    on, c_on = render(rt, sc, tail_max=1 << 20, tail_from=2, spp=16, fpp=4)
    assert c_on["tail_paths"] > 0
    assert np.array_equal(on, off)
    assert c_on["extend_rays"] == c_off["extend_rays"] and c_on["shadow_rays"] == c_off["shadow_rays"]
    # and frames-per-pass does not matter either
    one, _ = render(rt, sc, tail_max=1 << 20, tail_from=2, spp=16, fpp=1)
    assert np.array_equal(one, off)


def test_tail_is_not_offered_where_it_does_not_apply(rt, scene_cache):
    sc = scene_cache("atrium", scale=0.08, width=160, height=90)
    _, c = render(rt, sc, tail_max=1 << 20, tail_from=1, tail_slots=160 * 90 - 1)      # passes larger than the limit
    assert c["tail_paths"] == 0
    _, c = render(rt, sc, tail_max=1 << 20, tail_from=1, tail_slots=160 * 90)
    assert c["tail_paths"] > 0
    for flags in (rt.api.RT_EXACT_ORDER, rt.api.RT_KERNEL_TIMING, rt.api.RT_COUNT_TRAVERSAL):
        _, c = render(rt, sc, tail_max=1 << 20, tail_from=1, flags=flags)
        assert c["tail_paths"] == 0


def test_graph_replay_of_short_passes_changes_nothing(rt, scene_cache):
    """RT_HIP_GRAPH=1: a lane captures a short pass into a HIP graph the second time in a row it is given that shape and replays
    it afterwards, with the first frame index patched into the ray-generation node. Same image and counters as direct launches
    (RT_HIP_GRAPH=0), over runs of equal calls long enough for every lane to replay, calls of other sizes in between, a camera
    move (which invalidates the captured graphs) and a reset."""
    sc = scene_cache("lit_room", width=256, height=144)

    def run(graph):
        with env(RT_HIP_GRAPH=graph):
            r = rt.HipRenderer.create(sc, max_bounces=6, frames_per_pass=2)
        out = []
        for n in (1,) * 10 + (2, 5, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2):
            r.dispatch_compute(n)
        r.frame_complete()
        out.append(r.read_sum())
        cam = np.array(sc.camera, copy=True)
        cam["location"][0][0] += 0.25
        r.set_camera(cam)
        for n in (3,) + (1,) * 9:
            r.render(n)
        out.append(r.read_sum())
        r.reset_accumulator()
        for _ in range(7):
            r.render(4)
        out.append(r.read_sum())
        cnt = r.counters()
        r.destroy()
        return out, cnt

    direct, c0 = run(0)
    replayed, c1 = run(1)
    for a, b in zip(direct, replayed):
        assert np.array_equal(a, b)
    assert c0["extend_rays"] == c1["extend_rays"] and c0["shadow_rays"] == c1["shadow_rays"] and c0["camera_rays"] == c1["camera_rays"]
