"""rt_trace_closest_static: the closest-hit entry point that allocates no device memory (rays, hits and its work cursor
live in a fixed device array inside the library). Its hits equal rt_trace_closest's bit for bit, call after call, also
when a call follows a larger one (the array keeps the previous call's contents) and at the ray-count limit."""
import numpy as np
import pytest

from util import random_rays

pytestmark = pytest.mark.gpu


def _rays(sc, n, seed):
    rng = np.random.default_rng(seed)
    lo = sc.triangles["v"][..., :3].min(axis=(0, 1)) - 0.5
    hi = sc.triangles["v"][..., :3].max(axis=(0, 1)) + 0.5
    return random_rays(rng, n, lo, hi)


def test_static_entry_matches_allocating_entry_twice_second_smaller(rt, scene_cache):
    sc = scene_cache("atrium", scale=0.05, width=96, height=54)
    r = rt.HipRenderer.create(sc, max_bounces=2)
    for n, seed in ((50000, 1), (777, 2)):          # two calls in a row, the second with a smaller input
        ro, rd = _rays(sc, n, seed)
        want = r.trace_closest(ro, rd)
        got = r.trace_closest_static(ro, rd)
        assert np.array_equal(got["prim"], want["prim"]) and np.array_equal(got["t"], want["t"])
        assert np.array_equal(got["u"], want["u"]) and np.array_equal(got["v"], want["v"])
        assert (got["prim"] != 0xFFFFFFFF).mean() > 0.3
    r.destroy()


def test_static_entry_limit_and_second_context(rt, scene_cache):
    from ray_tracer_amd import api
    sc = scene_cache("cornell", width=64, height=64)
    a = rt.HipRenderer.create(sc, max_bounces=2, flags=api.RT_COUNT_TRAVERSAL)
    b = rt.HipRenderer.create(sc, max_bounces=2)          # a second context on the same GPU shares the array
    ro, rd = _rays(sc, api.RT_STATIC_MAX_RAYS, 3)
    want = a.trace_closest(ro, rd)
    a.reset_counters()
    got = a.trace_closest_static(ro, rd)                  # exactly at the limit
    assert np.array_equal(got, want)
    c = a.counters()
    assert c["extend_node_pops"] > 0 and c["extend_tri_tests"] > 0
    assert np.array_equal(b.trace_closest_static(ro[:100], rd[:100]), want[:100])
    assert len(a.trace_closest_static(ro[:0], rd[:0])) == 0
    ro2, rd2 = _rays(sc, api.RT_STATIC_MAX_RAYS + 1, 4)
    with pytest.raises(api.RtError) as e:
        a.trace_closest_static(ro2, rd2)
    assert e.value.code == -4 and "RT_STATIC_MAX_RAYS" in str(e.value)
    a.destroy(); b.destroy()
