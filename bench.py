#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (extend + shadow) and 1080p samples/s of the wavefront path tracer on the
Sponza-sized synthetic hall (`atrium-262k`, 1920x1080, 8 bounces) — BASELINE.json configs[1].

One step = one sample per pixel over the full frame. `python bench.py --gpus N --steps K --warmup W`;
for N > 1 launch with torch.distributed.run (one rank per GPU). Each rank renders the full frame for its
own slice of the global frame indices (rank, rank+N, ...) with the scene replicated, and the fp32 accumulation
buffers are sum-reduced with one RCCL all-reduce inside the timed region. Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# With HIP's default of 4 hardware queues the streams of a process that share a queue run one after the other. Matters for
# short passes only (the timed 20-frame passes use one stream at a time), and only takes effect if the HIP runtime has not
# been initialised yet in this process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--scene", default="atrium")
    ap.add_argument("--scale", type=float, default=1.0, help="triangle-count scale of the synthetic scene (1.0 = the named config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-eighth", action="store_true", help="skip the measured 32-CU (one eighth of the chip) leg")
    ap.add_argument("--eighth-child", action="store_true", help=argparse.SUPPRESS)   # internal: run only the 32-CU leg
    ap.add_argument("--exact-order", action="store_true", help="RT_EXACT_ORDER: the reference's tree and visiting order (bit-exact ids) instead of the default compressed 4-wide tree")
    ap.add_argument("--exact-arith", action="store_true", help="RT_EXACT_ARITH: shade stage with correctly rounded division / square root and no FMA (implied by --exact-order)")
    ap.add_argument("--profile-context", choices=("timing", "overlapped"), default=None,
                    help="profiling runs (tools/profile_r3.sh): timing = only the single-stream context that times every launch "
                         "(two passes of --steps frames); overlapped = only the timed region's context")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--frames-per-pass", type=int, default=0)
    ap.add_argument("--partition", choices=("samples", "tiles"), default="samples",
                    help="N > 1: samples = every rank renders whole frames of its own frame indices (per-GPU work fixed: weak scaling); "
                         "tiles = every rank renders its 8x8 tiles of the same frames (total work fixed: strong scaling, bit-identical to one GPU)")
    ap.add_argument("--bvh", choices=("spatial", "object"), default="spatial",
                    help="split policy of the host BVH build; spatial = BinSah::spatial(), what Scene::build_bvh uses upstream")
    return ap.parse_args()


def algorithmic_bytes_extend(rays, node_pops, tri_tests):
    # per ray: 32 B ray in + 16 B hit out + 32 B root node; per interior node popped: two 32-B child boxes; per triangle test: 48 B
    return rays * (32 + 16 + 32) + node_pops * 64 + tri_tests * 48


def algorithmic_bytes_connect(rays, node_pops, tri_tests, wide):
    # per shadow ray: 32 B ray + 16 B queued contribution in, 4 B result, 32 B root node; per node visit two 32-B child boxes,
    # or four 24-B boxes + four references (112 B) when the occlusion queries walk the uncompressed 4-wide tree (exact-order mode, scenes without glass)
    return rays * (32 + 16 + 4 + 32) + node_pops * (112 if wide else 64) + tri_tests * 48


def launch_ranks(n):
    """`bench.py --gpus N` started by hand (no RANK in the environment): run the same command line as N ranks under
    torch.distributed.run on 127.0.0.1 and return the launcher's exit code. Called before anything touches the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks through torch.distributed.run (port {port})", file=sys.stderr, flush=True)
    return subprocess.run(cmd).returncode


def _code_only(text: str) -> str:
    """C/C++ source without comments and with runs of white space collapsed: what the compiler sees, near enough."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":                       # string / character literal: copied whole
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c); i += 1
    return " ".join("".join(out).split())


def kernel_source_hash():
    """Hash of everything the device code of librt_hip.so is built from: the code of its sources and headers (comments and
    layout do not count) and the compiler flags."""
    import hashlib
    from ray_tracer_amd import build as rtbuild
    h = hashlib.sha256()
    for f in sorted(p for p in list(rtbuild.SOURCES) + list(rtbuild.HEADERS) if p.suffix in (".hip", ".h")):
        h.update(f.name.encode())
        h.update(_code_only(f.read_text()).encode())
    h.update(" ".join(rtbuild.FLAGS).encode())
    return h.hexdigest()[:16]


def measure_kernels(rt, api, scene, args, device, base_flags):
    """One pass of the timed region's size in a counting context (work per launch) and two in a context that brackets every
    launch with HIP events on its own stream (times of kernels running alone; the second pass is the one reported)."""
    lanes = int(os.environ.get("RT_HIP_LANES", "1"))
    fpp = args.frames_per_pass or 32
    k = max(1, min(fpp, -(-args.steps // max(1, lanes))))
    rc = rt.HipRenderer.create(scene, max_bounces=args.bounces, device=device, flags=api.RT_COUNT_TRAVERSAL | base_flags,
                               frames_per_pass=args.frames_per_pass)
    rc.render(k)
    c = rc.counters()
    rc.destroy()
    rtm = rt.HipRenderer.create(scene, max_bounces=args.bounces, device=device, flags=api.RT_KERNEL_TIMING | base_flags,
                                frames_per_pass=args.frames_per_pass)
    rtm.render(k)                # warm-up with a pass of the same size, so every timed launch is comparable
    rtm.reset_accumulator()
    rtm.render(k)
    tm = rtm.last_timer_perf()
    rtm.destroy()
    return c, tm, k


N_CU, SIMD_PER_CU, ISSUE_CYCLES_PER_WAVE_INSTR = 256, 4, 2     # a wave64 vector-ALU instruction occupies its SIMD (32 lanes wide) for 2 cycles


def roofline_blocks(args, scene, c, tm, k):
    prof, stale = {}, None
    tag = {"atrium": "", "interior": "_interior", "car": "_car", "glass_blob": "_glass"}.get(args.scene)
    std_size = (3840, 2160) if args.scene == "interior" else (1920, 1080)
    default_cfg = (tag is not None and args.scale == 1.0 and (args.width, args.height) == std_size and args.bounces == 8
                   and args.frames_per_pass == 0 and args.bvh == "spatial" and not args.exact_order and not args.exact_arith)
    pfile = ROOT / "profiles" / f"r3_roofline_counters{tag or ''}.json"
    if default_cfg and pfile.exists():
        prof = json.loads(pfile.read_text())
        stale = prof.get("kernel_source_hash") != kernel_source_hash()
        if stale:
            prof = {"source": prof.get("source"), "kernel_source_hash": prof.get("kernel_source_hash")}
    any_hit = not bool((scene.materials["spec_trans"] > 0.5).any())      # shadow rays may stop at the first hit
    wide = any_hit and args.exact_order                                   # 112-byte records only under RT_EXACT_ORDER
    items = {"k_extend": c["extend_rays"], "k_shade": c["extend_rays"], "k_connect": c["shadow_rays"]}
    ms = {"k_extend": tm["extend_ms"], "k_shade": tm["shade_ms"], "k_connect": tm["connect_ms"]}
    launches = {"k_extend": tm["extend_launches"], "k_shade": tm["shade_launches"], "k_connect": tm["connect_launches"]}
    algo = {"k_extend": algorithmic_bytes_extend(c["extend_rays"], c["extend_node_pops"], c["extend_tri_tests"]),
            "k_connect": algorithmic_bytes_connect(c["shadow_rays"], c["shadow_node_pops"], c["shadow_tri_tests"], wide),
            "k_shade": c["extend_rays"] * 380.0}
    clock_ghz = prof.get("shader_clock_ghz_measured")
    issue_peak = N_CU * SIMD_PER_CU * clock_ghz * 1e9 / ISSUE_CYCLES_PER_WAVE_INSTR if clock_ghz else None     # wave-instructions per second
    per_kernel = {}
    for kn in ("k_extend", "k_connect", "k_shade"):
        sec = ms[kn] / 1e3
        p = prof.get(kn, {})
        n_l = max(1, launches[kn])
        e = {"avg_launch_ms": ms[kn] / n_l, "launches": launches[kn], "items_per_launch": items[kn] / n_l,
             "item": "vertex" if kn == "k_shade" else "ray",
             "mitems_per_s": items[kn] / sec / 1e6 if sec > 0 else 0.0,
             "algorithmic_bytes_per_launch": algo[kn] / n_l,
             "achieved_algorithmic_GBs": algo[kn] / sec / 1e9 if sec > 0 else 0.0,
             # from the committed counter passes (null when they were taken from other kernel sources)
             "valu_wave_instr_per_s": None, "valu_issue_frac": None, "lanes_active_per_valu_instr": None, "lane_frac": None,
             "traffic_bytes_per_launch": None, "fabric_GBs": None, "hbm_frac_upper_bound": None}
        if p and sec > 0 and issue_peak:
            traffic = p["fabric_bytes_per_item"] * items[kn]            # bytes beyond L2 for the work this run did
            valu = p["valu_wave_instr_per_item"] * items[kn]
            e.update({"valu_wave_instr_per_item": p["valu_wave_instr_per_item"], "valu_wave_instr_per_s": valu / sec,
                      "valu_issue_frac": valu / sec / issue_peak,
                      "lanes_active_per_valu_instr": p["lanes_active_per_valu_instr"],
                      "lane_frac": valu / sec / issue_peak * p["lanes_active_per_valu_instr"] / 64.0,
                      "traffic_bytes_per_launch": traffic / n_l, "fabric_GBs": traffic / sec / 1e9,
                      "hbm_frac_upper_bound": traffic / sec / 1e9 / HBM_PEAK_GBS,
                      "valu_issue_frac_profiled": p.get("valu_issue_frac"),
                      "ta_busy_frac": p["ta_busy_frac"], "td_busy_frac": p["td_busy_frac"],
                      "td_stalled_on_cache_frac": p["td_stalled_on_cache_frac"], "l2_hit_rate": p["l2_hit_rate"],
                      "profiled_avg_launch_us": p.get("trace_avg_launch_us")})
        per_kernel[kn] = e
    ext = per_kernel["k_extend"]
    out = {}
    out["roofline"] = {
        "kernel": "k_extend",
        # The roof that binds is vector-ALU issue (a divergent gather of 64-byte records plus fp32 box / triangle tests: no
        # MFMA work, and HBM far from its limit), so `achieved` / `peak` are vector-ALU wave-instructions per second.
        "bound": "valu_issue", "unit": "wave-instructions/s",
        "achieved": ext["valu_wave_instr_per_s"], "peak": issue_peak, "frac": ext["valu_issue_frac"],
        "peak_note": f"{N_CU} CUs x {SIMD_PER_CU} SIMDs x measured shader clock / {ISSUE_CYCLES_PER_WAVE_INSTR} cycles per wave64 instruction",
        "shader_clock_ghz_measured": clock_ghz,
        "lanes_active_per_valu_instr": ext["lanes_active_per_valu_instr"], "lane_frac": ext["lane_frac"],
        "traffic": ext["traffic_bytes_per_launch"],
        # secondary: the memory side. Bytes that left L2 (Infinity-Cache hits included) per second bound HBM traffic from above.
        "hbm_upper_bound": {"achieved_GBs": ext["fabric_GBs"], "peak_GBs": HBM_PEAK_GBS, "frac": ext["hbm_frac_upper_bound"]},
        "achieved_algorithmic_GBs": ext["achieved_algorithmic_GBs"],
        "achieved_algorithmic_note": "bytes a one-thread-per-ray loop would request (64 B per interior node visit, 48 B per triangle test, "
                                     "80 B per ray) per second; most are served from LDS-resident records and L2, so this figure "
                                     "exceeds the HBM peak by construction and is reported for continuity only",
        "bytes_per_launch": ext["algorithmic_bytes_per_launch"], "avg_launch_ms": ext["avg_launch_ms"], "launches": ext["launches"],
        "samples": k,
        "counters_from": f"profiles/{pfile.name}" if prof else None, "counters_source_run": prof.get("source"),
        "counters_stale": stale, "counters_launch_frames": prof.get("launch_frames"),
        "live_in_this_run": ["avg_launch_ms", "launches", "items_per_launch", "algorithmic bytes", "node_pops_per_ray", "tri_tests_per_ray"],
        "from_committed_profile": ["valu_wave_instr_per_item", "lanes_active_per_valu_instr", "fabric bytes per item", "shader clock",
                                   "ta / td busy", "l2_hit_rate"],
        "tree": "binary, reference order (RT_EXACT_ORDER)" if args.exact_order else "compressed 4-wide",
        "node_pops_per_ray": c["extend_node_pops"] / max(1, c["extend_rays"]),
        "tri_tests_per_ray": c["extend_tri_tests"] / max(1, c["extend_rays"]),
        "extend_mrays_per_s": ext["mitems_per_s"],
        "occupancy": "8 waves/SIMD (64 VGPRs; two 1024-thread workgroups per CU, 80 KiB LDS each)",
    }
    out["roofline_kernels"] = per_kernel
    out["kernel_ms_per_sample"] = {s_: tm[f"{s_}_ms"] / k for s_ in ("raygen", "extend", "shade", "connect", "accumulate")}
    out["kernel_ms_per_sample"]["total"] = tm["total_ms"] / k
    out["shade_arithmetic"] = "exact (RT_EXACT_ARITH)" if (args.exact_arith or args.exact_order) else "fast (hardware rcp/rsq/sqrt, FMA contraction)"
    out["connect"] = {"mrays_per_s": per_kernel["k_connect"]["mitems_per_s"],
                      "tree": ("4-wide" if wide else ("binary" if args.exact_order else "compressed 4-wide")) + (", any hit" if any_hit else ", closest hit (glass rule)"),
                      "node_pops_per_ray": c["shadow_node_pops"] / max(1, c["shadow_rays"]),
                      "tri_tests_per_ray": c["shadow_tri_tests"] / max(1, c["shadow_rays"])}
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    dist = None
    # Library banners (RCCL prints one on stdout) must not mix with the single JSON line: park fd 1 on stderr until the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # RT_BENCH_REHEARSE_ONE_GPU=1: every rank uses GPU 0 and the collectives go through gloo, so that the N > 1 code path
    # (rank environment, frame slices, barrier, max-over-ranks clock, the reduce, one JSON line from rank 0) can be run end to
    # end on a box with a single GPU. The line it prints carries "rehearsal" and is not a measurement.
    rehearse = os.environ.get("RT_BENCH_REHEARSE_ONE_GPU") == "1"
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:   # launched by torch.distributed.run (any world size)
        import torch
        import torch.distributed as dist
        if rehearse:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    elif n_gpus > 1:
        # Started without a launcher: start one rank per GPU ourselves. This process has made no GPU call yet (argument
        # parsing only), and it never makes one: the ranks are fresh child processes of torch.distributed.run, this
        # process forwards their output (rank 0's JSON line) and leaves with the launcher's exit code.
        os.dup2(real_stdout, 1)
        sys.exit(launch_ranks(n_gpus))

    import ray_tracer_amd as rt
    from ray_tracer_amd import api, scenes

    def note(msg):                      # progress on stderr: a long silent run is taken to be hung
        if rank == 0:
            print(f"[bench +{time.time() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)
    t_start = time.time()

    base_flags = (api.RT_EXACT_ORDER if args.exact_order else 0) | (api.RT_EXACT_ARITH if args.exact_arith else 0)
    gen = scenes.GENERATORS[args.scene]
    kw = {"width": args.width, "height": args.height}
    if args.scene != "cornell":
        kw["scale"] = args.scale
    t0 = time.time()
    scene = gen(**kw).finalize(spatial=args.bvh == "spatial")
    t_scene = time.time() - t0

    if args.eighth_child:
        # The 32-CU leg, in a process of its own: render streams confined to 32 of the 256 compute units (RT_HIP_CUS in
        # rt_api.hip: a CU-masked stream, 4 CUs on each XCD, persistent grids sized for 32).
        os.environ["RT_HIP_CUS"] = "32"
        r8 = rt.HipRenderer.create(scene, max_bounces=args.bounces, device=local_rank, frames_per_pass=args.frames_per_pass, flags=base_flags)
        note("32-CU child: renderer created")
        r8.render(args.warmup)
        r8.frame_complete()
        note("32-CU child: warm-up done")
        r8.reset_accumulator()
        r8.reset_counters()
        r8.frame_complete()
        t8 = time.perf_counter()
        r8.dispatch_compute(args.steps)
        r8.frame_complete()
        e8 = time.perf_counter() - t8
        c8 = r8.counters()
        note(f"32-CU child: {args.steps} steps in {e8 * 1e3:.1f} ms")
        r8.destroy()
        note("32-CU child: renderer destroyed")
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps({"compute_units": 32, "how": "CU-masked HIP stream, 4 CUs on each of the 8 XCDs; L2, Infinity Cache and HBM whole",
                          "mrays_per_sec": (c8["extend_rays"] + c8["shadow_rays"]) / e8 / 1e6, "samples_per_sec": args.steps / e8,
                          "steps": args.steps, "warmup": args.warmup}), flush=True)
        return      # a normal interpreter exit, the HIP runtime's exit handler included (the parent notices if it does not return)

    if args.profile_context == "timing":
        # Profiling runs only: the single-stream context whose launches are bracketed by HIP events, two passes of --steps
        # frames, nothing else on the GPU, so that every traced or counted launch is one of these. Prints what the passes did.
        rtm = rt.HipRenderer.create(scene, max_bounces=args.bounces, device=local_rank, flags=api.RT_KERNEL_TIMING | base_flags,
                                    frames_per_pass=args.frames_per_pass)
        rtm.render(args.steps)
        rtm.reset_accumulator()
        rtm.render(args.steps)
        tm = rtm.last_timer_perf()
        cnt = rtm.counters()
        rtm.destroy()
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps({"profile_context": "timing", "passes": 2, "steps": args.steps, "extend_rays": cnt["extend_rays"], "shadow_rays": cnt["shadow_rays"],
                          "workload": f"{scene.name} {args.width}x{args.height}, {args.bounces} bounces", "kernel_source_hash": kernel_source_hash(),
                          "last_pass_ms": {s_: tm[f"{s_}_ms"] for s_ in ("raygen", "extend", "shade", "connect", "accumulate", "total")},
                          "last_pass_launches": {s_: tm[f"{s_}_launches"] for s_ in ("extend", "shade", "connect")}}), flush=True)
        return

    # BVH build-time baseline (stand-in for the reference's criterion bench, benches/bvh.rs:11-40: clone the triangles,
    # build, 10 samples): median wall time of the library's builders on this scene (binned SAH, binned SAH with
    # spatial splits, and the twin of the reference's plain `Bvh`), rank 0 only.
    bvh_bench = None
    if rank == 0 and not args.no_cpu_baseline:
        ts, ts_spatial, ts_sweep = [], [], []
        n_in = 0
        for i in range(10):
            clone = gen(**kw)
            n_in = len(clone.triangles)
            t1 = time.perf_counter()
            clone.build_bvh()
            ts.append((time.perf_counter() - t1) * 1e3)
            if i < 5:
                clone = gen(**kw)
                t1 = time.perf_counter()
                clone.build_bvh(spatial=True)
                ts_spatial.append((time.perf_counter() - t1) * 1e3)
                clone = gen(**kw)
                t1 = time.perf_counter()
                clone.build_bvh(sweep=True)
                ts_sweep.append((time.perf_counter() - t1) * 1e3)
        bvh_bench = {"builder": "binned SAH, object splits (BinSah::default parameters)", "triangles": int(n_in),
                     "median_ms": float(np.median(ts)), "min_ms": float(min(ts)), "samples": 10, "threads": min(64, os.cpu_count() or 1),
                     "mtris_per_s": n_in / float(np.median(ts)) / 1e3,
                     "spatial_splits_median_ms": float(np.median(ts_spatial)), "spatial_splits_samples": 5,
                     # the other half of benches/bvh.rs:20-37: the reference's plain `Bvh` (8 candidate planes per axis,
                     # one thread, bvh.rs:183-331)
                     "sweep_bvh_median_ms": float(np.median(ts_sweep)), "sweep_bvh_samples": 5, "sweep_bvh_threads": 1}

    note("scene and build-time baseline done; creating the renderer")
    r = rt.HipRenderer.create(scene, max_bounces=args.bounces, device=local_rank, frames_per_pass=args.frames_per_pass, flags=base_flags)
    from ray_tracer_amd.distributed import frame_slice, tile_slice
    if args.partition == "tiles":
        r.set_tile_slice(*tile_slice(rank, world))
    else:
        r.set_frame_slice(*frame_slice(rank, world))
    frames_total = args.steps if args.partition == "tiles" else world * args.steps     # distinct samples per pixel in the reduced image
    npix = args.width * args.height

    red = None
    if dist is not None:
        import torch
        from ray_tracer_amd.distributed import accumulator_tensor, reduce_accumulators
        red = accumulator_tensor(r, torch.device("cuda", local_rank))   # zero-copy view of the library's fp32 sum buffer

    def barrier():
        r.frame_complete()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warm-up, then the timed region: exactly K steps (+ the accumulation reduce when N > 1)
    r.render(args.warmup)
    if dist is not None:
        # warm-up of the collective as well: the first all-reduce of a buffer of this size sets up RCCL's channels
        import torch
        r.frame_complete()
        reduce_accumulators(red)
        torch.cuda.synchronize()
    r.reset_accumulator()
    r.reset_counters()
    barrier()
    t0 = time.perf_counter()
    r.dispatch_compute(args.steps)
    r.frame_complete()
    if dist is not None:
        import torch
        reduce_accumulators(red)          # one RCCL all-reduce of W*H*4 floats
        torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    cnt = r.counters()
    rays_local = cnt["extend_rays"] + cnt["shadow_rays"]

    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([float(rays_local), float(cnt["extend_rays"]), float(cnt["shadow_rays"])], dtype=torch.float64, device="cuda")
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays_total, ext_total, sh_total = (float(x) for x in rr.tolist())
    else:
        rays_total, ext_total, sh_total = float(rays_local), float(cnt["extend_rays"]), float(cnt["shadow_rays"])

    result = {
        "metric": "Mrays/s (extend+shadow), 1080p progressive path tracing",
        "value": rays_total / elapsed / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if args.partition == "tiles" and world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{scene.name} {args.width}x{args.height}, {args.bounces} bounces, NEE+MIS, 1 spp per step per GPU",
                   "triangles": int(len(scene.triangles)), "bvh_nodes": int(len(scene.bvh)), "emissive_tris": int(len(scene.emissive_tris)),
                   "bvh": f"binned SAH, {args.bvh} splits", "duplicated_triangles": int(scene.bvh_stats.get("duplicated_refs", 0)),
                   "parallelism": (f"tile-slice x{world}" if args.partition == "tiles" else f"sample-slice x{world}") if world > 1 else "single GPU",
                   "frames_per_pass": int(args.frames_per_pass)},
        "samples_per_sec": frames_total / elapsed,
        "samples_per_sec_vs_rtx3080_readme_5_to_10": [frames_total / elapsed / 10.0, frames_total / elapsed / 5.0],
        "value_div_8": rays_total / elapsed / 1e6 / 8.0,
        "extend_rays": ext_total, "shadow_rays": sh_total,
        "mpaths_per_sec": frames_total * npix / elapsed / 1e6,
        "scene_build_s": t_scene, "bvh_build_ms": scene.bvh_stats.get("build_ms"), "bvh_build_bench": bvh_bench,
    }
    if dist is not None:
        # what the collective layer itself saw: backend "nccl" is RCCL on ROCm
        result["collective"] = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "launched_as": "torch.distributed.run, one rank per GPU",
                                "partition": args.partition}
        if world != n_gpus:
            result["collective"]["warning"] = f"--gpus {n_gpus} but the launcher started {world} ranks"
    if rehearse and world > 1:
        result["rehearsal"] = f"{world} ranks sharing GPU 0, collectives over gloo: exercises the N > 1 code path, not a measurement"

    # ---- per-kernel resource fractions, rank 0 only. Measured live in this run: kernel times (HIP events around every launch,
    # on the stream the kernels run on, in a context that runs one pass of the size the timed region used) and rays / vertices /
    # node visits per launch (the library's counters). NOT measurable from inside this process: hardware counters. The
    # per-ray instruction counts, lane utilisation, bytes beyond L2 and the shader clock come from the committed rocprofv3
    # passes of this same workload at this same launch size (tools/profile_r3.sh -> tools/roofline_counters.py ->
    # profiles/r3_roofline_counters*.json), and are used only while the hash of the kernel sources stored in that file equals
    # the hash of the sources this library was built from; otherwise the line says "counters_stale" and carries nulls.
    note(f"timed region done: {elapsed * 1e3 / args.steps:.2f} ms per step")
    if args.profile_context == "overlapped":
        args.no_roofline = True
    if rank == 0 and not args.no_roofline:
        c, tm, k = measure_kernels(rt, api, scene, args, local_rank, base_flags)
        result.update(roofline_blocks(args, scene, c, tm, k))

    # ---- "1/8 MI355X", measured: the same workload with every stream of the context confined to 32 of the 256 compute units.
    # Caches and HBM are not partitioned, so this bounds a true one-eighth partition from above; `value_div_8` is the plain
    # division of the whole-chip figure. Runs as a child process after the timed region, so that the whole-chip contexts of
    # this process and the confined one never share a HIP runtime instance.
    if rank == 0 and world == 1 and not args.no_eighth:
        note("32-CU leg (child process)")
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--eighth-child", "--steps", str(max(2, args.steps // 4)),
               "--warmup", str(max(1, args.warmup // 4)), "--width", str(args.width), "--height", str(args.height),
               "--bounces", str(args.bounces), "--scene", args.scene, "--scale", str(args.scale),
               "--frames-per-pass", str(args.frames_per_pass), "--bvh", args.bvh] + (["--exact-order"] if args.exact_order else []) + (["--exact-arith"] if args.exact_arith else [])
        try:
            env = {k_: v_ for k_, v_ in os.environ.items()
                   if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE")}
            child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
            try:
                out, _ = child.communicate(timeout=150)
                how = f"returned {child.returncode}"
            except subprocess.TimeoutExpired:
                child.kill()
                out, _ = child.communicate()
                how = "did not exit within 150 s and was killed"
            e8 = json.loads(out.strip().splitlines()[-1])
            e8["child_process_exit"] = how
            e8["vs_value_div_8"] = e8["mrays_per_sec"] / result["value_div_8"]
            result["one_eighth_measured"] = e8
        except Exception as ex:   # the leg is a side measurement: report its failure, keep the bench line
            result["one_eighth_measured"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    # ---- CPU baseline: the oracle (a restatement of the reference shader) on a bounded sample of the same workload
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("CPU baseline")
        from oracle import pyoracle as po
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        o = po.OracleScene(scene)
        sw, sh = max(16, args.width // 4), max(9, args.height // 4)
        o.set_resolution(sw, sh)
        t0 = time.perf_counter()
        frames = 0
        crays = 0
        while True:
            _, _, cc = o.render(1, args.bounces, first=frames, threads=threads)
            crays += cc["rays"] + cc["shadow_rays"]
            frames += 1
            el = time.perf_counter() - t0
            if el > 12.0 or frames >= 512:
                break
        result["cpu_baseline"] = {"value": crays / el / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                                  "sample": f"{scene.name} at {sw}x{sh} (1/16 of the pixels), {frames} spp, {args.bounces} bounces, {el:.1f} s",
                                  "samples_per_sec_at_full_res": frames / el / 16.0}

    r.destroy()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
