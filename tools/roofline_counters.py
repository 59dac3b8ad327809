#!/usr/bin/env python3
"""Condenses one tools/profile_r2.sh run (rocprofv3 kernel trace + three separate counter passes of bench.py at the launch
size the driver times) into per-kernel, per-work-item figures that bench.py scales to whatever launch it measures:

    fabric bytes per item   = (FETCH_SIZE x 2 [gfx950 counts 128-byte requests as 64] + WRITE_SIZE) x 1024 / items
                              bytes that left L2 towards Infinity Cache / HBM: an UPPER bound on HBM traffic
    VALU wave-instructions per item, VALU wave-instructions per CU-cycle (cycles = GRBM_GUI_ACTIVE / 8 XCDs)
    TA / TD busy fractions  = TA_TA_BUSY_sum, TD_TD_BUSY_sum / (CUs x cycles);  TD stalled on cache data likewise
    L2 hit rate, lanes active per VALU instruction

Items: rays for k_extend / k_connect, path vertices (= extend rays) for k_shade. Usage: roofline_counters.py <prof dir> <out json>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

N_CU = 256
KERNELS = ("k_extend", "k_connect", "k_shade")


def short(name):
    for k in KERNELS:
        if k in name.split("(")[0]:
            return k
    return None


def counters(out, sub):
    agg = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in disp.items()}


def main(out, dst):
    bench = json.load(open(os.path.join(out, "bench_pmc_a.json")))
    steps, warm = bench["steps"], bench["warmup"]
    # every counted call (warm-up + timed) has the same size, so items per call x calls = items the counters saw
    calls = 2 if warm == steps else None
    assert calls, "profile with --warmup equal to --steps so that every counted launch has the timed size"
    items = {"k_extend": bench["extend_rays"] * calls, "k_shade": bench["extend_rays"] * calls, "k_connect": bench["shadow_rays"] * calls}
    a, da = counters(out, "pmc_a")
    b, db = counters(out, "pmc_b")
    c, dc = counters(out, "pmc_c")
    trace = {}
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k:
                trace.setdefault(k, []).append((int(r["Calls"]), float(r["AverageNs"]) / 1e3))
    res = {"source": os.path.basename(out.rstrip("/")), "bench_args": f"--steps {steps} --warmup {warm}", "launch_frames": steps,
           "workload": bench["config"]["workload"], "note": "per work item; fabric bytes include Infinity-Cache hits (upper bound on HBM bytes)"}
    for k in KERNELS:
        if k not in da or not items[k]:        # a scene without lights launches no k_connect
            continue
        n = items[k]
        cyc = c[k]["GRBM_GUI_ACTIVE"] / 8.0                     # summed over the pass's launches, per XCD
        res[k] = {
            "items_counted": n, "launches_counted": da[k],
            "fabric_bytes_per_item": (2.0 * a[k]["FETCH_SIZE"] + b[k]["WRITE_SIZE"]) * 1024.0 / n,
            "fetch_bytes_per_item": 2.0 * a[k]["FETCH_SIZE"] * 1024.0 / n, "write_bytes_per_item": b[k]["WRITE_SIZE"] * 1024.0 / n,
            "valu_wave_instr_per_item": a[k]["SQ_INSTS_VALU"] / n,
            "valu_wave_instr_per_cu_cycle": a[k]["SQ_INSTS_VALU"] / (N_CU * cyc),
            "lanes_active_per_valu_instr": a[k]["SQ_THREAD_CYCLES_VALU"] / max(1.0, a[k]["SQ_ACTIVE_INST_VALU"]),
            "ta_busy_frac": c[k]["TA_TA_BUSY_sum"] / (N_CU * cyc), "td_busy_frac": c[k]["TD_TD_BUSY_sum"] / (N_CU * cyc),
            "td_stalled_on_cache_frac": c[k]["TD_TC_STALL_sum"] / (N_CU * cyc),
            "l2_hit_rate": b[k]["TCC_HIT_sum"] / max(1.0, b[k]["TCC_HIT_sum"] + b[k]["TCC_MISS_sum"]),
            "lds_instr_per_item": b[k]["SQ_INSTS_LDS"] / n, "salu_instr_per_item": b[k]["SQ_INSTS_SALU"] / n,
            "gpu_cycles_per_item": cyc / n,
            # the instantiation the timed region runs has the most calls (the counting variant runs once, for the roofline block)
            "trace_avg_launch_us": max(trace[k], key=lambda x: x[0])[1] if k in trace else None,
        }
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
