#!/usr/bin/env python3
"""Condenses one tools/profile_r3.sh run (rocprofv3 kernel traces + three separate counter passes + a clock pass of bench.py's
single-stream timing context at the launch size the driver times) into per-kernel, per-work-item figures that bench.py scales
to whatever launch it measures, together with the hash of the kernel sources they were taken from:

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


def measured_clock(out):
    """Effective shader clock: GRBM_GUI_ACTIVE cycles (summed over the 8 XCDs) / traced duration, over the traversal kernels of
    the `clock` run (kernel trace and the one counter collected together)."""
    tr = glob.glob(os.path.join(out, "clock", "**", "*kernel_trace.csv"), recursive=True)
    cc = glob.glob(os.path.join(out, "clock", "**", "*counter_collection.csv"), recursive=True)
    if not tr or not cc:
        return None, {}
    dur = {r["Dispatch_Id"]: (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(tr[0]))}
    acc = defaultdict(lambda: [0.0, 0.0])
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
            continue
        name, ns = dur[r["Dispatch_Id"]]
        k = short(name)
        if k:
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += ns
    per = {k: c / 8.0 / ns for k, (c, ns) in acc.items() if ns > 0}
    tot_c = sum(c for c, _ in acc.values())
    tot_ns = sum(ns for _, ns in acc.values())
    return (tot_c / 8.0 / tot_ns if tot_ns > 0 else None), per


def main(out, dst):
    bench = json.load(open(os.path.join(out, "bench_pmc_a.json")))
    assert bench.get("profile_context") == "timing", "counter passes must run bench.py --profile-context timing"
    steps = bench["steps"]
    # every counted pass has the same size; the counters of the context cover both passes
    items = {"k_extend": bench["extend_rays"], "k_shade": bench["extend_rays"], "k_connect": bench["shadow_rays"]}
    a, da = counters(out, "pmc_a")
    b, db = counters(out, "pmc_b")
    c, dc = counters(out, "pmc_c")
    trace = {}
    for f in glob.glob(os.path.join(out, "trace_timing", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k:
                trace.setdefault(k, []).append((int(r["Calls"]), float(r["AverageNs"]) / 1e3))
    clock, clock_per_kernel = measured_clock(out)
    res = {"source": os.path.basename(out.rstrip("/")), "bench_args": f"--profile-context timing --steps {steps}", "launch_frames": steps,
           "workload": bench["workload"], "kernel_source_hash": bench.get("kernel_source_hash"),
           "shader_clock_ghz_measured": clock, "shader_clock_ghz_per_kernel": clock_per_kernel,
           "note": "per work item, single-stream context (kernels run alone); fabric bytes include Infinity-Cache hits (upper bound on HBM "
                   "bytes); valu_issue_frac = vector-ALU wave-instructions per CU-cycle / 2 (4 SIMDs, 2 cycles per wave64 instruction)"}
    for k in KERNELS:
        if k not in da or not items[k]:        # a scene without lights launches no k_connect
            continue
        n = items[k]
        cyc = c[k]["GRBM_GUI_ACTIVE"] / 8.0                     # summed over the pass's launches, per XCD
        per_cu_cycle = a[k]["SQ_INSTS_VALU"] / (N_CU * cyc)
        lanes = a[k]["SQ_THREAD_CYCLES_VALU"] / max(1.0, a[k]["SQ_ACTIVE_INST_VALU"])
        res[k] = {
            "items_counted": n, "launches_counted": da[k],
            "fabric_bytes_per_item": (2.0 * a[k]["FETCH_SIZE"] + b[k]["WRITE_SIZE"]) * 1024.0 / n,
            "fetch_bytes_per_item": 2.0 * a[k]["FETCH_SIZE"] * 1024.0 / n, "write_bytes_per_item": b[k]["WRITE_SIZE"] * 1024.0 / n,
            "valu_wave_instr_per_item": a[k]["SQ_INSTS_VALU"] / n,
            "valu_wave_instr_per_cu_cycle": per_cu_cycle,
            "valu_issue_frac": per_cu_cycle / 2.0,
            "lanes_active_per_valu_instr": lanes,
            "lane_frac": per_cu_cycle / 2.0 * lanes / 64.0,
            "wait_any_frac_of_wave_cycles": a[k]["SQ_WAIT_ANY"] / max(1.0, a[k]["SQ_WAVE_CYCLES"]),
            "valu_active_frac_of_wave_cycles": a[k]["SQ_ACTIVE_INST_VALU"] / max(1.0, a[k]["SQ_WAVE_CYCLES"]),
            "ta_busy_frac": c[k]["TA_TA_BUSY_sum"] / (N_CU * cyc), "td_busy_frac": c[k]["TD_TD_BUSY_sum"] / (N_CU * cyc),
            "td_stalled_on_cache_frac": c[k]["TD_TC_STALL_sum"] / (N_CU * cyc),
            "l2_hit_rate": b[k]["TCC_HIT_sum"] / max(1.0, b[k]["TCC_HIT_sum"] + b[k]["TCC_MISS_sum"]),
            "lds_instr_per_item": b[k]["SQ_INSTS_LDS"] / n, "salu_instr_per_item": b[k]["SQ_INSTS_SALU"] / n,
            "gpu_cycles_per_item": cyc / n,
            "trace_avg_launch_us": max(trace[k], key=lambda x: x[0])[1] if k in trace else None,
        }
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
