#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel trace stats + counter passes) into one per-kernel text table."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    for k in ("k_extend", "k_connect", "k_shade", "k_compact_queues", "k_raygen", "k_accumulate", "k_finish_pass", "k_trace_rays", "k_tonemap"):
        if k in n:
            counted = k in ("k_extend", "k_connect", "k_trace_rays") and ("ILb1" in n or "<true" in name)
            return k + ("<count>" if counted else "")
    return n[-40:]


TRAFFIC = {}


def main(out):
    for f in sorted(glob.glob(os.path.join(out, "trace*", "**", "*kernel_stats.csv"), recursive=True)):
        print("== kernel trace stats:", os.path.relpath(f, out), "(trace/ = single stream, trace_overlapped/ = default two-lane run)")
        rows = list(csv.DictReader(open(f)))
        print(f"{'kernel':28s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        for r in rows:
            print(f"{short(r['Name']):28s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} "
                  f"{float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
    for p in ("pmc_a", "pmc_b", "pmc_c", "pmc_d", "pmc_e", "pmc_f"):
        files = glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            print("\n== counters:", os.path.relpath(f, out), "(sum over dispatches, per kernel)")
            agg = defaultdict(lambda: defaultdict(float))
            calls = defaultdict(set)
            meta = {}
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                calls[k].add(r["Dispatch_Id"])
                meta[k] = (r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"), r.get("SGPR_Count", "?"), r.get("LDS_Block_Size", "?"), r.get("Scratch_Size", "?"))
            for k in sorted(agg):
                print(f"{k}: dispatches={len(calls[k])} vgpr={meta[k][0]} agpr={meta[k][1]} sgpr={meta[k][2]} lds={meta[k][3]} scratch={meta[k][4]}")
                for c in sorted(agg[k]):
                    print(f"    {c:32s} {agg[k][c]:.6g}   per-dispatch {agg[k][c]/max(1,len(calls[k])):.6g}")
                a = agg[k]
                if "TCC_HIT_sum" in a and a["TCC_HIT_sum"] + a["TCC_MISS_sum"] > 0:
                    print(f"    -> L2 hit rate {a['TCC_HIT_sum']/(a['TCC_HIT_sum']+a['TCC_MISS_sum']):.4f}")
                    TRAFFIC.setdefault(k, {})["l2_hit_rate"] = a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"])
                if "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"] > 0:
                    print(f"    -> wait_any/wave_cycles {a.get('SQ_WAIT_ANY',0)/a['SQ_WAVE_CYCLES']:.3f}  valu_active/wave_cycles {a.get('SQ_ACTIVE_INST_VALU',0)/a['SQ_WAVE_CYCLES']:.3f}"
                          f"  mean waves in flight (wave_cycles/busy_cycles) {a['SQ_WAVE_CYCLES']/max(1,a.get('SQ_BUSY_CYCLES',1)):.2f}")
                if "SQ_THREAD_CYCLES_VALU" in a and a.get("SQ_ACTIVE_INST_VALU", 0) > 0:
                    TRAFFIC.setdefault(k, {})["active_lanes_per_valu_instruction"] = a["SQ_THREAD_CYCLES_VALU"] / a["SQ_ACTIVE_INST_VALU"]
                if "SQ_WAVES" in a:
                    TRAFFIC.setdefault(k, {})["lds_bytes_per_workgroup"] = meta[k][3]
                if "FETCH_SIZE" in a:
                    TRAFFIC.setdefault(k, {})["fetch_kb_per_launch_raw"] = a["FETCH_SIZE"] / max(1, len(calls[k]))
                    TRAFFIC[k]["launches_fetch_pass"] = len(calls[k])
                    print(f"    -> FETCH_SIZE {a['FETCH_SIZE']:.6g} KB reported; x2 gfx950 correction for wide reads = {2*a['FETCH_SIZE']*1024/1e9:.4f} GB total, {2*a['FETCH_SIZE']*1024/max(1,len(calls[k]))/1e6:.3f} MB/launch")
                if "WRITE_SIZE" in a:
                    TRAFFIC.setdefault(k, {})["write_kb_per_launch"] = a["WRITE_SIZE"] / max(1, len(calls[k]))
                    print(f"    -> WRITE_SIZE {a['WRITE_SIZE']*1024/1e9:.4f} GB total, {a['WRITE_SIZE']*1024/max(1,len(calls[k]))/1e6:.3f} MB/launch")


if __name__ == "__main__":
    main(sys.argv[1])
    import json
    for k, v in TRAFFIC.items():
        # gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> double it (calibrated for wide streaming reads only)
        v["hbm_bytes_per_launch"] = (2.0 * v.get("fetch_kb_per_launch_raw", 0.0) + v.get("write_kb_per_launch", 0.0)) * 1024.0
    json.dump(TRAFFIC, open(os.path.join(sys.argv[1], "traffic.json"), "w"), indent=1)
