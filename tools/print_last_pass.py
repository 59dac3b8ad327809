"""Prints the launches of the last pass in a rocprofv3 kernel trace (start relative to the pass's first launch, duration, name).
usage: print_last_pass.py <kernel_trace.csv> [launches, default 28]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 28
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_extend", "k_connect", "k_shade", "k_raygen", "k_accumulate", "k_finish", "k_path_tail"))]
last = sel[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("rtdev::")[-1].split("(")[0].replace("fast_arith::", "")[:28]
    print("%-28s start %8.1f us  end %8.1f us  dur %7.1f us" % (name, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
