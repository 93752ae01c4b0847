#!/usr/bin/env python3
"""Launch-order timeline of the LAST bench step in a rocprofv3 --kernel-trace of bench.py (from the first persistent launch of the
previous step's end to the last persistent launch): start, gap to the previous kernel's end, duration, kernel."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "denoiser_persist" in r["Kernel_Name"]]
a, b = idx[-5], idx[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev = None
only = len(sys.argv) > 2
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+|\w+elementwise\w*)(<[^>]*>)?", r["Kernel_Name"])
    nm = m.group(0)[:60] if m else r["Kernel_Name"][:60]
    gap = (s - prev) / 1e3 if prev else 0
    if not only or (s - t0) / 1e3 > float(sys.argv[2]):
        print("%9.1f us +%6.1f gap %8.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, nm))
    prev = e
