#!/usr/bin/env python3
"""Per-kernel table of the LAST pass in a rocprofv3 kernel trace of tools/text_side_probe.py (in launch order)."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "embed_tokens" in n]
seg = rows[starts[-1]:]
t0 = int(seg[0]["Start_Timestamp"]); prev_end = t0
tot = gap = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
    nm = m.group(0) if m else r["Kernel_Name"][:50]
    wg = int(r["Workgroup_Size_X"]) or 1
    grid = "%dx%sx%s" % (int(r["Grid_Size_X"]) // wg, r["Grid_Size_Y"], r["Grid_Size_Z"])
    print("%8.1f us  +%5.1f gap  %6.1f us  %-12s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, grid, nm))
    tot += e - s; gap += max(0, s - prev_end); prev_end = e
print("kernels %d, busy %.1f us, gaps %.1f us, span %.1f us" % (len(seg), tot / 1e3, gap / 1e3, (prev_end - t0) / 1e3))
