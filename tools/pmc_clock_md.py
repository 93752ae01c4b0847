#!/usr/bin/env python3
"""Effective shader clock per kernel from a rocprofv3 --pmc pass with GRBM_GUI_ACTIVE (+ SQ_VALU_MFMA_BUSY_CYCLES when
present): clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (the counter is summed over the 8 XCDs; MI355X_MICROARCH.md,
DVFS give-back).  MFMA busy is then given twice: against the nominal 2.4 GHz and against the cycles the kernel actually
had.  Kernels shorter than 100 us are left out (the counter's window is not the kernel's)."""
XCDS = 8
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg, seen = collections.OrderedDict(), set()
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
    if not m or "at::native" in r["Kernel_Name"]:
        continue
    k = m.group(0)
    a = agg.setdefault(k, collections.defaultdict(float))
    a[r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"]))
        a["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a["_n"] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1]["_ns"])
tot = sum(v["_ns"] for _, v in rows)
print("| kernel | launches | avg µs | effective clock (GRBM_GUI_ACTIVE / time) | MFMA busy vs 2.4 GHz nominal | MFMA busy vs actual cycles |")
print("|---|---:|---:|---:|---:|---:|")
for k, v in rows:
    if v["_ns"] < 0.002 * tot or v["_ns"] / v["_n"] < 1e5:
        continue
    cyc = v.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
    ghz = cyc / v["_ns"]
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    print("| `%s` | %d | %.1f | %.2f GHz | %s | %s |" % (
        k, v["_n"], v["_ns"] / v["_n"] / 1e3, ghz,
        "%.1f %%" % (100 * busy / (v["_ns"] * 2.4 * 1024)) if busy else "–",
        "%.1f %%" % (100 * busy / max(cyc * 1024, 1.0)) if busy else "–"))
