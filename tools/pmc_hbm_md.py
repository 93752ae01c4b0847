#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per dispatch, read at the L2's
fabric side: Infinity-Cache hits are counted).  FETCH_SIZE is multiplied by the factor calibrated on known-byte-count streams
(profiles/pmc_traffic.json: x2.0 on gfx950 for 4-B and 16-B per lane alike; WRITE_SIZE x1.0).
Usage: pmc_hbm_md.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>"""
import csv, glob, json, os, re, sys, collections
try:
    _cal = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")))["calibration"]["dword_4B_per_lane"]
    FF, WF = float(_cal["fetch_factor"]), float(_cal["write_factor"])
except Exception:
    FF, WF = 2.0, 1.0
def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.OrderedDict(); seen = set()
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
        if not m or "at::native" in r["Kernel_Name"] or r["Counter_Name"] != counter: continue
        a = agg.setdefault(m.group(0), [0, 0.0, 0.0])
        a[1] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); a[0] += 1; a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg
fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print(f"| kernel | launches | fetch MB / launch (x{FF:g}) | write MB / launch | avg µs | (fetch + write) / time |")
print("|---|---:|---:|---:|---:|---:|")
rows = sorted(fe.items(), key=lambda kv: -kv[1][2])
for k, (n, kb, ns) in rows:
    if ns < 0.002 * sum(v[2] for v in fe.values()): continue
    w = wr.get(k, [n, 0.0, ns])
    fmb, wmb, us = FF * kb * 1024 / n / 1e6, WF * w[1] * 1024 / max(w[0], 1) / 1e6, ns / n / 1e3
    print("| `%s` | %d | %.1f | %.1f | %.1f | %.2f TB/s |" % (k, n, fmb, wmb, us, (fmb + wmb) / us))
