#!/usr/bin/env python3
"""profiles/pmc_traffic.json from four rocprofv3 --pmc passes: bench (FETCH_SIZE), bench (WRITE_SIZE), tools/pmc_calib.py
(FETCH_SIZE), tools/pmc_calib.py (WRITE_SIZE).  The calibration kernels move a known byte count, which gives the factor
rocprofv3's KB figures must be multiplied by for 4-B-per-lane and 16-B-per-lane streams on this box; the dominant
kernel's traffic is reported raw and corrected.  Usage: pmc_traffic.py <bench_fetch> <bench_write> <calib_fetch> <calib_write> <commit>"""
import csv, glob, json, re, sys, collections


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg, seen = collections.OrderedDict(), set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if not m:
            continue
        a = agg.setdefault(m.group(1), [0, 0.0])
        a[1] += float(r["Counter_Value"])
        if (m.group(1), r["Dispatch_Id"]) not in seen:
            seen.add((m.group(1), r["Dispatch_Id"]))
            a[0] += 1
    return {k: v[1] / v[0] for k, v in agg.items()}      # KB per launch


bf, bw, cf, cw = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), load(sys.argv[3], "FETCH_SIZE"), load(sys.argv[4], "WRITE_SIZE")
KNOWN = 64 * 1024 * 1024 * 4
cal = {}
for name, key in (("dword_4B_per_lane", "transpose_kernel"), ("dwordx4_16B_per_lane", "vectorized_elementwise_kernel")):
    cal[name] = {"kernel": key, "known_bytes_each_way": KNOWN,
                 "fetch_reported_bytes": round(cf[key] * 1024), "write_reported_bytes": round(cw[key] * 1024),
                 "fetch_factor": round(KNOWN / (cf[key] * 1024), 3), "write_factor": round(KNOWN / (cw[key] * 1024), 3)}
out = {"_comment": "HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, KB x 1024), with the "
                   "calibration MI355X_MICROARCH.md §HBM prescribes: known-byte-count streams of both access widths measured in the same "
                   "session give the factor to multiply the reported figures by.  bench.py reports `bytes_per_launch` (corrected) as "
                   "roofline.traffic.", "commit": sys.argv[5], "calibration": cal}
for k in ("denoiser_persist_kernel", "resblock_fused_kernel"):
    if k in bf:
        # the persistent kernel reads cp / x with 4-B-per-lane loads and its weights with 16-B-per-lane loads: correct with the
        # 4-B factor (the larger share) and give the raw figures next to it
        f4, w4 = cal["dword_4B_per_lane"]["fetch_factor"], cal["dword_4B_per_lane"]["write_factor"]
        out[k] = {"B": 32, "T": 512, "fetch_kb_raw": round(bf[k], 1), "write_kb_raw": round(bw.get(k, 0.0), 1),
                  "bytes_per_launch_raw": round((bf[k] + bw.get(k, 0.0)) * 1024),
                  "bytes_per_launch": round((bf[k] * f4 + bw.get(k, 0.0) * w4) * 1024)}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
