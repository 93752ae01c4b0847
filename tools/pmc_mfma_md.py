#!/usr/bin/env python3
"""Per-kernel MFMA-pipe occupancy and wave-state split from a rocprofv3 --pmc pass
(SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES [SQ_INSTS_VALU_MFMA_MOPS_F32]).
MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x clock x 1024 SIMDs); the clock is taken as 2.4 GHz,
so the fraction is against the nominal peak (a power-limited clock shows up as a lower fraction).  Markdown to stdout."""
import csv, glob, re, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
GHZ, SIMDS = 2.4, 1024
agg = collections.OrderedDict()
seen = set()
for r in csv.DictReader(open(f)):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
    if not m or "at::native" in r["Kernel_Name"]:
        continue
    k = m.group(0)
    a = agg.setdefault(k, collections.defaultdict(float))
    a[r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"],)
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"]))
        a["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a["_n"] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1]["_ns"])
tot = sum(v["_ns"] for _, v in rows)
print("| kernel | launches | time share | MFMA pipe busy (of 1024 SIMDs x 2.4 GHz) | waves: issuing / issue-stalled / parked |")
print("|---|---:|---:|---:|---|")
for k, v in rows:
    if v["_ns"] < 0.002 * tot:
        continue
    wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (v["_ns"] * GHZ * SIMDS)
    print("| `%s` | %d | %.1f %% | %.1f %% | %.0f / %.0f / %.0f %% |" % (
        k, v["_n"], 100 * v["_ns"] / tot, 100 * busy, 100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * v.get("SQ_WAIT_ANY", 0) / wc))
