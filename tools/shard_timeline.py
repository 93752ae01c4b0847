#!/usr/bin/env python3
"""Timeline of the LAST pass in a rocprofv3 --kernel-trace of tools/ragged_bench.py (MODE=ragged N=3): kernels between the last two
gaps > 300 us, with stream-overlap accounting: start, duration, kernel, and at the end busy time per kernel name."""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last pass: from the last embed_tokens launch of a pass... find starts of passes = launches of embed_tokens_kernel
starts = [i for i, r in enumerate(rows) if "embed_tokens" in r["Kernel_Name"]]
per_pass = int(sys.argv[2]) if len(sys.argv) > 2 else 1        # embed_tokens launches per pass
a = starts[-per_pass]
t0 = int(rows[a]["Start_Timestamp"])
agg = collections.OrderedDict()
end_max = t0
for r in rows[a:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+|\w+elementwise\w*)(<[^>]*>)?", r["Kernel_Name"])
    nm = m.group(0)[:70] if m else r["Kernel_Name"][:70]
    print("%9.1f us  %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), nm))
    k = agg.setdefault(nm, [0, 0.0]); k[0] += 1; k[1] += (e - s) / 1e3
    end_max = max(end_max, e)
print("span %.1f us" % ((end_max - t0) / 1e3))
for nm, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%8.1f us  x%-3d %s" % (t, n, nm))
