#!/usr/bin/env python3
"""Fold one round's committed counter tables into profiles/pmc_traffic.json, the file bench.py copies `roofline.traffic`, `mfma_busy` and
`effective_clock_ghz` from (PMC counters cannot be sampled from inside the benchmarked process).  Every entry carries the commit it was
measured at.  Usage: pmc_update_json.py <entry> <kernel-name-substring> <hbm.md> <mfma.md> <clock.md> <commit> [note]"""
import json, re, sys
entry, sub, hbm, mfma, clock, commit = sys.argv[1:7]
note = sys.argv[7] if len(sys.argv) > 7 else ""


def row(path, sub):
    for line in open(path):
        if line.startswith("| `") and sub in line:
            return [c.strip() for c in line.strip().strip("|").split("|")]
    raise SystemExit(f"{path}: no row for {sub}")


pj = json.load(open("profiles/pmc_traffic.json"))
h, m, c = row(hbm, sub), row(mfma, sub), row(clock, sub)
e = pj.setdefault(entry, {})
e.update({"B": 32, "T": 512, "kernel": h[0].strip("`"),
          "fetch_mb_x2": float(h[2]), "write_mb": float(h[3]), "bytes_per_launch": round((float(h[2]) + float(h[3])) * 1e6),
          "mfma_busy_nominal": round(float(m[3].rstrip(" %")) / 100, 4),
          "waves_issuing_stalled_parked": m[4],
          "effective_clock_ghz": float(c[3].split()[0]),
          "mfma_busy_actual": round(float(c[5].rstrip(" %")) / 100, 4),
          "commit": commit, "sources": [hbm, mfma, clock]})
if note:
    e["note"] = note
json.dump(pj, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(e, indent=1))
