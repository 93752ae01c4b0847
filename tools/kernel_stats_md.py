#!/usr/bin/env python3
"""rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv -> the markdown table kept under
profiles/.  Usage: tools/kernel_stats_md.py <dir with *_kernel_stats.csv> [--steps N]"""
import argparse, csv, glob, re

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:90]

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--steps", type=int, default=0, help="passes of the hot path in the trace (adds a per-step column)")
a = ap.parse_args()
f = glob.glob(a.dir + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("| kernel | calls | total_us | avg_us | % |" + (" ms/step |" if a.steps else ""))
print("|---|---:|---:|---:|---:|" + ("---:|" if a.steps else ""))
tot = 0.0
for r in rows:
    t = float(r["TotalDurationNs"]) / 1e3
    tot += t
    if float(r["Percentage"]) < 0.01:
        continue
    line = f"| `{short(r['Name'])}` | {r['Calls']} | {t:.1f} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |"
    if a.steps:
        line += f" {t / 1e3 / a.steps:.3f} |"
    print(line)
print(f"\ntotal kernel time {tot / 1e3:.2f} ms" + (f" = {tot / 1e3 / a.steps:.2f} ms/step over {a.steps} passes" if a.steps else ""))
