#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace into a small markdown/CSV table —
the equivalent of `--stats` kernel_stats.csv, which this rocprofv3 build only emits into the .db.
Usage: tools/rocpd_summary.py <results.db> [--steps N] > profiles/rNN_kernel_stats.md"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=0, help="bench steps in the trace (adds a per-step column)")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("| kernel | calls | total_us | avg_us | % |" + (" ms/step |" if a.steps else ""))
    print("|---|---:|---:|---:|---:|" + ("---:|" if a.steps else ""))
    for name, calls, tot, avg, pct in rows:
        if pct < 0.01:
            continue
        line = f"| `{short(name)}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |"
        if a.steps:
            line += f" {tot / 1e3 / a.steps:.3f} |"
        print(line)
    total = sum(r[2] for r in rows)
    print(f"\ntotal kernel time {total / 1e3:.2f} ms" + (f" = {total / 1e3 / a.steps:.2f} ms/step over {a.steps} steps" if a.steps else ""))


if __name__ == "__main__":
    main()
