"""Per-kernel breakdown of ONE headline bench step from a rocprofv3 --kernel-trace CSV (the 7th step)."""
import collections, csv, glob, re, sys

def main(d, list_upto=0):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "embed_tokens" in r["Kernel_Name"]]
    seg = rows[idx[6]:idx[7]]
    nm = lambda r: re.sub(r"\(.*", "", re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])))[:52]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    agg = collections.OrderedDict()
    for r in seg:
        c = agg.setdefault(nm(r), [0, 0.0]); c[0] += 1; c[1] += dur(r)
    tot = sum(v[1] for v in agg.values())
    print("kernels %d  wall %.1f us  kernel time %.1f us  idle %.1f us" % (len(seg), wall, tot, wall - tot))
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-54s %4d %9.1f us %5.1f%%" % (n, c, t, 100 * t / wall))
    for i, r in enumerate(seg[:list_upto]):
        print("%3d %-54s %7.1f  grid %s/%s/%s" % (i, nm(r), dur(r), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))

main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
