"""Per-stage summary of one HiFi-GAN forward from a rocprofv3 --kernel-trace CSV (last forward in the file)."""
import csv, glob, re, sys

def main(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "conv" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = 78
    last = rows[-n:]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
    names = [re.search(r"(conv1d_mfma16_kernel|conv1d_mfma_kernel|conv_post_kernel)(<[^>]*>)?", r["Kernel_Name"]).group(0) for r in last]
    print("conv_pre %.0f us" % dur[0])
    p = 1
    for st in range(4):
        print("stage %d: ups %.0f us (%s)" % (st + 1, dur[p], names[p]))
        p += 1
        for j, k in enumerate((3, 7, 11)):
            c1 = dur[p:p + 6:2]; c2 = dur[p + 1:p + 6:2]
            print("   k=%2d  c1 %s  c2 %s  (%s)" % (k, " ".join("%4.0f" % x for x in c1), " ".join("%4.0f" % x for x in c2), names[p]))
            p += 6
    print("conv_post %.0f us" % dur[p])
    print("total %.2f ms; resblock convs %.2f ms" % (sum(dur) / 1e3, sum(d for d, nm in zip(dur, names) if "mfma16" in nm) / 1e3))

main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/voc16")
