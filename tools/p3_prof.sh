#!/bin/bash
# fp16x3 HiFi-GAN generator: bitwise test of the X-resident kernels, A/B against the chunked path, per-kernel times (chains in line).
# Usage (GPU box): bash tools/p3_prof.sh  -> gpurun_out/p3prof/
O=/root/repo/gpurun_out/p3prof; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "pair16x3" 2>&1 | tail -3
VP=fp16x3 timeout 300 python tools/voc_switch_ab.py voc_pair3 0 1
cd /tmp; export TMPDIR=/tmp
VSTREAMS=0 VP=fp16x3 VN=3 timeout 600 rocprofv3 --kernel-trace --stats -d $O -o p3 --output-format csv -- python /root/repo/tools/voc_prof.py > /dev/null 2>&1
cd /root/repo
python - <<'PY'
import csv,re
rows=list(csv.DictReader(open('/root/repo/gpurun_out/p3prof/p3_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name'])[:70]
    print(f"{n:72s} {r['Calls']:>5s} {float(r['TotalDurationNs'])/1e3:10.1f} {float(r['AverageNs'])/1e3:9.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}")
print("ms per pass", tot/1e6/3)
PY
