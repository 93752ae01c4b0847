#!/bin/bash
# times of the experimental pair16x3 instances (tools/p3_exp.sh builds): bash tools/p3_exp_run.sh a b c ...
cd /tmp; export TMPDIR=/tmp
for t in "$@"; do
  O=/root/repo/gpurun_out/p3exp_$t; rm -rf $O; mkdir -p $O
  CMTTS_LIB=/root/repo/cm-tts_amd/libcmtts_hip_exp$t.so VSTREAMS=0 VP=fp16x3 VN=3 timeout 300 rocprofv3 --kernel-trace --stats -d $O -o p3 --output-format csv -- python /root/repo/tools/voc_prof.py > /dev/null 2>&1
  echo "variant $t: $(grep pair16x3 $O/p3_kernel_stats.csv | awk -F, '{print $1, $2, $4}' | tr -d '"' | sed 's/(anonymous namespace):://;s/void //')"
done
