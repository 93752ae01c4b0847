#!/bin/bash
# Timing-only ablation builds of conv_mfma16.hip (C16_ABL = 1: no weight stream, 2: no LDS operand reads, 3: both; results are
# WRONG): libcmtts_hip_c16ablN.so next to the real library.  Run e.g.
#   CMTTS_LIB=cm-tts_amd/libcmtts_hip_c16abl1.so VP=bf16 VSTREAMS=0 VPAIR=0 python tools/voc_prof.py   (under rocprofv3 --kernel-trace)
# with cmtts_set_option("voc_ring16", 0) (the hooks sit in the one-step-ahead loop).
set -e
cd "$(dirname "$0")/../cm-tts_amd/csrc"
make -j8 > /dev/null
OBJS=$(ls *.o | grep -v '^conv_mfma16.o$')
for n in 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DC16_ABL=$n -c conv_mfma16.hip -o /tmp/conv_mfma16_abl$n.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcmtts_hip_c16abl$n.so $OBJS /tmp/conv_mfma16_abl$n.o -ldl
done
ls -la ../libcmtts_hip_c16abl*.so
