#!/bin/bash
# Timing-only ablation builds of conv_xres.hip (XRES_ABL = 1: epilogue without stores, 2: without the activation; results are
# WRONG): libcmtts_hip_xablN.so next to the real library.  Run e.g.  CMTTS_LIB=cm-tts_amd/libcmtts_hip_xabl1.so python tools/xres_phases.py
set -e
cd "$(dirname "$0")/../cm-tts_amd/csrc"
make -j8 > /dev/null
OBJS=$(ls *.o | grep -v '^conv_xres.o$')
for n in 1 2; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DXRES_ABL=$n -c conv_xres.hip -o /tmp/conv_xres_abl$n.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcmtts_hip_xabl$n.so $OBJS /tmp/conv_xres_abl$n.o -ldl
done
ls -la ../libcmtts_hip_xabl*.so
