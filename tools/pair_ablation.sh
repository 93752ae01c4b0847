#!/bin/bash
# Timing-only ablation builds of resblock_pair.hip (PAIR_DBG = 1..4, see the source): libcmtts_hip_dbgN.so next to the
# real library (git-ignored, shipped by gpurun).  Run e.g.  CMTTS_LIB=cm-tts_amd/libcmtts_hip_dbg1.so python tools/voc_prof.py
set -e
cd "$(dirname "$0")/../cm-tts_amd/csrc"
make -j8 > /dev/null
OBJS=$(ls *.o | grep -v '^resblock_pair.o$')
for n in 1 2 3 4 5; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPAIR_DBG=$n -c resblock_pair.hip -o /tmp/resblock_pair_dbg$n.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcmtts_hip_dbg$n.so $OBJS /tmp/resblock_pair_dbg$n.o
done
ls -la ../libcmtts_hip_dbg*.so
