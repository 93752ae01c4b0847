#!/bin/bash
# Timing-only ablation builds of the 16-bit HiFi-GAN kernels (SRC = conv_xl16 [default] | resblock16 | resblock_pair16; X16_ABL bits: 1 no weight stream, 2 no LDS operand reads in the K loop, 4 no staging loads,
# 8 no epilogue loads / stores; value 1 ALONE sent the compiler into a >30-minute spin: use 3; results are WRONG): libcmtts_hip_x16ablN.so next to the real library, then tools/xl16_time.py on each.
set -e
cd "$(dirname "$0")/../cm-tts_amd/csrc"
make -j8 > /dev/null
SRC=${SRC:-conv_xl16}
OBJS=$(ls *.o | grep -v "^$SRC.o\$")
for n in ${ABLS:-2 3 4 8 12 15}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DX16_ABL=$n -c $SRC.hip -o /tmp/rp16_abl$n.o &
done
wait
for n in ${ABLS:-2 3 4 8 12 15}; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcmtts_hip_x16abl$n.so $OBJS /tmp/rp16_abl$n.o -ldl
done
ls ../libcmtts_hip_x16abl*.so
