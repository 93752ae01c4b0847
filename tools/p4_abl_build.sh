#!/bin/bash
# timing-only builds of libcmtts with -DP4_ABL=n (denoiser_persist4.hip: wrong results by construction) into tools/bin/libcmtts_p4_ablN.so
# (never the product library); extra -D flags per variant after the name: tools/p4_abl_build.sh "abl1:-DP4_ABL=1" "wr4:-DP4_WR=4"
set -e
cd "$(dirname "$0")/../cm-tts_amd/csrc"
mkdir -p ../../tools/bin
OBJS=$(ls *.o | grep -v '^denoiser_persist4.o$' | tr '\n' ' ')
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c denoiser_persist4.hip -o /tmp/p4_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libcmtts_p4_$name.so $OBJS /tmp/p4_$name.o -ldl
done
ls -la ../../tools/bin/
