#!/bin/bash
# timing-only builds of libcmtts with -DLP_STAMP [-DLP_ABL=n] into tools/bin/libcmtts_lpstamp[_ablN].so (never the product library)
set -e
cd "$(dirname "$0")/../cm-tts_amd/csrc"
OBJS=$(ls *.o | grep -v denoiser_persist_lp.o | tr '\n' ' ')
for abl in "" 1 2 4 3; do
  FLAGS="-DLP_STAMP"; SUF=""
  if [ -n "$abl" ]; then FLAGS="$FLAGS -DLP_ABL=$abl"; SUF="_abl$abl"; fi
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c denoiser_persist_lp.hip -o /tmp/lp_stamp$SUF.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libcmtts_lpstamp$SUF.so $OBJS /tmp/lp_stamp$SUF.o -ldl
done
ls -la ../../tools/bin/
