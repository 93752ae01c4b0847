#!/bin/bash
# timing-only builds of libcmtts with extra -D flags for ONE source file (wrong results by construction when an ablation flag is set) into
# tools/bin/libcmtts_<name>.so (never the product library): tools/abl_build.sh denoiser_persist "abl1:-DW43_ABL=1" "wr4:-DWINO43_RING=4"
set -e
cd "$(dirname "$0")/../cm-tts_amd/csrc"
mkdir -p ../../tools/bin
src="$1"; shift
OBJS=$(ls *.o | grep -v "^$src.o\$" | tr '\n' ' ')
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $src.hip -o /tmp/abl_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libcmtts_$name.so $OBJS /tmp/abl_$name.o -ldl
done
ls ../../tools/bin/
