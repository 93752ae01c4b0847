#!/bin/bash
# Quick experimental build of resblock_pairw16.hip (ONE bf16 instance: k = $K, default 11; extra -D flags from the command line, e.g. -DPW16_STAMP=1)
# linked with the other objects of the real library: cm-tts_amd/libcmtts_hip_exp$TAG.so.  Then: CMTTS_LIB=cm-tts_amd/libcmtts_hip_exp$TAG.so python tools/pw16_phases.py
set -e
cd "$(dirname "$0")/../cm-tts_amd/csrc"
K=${K:-11}
OBJS=$(ls *.o | grep -v "^resblock_pairw16_k$K.o$")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPW16_DEV=1 "$@" -c resblock_pairw16_k$K.hip -o /tmp/pw16_exp$TAG.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcmtts_hip_exp$TAG.so $OBJS /tmp/pw16_exp$TAG.o -ldl
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy -O binary --only-section=.hip_fatbin /tmp/pw16_exp$TAG.o /tmp/pw16_exp$TAG.fatbin
$B/clang-offload-bundler --unbundle --type=o --input=/tmp/pw16_exp$TAG.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/pw16_exp$TAG.co
$B/llvm-readelf --notes /tmp/pw16_exp$TAG.co | grep -E "\.name:|\.vgpr_count|vgpr_spill" | paste - - - | awk '{print $2, $4, $6}' | c++filt | sed 's/(anonymous namespace):://' | cut -c1-100
