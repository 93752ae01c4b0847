#!/bin/bash
# Quick experimental build of conv_xl16.hip (bf16 instances only, extra -D flags from the command line) linked with the other objects of the
# real library: cm-tts_amd/libcmtts_hip_exp$TAG.so.  Usage: TAG=a tools/xl16_exp.sh -DXL16_SOMETHING=1; then CMTTS_LIB=... python tools/xl16_time.py
set -e
cd "$(dirname "$0")/../cm-tts_amd/csrc"
OBJS=$(ls *.o | grep -v '^conv_xl16.o$')
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DXL16_BF16_ONLY "$@" -c conv_xl16.hip -o /tmp/xl16_exp$TAG.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcmtts_hip_exp$TAG.so $OBJS /tmp/xl16_exp$TAG.o -ldl
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy -O binary --only-section=.hip_fatbin /tmp/xl16_exp$TAG.o /tmp/xl16_exp$TAG.fatbin
$B/clang-offload-bundler --unbundle --type=o --input=/tmp/xl16_exp$TAG.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/xl16_exp$TAG.co
$B/llvm-readelf --notes /tmp/xl16_exp$TAG.co | grep -E "\.name:|\.vgpr_count|vgpr_spill" | paste - - - | awk '{print $2, $4, $6}' | c++filt | sed 's/(anonymous namespace):://' | cut -c1-100
