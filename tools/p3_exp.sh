#!/bin/bash
# Quick experimental build of resblock_pair16x3.inc: ONE pair instance (C = $C, default 128; k = $K, default 7; tile -DP3_DEV_N1/_WN/_OCC, ring -DP3_RING,
# -DP3_B128=1 ...) linked with the other objects of the real library: cm-tts_amd/libcmtts_hip_exp$TAG.so (the other shapes fall back to the chunked path).
# Then: CMTTS_LIB=cm-tts_amd/libcmtts_hip_exp$TAG.so VP=fp16x3 python tools/voc_prof.py under rocprofv3.
set -e
cd "$(dirname "$0")/../cm-tts_amd/csrc"
K=${K:-7}; C=${C:-128}
OBJS=$(ls *.o | grep -v "^resblock_pair16x3_k$K.o$")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DP3_DEV=$C "$@" -c resblock_pair16x3_k$K.hip -o /tmp/p3_exp$TAG.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcmtts_hip_exp$TAG.so $OBJS /tmp/p3_exp$TAG.o -ldl
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy -O binary --only-section=.hip_fatbin /tmp/p3_exp$TAG.o /tmp/p3_exp$TAG.fatbin
$B/clang-offload-bundler --unbundle --type=o --input=/tmp/p3_exp$TAG.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/p3_exp$TAG.co
$B/llvm-readelf --notes /tmp/p3_exp$TAG.co | grep -E "\.name:|\.vgpr_count|vgpr_spill|agpr_count" | sed 's/^ *//' | paste - - - - | c++filt | sed 's/(anonymous namespace):://;s/void //' | cut -c1-200
