#!/bin/bash
# Quick experimental build of resblock16.hip: ONE bf16 instance (C=$C, K=$K; default 64, 7) + extra -D flags (e.g. -DRB16_STAMP=1), linked with the
# other objects of the real library: cm-tts_amd/libcmtts_hip_exp$TAG.so.  Then: CMTTS_LIB=cm-tts_amd/libcmtts_hip_exp$TAG.so python tools/rb16_phases.py
set -e
cd "$(dirname "$0")/../cm-tts_amd/csrc"
OBJS=$(ls *.o | grep -v '^resblock16.o$')
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DRB16_DEV=1 -DRB16_DEV_C=${C:-64} -DRB16_DEV_K=${K:-7} "$@" -c resblock16.hip -o /tmp/rb16_exp$TAG.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcmtts_hip_exp$TAG.so $OBJS /tmp/rb16_exp$TAG.o -ldl
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy -O binary --only-section=.hip_fatbin /tmp/rb16_exp$TAG.o /tmp/rb16_exp$TAG.fatbin
$B/clang-offload-bundler --unbundle --type=o --input=/tmp/rb16_exp$TAG.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/rb16_exp$TAG.co
$B/llvm-readelf --notes /tmp/rb16_exp$TAG.co | grep -E "\.name:|\.vgpr_count|vgpr_spill" | paste - - - | awk '{print $2, $4, $6}' | c++filt | sed 's/(anonymous namespace):://' | cut -c1-100
