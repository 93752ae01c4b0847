#!/bin/bash
# Timing-only build of the library with -DLP_STAMP (cycle stamps in denoiser_persist_lp.hip), loaded through CMTTS_LIB; prints the phase
# table of the bf16 persistent denoiser (tools/lp_phases.py).  Run on the GPU box: bash tools/lp_phases.sh
set -e
cd "$(dirname "$0")/.."
python tools/lp_phases.py
