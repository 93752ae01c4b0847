#!/usr/bin/env python3
"""One precision / one voc_pair setting of the HiFi-GAN generator, a few passes: the target of rocprofv3 runs.
Env: VB, VT, VP (fp32|bf16|fp16), VPAIR (0|1), VN (passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict

lib = _lib.load()
B, T = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512))
_lib.internal_set(b"voc_pair", int(os.environ.get("VPAIR", 1)))
_lib.internal_set(b"voc_xl", int(os.environ.get("VXL", os.environ.get("VPAIR", 1))))
lib.cmtts_set_option(b"branch_streams", int(os.environ.get("VSTREAMS", 1)))      # 0: the three ResBlock chains in line (clean per-kernel times)
_lib.internal_set(b"voc_rb16", int(os.environ.get("VRB", 1)))       # 2 = whole-ResBlock kernel for every (C, k) of the narrow stages
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
voc.set_precision(os.environ.get("VP", "fp32"))
mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
for _ in range(int(os.environ.get("VN", 3))):
    w = voc(mel)
torch.cuda.synchronize()
