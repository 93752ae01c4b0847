#!/usr/bin/env python3
"""A few T = 4 samples of a 32 x 512 batch with 16-bit residual-block operands (LP = bf16 | fp16 | fp16x3): the target of rocprofv3
runs on denoiser_persist_lp_kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0))
model.set_precision(os.environ.get("LP", "bf16"))
B, T = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512))
cond = torch.randn(B, 256, T, device="cuda"); noise = torch.randn(5, B, 1, T, 80, device="cuda")
for _ in range(int(os.environ.get("VN", 3))):
    mel = host.sample_with_cond(model, cond, None, 4, noise)
torch.cuda.synchronize()
