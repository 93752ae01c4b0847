#!/usr/bin/env python3
"""cfg2-shaped sampler timing per operand precision of the residual blocks (fp32 / bf16 / fp16)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0))
B, T = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512))

cond = torch.randn(B, 256, T, device="cuda"); noise = torch.randn(5, B, 1, T, 80, device="cuda")
ref = None
for dt in os.environ.get("LP", "fp32,bf16,fp16,fp16x3").split(","):
    model.set_precision(dt)
    for _ in range(2):
        mel = host.sample_with_cond(model, cond, None, 4, noise)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        mel = host.sample_with_cond(model, cond, None, 4, noise)
    torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 5
    if ref is None: ref = mel
    print(f"{dt}: {d*1e3:.2f} ms per T=4 sample of {B}x{T} frames = {B*T/d:.0f} frames/s; max|d vs fp32| {float((mel-ref).abs().max()):.2e}")
model.set_precision("fp32")
