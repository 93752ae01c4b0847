#!/usr/bin/env python3
"""HiFi-GAN generator alone: time per batch and achieved TFLOP/s (614.1 MFLOP per mel frame)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict

B, T = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512))
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
for prec in os.environ.get("VP", "fp32").split(","):
    voc.set_precision(prec)
    for _ in range(2):
        w = voc(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        w = voc(mel)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{prec} B={B} T={T}: {dt*1e3:.2f} ms/batch, {B*T/dt:.0f} mel-frames/s, {B*T*614.105088e6/dt/1e12:.1f} TFLOP/s, finite={bool(torch.isfinite(w).all())}")
