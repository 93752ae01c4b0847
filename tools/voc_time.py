#!/usr/bin/env python3
"""Wall time of the HiFi-GAN generator per precision (32 x 512 mel frames by default).  Env: VB, VT, VN, VPS (comma list of fp32,bf16,fp16,fp16x3)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict

_lib.load()
B, T, N = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512)), int(os.environ.get("VN", 5))
mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
for prec in os.environ.get("VPS", "fp32,bf16,fp16x3").split(","):
    voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
    voc.set_precision(prec)
    for rnd in range(3):
        w = voc(mel)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(N):
            w = voc(mel)
        torch.cuda.synchronize()
        print(f"{prec} round {rnd}: {(time.perf_counter() - t) / N * 1e3:.3f} ms per {B} x {T}-frame batch", flush=True)
