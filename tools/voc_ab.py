#!/usr/bin/env python3
"""HiFi-GAN generator A/B: the fused ResBlock pair kernels (voc_pair) on / off, per precision.  VB / VT / VP env as voc_bench.py."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict

lib = _lib.load()
B, T = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512))
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
for prec in os.environ.get("VP", "fp32").split(","):
    voc.set_precision(prec)
    outs = []
    for vp in (0, 1):
        _lib.internal_set(b"voc_pair", vp)
        _lib.internal_set(b"voc_xl", vp)
        for _ in range(2):
            w = voc(mel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            w = voc(mel)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        outs.append(w.clone())
        print(f"{prec} voc_pair={vp} B={B} T={T}: {dt*1e3:.2f} ms/batch, {B*T*614.105088e6/dt/1e12:.1f} TFLOP/s, finite={bool(torch.isfinite(w).all())}", flush=True)
    print(f"{prec}: pair vs two-launch max|d| = {float((outs[0]-outs[1]).abs().max()):.3e}, bitwise={bool(torch.equal(outs[0], outs[1]))}")
