#!/usr/bin/env python3
"""HiFi-GAN generator with 16-bit ResBlock-conv operands: deep weight ring of the wide (C >= 128) convs on / off
(cmtts_set_option("voc_ring16")), bit equality and time per batch.  VB / VT / VP env as voc_bench.py."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict

lib = _lib.load()
lib.cmtts_set_option(b"voc_ring16", 1)     # the iteration-order fragment copies are built at finalize only when this is set
lib.cmtts_set_option(b"voc_xl16", 0)       # the ring lives in the chunked kernel, which conv_xl16 replaced for C >= 128
B, T = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512))
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
for prec in os.environ.get("VP", "bf16,fp16").split(","):
    voc.set_precision(prec)
    outs = []
    for ring in (0, 1):
        lib.cmtts_set_option(b"voc_ring16", ring)
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
        print(f"{prec} voc_ring16={ring} B={B} T={T}: {dt*1e3:.2f} ms/batch, finite={bool(torch.isfinite(w).all())}", flush=True)
    print(f"{prec}: ring vs one-step-ahead max|d| = {float((outs[0]-outs[1]).abs().max()):.3e}, bitwise={bool(torch.equal(outs[0], outs[1]))}")
lib.cmtts_set_option(b"voc_ring16", 1)
