#!/usr/bin/env python3
"""HiFi-GAN generator A/B over one internal switch (csrc/internal_hooks.h): time per batch and bit equality of the wav.
Usage: VP=bf16 python tools/voc_switch_ab.py voc_pairw 0 1   (env VB, VT, VN as voc_bench.py)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict

name = sys.argv[1].encode()
values = [int(v) for v in sys.argv[2:]] or [0, 1]
B, T, N = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512)), int(os.environ.get("VN", 10))
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
for prec in os.environ.get("VP", "bf16").split(","):
    voc.set_precision(prec)
    outs = []
    for v in values:
        prev = _lib.internal_set(name, v)
        for _ in range(3):
            w = voc(mel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            w = voc(mel)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / N
        outs.append(w.clone())
        _lib.internal_set(name, prev)
        print(f"{prec} {name.decode()}={v} B={B} T={T}: {dt*1e3:.2f} ms/batch, {B*T*614.105088e6/dt/1e12:.1f} TFLOP/s", flush=True)
    for i in range(1, len(outs)):
        print(f"{prec}: {values[i]} vs {values[0]}: max|d| = {float((outs[i]-outs[0]).abs().max()):.3e}, bitwise={bool(torch.equal(outs[i], outs[0]))}")
