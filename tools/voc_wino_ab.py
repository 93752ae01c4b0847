#!/usr/bin/env python3
"""fp32 HiFi-GAN generator: ResBlock convs of the C >= 128 stages in their direct form vs the Winograd form (cmtts_internal_set("voc_wino")).
Same process, interleaved rounds: time per batch and the difference of the waveforms.  Env: VB, VT, ROUNDS."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import HifiGanConfig
from cmtts_amd.weights import synth_hifigan_state_dict

B, T = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512))
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=0))
mel = torch.randn(B, 80, T, device="cuda") * 1.5 - 4
out, times = {}, {0: [], 1: []}
for rnd in range(int(os.environ.get("ROUNDS", 3))):
    for wn in (0, 1):
        _lib.internal_set(b"voc_wino", wn)
        for _ in range(2 if rnd == 0 else 1):
            w = voc(mel)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            w = voc(mel)
        torch.cuda.synchronize(); times[wn].append((time.perf_counter() - t0) / 3)
        out[wn] = w
_lib.internal_set(b"voc_wino", 1)
d = (out[1] - out[0]).double()
fl = B * T * 614.105088e6
print(f"B={B} T={T}: direct {min(times[0])*1e3:.2f} ms ({fl/min(times[0])/1e12:.1f} TFLOP/s), winograd {min(times[1])*1e3:.2f} ms ({fl/min(times[1])/1e12:.1f} TFLOP/s of the direct form's FLOPs)")
print(f"  wav: max|d| {float(d.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e} (wav rms {float(out[0].double().pow(2).mean().sqrt()):.3f}), finite {bool(torch.isfinite(out[1]).all())}")
