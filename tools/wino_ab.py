#!/usr/bin/env python3
"""A/B of the fp32 persistent denoiser's k = 3 conv: direct form vs Winograd F(2,3) (cmtts_internal_set("persist_wino")).
Same process, interleaved rounds: time per T = 4 sample and the difference of the mels.  Env: VB, VT, ROUNDS."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0))
B, T = int(os.environ.get("VB", 32)), int(os.environ.get("VT", 512))
cond = torch.randn(B, 256, T, device="cuda"); noise = torch.randn(5, B, 1, T, 80, device="cuda")
_lib.load().cmtts_set_persistent_denoiser(2)
mels, times = {}, {0: [], 1: []}
for rnd in range(int(os.environ.get("ROUNDS", 4))):
    for wn in (0, 1):
        _lib.internal_set(b"persist_wino", wn)
        for _ in range(2 if rnd == 0 else 1):
            mel = host.sample_with_cond(model, cond, None, 4, noise)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            mel = host.sample_with_cond(model, cond, None, 4, noise)
        torch.cuda.synchronize(); times[wn].append((time.perf_counter() - t0) / 5)
        mels[wn] = mel
_lib.internal_set(b"persist_wino", 0)
d = (mels[1] - mels[0]).double()
print(f"B={B} T={T}: direct {min(times[0])*1e3:.3f} ms (rounds {[round(t*1e3,3) for t in times[0]]}), winograd {min(times[1])*1e3:.3f} ms ({[round(t*1e3,3) for t in times[1]]})")
print(f"  mel: max|d| {float(d.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e} (mel rms {float(mels[0].double().pow(2).mean().sqrt()):.3f}), finite {bool(torch.isfinite(mels[1]).all())}")
# one evaluation of the network (no sampler feedback): the difference a single denoiser call makes
x = torch.randn(B, 1, T, 80, device="cuda"); c2 = torch.randn(B, T, 256, device="cuda"); t = torch.full((B,), 1095.5, device="cuda")
outs = {}
for wn in (0, 1):
    _lib.internal_set(b"persist_wino", wn)
    outs[wn] = model.net(x, t, c2, None).double()
_lib.internal_set(b"persist_wino", 0)
d = outs[1] - outs[0]
print(f"  one network evaluation: max|d| {float(d.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt()):.3e} (output rms {float(outs[0].pow(2).mean().sqrt()):.3f})")
