#!/usr/bin/env python3
"""Odd large shapes through every denoiser mode (bitwise) and the vocoder (finite, in-line == streamed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config, HifiGanConfig
from cmtts_amd.weights import synth_cmtts_state_dict, synth_hifigan_state_dict
lib = _lib.load()
cfg = get_config("VCTK")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=3))
voc = host.Generator(HifiGanConfig(), "cuda:0").load_state_dict(synth_hifigan_state_dict(HifiGanConfig(), seed=3))
ok = True
for B, T in [(3, 1000), (70, 300), (1, 5000), (5, 65), (2, 16385 // 8), (33, 513)]:
    g = torch.Generator().manual_seed(B * 7 + T)
    x = torch.randn(B, 1, T, 80, generator=g).cuda(); cond = torch.randn(B, T, 256, generator=g).cuda()
    spk = torch.randn(B, 256, generator=g).cuda(); t = torch.full((B,), 1095.5).cuda()
    outs = {}
    _lib.internal_set(b"persist_wino", 0)          # the direct conv form: every mode bit for bit
    for name, fused, pers in (("3-launch", 0, 0), ("per-layer", 1, 0), ("persistent", 1, 2), ("default", 1, 1)):
        lib.cmtts_set_fused_resblock(fused); lib.cmtts_set_persistent_denoiser(pers)
        outs[name] = model.net(x, t, cond, spk).clone()
    _lib.internal_set(b"persist_wino", 1)          # the default (Winograd) form of the persistent stack: fp32 rounding only
    lib.cmtts_set_persistent_denoiser(2)
    wino = model.net(x, t, cond, spk).clone()
    torch.cuda.synchronize()
    dw = float((wino - outs["3-launch"]).abs().max())
    same = all(torch.equal(outs["3-launch"], v) for v in outs.values()) and 0 < dw <= 3e-5
    fin = bool(torch.isfinite(outs["default"]).all())
    lib.cmtts_set_fused_resblock(1); lib.cmtts_set_persistent_denoiser(1)
    Tv = min(T, 700)
    mel = (torch.randn(min(B, 6), 80, Tv, generator=g) * 1.5 - 4).cuda()
    lib.cmtts_set_option(b"branch_streams", 0); w0 = voc(mel).clone()
    lib.cmtts_set_option(b"branch_streams", 1); w1 = voc(mel); torch.cuda.synchronize()
    vs = torch.equal(w0, w1) and bool(torch.isfinite(w1).all())
    print(f"B={B} T={T}: denoiser modes bitwise (direct form) and Winograd stack within 3e-5 (max|d| {dw:.1e}) {same}, finite {fin}; vocoder ({mel.shape[0]}x{Tv}) streamed == in-line and finite {vs}", flush=True)
    ok &= same and fin and vs
print("ALL OK" if ok else "FAILED")
