#!/usr/bin/env python3
"""Winograd F(4,3) persistent stack (denoiser_persist.hip WINO == 2, persist_wino = 3) against the direct form and F(2,3): error against the
direct fp32 form at even / odd / huge shapes, with cp and with factors; then an interleaved timing A/B against F(2,3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
lib = _lib.load()
ok = True
for variant in ("VCTK", "LJSpeech"):
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=3))
    lib.cmtts_set_persistent_denoiser(2)
    for B, T in [(2, 200), (3, 64), (1, 130), (5, 1000), (32, 512), (33, 513), (5, 65), (1, 5000), (7, 1), (3, 63)]:
        g = torch.Generator().manual_seed(B * 7 + T)
        x = torch.randn(B, 1, T, 80, generator=g).cuda(); cond = torch.randn(B, T, 256, generator=g).cuda()
        spk = torch.randn(B, 256, generator=g).cuda() if cfg.multi_speaker else None
        t = torch.full((B,), 1095.5).cuda()
        noise = torch.randn(3, B, 1, T, 80, generator=g).cuda()
        cond_ct = cond.transpose(1, 2).contiguous()
        outs, mels = {}, {}
        for wn in (0, 1, 3):
            _lib.internal_set(b"persist_wino", wn)
            outs[wn] = model.net(x, t, cond, spk).clone()
            mels[wn] = host.sample_with_cond(model, cond_ct, spk, 2, noise).clone()
        torch.cuda.synchronize()
        d3 = float((outs[3] - outs[0]).abs().max()); d1 = float((outs[1] - outs[0]).abs().max())
        m3 = float((mels[3] - mels[0]).abs().max()); m1 = float((mels[1] - mels[0]).abs().max())
        fin = bool(torch.isfinite(outs[3]).all() and torch.isfinite(mels[3]).all())
        print(f"{variant} B={B} T={T}: net F(4,3) vs direct {d3:.1e} (F(2,3) {d1:.1e}); T=2 mel {m3:.1e} (F(2,3) {m1:.1e}); scale {float(outs[0].abs().max()):.2f}; finite {fin}", flush=True)
        ok &= fin and d3 <= 1e-4 and m3 <= 1e-4
print("ALL OK" if ok else "FAILED", flush=True)
cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0))
B, T = 32, 512
cond = torch.randn(B, 256, T, device="cuda"); noise = torch.randn(5, B, 1, T, 80, device="cuda")
times = {1: [], 3: []}
for rnd in range(4):
    for wn in (1, 3):
        _lib.internal_set(b"persist_wino", wn)
        for _ in range(2 if rnd == 0 else 1):
            host.sample_with_cond(model, cond, None, 4, noise)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            host.sample_with_cond(model, cond, None, 4, noise)
        torch.cuda.synchronize(); times[wn].append((time.perf_counter() - t0) / 5)
print(f"T=4 sample B=32 x 512: F(2,3) {min(times[1])*1e3:.3f} ms {[round(t*1e3,3) for t in times[1]]}, F(4,3) {min(times[3])*1e3:.3f} ms {[round(t*1e3,3) for t in times[3]]}")
