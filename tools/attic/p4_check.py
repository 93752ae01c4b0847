#!/usr/bin/env python3
"""One-wave-per-SIMD persistent stack (denoiser_persist4.hip, persist_wino = 2) against the 8-wave Winograd instances (persist_wino = 1):
same arithmetic per element => bit for bit, at even / odd / huge shapes, with cp and with factors; then an interleaved timing A/B."""
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
    for B, T in [(2, 200), (3, 64), (1, 130), (5, 1000), (32, 512), (40, 300), (33, 513), (5, 65), (1, 5000), (70, 300), (7, 1), (3, 63)]:
        g = torch.Generator().manual_seed(B * 7 + T)
        x = torch.randn(B, 1, T, 80, generator=g).cuda(); cond = torch.randn(B, T, 256, generator=g).cuda()
        spk = torch.randn(B, 256, generator=g).cuda() if cfg.multi_speaker else None
        t = torch.full((B,), 1095.5).cuda()
        noise = torch.randn(3, B, 1, T, 80, generator=g).cuda()
        cond_ct = cond.transpose(1, 2).contiguous()
        outs, mels = {}, {}
        for wn in (0, 1, 2):
            _lib.internal_set(b"persist_wino", wn)
            outs[wn] = model.net(x, t, cond, spk).clone()
            mels[wn] = host.sample_with_cond(model, cond_ct, spk, 2, noise).clone()
        torch.cuda.synchronize()
        same = torch.equal(outs[1], outs[2]) and torch.equal(mels[1], mels[2])
        d = float((outs[2] - outs[0]).abs().max()); d1 = float((outs[2] - outs[1]).abs().max())
        fin = bool(torch.isfinite(outs[2]).all())
        print(f"{variant} B={B} T={T}: wino4 == wino8 bitwise {same} (max|d| {d1:.1e}); vs direct {d:.1e}; finite {fin}", flush=True)
        ok &= same and fin and d <= 3e-5
_lib.internal_set(b"persist_wino", 2)
print("ALL OK" if ok else "FAILED", flush=True)
# timing A/B (LJSpeech, B = 32 x 512, T = 4 sample)
cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0))
B, T = 32, 512
cond = torch.randn(B, 256, T, device="cuda"); noise = torch.randn(5, B, 1, T, 80, device="cuda")
times = {1: [], 2: []}
for rnd in range(4):
    for wn in (1, 2):
        _lib.internal_set(b"persist_wino", wn)
        for _ in range(2 if rnd == 0 else 1):
            host.sample_with_cond(model, cond, None, 4, noise)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            host.sample_with_cond(model, cond, None, 4, noise)
        torch.cuda.synchronize(); times[wn].append((time.perf_counter() - t0) / 5)
print(f"T=4 sample B=32 x 512: 8-wave {min(times[1])*1e3:.3f} ms {[round(t*1e3,3) for t in times[1]]}, one wave per SIMD {min(times[2])*1e3:.3f} ms {[round(t*1e3,3) for t in times[2]]}")
