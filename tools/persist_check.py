#!/usr/bin/env python3
"""Persistent denoiser stack vs per-layer kernels: bitwise comparison and timing (GPU only)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict

lib = _lib.load()
shapes = [(2, 200), (3, 64), (1, 130), (5, 1000), (32, 512), (40, 512), (16, 1024)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
for variant in ("LJSpeech", "VCTK"):
    cfg = get_config(variant)
    model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=3))
    for B, T in shapes:
        g = torch.Generator(device="cuda").manual_seed(B * 1000 + T)
        x = torch.randn(B, 1, T, 80, device="cuda", generator=g)
        cond = torch.randn(B, T, 256, device="cuda", generator=g)
        spk = torch.randn(B, 256, device="cuda", generator=g) if cfg.multi_speaker else None
        t = torch.full((B,), 1095.5, device="cuda")
        res = {}
        for mode in (0, 1):
            lib.cmtts_set_persistent_denoiser(mode)
            y = model.net(x, t, cond, spk)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                y = model.net(x, t, cond, spk)
            torch.cuda.synchronize()
            res[mode] = (y, (time.perf_counter() - t0) / 3 * 1e3)
        lib.cmtts_set_persistent_denoiser(1)
        same = torch.equal(res[0][0], res[1][0])
        d = (res[0][0] - res[1][0]).abs().max().item()
        print(f"{variant} B={B} T={T}: bitwise {same} (max diff {d:.3e}); per-layer {res[0][1]:.2f} ms, persistent {res[1][1]:.2f} ms", flush=True)
