#!/usr/bin/env python3
"""One denoiser evaluation, per-layer path: one workgroup per 32-frame tile (resblock_fused) vs four workgroups per tile
in two launches (resblock_split), over batch sizes — where does the split form stop winning?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
lib = _lib.load()
cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=3))
lib.cmtts_set_persistent_denoiser(0)
for B, T in [(1, 150), (1, 510), (2, 510), (4, 256), (4, 510), (8, 256), (8, 510), (12, 510), (16, 510)]:
    x = torch.randn(B, 1, T, 80, device="cuda"); cond = torch.randn(B, T, 256, device="cuda"); t = torch.full((B,), 1095.5, device="cuda")
    r = {}
    for mode in (0, 2):
        lib.cmtts_set_option(b"resblock_split", mode)
        for _ in range(3): model.net(x, t, cond, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): model.net(x, t, cond, None)
        torch.cuda.synchronize(); r[mode] = (time.perf_counter() - t0) / 10 * 1e3
    print(f"B={B} T={T} tiles32={B*((T+31)//32)}: one workgroup per tile {r[0]:.3f} ms, split {r[2]:.3f} ms", flush=True)
lib.cmtts_set_option(b"resblock_split", 1); lib.cmtts_set_persistent_denoiser(1)
