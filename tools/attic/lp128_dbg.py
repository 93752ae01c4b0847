#!/usr/bin/env python3
"""denoiser_persist_lp128 vs the 64-frame 16-bit persistent kernel: where do they differ?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
import dataclasses
lib = _lib.load()
for NL in (3, 4, 8, 20):
    for B, T in ((1, 64), (1, 96), (2, 128), (2, 64), (1, 30)):
        cfg = dataclasses.replace(get_config("LJSpeech"), res_layers=NL)
        model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=6))
        gen = torch.Generator(device="cpu").manual_seed(B * 1000 + T)
        cond = torch.randn(B, T, cfg.hidden, generator=gen); x = torch.randn(B, 1, T, cfg.n_mels, generator=gen); t = torch.full((B,), 1095.5)
        lib.cmtts_set_persistent_denoiser(2); model.set_precision("bf16")
        _lib.internal_set(b"persist_lp128", 2); a = model.net(x, t, cond, None).clone()
        _lib.internal_set(b"persist_lp128", 0); r = model.net(x, t, cond, None).clone()
        torch.cuda.synchronize()
        d = (a - r).abs()[:, 0].cpu().numpy()          # [B][T][80]
        fr = np.nonzero(d.max(axis=(0, 2)) > 0)[0]
        print(f"NL={NL} B={B} T={T}: max {d.max():.3e}; frames with a difference: {fr[:3].tolist()} .. {fr[-3:].tolist()} count {len(fr)}; per utterance max {[float(d[i].max()) for i in range(B)]}")
