#!/usr/bin/env python3
"""Round 6: the 16-bit persistent stack with 128-frame tiles (denoiser_persist_lp128.hip) against the 64-frame kernel: ms per T = 4 sample (four
launches + conditioner GEMM + input projections) at shapes with more 64-frame tiles than CUs and at the bench shape."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host, _lib
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0))
for B, T in ((64, 512), (32, 1024), (32, 512), (16, 1024), (128, 512)):
    cond = torch.randn(B, 256, T, device="cuda"); noise = torch.randn(5, B, 1, T, 80, device="cuda")
    for dt in ("bf16", "fp16"):
        model.set_precision(dt)
        res = {}
        for sw in (0, 2, 0, 2):
            _lib.internal_set(b"persist_lp128", sw)
            for _ in range(2):
                mel = host.sample_with_cond(model, cond, None, 4, noise)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                mel = host.sample_with_cond(model, cond, None, 4, noise)
            torch.cuda.synchronize(); res.setdefault(sw, []).append((time.perf_counter() - t0) / 5 * 1e3)
            res[("mel", sw)] = mel
        _lib.internal_set(b"persist_lp128", 1)
        print(f"{dt} {B} x {T} ({B * ((T + 63) // 64)} 64-frame tiles): 64-frame tiles {min(res[0]):.3f} ms, 128-frame tiles {min(res[2]):.3f} ms per T = 4 sample "
              f"({min(res[0]) / min(res[2]):.2f} x); equal {torch.equal(res[('mel', 0)], res[('mel', 2)])}", flush=True)
model.set_precision("fp32")
