#!/usr/bin/env python3
"""A/B in one process: the FFT blocks' in-projection (conv_xres with the LayerNorm prologue) with the launcher's tile rule (qkv_nt = 0) against its
32-column instance forced (1); the same script measured the out-projection on the generic kernel against conv_xres (text_xres 5 / 7) in round 4."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import _lib, host
from cmtts_amd.config import get_config
from cmtts_amd.weights import synth_cmtts_state_dict
cfg = get_config("LJSpeech")
model = host.CMTotalTTS(cfg, "cuda:0").load_state_dict(synth_cmtts_state_dict(cfg, seed=0, dur_frames=6.0, dur_spread=0.0))
rs = np.random.RandomState(0)
for B, L in ((32, 85), (64, 85), (32, 171), (8, 85)):
    texts = torch.from_numpy(rs.randint(1, cfg.n_symbols, size=(B, L)).astype(np.int64)).cuda()
    lens = torch.full((B,), L, dtype=torch.int64, device="cuda")
    ref, res = None, {}
    for r in range(4):
        for bits in (0, 1):
            _lib.internal_set("qkv_nt", bits)
            for _ in range(2): o = model.duration_pitch_energy_net(None, texts, lens, max_mel_len=6 * L)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): o = model.duration_pitch_energy_net(None, texts, lens, max_mel_len=6 * L)
            torch.cuda.synchronize(); res.setdefault(bits, []).append((time.perf_counter() - t0) / 20 * 1e3)
            if ref is None: ref = o["cond_ct"].clone()
            assert torch.equal(o["cond_ct"], ref)
    _lib.internal_set("qkv_nt", 0)
    print(f"B={B} L={L}: QKV rule {np.median(res[0]):.3f} ms, 32-column {np.median(res[1]):.3f} ms (same bits)")
